"""The on-disk contract between the CLIP encoder path and the caption trainer:

  extract_embeddings  = /root/reference/CLIP_prefix_caption/parse_coco.py:15-68 - per annotation: `encode_image` (the prefix),
                        2-way caption-type and 9-way violation-type zero-shot -> `attribute = f'{caption_type} {violation_type} '`,
                        `clip_embedding = i`; batched here, prompt features encoded once.
  save_embeddings / load_embeddings: the pickle `{"clip_embedding": Tensor[N,512], "captions": [annotation dicts]}`
                        (parse_coco.py:64-65; the reference's periodic dump uses the typo key "clip_embeddings", :62 - not reproduced).
  ClipCocoDataset     = /root/reference/CLIP_prefix_caption/train.py:27-107: tokenise caption + attribute, pad/truncate to
                        max_seq_len = min(int(mean + 10*std), max), attribute to attribute_length, mask = cat(ones(P+A), tokens>=0),
                        side-effect `<data>_tokens.pkl`.
Pickles here are files this code (or the user's own run of it) wrote; nothing shipped by the reference is unpickled.
"""
from __future__ import annotations

import pickle
import sys
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch.utils.data import Dataset

CAPTION_TYPES = {"status": "現況", "violation": "缺失"}                                     # parse_coco.py:24-27
VIOLATION_TYPES = ["墜落", "防護具", "感電", "工作場所", "物料", "爆炸", "穿刺", "機械", "搬運"]   # parse_coco.py:28


def save_embeddings(path: str, clip_embedding: torch.Tensor, captions: List[dict]) -> None:
    with open(path, "wb") as f:
        pickle.dump({"clip_embedding": clip_embedding.detach().cpu(), "captions": captions}, f)


def load_embeddings(path: str) -> dict:
    with open(path, "rb") as f:
        return pickle.load(f)


@torch.no_grad()
def extract_embeddings(model, annotations: Sequence[dict], load_image: Callable[[dict], torch.Tensor], tokenize: Callable,
                       batch_size: int = 256) -> Tuple[torch.Tensor, List[dict]]:
    """load_image(annotation) -> preprocessed [3,R,R] tensor.  Returns (embeddings [N,E] on CPU, annotated captions)."""
    from clip.data import ZeroShotClassifier
    dev = model.logit_scale.device
    cap_cls = ZeroShotClassifier(model, tokenize(list(CAPTION_TYPES.keys())), list(CAPTION_TYPES.values()))
    vio_cls = ZeroShotClassifier(model, tokenize(VIOLATION_TYPES), VIOLATION_TYPES)
    feats, out = [], []
    for s in range(0, len(annotations), batch_size):
        chunk = annotations[s:s + batch_size]
        images = torch.stack([load_image(a) for a in chunk]).to(dev)
        f = model.encode_image(images)
        _, _, cap = cap_cls(image_features=f)
        _, _, vio = vio_cls(image_features=f)
        for j, a in enumerate(chunk):
            a = dict(a)
            a["clip_embedding"] = s + j
            a["attribute"] = f"{cap[j]} {vio[j]} "
            out.append(a)
        feats.append(f.cpu())
    return torch.cat(feats, dim=0), out


class ClipCocoDataset(Dataset):
    def __init__(self, data_path: str, prefix_length: int, attribute_length: int, gpt2_type: str = "", normalize_prefix=False,
                 tokenizer=None, write_tokens_cache: bool = True):
        if tokenizer is None:
            from transformers import AutoTokenizer                      # train.py:67 (needs local files: no network here)
            tokenizer = AutoTokenizer.from_pretrained(gpt2_type, local_files_only=True)
        self.tokenizer = tokenizer
        self.prefix_length, self.attribute_length, self.normalize_prefix = prefix_length, attribute_length, normalize_prefix
        all_data = load_embeddings(data_path)
        print("Data size is %0d" % len(all_data["clip_embedding"]))
        sys.stdout.flush()
        self.prefixes = all_data["clip_embedding"]
        self.captions, self.captions_tokens, self.attributes_tokens, self.caption2embedding = [], [], [], []
        max_seq_len = 0
        for caption in all_data["captions"]:
            if caption["caption"] == "":
                caption["caption"] = caption["violation_list"]          # train.py:85-86
            self.captions.append(caption["caption"])
            self.captions_tokens.append(torch.tensor(self.tokenizer.encode(caption["caption"]), dtype=torch.int64))
            self.attributes_tokens.append(torch.tensor(self.tokenizer.encode(caption["attribute"]), dtype=torch.int64))
            self.caption2embedding.append(caption["clip_embedding"])
            max_seq_len = max(max_seq_len, self.captions_tokens[-1].shape[0])
        if write_tokens_cache:
            with open(f"{data_path[:-4]}_tokens.pkl", "wb") as f:      # train.py:103-104
                pickle.dump([self.captions_tokens, self.caption2embedding, max_seq_len], f)
        all_len = torch.tensor([len(t) for t in self.captions_tokens]).float()
        self.max_seq_len = min(int(all_len.mean() + all_len.std() * 10), int(all_len.max()))

    def __len__(self) -> int:
        return len(self.captions_tokens)

    def pad_tokens(self, item: int):
        tokens = self.captions_tokens[item]
        padding = self.max_seq_len - tokens.shape[0]
        if padding > 0:
            tokens = torch.cat((tokens, torch.zeros(padding, dtype=torch.int64)))
        elif padding < 0:
            tokens = tokens[:self.max_seq_len]
        self.captions_tokens[item] = tokens
        attribute = self.attributes_tokens[item]
        padding = self.attribute_length - attribute.shape[0]
        if padding > 0:
            attribute = torch.cat((attribute, torch.zeros(padding, dtype=torch.int64)))
        elif padding < 0:
            attribute = attribute[:self.attribute_length]
        self.attributes_tokens[item] = attribute
        mask = tokens.ge(0)
        tokens[~mask] = 0
        mask = torch.cat((torch.ones(self.prefix_length + self.attribute_length), mask.float()), dim=0)
        return tokens, attribute, mask

    def __getitem__(self, item: int):
        tokens, attribute, mask = self.pad_tokens(item)
        prefix = self.prefixes[self.caption2embedding[item]]
        if self.normalize_prefix:
            prefix = prefix.float()
            prefix = prefix / prefix.norm(2, -1)
        return tokens, mask, prefix, attribute
