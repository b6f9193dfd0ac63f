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


def _pad_rows(rows: Sequence[torch.Tensor], width: int) -> torch.Tensor:
    """[len(rows), width] int64: every row cut to `width` and right-padded with id 0."""
    out = torch.zeros(len(rows), width, dtype=torch.int64)
    for i, r in enumerate(rows):
        n = min(int(r.shape[0]), width)
        out[i, :n] = r[:n]
    return out


class ClipCocoDataset(Dataset):
    """Caption-trainer dataset with the reference's interface (train.py:27-107) and a dense layout: everything
    `pad_tokens` decides per visit in the reference is decided ONCE here, into three tensors

        tokens     [N, max_seq_len]            int64   caption ids cut / zero-padded to max_seq_len, negative ids -> 0
        attributes [N, attribute_length]       int64   attribute ids cut / zero-padded
        masks      [N, P + A + max_seq_len]    float   ones over prefix + attribute, then (original id >= 0)

    so `__getitem__` is four row views (a DataLoader worker collates them without touching Python lists) and a whole epoch can
    be moved to the device with `.tensors()`.  `max_seq_len = min(int(mean + 10 std), max)` over the caption lengths.
    Pinned against the reference's own `pad_tokens` / `__getitem__` on hand-made token lists
    (tests/golden/ref_pad_tokens.pt).  One reference behaviour is deliberately not reproduced: it zeroes negative ids inside
    the list it caches, so from the SECOND visit of such an item on its mask is all ones; here the mask of an item is the same
    every epoch (tokenizer ids are never negative, so the two only differ on hand-made inputs)."""

    def __init__(self, data_path: str, prefix_length: int, attribute_length: int, gpt2_type: str = "", normalize_prefix=False,
                 tokenizer=None, write_tokens_cache: bool = True):
        if tokenizer is None:
            from transformers import AutoTokenizer                      # train.py:67 (needs local files: no network here)
            tokenizer = AutoTokenizer.from_pretrained(gpt2_type, local_files_only=True)
        self.tokenizer = tokenizer
        all_data = load_embeddings(data_path)
        print("Data size is %0d" % len(all_data["clip_embedding"]))
        sys.stdout.flush()
        records = all_data["captions"]
        for rec in records:
            if rec["caption"] == "":
                rec["caption"] = rec["violation_list"]                  # train.py:85-86
        self.captions = [rec["caption"] for rec in records]
        caps = [torch.tensor(tokenizer.encode(rec["caption"]), dtype=torch.int64) for rec in records]
        attrs = [torch.tensor(tokenizer.encode(rec["attribute"]), dtype=torch.int64) for rec in records]
        index = [rec["clip_embedding"] for rec in records]
        if write_tokens_cache:                                          # train.py:103-104 side effect, same payload
            with open(f"{data_path[:-4]}_tokens.pkl", "wb") as f:
                pickle.dump([caps, index, max((int(c.shape[0]) for c in caps), default=0)], f)
        self._build(caps, attrs, index, all_data["clip_embedding"], prefix_length, attribute_length, normalize_prefix, None)

    @classmethod
    def from_token_lists(cls, captions_tokens: Sequence[torch.Tensor], attributes_tokens: Sequence[torch.Tensor],
                         caption2embedding: Sequence[int], prefixes: torch.Tensor, prefix_length: int, attribute_length: int,
                         normalize_prefix: bool = False, max_seq_len: Optional[int] = None) -> "ClipCocoDataset":
        """The same dataset from already tokenised captions (no tokenizer / pickle involved)."""
        ds = cls.__new__(cls)
        ds.tokenizer, ds.captions = None, []
        ds._build(list(captions_tokens), list(attributes_tokens), list(caption2embedding), prefixes, prefix_length,
                  attribute_length, normalize_prefix, max_seq_len)
        return ds

    def _build(self, caps, attrs, index, prefixes, prefix_length, attribute_length, normalize_prefix, max_seq_len):
        self.prefix_length, self.attribute_length, self.normalize_prefix = prefix_length, attribute_length, normalize_prefix
        self.prefixes = prefixes
        self.caption2embedding = torch.as_tensor(index, dtype=torch.int64)
        if max_seq_len is None:
            lens = torch.tensor([float(c.shape[0]) for c in caps])
            max_seq_len = min(int(lens.mean() + lens.std() * 10), int(lens.max()))      # train.py:105-106
        self.max_seq_len = max_seq_len
        raw = _pad_rows(caps, max_seq_len)
        valid = raw.ge(0)                                               # "mask is zero where we out of sequence"
        self.tokens = raw * valid
        self.attributes = _pad_rows(attrs, attribute_length)
        self.masks = torch.cat((torch.ones(len(caps), prefix_length + attribute_length), valid.float()), dim=1)

    def __len__(self) -> int:
        return self.tokens.shape[0]

    def pad_tokens(self, item: int):
        return self.tokens[item], self.attributes[item], self.masks[item]

    def __getitem__(self, item: int):
        prefix = self.prefixes[self.caption2embedding[item]]
        if self.normalize_prefix:
            prefix = prefix.float()
            prefix = prefix / prefix.norm(2, -1)
        return self.tokens[item], self.masks[item], prefix, self.attributes[item]

    def tensors(self, device=None):
        """(tokens, masks, prefixes-per-item, attributes) of the whole set, optionally on `device`: with 288 GB of HBM the
        caption trainer can keep the dataset resident and index batches on the GPU."""
        prefix = self.prefixes[self.caption2embedding]
        if self.normalize_prefix:
            prefix = prefix.float()
            prefix = prefix / prefix.norm(2, -1, keepdim=True)
        out = (self.tokens, self.masks, prefix, self.attributes)
        return tuple(t.to(device) for t in out) if device is not None else out
