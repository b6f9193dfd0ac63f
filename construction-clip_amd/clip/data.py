"""Host-side datasets feeding the contrastive step - the callers' side of the hot path.

ClipPairDataset       /root/reference/CLIP/train.py:36-91: class-balanced K-way groups.  For every combination of
                      `combination_num` classes, item i yields one (image, label text) per class, cycling through each
                      class's annotations (`pair_dict[k][item % len]`, train.py:53); every combination contributes a
                      fixed 50 items (train.py:91); train/test split per class at int(count * train_ratio) (train.py:73-87).
ClipCaptionPairDataset /root/reference/CLIP/train_caption.py:36-64: one (image, caption) per annotation, split at
                      int(len * train_ratio).
ZeroShotClassifier    the two zero-shot heads of parse_coco.py:24-56 / application.py:80-90 with the prompt features
                      encoded ONCE and images batched (the reference re-encodes the prompts per image at batch 1).

Image decoding / preprocess / tokenize stay on the host (DataLoader workers), exactly as in the reference; they are
injected (`image_loader`, `tokenize`) so the logic is testable without image files or the BPE vocabulary.
"""
from __future__ import annotations

import collections
import json
import os
from itertools import combinations
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

ITEMS_PER_COMBINATION = 50          # train.py:91 `self.cumulative_sizes = [50 for p in self.pair_list]`


def _default_loader(path: str):
    from PIL import Image
    return Image.open(path)


class ClipPairDataset(Dataset):
    def __init__(self, preprocess: Callable, json_path: str, image_path: str, train_ratio: float, key: str, split: str,
                 combination_num: int, image_loader: Callable = _default_loader, tokenize: Optional[Callable] = None):
        if tokenize is None:
            from .clip import tokenize as _tok
            tokenize = _tok
        data = json.load(open(json_path, "r"))
        self.preprocess, self.image_path, self.key = preprocess, image_path, key
        self.image_loader, self.tokenize = image_loader, tokenize
        annotations = [a for a in data["annotations"] if a[key] != ""]
        self.pair = [a[key] for a in annotations]
        c = collections.Counter(self.pair)                      # first-seen class order
        self.combination = list(combinations(c.keys(), combination_num))
        self.train_count = {k: int(v * train_ratio) for k, v in c.items()}
        pair_list = {"train": [], "test": []}
        for combine in self.combination:
            per_class = {k: [a for a in annotations if a[key] == k] for k in combine}
            pair_list["train"].append({k: v[:self.train_count[k]] for k, v in per_class.items()})
            pair_list["test"].append({k: v[self.train_count[k]:] for k, v in per_class.items()})
        self.pair_list = pair_list[split]
        self.cumulative_sizes = [ITEMS_PER_COMBINATION for _ in self.pair_list]

    def __len__(self) -> int:
        return int(np.sum(self.cumulative_sizes))

    def locate(self, item: int) -> Tuple[int, int]:
        """(combination index, index inside it) - the walk of train.py:45-49."""
        i = 0
        for i, length in enumerate(self.cumulative_sizes):
            if length <= item:
                item -= length
            else:
                break
        return i, item

    def annotations_for(self, item: int) -> List[dict]:
        i, item = self.locate(item)
        pair_dict = self.pair_list[i]
        return [pair_dict[k][item % len(pair_dict[k])] for k in pair_dict.keys()]

    def __getitem__(self, item: int):
        anns = self.annotations_for(item)
        images = [self.preprocess(self.image_loader(os.path.join(self.image_path, a["file_name"]))) for a in anns]
        text = [a[self.key] for a in anns]
        return torch.tensor(np.stack([np.asarray(im) for im in images])), self.tokenize(text)


class ClipCaptionPairDataset(Dataset):
    def __init__(self, preprocess: Callable, json_path: str, image_path: str, train_ratio: float, key: str, split: str,
                 image_loader: Callable = _default_loader, tokenize: Optional[Callable] = None):
        if tokenize is None:
            from .clip import tokenize as _tok
            tokenize = _tok
        self.preprocess, self.image_path, self.key = preprocess, image_path, key
        self.image_loader, self.tokenize = image_loader, tokenize
        annotations = [a for a in json.load(open(json_path, "r"))["annotations"] if a[key] != ""]
        n_train = int(len(annotations) * train_ratio)
        self.pair_list = {"train": annotations[:n_train], "test": annotations[n_train:]}[split]

    def __len__(self) -> int:
        return len(self.pair_list)

    def __getitem__(self, item: int):
        a = self.pair_list[item]
        image = self.preprocess(self.image_loader(os.path.join(self.image_path, a["file_name"])))
        return image, self.tokenize(a[self.key])[0]


class ZeroShotClassifier:
    """softmax(logit_scale * I @ T^T) -> argmax over a FIXED prompt set, as parse_coco.py:45-53 does per image."""

    def __init__(self, model, prompts_tokens: torch.Tensor, labels: Sequence[str]):
        assert prompts_tokens.shape[0] == len(labels)
        self.model, self.labels = model, list(labels)
        with torch.no_grad():
            self.text_features = model.encode_text(prompts_tokens.to(model.logit_scale.device))   # encoded once

    @torch.no_grad()
    def __call__(self, images: torch.Tensor = None, image_features: torch.Tensor = None):
        """Returns (similarity [N, P] softmax, indices [N], labels list).  Pass image_features to reuse an encode."""
        from .model import normalized_logits
        if image_features is None:
            image_features = self.model.encode_image(images)
        ls = self.model.logit_scale.detach().float().reshape(1).contiguous()
        logits = normalized_logits(image_features.contiguous().float(), self.text_features.contiguous().float(), ls)[0]
        sim = logits.softmax(dim=-1)
        idx = sim.argmax(dim=1)
        return sim, idx, [self.labels[i] for i in idx.tolist()]
