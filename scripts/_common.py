"""Shared pieces of the entry-point scripts: path setup, an offline stand-in tokenizer, and synthetic fixtures (annotation
JSON + generated images, an embedding pickle) so that every script runs end to end on a box with no network, no dataset, no
BPE vocabulary and no pretrained weights (`--synthetic`).  The scripts themselves are this repo's own statement of the
loops of CLIP/train.py, CLIP/predict.py and CLIP_prefix_caption/train.py on top of the MI355X `clip` / `clip_caption` packages."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "construction-clip_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

SOT, EOT = 49406, 49407


def byte_tokenize(texts, context_length: int = 77, vocab_size: int = 49408) -> torch.Tensor:
    """Stand-in for clip.tokenize when the BPE vocabulary (bpe_simple_vocab_16e6.txt.gz) is not on disk: [SOT] + one id per
    UTF-8 byte + [EOT], zero padded, over-long texts cut (the real tokenizer raises instead).  SOT / EOT are the model's two
    largest ids (49406 / 49407 for the real vocabulary), so EOT stays the largest id of a row - all encode_text relies on."""
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = vocab_size - 2, vocab_size - 1
    out = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, t in enumerate(texts):
        ids = [sot] + [1 + b % (vocab_size - 3) for b in t.encode("utf-8")][: context_length - 2] + [eot]
        out[i, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return out


def get_tokenize(model=None):
    """clip.tokenize when a vocabulary file is configured (CCLIP_BPE_PATH), the byte stand-in otherwise; sized for `model`."""
    import functools
    ctx = getattr(model, "context_length", 77)
    if os.environ.get("CCLIP_BPE_PATH"):
        import clip
        return functools.partial(clip.tokenize, context_length=ctx)
    return functools.partial(byte_tokenize, context_length=ctx, vocab_size=getattr(model, "vocab_size", 49408))


class ByteCaptionTokenizer:
    """Stand-in for AutoTokenizer.from_pretrained(<gpt2 type>) in --synthetic runs: ids 1 + byte (mod vocab), decode inverts it."""

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size

    def encode(self, text: str):
        return [1 + (b % (self.vocab_size - 1)) for b in text.encode("utf-8")]

    def decode(self, ids):
        return bytes(max(0, int(i) - 1) % 256 for i in ids if int(i) > 0).decode("utf-8", errors="replace")


CLASSES = ["fall", "machine", "material", "shock", "gear", "puncture", "blast", "site", "carry"]


def make_synthetic_annotations(out_dir: str, per_class: int = 12, size: int = 96, seed: int = 567) -> str:
    """`<out_dir>/all.json` in the reference's annotation layout ({"annotations": [{id, caption_type, violation_type,
    violation_list, caption, file_name}]}) plus one generated RGB image per annotation (class-tinted noise)."""
    from PIL import Image
    os.makedirs(os.path.join(out_dir, "images"), exist_ok=True)
    rng = np.random.RandomState(seed)
    anns = []
    for c, name in enumerate(CLASSES):
        tint = rng.randint(40, 215, size=3)
        for j in range(per_class):
            fn = f"images/{name}_{j:03d}.png"
            px = np.clip(rng.normal(0, 40, size=(size + 8 * (j % 3), size, 3)) + tint, 0, 255).astype(np.uint8)
            Image.fromarray(px).save(os.path.join(out_dir, fn))
            anns.append({"id": len(anns), "caption_type": "violation" if j % 2 else "status", "violation_type": name,
                         "violation_list": f"{name} hazard {j % 4}", "caption": f"worker near {name} zone {j}", "file_name": fn})
    path = os.path.join(out_dir, "all.json")
    with open(path, "w") as f:
        json.dump({"type": "synthetic", "annotations": anns}, f)
    return path


def log_line(**kw) -> None:
    print(json.dumps(kw), flush=True)
