"""Module-level API of the `clip` drop-in: `available_models`, `load`, `tokenize` - the four names
the reference's scripts use (`clip.load("ViT-B/32", device=device)` at /root/reference/CLIP/train.py:105,
/root/reference/CLIP/predict.py:12, parse_coco.py:20 with jit=False; `clip.tokenize(texts)` at
CLIP/train.py:60, CLIP/predict.py:40, parse_coco.py:29-30).  Same argument meaning and error behaviour
as openai/CLIP; differences forced by the offline box are stated where they occur.
"""
from __future__ import annotations

import os
import warnings
from typing import List, Union

import numpy as np
import torch

from .model import CLIP, build_model
from .simple_tokenizer import SimpleTokenizer
from .weights import MODELS, init_state_dict

__all__ = ["available_models", "load", "tokenize"]

_tokenizer = None


def available_models() -> List[str]:
    return [m for m in MODELS if not m.startswith("test-")]


class _Transform:
    """openai/CLIP `_transform(n_px)`: Resize(n_px, BICUBIC) on the shorter side -> CenterCrop(n_px) -> RGB ->
    ToTensor -> Normalize(CLIP mean/std).  Host-side (PIL), as in the reference's DataLoader workers
    (CLIP/train.py:56,138); torchvision is not installed here so the four steps are restated on PIL + numpy."""
    MEAN = (0.48145466, 0.4578275, 0.40821073)
    STD = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, n_px: int):
        self.n_px = n_px

    def __call__(self, image) -> torch.Tensor:
        from PIL import Image
        n = self.n_px
        w, h = image.size
        if (w <= h and w != n) or (h <= w and h != n):
            if w < h:
                nw, nh = n, int(n * h / w)
            else:
                nh, nw = n, int(n * w / h)
            image = image.resize((nw, nh), Image.BICUBIC)
        w, h = image.size
        left, top = int(round((w - n) / 2.0)), int(round((h - n) / 2.0))
        image = image.crop((left, top, left + n, top + n)).convert("RGB")
        arr = np.asarray(image, dtype=np.float32) / 255.0
        arr = (arr - np.asarray(self.MEAN, dtype=np.float32)) / np.asarray(self.STD, dtype=np.float32)
        return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))

    def __repr__(self):
        return f"_Transform(n_px={self.n_px})"


def _transform(n_px: int) -> _Transform:
    return _Transform(n_px)


def _load_dtype() -> torch.dtype:
    """MFMA operand type of a model returned by clip.load: fp16, as openai/CLIP's own clip.load gives on a CUDA device (its
    build_model() converts the weights to fp16) and the mode in which this build meets the <= 1e-3 parity target with
    bit-exact arg-max; masters, residual stream, LayerNorm statistics and the head stay fp32 either way.
    CCLIP_COMPUTE_DTYPE=bf16 (or model.bfloat16()) selects bf16 operands - the training benchmark's choice (wider range)."""
    return torch.bfloat16 if os.environ.get("CCLIP_COMPUTE_DTYPE", "fp16").lower() in ("bf16", "bfloat16") else torch.float16


def load(name: str, device: Union[str, torch.device] = "cuda" if torch.cuda.is_available() else "cpu",
         jit: bool = False, download_root: str = None):
    """Returns (model, preprocess).  `name` is a model name from available_models() or a path to a
    state_dict checkpoint (as openai/CLIP accepts).  openai/CLIP downloads named models; there is no
    network here, so a named model is read from `<download_root or $CCLIP_WEIGHTS_DIR or ~/.cache/clip>/
    <name with / replaced by ->.pt` if that file exists, and otherwise gets seeded synthetic weights (a
    warning says so) - the reference overwrites them right away with its fine-tuned checkpoint
    (CLIP/train.py:108-111, CLIP/predict.py:14-16).  jit=True is not supported (TorchScript archives
    execute code on load); the reference only ever passes jit=False."""
    if jit:
        raise RuntimeError("clip.load(jit=True) is not supported by the MI355X build; use jit=False")
    sd = None
    if os.path.isfile(name):
        sd = torch.load(name, map_location="cpu", weights_only=True)
    elif name in MODELS:
        root = download_root or os.environ.get("CCLIP_WEIGHTS_DIR") or os.path.expanduser("~/.cache/clip")
        path = os.path.join(root, name.replace("/", "-") + ".pt")
        if os.path.isfile(path):
            sd = torch.load(path, map_location="cpu", weights_only=True)
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    if sd is not None:
        if "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
        model = build_model(sd, _load_dtype())
    else:
        warnings.warn(f"clip.load({name!r}): no local checkpoint and no network - using seeded synthetic weights "
                      "(seed 567); load a state_dict to get a trained model")
        model = CLIP(MODELS[name], _load_dtype())
        model.load_state_dict(init_state_dict(MODELS[name], 567))
        model.eval()
    model = model.to(device)
    return model, _transform(model.visual.input_resolution)


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
    """[SOT] + bpe(text) + [EOT], zero padded to context_length; RuntimeError when too long (truncate=False),
    exactly as openai/CLIP.  Returns an int32 CPU tensor [N, context_length] (openai/CLIP returns int32 on
    torch >= 1.8)."""
    global _tokenizer
    if isinstance(texts, str):
        texts = [texts]
    if _tokenizer is None:
        _tokenizer = SimpleTokenizer()
    sot, eot = _tokenizer.encoder["<|startoftext|>"], _tokenizer.encoder["<|endoftext|>"]
    all_tokens = [[sot] + _tokenizer.encode(t) + [eot] for t in texts]
    result = torch.zeros(len(all_tokens), context_length, dtype=torch.int32)
    for i, tokens in enumerate(all_tokens):
        if len(tokens) > context_length:
            if truncate:
                tokens = tokens[:context_length]
                tokens[-1] = eot
            else:
                raise RuntimeError(f"Input {texts[i]} is too long for context length {context_length}")
        result[i, :len(tokens)] = torch.tensor(tokens, dtype=torch.int32)
    return result
