"""Device-side `preprocess`: the resize / crop / normalise part of openai/CLIP's `_transform(n_px)` on
the GPU, bit-identical to the PIL + numpy pipeline of clip.clip._Transform (which stays the host fallback).

Host side (this file): PIL's resampling windows and integer coefficients restated from its published algorithm
(libImaging/Resample.c: precompute_coeffs + normalize_coeffs_8bpc; bicubic a = -0.5; 8-bit fixed point with 22 fraction
bits) - a few hundred numbers per image size, cached.  Device side: csrc/preprocess.hip (two launches per image).
JPEG decoding stays on the host (PIL), as in the reference's DataLoader workers.
"""
from __future__ import annotations

import functools
import math
from typing import Sequence, Tuple

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=256)
def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """(bounds int32 [out, 2] = (first input index, count), coefficients int32 [out, ksize], ksize) of PIL's BICUBIC resampler."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)            # C cast: truncation
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.array([_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)], dtype=np.float64)
        ww = 0.0
        for v in w:                                   # sequential double sum, as the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        kq = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
        kk[xx, :xmax] = kq
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resized_size(w: int, h: int, n: int) -> Tuple[int, int]:
    """torchvision Resize(n) on the shorter side (what clip.clip._Transform does)."""
    if (w <= h and w == n) or (h <= w and h == n):
        return w, h
    if w < h:
        return n, int(n * h / w)
    return int(n * w / h), n


def plan(w: int, h: int, n: int):
    """Everything the two launches need for a w x h image: crop offsets, the horizontal windows of the n surviving columns,
    the vertical windows of the n surviving rows and the input-row range those touch."""
    nw, nh = resized_size(w, h, n)
    left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
    bh, kh, ksh = resample_coeffs(w, nw)
    bv, kv, ksv = resample_coeffs(h, nh)
    bh, kh = np.ascontiguousarray(bh[left:left + n]), np.ascontiguousarray(kh[left:left + n])
    bv, kv = np.ascontiguousarray(bv[top:top + n]), np.ascontiguousarray(kv[top:top + n])
    row0 = int(bv[:, 0].min())
    row1 = int((bv[:, 0] + bv[:, 1]).max())
    return dict(bh=bh, kh=kh, ksh=ksh, bv=bv, kv=kv, ksv=ksv, row0=row0, rows=row1 - row0)


def reference_numpy(img_u8: np.ndarray, n: int) -> np.ndarray:
    """The same two integer passes in numpy (host check of the coefficient restatement against PIL; tests only)."""
    h, w, _ = img_u8.shape
    p = plan(w, h, n)
    src = img_u8[p["row0"]:p["row0"] + p["rows"]].astype(np.int64)
    tmp = np.zeros((p["rows"], n, 3), dtype=np.uint8)
    for xx in range(n):
        x0, cnt = p["bh"][xx]
        s = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + cnt, :], p["kh"][xx, :cnt].astype(np.int64), axes=([1], [0]))
        tmp[:, xx, :] = np.clip(s >> PRECISION_BITS, 0, 255)
    out = np.zeros((n, n, 3), dtype=np.uint8)
    t64 = tmp.astype(np.int64)
    for yy in range(n):
        y0, cnt = p["bv"][yy]
        s = (1 << (PRECISION_BITS - 1)) + np.tensordot(p["kv"][yy, :cnt].astype(np.int64), t64[y0 - p["row0"]:y0 - p["row0"] + cnt], axes=([0], [0]))
        out[yy] = np.clip(s >> PRECISION_BITS, 0, 255)
    return out


class DevicePreprocess:
    """Callable with the semantics of clip's `preprocess` but producing a CUDA tensor: PIL image (mode RGB) or uint8 HWC
    array / tensor -> fp32 [3, n, n] on `device`.  Non-RGB PIL images go through the host pipeline (the reference converts
    to RGB only AFTER resizing, which an RGB-first device path would not reproduce bit for bit)."""

    def __init__(self, n_px: int, device="cuda"):
        self.n_px, self.device = n_px, torch.device(device)
        self._plans = {}

    def _device_plan(self, w, h):
        key = (w, h)
        if key not in self._plans:
            p = plan(w, h, self.n_px)
            dev = {k: torch.from_numpy(p[k]).to(self.device) for k in ("bh", "kh", "bv", "kv")}
            dev.update(ksh=p["ksh"], ksv=p["ksv"], row0=p["row0"], rows=p["rows"])
            self._plans[key] = dev
        return self._plans[key]

    def __call__(self, image) -> torch.Tensor:
        from cclip_hip import ops
        if hasattr(image, "mode"):                      # PIL
            if image.mode != "RGB":
                from .clip import _Transform
                return _Transform(self.n_px)(image).to(self.device)
            arr = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy())
        else:
            arr = torch.as_tensor(image)
        assert arr.dtype == torch.uint8 and arr.dim() == 3 and arr.shape[2] == 3, "expected a uint8 HWC RGB image"
        h, w, _ = arr.shape
        p = self._device_plan(w, h)
        src = arr.to(self.device, non_blocking=True).contiguous()
        n = self.n_px
        tmp = torch.empty(p["rows"], n, 3, device=self.device, dtype=torch.uint8)
        out = torch.empty(3, n, n, device=self.device, dtype=torch.float32)
        ops.resample_h_u8(src[p["row0"]:p["row0"] + p["rows"]], p["bh"], p["kh"], p["ksh"], tmp)
        ops.resample_v_norm(tmp, p["row0"], p["bv"], p["kv"], p["ksv"], MEAN, STD, out)
        return out

    def batch(self, images: Sequence) -> torch.Tensor:
        return torch.stack([self(im) for im in images])
