"""Thin torch-tensor -> raw-pointer shims over the C ABI (include/cclip_hip.h).

torch is plumbing here: it owns device memory and the stream; every FLOP happens in
libcclip_hip.so.  All functions enqueue on torch's current stream and return immediately.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ._lib import lib, check

c_void_p, c_int, c_long, c_float = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

ACT_NONE, ACT_QUICKGELU, ACT_TANH, ACT_GELU_NEW, ACT_RELU = 0, 1, 2, 3, 4
ACT_DQUICKGELU, ACT_DTANH, ACT_DGELU_NEW, ACT_DRELU = 16, 17, 18, 19


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]) -> c_void_p:
    return c_void_p(0 if t is None else t.data_ptr())


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError(f"{name}: expected cuda {dtype}, got {t.device} {t.dtype}")


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p),
        ("a_kcontig", c_int), ("b_kcontig", c_int),
        ("lda", c_long), ("ldb", c_long),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("alpha", c_float),
        ("bias", c_void_p),
        ("act", c_int),
        ("aux", c_void_p), ("ldaux", c_long),
        ("residual", c_void_p), ("ldr", c_long),
        ("out_f32", c_void_p), ("out_bf16", c_void_p), ("out_pre_bf16", c_void_p), ("ldc", c_long),
        ("split_k", c_int), ("split_ws", c_void_p),
    ]


def gemm_bf16(A: torch.Tensor, B: torch.Tensor, *, a_kcontig: bool = True, b_kcontig: bool = True,
              alpha: float = 1.0, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
              aux: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
              out_f32: Optional[torch.Tensor] = None, out_bf16: Optional[torch.Tensor] = None,
              out_pre: Optional[torch.Tensor] = None, split_k: int = 1,
              split_ws: Optional[torch.Tensor] = None, M: Optional[int] = None) -> None:
    """C[m][n] = epi(alpha * sum_k A(m,k) B(n,k)); see cclip_gemm_bf16 in include/cclip_hip.h.
    A: [M,K] (a_kcontig) or [K,M]; B: [N,K] (b_kcontig) or [K,N]; 2-D, inner stride 1."""
    _req(A, torch.bfloat16, "A"); _req(B, torch.bfloat16, "B")
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    Mx, K = (A.shape[0], A.shape[1]) if a_kcontig else (A.shape[1], A.shape[0])
    N, Kb = (B.shape[0], B.shape[1]) if b_kcontig else (B.shape[1], B.shape[0])
    if M is None:
        M = Mx
    assert K == Kb, (A.shape, B.shape, a_kcontig, b_kcontig)
    outs = [t for t in (out_f32, out_bf16, out_pre) if t is not None]
    assert outs, "no output"
    ldc = outs[0].stride(0)
    for t in outs:
        assert t.stride(0) == ldc and t.stride(1) == 1 and t.shape[0] >= M and t.shape[1] == N
    d = GemmDesc()
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.a_kcontig, d.b_kcontig = int(a_kcontig), int(b_kcontig)
    d.lda, d.ldb = A.stride(0), B.stride(0)
    d.M, d.N, d.K = M, N, K
    d.alpha = alpha
    d.bias = 0 if bias is None else bias.data_ptr()
    d.act = act
    d.aux = 0 if aux is None else aux.data_ptr()
    d.ldaux = 0 if aux is None else aux.stride(0)
    d.residual = 0 if residual is None else residual.data_ptr()
    d.ldr = 0 if residual is None else residual.stride(0)
    d.out_f32 = 0 if out_f32 is None else out_f32.data_ptr()
    d.out_bf16 = 0 if out_bf16 is None else out_bf16.data_ptr()
    d.out_pre_bf16 = 0 if out_pre is None else out_pre.data_ptr()
    d.ldc = ldc
    d.split_k = split_k
    d.split_ws = 0 if split_ws is None else split_ws.data_ptr()
    if bias is not None:
        _req(bias, torch.float32, "bias")
    if residual is not None:
        _req(residual, torch.float32, "residual")
    if out_f32 is not None:
        _req(out_f32, torch.float32, "out_f32")
    if split_k > 1:
        assert split_ws is not None and split_ws.numel() >= split_k * M * N and split_ws.dtype == torch.float32
    check(lib.cclip_gemm_bf16(ctypes.byref(d), _stream()), "cclip_gemm_bf16")
