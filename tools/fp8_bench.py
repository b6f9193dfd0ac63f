"""fp8 (e4m3, block-scaled MFMA) GEMM vs the bf16 tile kernels on the ViT-L/14@336px projection shapes (bs 256: M = 147712)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


M = 256 * 577
for name, N, K, act in (("qkv", 3072, 1024, 0), ("fc", 4096, 1024, 1), ("out", 1024, 1024, 0), ("proj", 1024, 4096, 0)):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    A8, sa = torch.empty(M, K, device="cuda", dtype=torch.uint8), torch.empty(M, device="cuda")
    W8, sw = torch.empty(N, K, device="cuda", dtype=torch.uint8), torch.empty(N, device="cuda")
    ops.quantize_rows_fp8(W, W8, sw)
    ops.quantize_rows_fp8(A, A8, sa)
    tb = min(timeit(lambda c=c: ops.gemm_bf16(A, W, bias=bias, act=act, out_bf16=out, tile_config=c)) for c in (1, 2, 3))
    tq = timeit(lambda: ops.quantize_rows_fp8(A, A8, sa))
    t8 = timeit(lambda: ops.gemm_fp8(A8, sa, W8, sw, out, bias=bias, act=act))
    fl = 2.0 * M * N * K
    print(f"{name:5s} M={M} N={N} K={K}: bf16 best {tb * 1e3:7.1f} us ({fl / tb / 1e9:6.0f} TF) | fp8 gemm {t8 * 1e3:7.1f} us ({fl / t8 / 1e9:6.0f} TF) "
          f"+ quantise rows {tq * 1e3:6.1f} us ({M * K * 3 / tq / 1e9:5.2f} TB/s) -> {(t8 + tq) / tb:5.2f}x of bf16 with the separate quantise pass, {t8 / tb:5.2f}x without", flush=True)
