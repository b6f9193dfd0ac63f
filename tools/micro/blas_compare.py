"""Headroom check: the step's GEMM shapes through this library (best tile configuration, bias epilogue) against the
vendor BLAS torch dispatches bf16 matmuls to (hipBLASLt / rocBLAS), plain C = A @ B^T, in ONE process.  Measurement tool only -
nothing in the product path calls the vendor library for these shapes.

    python tools/micro/blas_compare.py
"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd"), os.path.join(ROOT, "tools")]
from cclip_hip import ops  # noqa: E402
from cclip_hip.ops import GemmDesc  # noqa: E402

LIB = ctypes.CDLL(os.path.join(ROOT, "construction-clip_amd/cclip_hip/libcclip_hip.so"))
PEAK = 2500.0


def ev_time(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3                                # us


def ours(M, N, K, akc, bkc, split):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn((M, K) if akc else (K, M), device="cuda", generator=g).bfloat16()
    B = torch.randn((N, K) if bkc else (K, N), device="cuda", generator=g).bfloat16()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    best, bestcfg = 1e9, None
    ncol = (N + 255) // 256
    cfgs = (1, 2, 3, 4, 5, 7, 8, 10) + tuple(c + 256 * g for c in (3, 8) for g in (3, 4, 6) if ncol > g) if not split else (2, 3, 11)
    for cfg in cfgs:
        for sk in ((1,) if not split else (4, 6, 7, 8, 9, 12)):
            d = GemmDesc()
            d.A, d.B, d.a_kcontig, d.b_kcontig, d.lda, d.ldb = A.data_ptr(), B.data_ptr(), akc, bkc, A.stride(0), B.stride(0)
            d.M, d.N, d.K, d.alpha, d.ldc, d.split_k, d.tile_config = M, N, K, 1.0, N, sk, cfg
            if split:
                o = torch.empty(M, N, device="cuda"); ws = torch.empty(sk * (M * N + max(M, N)), device="cuda")
                d.out_f32, d.split_ws = o.data_ptr(), ws.data_ptr()
            else:
                o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); d.out_bf16 = o.data_ptr()
            if LIB.cclip_gemm_bf16(ctypes.byref(d), st) != 0:
                continue
            t = ev_time(lambda: LIB.cclip_gemm_bf16(ctypes.byref(d), st))
            if t < best:
                best, bestcfg = t, (cfg, sk)
    return best, bestcfg, A, B


if __name__ == "__main__":
    Mi, Mt = 51200, 78848
    shapes = [("img qkv", Mi, 2304, 768, 1, 1, 0), ("img out", Mi, 768, 768, 1, 1, 0), ("img fc", Mi, 3072, 768, 1, 1, 0),
              ("img proj", Mi, 768, 3072, 1, 1, 0), ("txt qkv", Mt, 1536, 512, 1, 1, 0), ("txt fc", Mt, 2048, 512, 1, 1, 0),
              ("txt proj", Mt, 512, 2048, 1, 1, 0),
              ("img wgrad qkv", 2304, 768, Mi, 0, 0, 1), ("img wgrad fc", 3072, 768, Mi, 0, 0, 1),
              ("square 4096", 4096, 4096, 4096, 1, 1, 0), ("square 8192", 8192, 8192, 8192, 1, 1, 0)]
    print("(cfg = (tile_config, split_k); tile_config = configuration + 256 * column-group width of the tile order)")
    print(f"{'shape':16s} {'ours us':>9s} {'cfg':>8s} {'TF/s':>7s} {'frac':>6s} | {'blas us':>9s} {'TF/s':>7s} {'frac':>6s} | ours/blas")
    for name, M, N, K, akc, bkc, split in shapes:
        t, cfg, A, B = ours(M, N, K, akc, bkc, split)
        a = A if akc else A.t()
        b = (B if bkc else B.t()).t()                 # [K, N] view, the layout the operands already have
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        tb = ev_time(lambda: torch.matmul(a, b, out=out))
        fl = 2.0 * M * N * K
        print(f"{name:16s} {t:9.1f} {str(cfg):>8s} {fl / t * 1e-6:7.0f} {fl / t * 1e-6 / PEAK:6.3f} | {tb:9.1f} {fl / tb * 1e-6:7.0f} {fl / tb * 1e-6 / PEAK:6.3f} | {t / tb:5.2f}",
              flush=True)
