"""Micro-benchmark of cclip_gemm_bf16 on the step's real shapes (interleaved A/B of tile configs in one process)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops

def bench(M, N, K, akc, bkc, cfg, kind, iters=20, split=1):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn((M, K) if akc else (K, M), device="cuda", generator=g).bfloat16()
    B = torch.randn((N, K) if bkc else (K, N), device="cuda", generator=g).bfloat16()
    kw = {}
    if kind == "bf16": kw = dict(out_bf16=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), bias=torch.randn(N, device="cuda"))
    elif kind == "res": x = torch.randn(M, N, device="cuda"); kw = dict(out_f32=x, residual=x, bias=torch.randn(N, device="cuda"))
    elif kind == "gelu": kw = dict(out_bf16=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), out_pre=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), bias=torch.randn(N, device="cuda"), act=1)
    elif kind == "f32": kw = dict(out_f32=torch.empty(M, N, device="cuda"))
    elif kind == "split":
        kw = dict(out_f32=torch.empty(M, N, device="cuda"), split_k=split, split_ws=torch.empty(split * M * N, device="cuda"))
    for _ in range(3):
        ops.gemm_bf16(A, B, a_kcontig=akc, b_kcontig=bkc, tile_config=cfg, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_bf16(A, B, a_kcontig=akc, b_kcontig=bkc, tile_config=cfg, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

if __name__ == "__main__":
    Mi, Mt = 51200, 78848
    shapes = [("img qkv", Mi, 2304, 768, 1, 1, "bf16"), ("img out", Mi, 768, 768, 1, 1, "res"), ("img fc", Mi, 3072, 768, 1, 1, "gelu"),
              ("img proj", Mi, 768, 3072, 1, 1, "res"), ("txt qkv", Mt, 1536, 512, 1, 1, "bf16"), ("txt fc", Mt, 2048, 512, 1, 1, "gelu"),
              ("txt proj", Mt, 512, 2048, 1, 1, "res"),
              ("img dgrad fc", Mi, 768, 3072, 1, 0, "bf16"), ("img dgrad proj", Mi, 3072, 768, 1, 0, "bf16"),
              ("img wgrad qkv", 2304, 768, Mi, 0, 0, "split"), ("img wgrad proj", 768, 3072, Mi, 0, 0, "split"),
              ("square 4096", 4096, 4096, 4096, 1, 1, "bf16"), ("square 8192 f32out", 8192, 8192, 8192, 1, 1, "f32")]
    only = sys.argv[1:] 
    for name, M, N, K, akc, bkc, kind in shapes:
        if only and not any(o in name for o in only): continue
        row = []
        for cfg in (1, 2, 3):
            sp = 8 if kind == "split" else 1
            ms, tf = bench(M, N, K, bool(akc), bool(bkc), cfg, kind, split=sp)
            row.append(f"cfg{cfg}: {ms:7.3f} ms {tf:7.1f} TF")
        print(f"{name:18s} M={M:6d} N={N:5d} K={K:6d} | " + " | ".join(row), flush=True)
