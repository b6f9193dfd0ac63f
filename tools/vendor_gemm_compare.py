"""Context for the roofline numbers: the vendor library (hipBLASLt / rocBLAS through torch.matmul) on the step's GEMM shapes, next
to cclip_gemm_bf16's best tile configuration.  torch is used here ONLY as a yardstick; nothing in the product calls it for math."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops
ops.AUTOTUNE = False


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


Mi, Mt = 51200, 78848
for name, M, N, K in [("img qkv", Mi, 2304, 768), ("img out", Mi, 768, 768), ("img fc", Mi, 3072, 768), ("img proj", Mi, 768, 3072),
                      ("txt qkv", Mt, 1536, 512), ("txt fc", Mt, 2048, 512), ("txt proj", Mt, 512, 2048), ("square 4096", 4096, 4096, 4096),
                      ("L/14 qkv", 147712, 3072, 1024), ("L/14 proj", 147712, 1024, 4096)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tv = timeit(lambda: torch.matmul(A, W.t(), out=out))
    tm = min(timeit(lambda c=c: ops.gemm_bf16(A, W, out_bf16=out, tile_config=c)) for c in (1, 2, 3))
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d} | vendor (torch.matmul) {tv * 1e3:7.1f} us {fl / tv / 1e9:6.0f} TF | cclip best cfg {tm * 1e3:7.1f} us {fl / tm / 1e9:6.0f} TF | {tv / tm:5.2f}x", flush=True)
