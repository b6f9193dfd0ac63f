"""Ablation timing of tile configuration 5 (gemm_bf16_cfg5.hip built with -DCCLIP_CFG5_ABLATE=mask; results are numerically
wrong by construction): 1 = no in-loop DMA, 2 = no fragment refresh (LDS reads), 4 = no per-step wait + barrier,
8 = no epilogue.  python tools/micro/cfg5_ablate.py   (libraries: tools/micro/_bin/libcclip_abl_<mask>.so, each the normal object
list with gemm_bf16_cfg5.o rebuilt under the mask).

Round-1 result, 1x MI355X, us (mask: 0 | 1 | 2 | 3 | 4 | 8 | 15):
  img qkv 51200x2304x768:  232 | 194 | 210 | 169 | 199 | 164 | 120
  img fc  51200x3072x768:  278 | 237 | 266 | 216 | 266 | 205 | 151
  4096^3:                  105 |  91 |  95 |  76 | 105 | 100 |  70
i.e. at K = 768 the un-overlapped epilogue (a synchronized 33 MB store burst per round of tiles) costs 30 % of the kernel, the
operand DMA 16 %, the fragment reads 10 %, the barrier 14 %; with all four removed the MFMAs alone run at 1.5 PFLOP/s."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip.ops import GemmDesc
BIN = os.path.join(ROOT, "tools", "micro", "_bin")
masks = [0, 1, 2, 3, 4, 8, 15]
libs = {m: ctypes.CDLL(os.path.join(BIN, f"libcclip_abl_{m}.so")) for m in masks}
for (name, M, N, K) in (("img qkv", 51200, 2304, 768), ("img fc", 51200, 3072, 768), ("square", 4096, 4096, 4096)):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16(); B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    bias = torch.randn(N, device="cuda"); o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    d = GemmDesc()
    d.A, d.B, d.a_kcontig, d.b_kcontig, d.lda, d.ldb = A.data_ptr(), B.data_ptr(), 1, 1, K, K
    d.M, d.N, d.K, d.alpha, d.ldc, d.split_k, d.tile_config = M, N, K, 1.0, N, 1, 5
    d.bias, d.out_bf16 = bias.data_ptr(), o.data_ptr()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cells = []
    for m in masks:
        lib = libs[m]
        for _ in range(3): assert lib.cclip_gemm_bf16(ctypes.byref(d), st) == 0
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): lib.cclip_gemm_bf16(ctypes.byref(d), st)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        cells.append(f"mask {m:2d}: {best * 1e3:6.1f} us")
    print(f"{name:8s} " + " | ".join(cells), flush=True)
