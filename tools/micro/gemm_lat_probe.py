"""Is the K loop of configurations 8 / 10 waiting for HBM?  The same GEMM with lda = 0 (every A row aliases row 0: the whole A
operand is L2-resident) against the real row stride, with and without the epilogue (CCLIP_GEMM_DBG=1).

    python tools/micro/gemm_lat_probe.py
"""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip.ops import GemmDesc  # noqa: E402
LIB = ctypes.CDLL(os.path.join(ROOT, "construction-clip_amd/cclip_hip/libcclip_hip.so"))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, M, N, K in [("img qkv", 51200, 2304, 768), ("img fc", 51200, 3072, 768), ("img out", 51200, 768, 768), ("img proj", 51200, 768, 3072)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = f"{name:9s}"
    for cfg in (8, 10):
        for lda in (K, 0):
            d = GemmDesc()
            d.A, d.B, d.a_kcontig, d.b_kcontig, d.lda, d.ldb = A.data_ptr(), B.data_ptr(), 1, 1, lda, K
            d.M, d.N, d.K, d.alpha, d.ldc, d.split_k, d.tile_config, d.out_bf16 = M, N, K, 1.0, N, 1, cfg, o.data_ptr()
            ts = []
            for _ in range(3):
                LIB.cclip_gemm_bf16(ctypes.byref(d), st)
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    LIB.cclip_gemm_bf16(ctypes.byref(d), st)
                e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 5 * 1e3)
            row += f"  cfg{cfg} lda={lda:4d}: {statistics.median(ts):7.1f}"
    print(row, flush=True)
