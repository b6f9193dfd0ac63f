"""How much of a launch is the operands' trip through the CU's L1?  The same GEMM with the A and / or B rows ALIASED onto one
row (leading dimension 0: every row of a tile is the same 128-byte line per K step, so the L1 serves it and the L2 sees one
request) - results are garbage, times are not.  tools/micro/l1_alias_probe.py [cfg]"""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip.ops import GemmDesc  # noqa: E402
LIB = ctypes.CDLL(os.path.join(ROOT, "construction-clip_amd/cclip_hip/libcclip_hip.so"))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, M, N, K in (("img qkv", 51200, 2304, 768), ("img fc", 51200, 3072, 768), ("img proj", 51200, 768, 3072), ("sq 4096", 4096, 4096, 4096)):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = f"{name:9s}"
    for la, lb in ((K, K), (0, K), (K, 0), (0, 0)):
        d = GemmDesc()
        d.A, d.B, d.a_kcontig, d.b_kcontig, d.lda, d.ldb = A.data_ptr(), B.data_ptr(), 1, 1, la, lb
        d.M, d.N, d.K, d.alpha, d.ldc, d.split_k, d.tile_config, d.out_bf16 = M, N, K, 1.0, N, 1, cfg, o.data_ptr()
        if LIB.cclip_gemm_bf16(ctypes.byref(d), st) != 0:
            row += "      n/a"; continue
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                LIB.cclip_gemm_bf16(ctypes.byref(d), st)
            e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        row += f"  lda={'K' if la else '0'} ldb={'K' if lb else '0'}: {statistics.median(ts):6.1f}"
    print(row, flush=True)
