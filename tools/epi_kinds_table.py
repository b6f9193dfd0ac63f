"""The epilogue FORMS of the hot path (residual, GELU + saved pre-activation, activation derivative) per tile configuration, one
process, interleaved:   python tools/epi_kinds_table.py [cfg ...]   (default 3 7 8)"""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
sys.argv, argv = sys.argv[:1], sys.argv[1:]
import gemm_ab as G   # noqa: E402
cfgs = [int(a) for a in argv] or [3, 7, 8]
Mi, Mt = 51200, 78848
shapes = [("img out res", Mi, 768, 768, "res"), ("img proj res", Mi, 768, 3072, "res"), ("img fc gelu+pre", Mi, 3072, 768, "gelu"),
          ("img dproj dact", Mi, 3072, 768, "dact"), ("txt fc gelu+pre", Mt, 2048, 512, "gelu"), ("txt dproj dact", Mt, 2048, 512, "dact"),
          ("txt proj res", Mt, 512, 2048, "res"),
          ("img fc gelu (inf)", Mi, 3072, 768, "geluinf"), ("lane fc gelu (inf)", Mi // 2, 3072, 768, "geluinf"), ("img fc bare", Mi, 3072, 768, "bf16")]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, M, N, K, kind in shapes:
    ds = [G.desc(M, N, K, 1, 1, kind, c, 1) for c in cfgs]
    times = [[] for _ in ds]
    for d, _ in ds:
        for _ in range(2):
            G.LB.cclip_gemm_bf16(ctypes.byref(d), st)
    for _ in range(5):
        for i, (d, _) in enumerate(ds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                G.LB.cclip_gemm_bf16(ctypes.byref(d), st)
            e1.record(); e1.synchronize(); times[i].append(e0.elapsed_time(e1) / 4 * 1e3)
    print(f"{name:18s} " + "  ".join(f"cfg{c}: {statistics.median(t):7.1f}" for c, t in zip(cfgs, times)), flush=True)
