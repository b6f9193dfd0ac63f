"""Per-configuration timing table of cclip_gemm_bf16 on the forward-layout shapes of the hot path (bare GEMM, bf16 out), all
configurations interleaved in ONE process (rounds of: every configuration once), median and min over the rounds.

    python tools/gemm_cfg_table.py [cfg ...]         (default: 3 7 8 10)
"""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops  # noqa: E402
from cclip_hip.ops import GemmDesc  # noqa: E402

LIB = ctypes.CDLL(os.path.join(ROOT, "construction-clip_amd/cclip_hip/libcclip_hip.so"))
PEAK = 2500.0
SHAPES = [("img qkv", 51200, 2304, 768), ("img out", 51200, 768, 768), ("img fc", 51200, 3072, 768), ("img proj", 51200, 768, 3072),
          ("lane qkv", 25600, 2304, 768), ("lane fc", 25600, 3072, 768),
          ("txt qkv", 78848, 1536, 512), ("txt fc", 78848, 2048, 512), ("txt proj", 78848, 512, 2048),
          ("sq 4096", 4096, 4096, 4096), ("sq 8192", 8192, 8192, 8192)]


def main():
    cfgs = [int(a) for a in sys.argv[1:]] or [3, 7, 8, 10]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    print(f"{'shape':10s} " + " ".join(f"{'cfg' + str(c) + ' us(med/min)':>20s} {'frac':>6s}" for c in cfgs) + f" {'blas us':>9s} {'frac':>6s}")
    for name, M, N, K in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
        B = torch.randn((N, K), device="cuda", generator=g).bfloat16()
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        descs = []
        for cfg in cfgs:
            d = GemmDesc()
            d.A, d.B, d.a_kcontig, d.b_kcontig, d.lda, d.ldb = A.data_ptr(), B.data_ptr(), 1, 1, A.stride(0), B.stride(0)
            d.M, d.N, d.K, d.alpha, d.ldc, d.split_k, d.tile_config = M, N, K, 1.0, N, 1, cfg
            d.out_bf16 = o.data_ptr()
            descs.append(d)
        fns = [(lambda d=d: LIB.cclip_gemm_bf16(ctypes.byref(d), st)) for d in descs]
        fns.append(lambda: torch.matmul(A, B.t(), out=o))
        times = [[] for _ in fns]
        iters = 5
        for f in fns:
            for _ in range(3):
                f()
        for _ in range(7):
            for i, f in enumerate(fns):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    f()
                e1.record(); e1.synchronize()
                times[i].append(e0.elapsed_time(e1) / iters * 1e3)
        fl = 2.0 * M * N * K
        row = f"{name:10s} "
        for t in times[:-1]:
            med, mn = statistics.median(t), min(t)
            row += f"{med:11.1f}/{mn:8.1f} {fl / med * 1e-6 / PEAK:6.3f} "
        tb = statistics.median(times[-1])
        row += f"{tb:9.1f} {fl / tb * 1e-6 / PEAK:6.3f}"
        print(row, flush=True)


if __name__ == "__main__":
    main()
