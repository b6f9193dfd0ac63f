"""Probe: tile-count quantisation.  A GEMM whose tile count is 7.03 x the 256 CU slots pays 8 rounds.  Compare one launch against
a ROW-PARTITIONED pair: rows [0, M1) = whole rounds of the big tile, rows [M1, M) with a smaller tile.  Measurement tool only."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops  # noqa: E402
from cclip_hip.ops import GemmDesc  # noqa: E402
LIB = ctypes.CDLL(os.path.join(ROOT, "construction-clip_amd/cclip_hip/libcclip_hip.so"))
TILE = {1: (128, 128, 512), 2: (256, 128, 256), 3: (256, 256, 256), 5: (192, 256, 256), 7: (256, 256, 256)}


def mk(A, B, out, M, row0, cfg):
    d = GemmDesc()
    N, K = B.shape
    d.A, d.B, d.a_kcontig, d.b_kcontig, d.lda, d.ldb = A.data_ptr() + row0 * K * 2, B.data_ptr(), 1, 1, K, K
    d.M, d.N, d.K, d.alpha, d.ldc, d.split_k, d.tile_config = M, N, K, 1.0, N, 1, cfg
    d.out_bf16 = out.data_ptr() + row0 * N * 2
    return d


def run(ds):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for d in ds:
        assert LIB.cclip_gemm_bf16(ctypes.byref(d), st) == 0


def ev(ds, iters=10):
    for _ in range(3):
        run(ds)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            run(ds)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3


Mi, Mt = 51200, 78848
for name, M, N, K in [("img qkv", Mi, 2304, 768), ("img out", Mi, 768, 768), ("img fc", Mi, 3072, 768), ("img proj", Mi, 768, 3072),
                      ("img dqkv", Mi, 768, 2304), ("txt qkv", Mt, 1536, 512), ("txt out", Mt, 512, 512), ("txt fc", Mt, 2048, 512),
                      ("txt proj", Mt, 512, 2048)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16(); B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    single = {c: ev([mk(A, B, out, M, 0, c)]) for c in (1, 2, 3, 5, 7)}
    bc = min(single, key=single.get)
    res = []
    for ca in (2, 3, 5, 7):
        bm, bn, P = TILE[ca]
        tn = -(-N // bn); T = -(-M // bm) * tn
        full = T // P
        if full == 0:
            continue
        m1 = (full * P // tn) * bm
        if m1 >= M:
            continue
        for cb in (1, 2, 3):
            t = ev([mk(A, B, out, m1, 0, ca), mk(A, B, out, M - m1, m1, cb)])
            res.append((t, ca, cb, m1))
    res.sort()
    fl = 2.0 * M * N * K
    print(f"{name:9s} single best cfg{bc} {single[bc]:7.1f} us ({fl / single[bc] * 1e-6 / 2500:.3f}) | " +
          " ".join(f"[{ca}+{cb} rows {m1}: {t:6.1f} ({(t / single[bc] - 1) * 100:+.1f}%)]" for t, ca, cb, m1 in res[:3]), flush=True)
