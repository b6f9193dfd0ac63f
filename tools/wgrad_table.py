"""Weight-gradient GEMM timing per (configuration, split) on the image tower's shapes, interleaved in one process.

    python tools/wgrad_table.py
"""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip.ops import GemmDesc  # noqa: E402
LIB = ctypes.CDLL(os.path.join(ROOT, "construction-clip_amd/cclip_hip/libcclip_hip.so"))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
PEAK = 2500.0
for name, M, N, K in [("img wgrad qkv", 2304, 768, 51200), ("img wgrad out", 768, 768, 51200), ("img wgrad fc", 3072, 768, 51200),
                      ("img wgrad proj", 768, 3072, 51200), ("txt wgrad qkv", 1536, 512, 78848), ("txt wgrad fc", 2048, 512, 78848)]:
    A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda"); cs = torch.empty(M, device="cuda")
    ws = torch.empty(32 * (M * N + max(M, N)), device="cuda")
    rows = []
    for cfg in (2, 3, 11):
        tiles = -(-M // 256) * -(-N // (128 if cfg == 2 else 256))
        sps = {min(32, max(1, 256 // tiles)), min(32, max(1, 512 // tiles)), min(32, max(1, 768 // tiles)), min(32, max(1, 1024 // tiles))}   # (ws holds 32 slabs)
        if cfg == 11:
            sps |= {8, 16}
        for sp in sorted(sps):
            d = GemmDesc()
            d.A, d.B, d.a_kcontig, d.b_kcontig, d.lda, d.ldb = A.data_ptr(), B.data_ptr(), 0, 0, M, N
            d.M, d.N, d.K, d.alpha, d.ldc, d.split_k, d.tile_config = M, N, K, 1.0, N, sp, cfg
            d.out_f32, d.split_ws = out.data_ptr(), ws.data_ptr()
            if cfg != 3:
                d.colsum_out = cs.data_ptr()
            if LIB.cclip_gemm_bf16(ctypes.byref(d), st) != 0:
                continue
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    LIB.cclip_gemm_bf16(ctypes.byref(d), st)
                e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 4 * 1e3)
            rows.append((statistics.median(ts), cfg, sp))
    fl = 2.0 * M * N * K
    if "-v" in sys.argv:
        print("   " + "  ".join(f"cfg{c}/s{sp}: {t:6.1f}" for t, c, sp in sorted(rows, key=lambda r: (r[1], r[2]))))
    best = {c: min((r for r in rows if r[1] == c), default=None) for c in (2, 3, 11)}
    print(f"{name:15s} " + "  ".join(f"cfg{c}: {b[0]:7.1f} us (split {b[2]:2d}) {fl / b[0] * 1e-6 / PEAK:5.3f}" for c, b in best.items() if b), flush=True)
