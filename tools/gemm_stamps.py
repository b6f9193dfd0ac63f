"""Per-tile phase timeline of one cclip_gemm_bf16 launch (diagnostics build of the library with -DCCLIP_GEMM_STAMPS).

Build first (in the build container):  CCLIP_BUILD_VARIANT=stamps python construction-clip_amd/csrc/build.py
Run on the GPU box:                    python tools/gemm_stamps.py [shape ...]

Every workgroup records wall_clock64() (100 MHz) at: start, operand DMA prologue issued, first K-tile landed
(first barrier passed), K loop done, epilogue stores issued, stores drained; plus its XCC / CU id.  The report
gives the median phase durations and, per CU, how the phases of co-resident / successive tiles line up.
"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "micro", "_bin", "libcclip_hip_stamps.so")
from cclip_hip import ops
import numpy as np

SHAPES = {"out": (51200, 768, 768, "res"), "qkv": (51200, 2304, 768, "bf16"), "fc": (51200, 3072, 768, "gelu"),
          "proj": (51200, 768, 3072, "res")}


def run(name, cfg):
    M, N, K, kind = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    bias = torch.randn(N, device="cuda")
    if kind == "res":
        x = torch.randn(M, N, device="cuda"); kw = dict(out_f32=x, residual=x, bias=bias)
    elif kind == "gelu":
        kw = dict(out_bf16=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), out_pre=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), bias=bias, act=1)
    else:
        kw = dict(out_bf16=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), bias=bias)
    bm, bn = (128, 128) if cfg == 1 else (256, 128) if cfg == 2 else (256, 256)
    tiles = -(-M // bm) * -(-N // bn)
    st = torch.zeros(tiles * 9, 8, device="cuda", dtype=torch.int64)        # [tiles][8] phase records + [tiles][8 waves][8] iteration records
    for _ in range(3):
        ops.gemm_bf16(A, B, tile_config=cfg, **kw)
    _lib.lib.cclip_gemm_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_bf16(A, B, tile_config=cfg, **kw); e1.record(); torch.cuda.synchronize()
    _lib.lib.cclip_gemm_debug_set_stamps(ctypes.c_void_p(0))
    sall = st.cpu().numpy()
    s, it = sall[:tiles], sall[tiles:].reshape(tiles, 8, 8)
    hw, xcc = s[:, 0] & 0xffffffff, s[:, 0] >> 32
    cu = ((hw >> 8) & 0xff) | ((xcc & 0xf) << 8)                                      # CU_ID, SH_ID, SE_ID | XCC
    t = (s[:, 1:7] - s[:, 1].min()) / 100.0                                          # microseconds
    ph = np.diff(t, axis=1)
    names = ["issue prologue", "first tile lands", "K loop", "epilogue issue", "store drain"]
    print(f"== {name} cfg{cfg}: M={M} N={N} K={K} {kind}; {tiles} tiles on {len(np.unique(cu))} CUs; kernel {e0.elapsed_time(e1) * 1e3:.1f} us (with stamps); "
          f"last tile ends at {t[:, 5].max():.1f} us")
    for i, n in enumerate(names):
        print(f"   {n:18s} median {np.median(ph[:, i]):6.2f} us   p10 {np.percentile(ph[:, i], 10):6.2f}   p90 {np.percentile(ph[:, i], 90):6.2f}")
    print(f"   tile lifetime      median {np.median(t[:, 5] - t[:, 0]):6.2f} us")
    # chip-level: fraction of the kernel during which >= 1 tile per CU is in its K loop
    ends = t[:, 5].max()
    grid = np.linspace(0, ends, 2000)
    ink = ((t[:, 2][None, :] <= grid[:, None]) & (grid[:, None] < t[:, 3][None, :])).sum(1)
    inep = ((t[:, 3][None, :] <= grid[:, None]) & (grid[:, None] < t[:, 5][None, :])).sum(1)
    infill = ((t[:, 0][None, :] <= grid[:, None]) & (grid[:, None] < t[:, 2][None, :])).sum(1)
    print(f"   time-averaged tiles in flight: filling {infill.mean():.0f}, in K loop {ink.mean():.0f}, in epilogue/drain {inep.mean():.0f}")
    nw = 4 if cfg == 1 else 8
    d = np.diff(it[:, :nw, :5].astype(np.int64), axis=2).reshape(-1, 4)
    d = d[(d >= 0).all(1) & (it[:, :nw, 0].reshape(-1) > 0)]
    if len(d):
        print("   K iteration 5, per wave, core clocks (median | p10 | p90):  " + "   ".join(
            f"{n} {np.median(d[:, i]):.0f}|{np.percentile(d[:, i], 10):.0f}|{np.percentile(d[:, i], 90):.0f}"
            for i, n in enumerate(["wait DMA", "barrier", "issue next DMA", "LDS reads + MFMAs"])) + f"   total {np.median(d.sum(1)):.0f}")
    one = cu == cu[0]
    order = np.argsort(t[one, 0])
    print("   one CU's tiles (start, landed, kdone, issued, drained):")
    for r in t[one][order][:8]:
        print("      " + "  ".join(f"{v:7.2f}" for v in (r[0], r[2], r[3], r[4], r[5])))


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in SHAPES] or ["out", "qkv"]
    cfgs = [int(a[3:]) for a in sys.argv[1:] if a.startswith("cfg")] or [1, 2, 3]
    for n in which:
        for cfg in cfgs:
            run(n, cfg)
