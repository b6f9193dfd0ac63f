"""LayerNorm forward / backward timing on the train step's row counts (HIP events, median of 7 x 10 launches) with the bytes
each launch moves and the HBM rate that is:   python tools/ln_time.py"""
import os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops  # noqa: E402

OLD = os.path.join(ROOT, "tools/micro/ab_old/libln_old.so")      # optional: a build of the previous kernels to A/B against


def med(fn):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return statistics.median(ts)


libs = [("new", ops.lib)]
if os.path.isfile(OLD):
    import ctypes
    libs.append(("old", ctypes.CDLL(OLD)))
for (tag, L), (name, R, D) in [(l, s) for s in (("img", 51200, 768), ("txt dense", 78848, 512), ("txt packed", 40311, 512)) for l in libs]:
    ops.lib = L
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(R, D, device="cuda", generator=g)
    gamma, beta = torch.rand(D, device="cuda", generator=g) + 0.5, torch.randn(D, device="cuda", generator=g)
    y = torch.empty(R, D, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    tf = med(lambda: ops.layernorm_fwd(x, gamma, beta, rows=R, out_bf16=y, mean=mean, rstd=rstd))
    bf = R * D * 6
    dy = torch.randn(R, D, device="cuda", generator=g).bfloat16()
    dres = torch.randn(R, D, device="cuda", generator=g)
    dx, dxb = torch.empty(R, D, device="cuda"), torch.empty(R, D, device="cuda", dtype=torch.bfloat16)
    dgm, dbt = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ws = torch.empty(ops.layernorm_bwd_ws_floats(R, D), device="cuda")
    tb = med(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, rows=R, dx_res=dres, dx_out=dx, dx_out_bf16=dxb, dgamma=dgm, dbeta=dbt, ws=ws))
    bb = R * D * 16
    print(f"{tag} {name:11s} R={R:6d} D={D:4d}  fwd {tf:6.1f} us = {bf / tf * 1e-6:5.2f} TB/s   bwd {tb:6.1f} us = {bb / tb * 1e-6:5.2f} TB/s", flush=True)
