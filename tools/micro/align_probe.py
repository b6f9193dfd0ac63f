import sys, torch
sys.path[:0] = ["/root/repo", "/root/repo/construction-clip_amd"]
from cclip_hip import ops
from cclip_hip.stack import wgrad_candidates
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
D = 768
for M in (20480, 16923, 16896, 16928):
    L = 4
    x = torch.randn(L, 2, M, D, device="cuda")
    st = torch.randn(L, 4, M, device="cuda").abs() + 0.5
    dy = torch.randn(M, D, device="cuda").bfloat16()
    dx = torch.zeros(M, D, device="cuda"); dxb = torch.zeros(M, D, device="cuda", dtype=torch.bfloat16)
    g = torch.ones(D, device="cuda"); dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    ws = torch.empty(ops.layernorm_bwd_ws_floats(M, D), device="cuda")
    t_ln = timeit(lambda: ops.layernorm_bwd(dy, x[1, 1], g, st[1, 2], st[1, 3], rows=M, dx_res=dx, dx_out=dx, dx_out_bf16=dxb, dgamma=dg, dbeta=db, accumulate=False, ws=ws))
    dyw = torch.randn(L, M, 2304, device="cuda").bfloat16(); xn = torch.randn(L, M, 6 * D + 2 * 3072, device="cuda").bfloat16()
    gw = torch.zeros(768, 2304, device="cuda")
    sc = torch.empty(64 * 2304 * 800, device="cuda")
    t_wg = timeit(lambda: ops.gemm_bf16(xn[1][:, 0:D], dyw[1], a_kcontig=False, b_kcontig=False, out_f32=gw, split_candidates=wgrad_candidates(768, 2304, M), scratch=lambda n: sc))
    print(f"M={M}: ln_bwd {t_ln:7.1f} us   wgrad qkv (Conv1D layout) {t_wg:7.1f} us", flush=True)
