"""GPU parity of every non-GEMM HIP kernel against a plain torch fp32 reference of the same op
(the model-level comparison against the CPU oracle lives in test_clip_parity_gpu.py).
Tolerances are written per test: fp32 kernels 1e-5..1e-4 relative; kernels with bf16 I/O are
limited by one bf16 rounding (2^-8) of their inputs/outputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def ops():
    from cclip_hip import ops as o
    return o


def close(name, got, ref, rtol, atol=0.0):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs().max().clamp_min(1e-12)
    if not bool((err <= bound).all()) or not bool(torch.isfinite(got).all()):
        bad = (err > bound) | ~torch.isfinite(got)
        idx = bad.nonzero()[:6].tolist()
        pytest.fail(f"{name}: {int(bad.sum())}/{bad.numel()} bad; max err {err[torch.isfinite(err)].max().item() if torch.isfinite(err).any() else float('nan'):.4g} "
                    f"bound {bound.item():.4g}; first {idx}; got {[got[tuple(i)].item() for i in idx[:3]]} ref {[ref[tuple(i)].item() for i in idx[:3]]}")


def G(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


@pytest.mark.parametrize("rows,D", [(7, 128), (1000, 512), (513, 768), (64, 1024)])
def test_layernorm_fwd_bwd(rows, D):
    o = ops()
    g = G(rows + D)
    x = torch.randn(rows, D, device="cuda", generator=g) * 2 + 0.5
    gamma = 1 + 0.1 * torch.randn(D, device="cuda", generator=g)
    beta = 0.1 * torch.randn(D, device="cuda", generator=g)
    dy = torch.randn(rows, D, device="cuda", generator=g)
    res = torch.randn(rows, D, device="cuda", generator=g)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    y.backward(dy)
    out_b = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    out_f = torch.empty(rows, D, device="cuda")
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    o.layernorm_fwd(x, gamma, beta, rows=rows, out_bf16=out_b, out_f32=out_f, mean=mean, rstd=rstd)
    close("ln f32", out_f, y, 1e-5, 1e-5)
    close("ln bf16", out_b, y, 2 ** -7)
    close("mean", mean, x.mean(1), 1e-5, 1e-6)
    dx = torch.empty(rows, D, device="cuda"); dxb = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    dg = torch.ones(D, device="cuda"); db = torch.ones(D, device="cuda")
    ws = torch.empty(o.layernorm_bwd_ws_floats(rows, D), device="cuda")
    o.layernorm_bwd(dy, x, gamma, mean, rstd, rows=rows, dx_res=res, dx_out=dx, dx_out_bf16=dxb, dgamma=dg, dbeta=db,
                    accumulate=True, ws=ws)
    close("ln dx", dx, xr.grad + res, 1e-4, 1e-5)
    close("ln dx bf16", dxb, xr.grad + res, 2 ** -7)
    close("ln dgamma", dg, gr.grad + 1, 1e-4, 1e-4)
    close("ln dbeta", db, br.grad + 1, 1e-4, 1e-4)
    # bf16 upstream gradient, no accumulate
    o.layernorm_bwd(dy.bfloat16(), x, gamma, mean, rstd, rows=rows, dx_out=dx, dgamma=dg, dbeta=db, ws=ws)
    xr.grad = None; gr.grad = None
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5).backward(dy.bfloat16().float())
    close("ln dx (bf16 dy)", dx, xr.grad, 1e-4, 1e-5)
    close("ln dgamma (bf16 dy)", dg, gr.grad, 1e-4, 1e-4)


def test_layernorm_row_index():
    o = ops()
    g = G(3)
    x = torch.randn(40, 256, device="cuda", generator=g)
    idx = torch.tensor([0, 10, 39, 5], device="cuda", dtype=torch.int32)
    gamma = torch.rand(256, device="cuda", generator=g) + 0.5
    beta = torch.randn(256, device="cuda", generator=g)
    out = torch.empty(4, 256, device="cuda")
    mean = torch.empty(4, device="cuda"); rstd = torch.empty(4, device="cuda")
    o.layernorm_fwd(x, gamma, beta, rows=4, row_index=idx, out_f32=out, mean=mean, rstd=rstd)
    ref = torch.nn.functional.layer_norm(x[idx.long()], (256,), gamma, beta, 1e-5)
    close("ln gather", out, ref, 1e-5, 1e-5)
    dy = torch.randn(4, 256, device="cuda", generator=g)
    dx = torch.zeros(40, 256, device="cuda")
    o.layernorm_bwd(dy, x, gamma, mean, rstd, rows=4, row_index=idx, dx_out=dx)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr[idx.long()], (256,), gamma, beta, 1e-5).backward(dy)
    close("ln scatter dx", dx, xr.grad, 1e-4, 1e-5)


def _attn_scores(q, k, causal, keep):
    s = (q @ k.transpose(-1, -2)) * 0.125
    T = q.shape[2]
    if causal:
        s = s + torch.full((T, T), float("-inf"), device=q.device).triu_(1)
    if keep is not None:
        s = s.masked_fill(keep[:, None, None, :] == 0, float("-inf"))
    return s


def _attn_ref(q, k, v, causal, keep):
    # q,k,v: [B,H,T,64] fp32
    return torch.softmax(_attn_scores(q, k, causal, keep), dim=-1) @ v


@pytest.mark.parametrize("B,T,H,causal,pad", [(3, 50, 12, False, False), (2, 77, 8, True, False), (2, 5, 2, False, False),
                                              (3, 80, 12, True, True), (1, 128, 3, True, False), (2, 33, 2, False, True),
                                              (2, 17, 1, True, False), (1, 100, 2, False, False),
                                              (2, 129, 2, False, False), (1, 197, 3, True, False), (2, 257, 2, False, True),
                                              (1, 577, 4, False, False), (1, 300, 1, True, True)])
def test_attention_fwd_bwd(B, T, H, causal, pad):
    o = ops()
    g = G(B * 1000 + T)
    D = H * 64
    qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).bfloat16()
    keep = None
    if pad:
        keep = torch.ones(B, T, device="cuda")
        keep[0, T - 3:] = 0
        keep[-1, T // 2:] = 0
    out = torch.full((B * T, D), float("nan"), device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device="cuda")
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o.attention_fwd(q, k, v, out, B=B, T=T, H=H, causal=causal, key_keep=keep, lse=lse)
    f = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)
    ref = _attn_ref(f[0], f[1], f[2], causal, keep)            # [B,H,T,64]
    ref2 = ref.permute(0, 2, 1, 3).reshape(B * T, D)
    close("attn fwd", out, ref2, 1.5e-2)
    close("attn lse", lse, torch.logsumexp(_attn_scores(f[0], f[1], causal, keep), dim=-1), 1e-3, 2e-3)
    dout = torch.randn(B * T, D, device="cuda", generator=g).bfloat16()
    dqkv = torch.full((B * T, 3 * D), float("nan"), device="cuda", dtype=torch.bfloat16)
    o.attention_bwd(q, k, v, out, lse, dout, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B=B, T=T, H=H, causal=causal,
                    key_keep=keep)
    ref.backward(dout.float().view(B, T, H, 64).permute(0, 2, 1, 3))
    gref = f.grad.permute(1, 3, 0, 2, 4).reshape(B * T, 3 * D)
    close("attn dq", dqkv[:, :D], gref[:, :D], 3e-2)
    close("attn dk", dqkv[:, D:2 * D], gref[:, D:2 * D], 3e-2)
    close("attn dv", dqkv[:, 2 * D:], gref[:, 2 * D:], 3e-2)


@pytest.mark.parametrize("B,T,H,dh,dt", [(3, 40, 8, 96, torch.bfloat16), (2, 10, 8, 16, torch.bfloat16), (2, 48, 2, 128, torch.float16), (2, 64, 4, 64, torch.bfloat16),
                                         (1, 1, 1, 8, torch.bfloat16), (2, 33, 3, 40, torch.float16)])
def test_attention_small_fwd_bwd(B, T, H, dh, dt):
    """Generic-head_dim attention (TransformerMapper: 8 heads x 96, 40 tokens) against fp32 torch on the same 16-bit inputs."""
    o = ops()
    g = G(B * 1000 + T + dh)
    D = H * dh
    qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).to(dt)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    out = torch.full((B * T, D), float("nan"), device="cuda", dtype=dt)
    lse = torch.empty(B, H, T, device="cuda")
    o.attention_small_fwd(q, k, v, out, B=B, T=T, H=H, head_dim=dh, lse=lse)
    f = qkv.float().view(B, T, 3, H, dh).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)
    sc = (f[0] @ f[1].transpose(-1, -2)) * dh ** -0.5
    ref = torch.softmax(sc, dim=-1) @ f[2]
    tol = 1.5e-2 if dt == torch.bfloat16 else 2e-3
    close("attn_small fwd", out, ref.permute(0, 2, 1, 3).reshape(B * T, D), tol)
    close("attn_small lse", lse, torch.logsumexp(sc, dim=-1), 1e-4, 1e-4)
    dout = torch.randn(B * T, D, device="cuda", generator=g).to(dt)
    dqkv = torch.full((B * T, 3 * D), float("nan"), device="cuda", dtype=dt)
    o.attention_small_bwd(q, k, v, out, lse, dout, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B=B, T=T, H=H, head_dim=dh)
    ref.backward(dout.float().view(B, T, H, dh).permute(0, 2, 1, 3))
    gref = f.grad.permute(1, 3, 0, 2, 4).reshape(B * T, 3 * D)
    for i, nm in enumerate(("dq", "dk", "dv")):
        close("attn_small " + nm, dqkv[:, i * D:(i + 1) * D], gref[:, i * D:(i + 1) * D], 2 * tol)


def test_attention_small_rejects_unsupported_shapes():
    from cclip_hip._lib import CclipError
    o = ops()
    x = torch.zeros(65 * 2, 3 * 64, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(65 * 2, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(CclipError):
        o.attention_small_fwd(x[:, :64], x[:, 64:128], x[:, 128:], out, B=2, T=65, H=1, head_dim=64)     # T > 64
    x = torch.zeros(8, 3 * 12, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(8, 12, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(CclipError):
        o.attention_small_fwd(x[:, :12], x[:, 12:24], x[:, 24:], out, B=1, T=8, H=1, head_dim=12)          # head_dim % 8


@pytest.mark.parametrize("M,N,K,ta,tb", [(9, 9, 512, False, False), (1024, 1024, 512, False, False), (100, 70, 33, True, False),
                                         (65, 130, 768, False, True), (512, 768, 100, True, True)])
def test_gemm_f32(M, N, K, ta, tb):
    o = ops()
    g = G(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g)
    B = torch.randn((K, N) if tb else (N, K), device="cuda", generator=g)
    Av = A.t() if ta else A
    Bv = B.t() if tb else B
    C = torch.randn(M, N, device="cuda", generator=g)
    ref = 0.5 * (Av.double() @ Bv.double().t()) + 2.0 * C.double()
    o.gemm_f32(Av, Bv, C, alpha=0.5, beta=2.0)
    close("gemm_f32", C, ref.float(), 2e-6 * math.sqrt(K) + 1e-6)


@pytest.mark.parametrize("P,grid", [(14, 3), (14, 24), (16, 2), (6, 4)])
def test_patchify_any_patch_size(P, grid):
    """ViT-L/14: 3*14*14 = 588 is not a multiple of 8 -> rows zero-padded to 592; bit-exact against unfold."""
    o = ops()
    R, B = P * grid, 2
    img = torch.randn(B, 3, R, R, device="cuda", generator=G(P * 100 + grid))
    KP = 3 * P * P
    KPAD = (KP + 7) // 8 * 8
    out = torch.full((B * (grid * grid + 1), KPAD), float("nan"), device="cuda", dtype=torch.bfloat16)
    o.patchify(img, out, P)
    ref = torch.nn.functional.unfold(img, kernel_size=P, stride=P).transpose(1, 2)          # [B, grid*grid, 3*P*P] in (c, ky, kx) order
    got = out.view(B, grid * grid + 1, KPAD)
    assert torch.equal(got[:, 1:, :KP], ref.bfloat16())
    assert (got[:, 0] == 0).all() and (got[:, :, KP:] == 0).all()


def test_patchify_and_embeds():
    o = ops()
    g = G(11)
    B, R, P, W = 3, 64, 32, 128
    Gd = R // P
    T = Gd * Gd + 1
    img = torch.randn(B, 3, R, R, device="cuda", generator=g)
    out = torch.full((B * T, 3 * P * P), float("nan"), device="cuda", dtype=torch.bfloat16)
    o.patchify(img, out, P)
    ref = torch.nn.functional.unfold(img, kernel_size=P, stride=P).transpose(1, 2)      # [B, G*G, 3*P*P]
    ref = torch.cat([torch.zeros(B, 1, 3 * P * P, device="cuda"), ref], 1).reshape(B * T, -1)
    close("patchify", out, ref.bfloat16().float(), 0.0, 0.0)
    po = torch.randn(B * T, W, device="cuda", generator=g)
    cls = torch.randn(W, device="cuda", generator=g); pos = torch.randn(T, W, device="cuda", generator=g)
    gamma = torch.rand(W, device="cuda", generator=g) + 0.5; beta = torch.randn(W, device="cuda", generator=g)
    x = torch.empty(B * T, W, device="cuda"); x0 = torch.empty(B * T, W, device="cuda")
    mean = torch.empty(B * T, device="cuda"); rstd = torch.empty(B * T, device="cuda")
    o.vit_embed_ln(po, cls, pos, gamma, beta, x, rows=B * T, T=T, x0=x0, mean=mean, rstd=rstd)
    r0 = po.view(B, T, W) + pos
    r0[:, 0] += cls
    close("x0", x0, r0.reshape(B * T, W), 1e-6, 1e-6)
    close("ln_pre", x, torch.nn.functional.layer_norm(r0, (W,), gamma, beta, 1e-5).reshape(B * T, W), 1e-5, 1e-5)
    # text embed + scatter
    V, L = 300, 16
    text = torch.randint(0, V, (B * L,), device="cuda", generator=g, dtype=torch.int32)
    emb = torch.randn(V, W, device="cuda", generator=g); posl = torch.randn(L, W, device="cuda", generator=g)
    xt = torch.empty(B * L, W, device="cuda")
    o.text_embed(text, emb, posl, xt, rows=B * L, L=L)
    close("text_embed", xt, (emb[text.long()].view(B, L, W) + posl).reshape(B * L, W), 1e-6, 1e-6)
    dx = torch.randn(B * L, W, device="cuda", generator=g)
    demb = torch.zeros(V, W, device="cuda")
    o.embed_scatter_add(text, dx, demb, rows=B * L)
    ref = torch.zeros(V, W, device="cuda").index_add_(0, text.long(), dx)
    close("scatter", demb, ref, 1e-5, 1e-5)


def test_embedding_gradient_is_deterministic_and_handles_long_runs():
    """csrc/embed.hip embed_segsum: the embedding-table gradient summed WITHOUT atomics - rows sorted by token id, one owner
    per id, runs longer than 64 rows through ordered partials.  A hot id (thousands of rows: the SOT / EOT tokens of a batch),
    dropped rows (`keep`), the GPT-2 row mapping (seq_stride / seq_off), accumulation into a non-zero table; two launches agree
    bit for bit, and with the atomics kernel to rounding."""
    o = ops()
    g = G(11)
    V, W, B, L = 500, 512, 64, 77
    text = torch.randint(1, V, (B * L,), device="cuda", generator=g, dtype=torch.int32)
    text.view(B, L)[:, 0] = V - 2                                       # every sequence starts with the same id: runs of B rows
    text.view(B, L)[:, 5::7] = 3                                        # a hot id with hundreds of rows
    dx = torch.randn(B * L, W, device="cuda", generator=g)
    base = torch.randn(V, W, device="cuda", generator=g)
    keep = torch.rand(B * L, device="cuda", generator=g) > 0.25
    outs = []
    for _ in range(2):
        demb = base.clone()
        o.embed_scatter_add(text, dx, demb, rows=B * L, keep=keep)
        outs.append(demb)
    assert torch.equal(outs[0], outs[1])
    ref = base.double().index_add_(0, text[keep].long(), dx[keep].double())
    close("segsum", outs[0], ref.float(), 1e-5, 1e-4)
    o.SCATTER_DETERMINISTIC = False
    try:
        d2 = base.clone()
        o.embed_scatter_add(text, dx * keep[:, None], d2, rows=B * L)
    finally:
        o.SCATTER_DETERMINISTIC = True
    close("segsum vs atomics", outs[0], d2, 1e-5, 1e-4)
    # GPT-2 mapping: ids [n_seq, Lt] sit at rows b*S + P + j of the gradient
    S, P, Lt = 30, 6, 20
    ids = torch.randint(0, V, (B * Lt,), device="cuda", generator=g, dtype=torch.int32)
    dxs = torch.randn(B * S, W, device="cuda", generator=g)
    d3 = torch.zeros(V, W, device="cuda")
    o.embed_scatter_add(ids, dxs, d3, rows=B * Lt, L=Lt, seq_stride=S, seq_off=P)
    ref3 = torch.zeros(V, W, device="cuda", dtype=torch.float64).index_add_(0, ids.long(), dxs.view(B, S, W)[:, P:P + Lt].reshape(-1, W).double())
    close("segsum gpt2 rows", d3, ref3.float(), 1e-5, 1e-4)


@pytest.mark.parametrize("n,V,drop", [(64 * 77, 500, 0.25), (1000, 49408, 0.0), (5000, 7, 0.5), (65, 3, 0.0), (1, 10, 0.0)])
def test_embedding_gradient_tables(n, V, drop):
    """cclip_embed_tables (csrc/embed.hip): chunk / run bookkeeping of the sorted id list from binary searches - against a plain
    host walk of the same list: every run of a kept id is cut into chunks of <= 64 rows, a chunk start carries its end, a run
    start its chunk count, chunk slots are unique, consecutive inside a run and below V + n/64 + 1; dropped rows form no chunk."""
    o = ops()
    g = G(n + V)
    text = torch.randint(0, V, (n,), device="cuda", generator=g, dtype=torch.int32)
    if n > 100:
        text[::3] = V // 2                                              # a hot id: runs of many chunks
    keep = (torch.rand(n, device="cuda", generator=g) >= drop) if drop else None
    order, st, cend, cidx, rlen = o.embed_scatter_tables(text, V, rows=n, keep=keep)
    tok = text.cpu().tolist(); kp = keep.cpu().tolist() if keep is not None else [True] * n
    key = [t if k else V for t, k in zip(tok, kp)]
    want_order = sorted(range(n), key=lambda r: (key[r], r))
    assert order.cpu().tolist() == want_order
    stl = [key[r] for r in want_order]
    assert st.cpu().tolist() == stl
    ce, ci, rl = cend.cpu().tolist(), cidx.cpu().tolist(), rlen.cpu().tolist()
    p, slots = 0, set()
    while p < n:
        q = p
        while q < n and stl[q] == stl[p]:
            q += 1
        if stl[p] >= V:                                                  # dropped rows
            assert all(ce[i] == 0 and rl[i] == 0 for i in range(p, q))
        else:
            k = (q - p + 63) // 64
            assert rl[p] == k and all(rl[i] == 0 for i in range(p + 1, q))
            for j in range(k):
                c = p + 64 * j
                assert ce[c] == min(c + 64, q)
                assert ci[c] == ci[p] + j and ci[c] not in slots and 0 <= ci[c] < V + n // 64 + 1
                slots.add(ci[c])
            assert all(ce[i] == 0 for i in range(p, q) if (i - p) % 64)
        p = q


@pytest.mark.parametrize("R,C,bf", [(1000, 768, True), (50, 2304, False), (4097, 104, True), (300, 100, False)])
def test_colsum(R, C, bf):
    o = ops()
    g = G(R)
    x = torch.randn(R, C, device="cuda", generator=g)
    if bf:
        x = x.bfloat16()
    out = torch.ones(C, device="cuda")
    ws = torch.empty(o.colsum_ws_floats(R, C), device="cuda")
    o.colsum(x, out, ws, R=R, C=C, ld=C, accumulate=True)
    close("colsum", out, 1 + x.float().sum(0), 1e-5, 1e-4)


def test_l2norm_and_xent():
    o = ops()
    g = G(21)
    x = torch.randn(37, 512, device="cuda", generator=g)
    y = torch.empty_like(x); inv = torch.empty(37, device="cuda")
    o.l2norm_fwd(x, y, inv)
    xr = x.clone().requires_grad_(True)
    yr = xr / xr.norm(dim=1, keepdim=True)
    close("l2norm", y, yr, 1e-6, 1e-7)
    dy = torch.randn_like(x)
    yr.backward(dy)
    dx = torch.empty_like(x)
    o.l2norm_bwd(dy, y, inv, dx)
    close("l2norm bwd", dx, xr.grad, 1e-5, 1e-7)
    for R, C in [(9, 9), (1, 2), (300, 1000), (64, 21128)]:
        lg = torch.randn(R, C, device="cuda", generator=g) * 3
        labels = torch.randint(0, C, (R,), device="cuda", generator=g, dtype=torch.int32)
        if R > 8:
            labels[3] = 0
        lr = lg.clone().requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(lr, labels.long(), ignore_index=0, reduction="sum")
        loss.backward()
        lrow = torch.empty(R, device="cuda"); pred = torch.empty(R, device="cuda", dtype=torch.int32)
        d = torch.empty_like(lg); db = torch.empty(R, C, device="cuda", dtype=torch.bfloat16)
        o.xent_rows(lg, labels, loss_row=lrow, pred=pred, dlogits=d, grad_scale=0.5, ignore_index=0)
        o.xent_rows(lg, labels, dlogits=db, grad_scale=0.5, ignore_index=0)
        close(f"xent loss {R}x{C}", lrow.sum(), loss.detach(), 1e-5, 1e-5)
        close(f"xent grad {R}x{C}", d, 0.5 * lr.grad, 1e-5, 1e-7)
        close(f"xent grad bf16 {R}x{C}", db, 0.5 * lr.grad, 2 ** -7, 1e-7)
        assert torch.equal(pred.long(), lg.argmax(1)), "argmax mismatch"
        lg2 = lg.clone()
        o.xent_rows(lg2, labels, dlogits=lg2, grad_scale=0.5, ignore_index=0)   # in place
        close("xent inplace", lg2, d, 0.0, 0.0)


def test_adamw_matches_hf_form():
    o = ops()
    g = G(5)
    n = 4096 + 64
    p = torch.randn(n, device="cuda", generator=g); gr = torch.randn(n, device="cuda", generator=g)
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    pr, mr, vr = p.clone().double(), m.clone().double(), v.clone().double()
    sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-6, 0.01
    for step in range(1, 4):
        o.adamw_step(p, gr, m, v, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step, bf16_shadow=sh)
        mr = b1 * mr + (1 - b1) * gr.double()
        vr = b2 * vr + (1 - b2) * gr.double() ** 2
        ss = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        pr = pr - ss * mr / (vr.sqrt() + eps)
        pr = pr - lr * wd * pr
    close("adamw p", p, pr.float(), 1e-5, 1e-6)
    close("adamw shadow", sh, pr.float(), 2 ** -8)
    # torch.optim.AdamW form
    p2 = torch.randn(n, device="cuda", generator=g)
    tp = torch.nn.Parameter(p2.clone()); tp.grad = gr.clone()
    opt = torch.optim.AdamW([tp], lr=lr, betas=(b1, b2), eps=1e-8, weight_decay=wd)
    m.zero_(); v.zero_()
    for step in range(1, 3):
        opt.step()
        o.adamw_step(p2, gr, m, v, lr=lr, beta1=b1, beta2=b2, eps=1e-8, weight_decay=wd, step=step, mode=1)
    close("adamw torch form", p2, tp.detach(), 1e-5, 1e-6)
    src = torch.randn(1024, device="cuda", generator=g); dst = torch.empty(1024, device="cuda", dtype=torch.bfloat16)
    o.cast_f32_to_bf16(src, dst)
    assert torch.equal(dst, src.bfloat16())


def test_transpose16_batched_and_arena_transposed_shadows():
    """csrc/transpose16.hip: batched 16-bit transposes (full 64x64 tiles and ragged edges), and the arena's transposed
    weight shadows following the masters through an optimiser step."""
    from cclip_hip import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(768, 2304), (64, 64), (200, 72), (1, 9), (3072, 768)]
    mats = [torch.randn(r, c, device="cuda", generator=g).bfloat16() for r, c in shapes]
    src = torch.cat([m.flatten() for m in mats] + [torch.zeros(8, device="cuda", dtype=torch.bfloat16)])
    dst = torch.full_like(src, float("nan"))
    offs, table, tiles = 0, [], 0
    for r, c in shapes:
        table.append([offs, offs, r, c]); offs += r * c
        tiles = max(tiles, ((r + 63) // 64) * ((c + 63) // 64))
    ops.transpose16_batched(src, dst, torch.tensor(table, dtype=torch.int64, device="cuda"), tiles)
    for (off, _, r, c), m in zip(table, mats):
        assert torch.equal(dst[off:off + r * c].view(c, r), m.t().contiguous()), (r, c)
    # arena: transposed shadows == shadows^T, before and after a fused AdamW step
    import clip
    from clip import optim as coptim
    from clip.weights import MODELS, init_state_dict, synthetic_images, synthetic_text
    geo = MODELS["test-small"]
    model = clip.build_model(init_state_dict(geo, 3)).cuda().train()
    img, txt = synthetic_images(6, geo, 4).cuda(), synthetic_text(6, geo, 5).cuda()
    opt = coptim.AdamW(model, lr=1e-2)
    for _ in range(2):
        opt.zero_grad()
        li, lt = model(img, txt)
        lab = torch.arange(6, device="cuda")
        ((torch.nn.functional.cross_entropy(li, lab) + torch.nn.functional.cross_entropy(lt, lab)) / 2).backward()
        ar = model.arena
        assert ar.t, "transposed shadows are on by default"
        for n, tv in ar.t.items():
            assert torch.equal(tv, ar.b[n].t()), n                       # what backward just used
        opt.step()
    ar.refresh_transposed()
    for n, tv in ar.t.items():
        assert torch.equal(tv, ar.b[n].t()) and torch.equal(ar.b[n].float(), ar.params[n].data.to(ar.b[n].dtype).float()), n
