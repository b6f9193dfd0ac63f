"""GPU parity of the bf16 MFMA GEMM (cclip_gemm_bf16) against a plain torch fp32 matmul of the
same bf16-rounded operands.  Tolerance: fp32 accumulation of exact bf16 products differs from
torch's only by summation order -> 2e-3 relative to the row's |a|.|b| bound is generous; outputs
rounded to bf16 get one extra 2^-8 relative rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from cclip_hip import ops
    return ops


def _ref_act(v, act, aux):
    import math
    if act == 0:
        return v
    if act == 1:
        return v * torch.sigmoid(1.702 * v)
    if act == 2:
        return torch.tanh(v)
    if act == 3:
        return 0.5 * v * (1 + torch.tanh(math.sqrt(2 / math.pi) * (v + 0.044715 * v ** 3)))
    if act == 4:
        return torch.relu(v)
    a = aux.float()
    if act == 16:
        s = torch.sigmoid(1.702 * a)
        return v * s * (1 + 1.702 * a * (1 - s))
    if act == 17:
        return v * (1 - a * a)
    if act == 18:
        a = a.clone().requires_grad_(True)
        y = 0.5 * a * (1 + torch.tanh(math.sqrt(2 / math.pi) * (a + 0.044715 * a ** 3)))
        (g,) = torch.autograd.grad(y.sum(), a)
        return v * g
    if act == 19:
        return torch.where(a > 0, v, torch.zeros_like(v))
    raise ValueError(act)


def _report(name, got, ref, tol):
    err = (got.float() - ref).abs()
    scale = ref.abs().max().clamp_min(1e-6)
    bad = err > tol * scale
    if bad.any():
        idx = bad.nonzero()[:8].tolist()
        rows = bad.any(dim=1).nonzero().flatten()[:16].tolist()
        cols = bad.any(dim=0).nonzero().flatten()[:16].tolist()
        pytest.fail(f"{name}: {int(bad.sum())}/{bad.numel()} bad, max err {err.max().item():.4g} (scale {scale.item():.4g}); "
                    f"first idx {idx}; bad rows {rows}; bad cols {cols}; got {got[idx[0][0], idx[0][1]].item()} ref {ref[idx[0][0], idx[0][1]].item()}")


LAYOUTS = [(True, True), (True, False), (False, False)]
SHAPES = [(128, 128, 64), (256, 384, 192), (450, 768, 768), (264, 200, 72), (1000, 2304, 768), (72, 136, 3072)]


@pytest.mark.parametrize("cfg", [1, 2, 3, 5, 7])
@pytest.mark.parametrize("akc,bkc", LAYOUTS)
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_plain(M, N, K, akc, bkc, cfg):
    ops = _ops()
    if not akc and M % 8:
        pytest.skip("a_kcontig=0 needs M % 8 == 0")
    if cfg == 7 and not (akc and bkc):
        pytest.skip("configuration 7 (rotated K loop) is built for the forward layout")
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K) if akc else (K, M), device="cuda", generator=g).bfloat16()
    B = torch.randn((N, K) if bkc else (K, N), device="cuda", generator=g).bfloat16()
    Am = A.float() if akc else A.float().t()
    Bm = B.float() if bkc else B.float().t()
    ref = Am @ Bm.t()
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16(A, B, a_kcontig=akc, b_kcontig=bkc, out_f32=out, tile_config=cfg)
    torch.cuda.synchronize()
    _report(f"plain {M}x{N}x{K} {akc}{bkc} cfg{cfg}", out, ref, 2e-3)


@pytest.mark.parametrize("act,akc,bkc", [(1, True, True), (2, True, True), (4, True, True), (18, True, True),
                                         (3, True, False), (16, True, False), (17, True, False), (19, True, False)])
@pytest.mark.parametrize("cfg", [0, 2, 3, 5, 7])
def test_gemm_epilogues(act, akc, bkc, cfg):
    ops = _ops()
    if cfg == 7 and not (akc and bkc):
        pytest.skip("configuration 7 (rotated K loop) is built for the forward layout")
    M, N, K = 300, 264, 256
    g = torch.Generator(device="cuda").manual_seed(act)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.1).bfloat16()
    B = torch.randn((N, K) if bkc else (K, N), device="cuda", generator=g).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    aux = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    if act == 17:
        aux = torch.tanh(aux.float()).bfloat16()
    Bm = B.float() if bkc else B.float().t()
    pre = 0.5 * (A.float() @ Bm.t()) + bias
    ref = _ref_act(pre, act, aux) + res
    out = torch.full((M, N), float("nan"), device="cuda")
    outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    outp = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_bf16(A, B, a_kcontig=akc, b_kcontig=bkc, alpha=0.5, bias=bias, act=act, aux=aux if act >= 16 else None,
                  residual=res, out_f32=out, out_bf16=outb, out_pre=outp, tile_config=cfg)
    torch.cuda.synchronize()
    _report(f"act{act} f32", out, ref, 2e-3)
    _report(f"act{act} bf16", outb, ref, 1e-2)
    _report(f"act{act} pre", outp, pre, 1e-2)


@pytest.mark.parametrize("cfg", [8])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 768), (450, 768, 768), (1000, 2304, 768), (264, 200, 192),
                                   (300, 264, 3072), (2048, 512, 512)])
def test_gemm_cfg8_plain_and_bit_exact(M, N, K, cfg):
    """Configuration 8 / 9 (four 128x128 waves, hand-scheduled asm K loop): forward layout, K % 64 == 0.  It sums each output's
    products in the same order as the 256x256 configuration 3 (k ascending, 32 per MFMA) -> bit-identical results."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    B = torch.randn((N, K), device="cuda", generator=g).bfloat16()
    ref = A.float() @ B.float().t()
    out = torch.full((M, N), float("nan"), device="cuda")
    out3 = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, out_f32=out, tile_config=cfg)
    ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, out_f32=out3, tile_config=3)
    torch.cuda.synchronize()
    _report(f"plain {M}x{N}x{K} cfg{cfg}", out, ref, 2e-3)
    assert torch.equal(out, out3), "configuration 8 must reproduce configuration 3 bit for bit"


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 768), (450, 768, 768), (1000, 2304, 768), (264, 200, 384),
                                   (300, 264, 3072), (2048, 512, 512), (70000, 512, 256), (25600, 768, 768)])
def test_gemm_cfg10_persistent_ring_bit_exact(M, N, K):
    """Configuration 10 (persistent 4-wave ring kernel; K % 128 == 0): every work-group walks several tiles when there are more
    tiles than CUs (the last two shapes), the DMA stream crossing tile boundaries.  Same summation order as configuration 3."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    B = torch.randn((N, K), device="cuda", generator=g).bfloat16()
    out = torch.full((M, N), float("nan"), device="cuda")
    out3 = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, out_f32=out, tile_config=10)
    ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, out_f32=out3, tile_config=3)
    torch.cuda.synchronize()
    if M * N <= 1 << 22:
        _report(f"plain {M}x{N}x{K} cfg10", out, A.float() @ B.float().t(), 2e-3)
    assert torch.equal(out, out3), "configuration 10 must reproduce configuration 3 bit for bit"
    # run it again into the same buffer behind another launch: no state may leak from launch to launch
    out.fill_(float("nan"))
    ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, out_f32=out, tile_config=10)
    assert torch.equal(out, out3)


@pytest.mark.parametrize("cfg", [8, 10])
@pytest.mark.parametrize("act", [0, 1, 16])
def test_gemm_cfg8_epilogues(act, cfg):
    ops = _ops()
    for (M, N, K) in ((300, 264, 256), (512, 512, 256)):       # ragged edges (general epilogue) and interior tiles (fast paths)
        g = torch.Generator(device="cuda").manual_seed(act)
        A = (torch.randn(M, K, device="cuda", generator=g) * 0.1).bfloat16()
        B = torch.randn((N, K), device="cuda", generator=g).bfloat16()
        bias = torch.randn(N, device="cuda", generator=g)
        res = torch.randn(M, N, device="cuda", generator=g)
        aux = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        pre = 0.5 * (A.float() @ B.float().t()) + bias
        ref = _ref_act(pre, act, aux) + res
        out = torch.full((M, N), float("nan"), device="cuda")
        outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        outp = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, alpha=0.5, bias=bias, act=act, aux=aux if act >= 16 else None,
                      residual=res, out_f32=out, out_bf16=outb, out_pre=outp, tile_config=cfg)
        torch.cuda.synchronize()
        _report(f"act{act} f32", out, ref, 2e-3)
        _report(f"act{act} bf16", outb, ref, 1e-2)
        _report(f"act{act} pre", outp, pre, 1e-2)
        # the fast-path forms the hot path issues: 16-bit out (+ pre-activation), fp32 residual in place, activation derivative -
        # configurations 8 / 10 stage them through LDS for row-contiguous stores: same arithmetic, so bit-identical to configuration 3
        if act in (0, 1):
            got = []
            for c in (cfg, 3):
                ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
                op = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if act else None
                ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, alpha=0.5, bias=bias, act=act, out_bf16=ob, out_pre=op, tile_config=c)
                got.append((ob, op))
            _report(f"fast16 act{act}", got[0][0], _ref_act(pre, act, None), 1e-2)
            assert torch.equal(got[0][0], got[1][0])
            if act:
                _report(f"fast16 pre act{act}", got[0][1], pre, 1e-2)
                assert torch.equal(got[0][1], got[1][1])
                ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)          # inference form: activation, no saved pre-activation
                ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, alpha=0.5, bias=bias, act=act, out_bf16=ob, tile_config=cfg)
                assert torch.equal(ob, got[1][0])
        if act == 0:
            x, x3 = res.clone(), res.clone()
            ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, alpha=0.5, bias=bias, residual=x, out_f32=x, tile_config=cfg)
            ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, alpha=0.5, bias=bias, residual=x3, out_f32=x3, tile_config=3)
            _report("fast residual", x, pre + res, 2e-3)
            assert torch.equal(x, x3)
        if act == 16:
            ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            ob3 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, alpha=0.5, bias=bias, act=act, aux=aux, out_bf16=ob, tile_config=cfg)
            ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, alpha=0.5, bias=bias, act=act, aux=aux, out_bf16=ob3, tile_config=3)
            _report("fast dact", ob, _ref_act(pre, act, aux), 1e-2)
            assert torch.equal(ob, ob3)


@pytest.mark.parametrize("M,N,K", [(1100, 2304, 256), (600, 1792, 128), (300, 3072, 192)])
@pytest.mark.parametrize("cfg", [2, 3, 8])
@pytest.mark.parametrize("g", [3, 4, 6])
def test_gemm_grouped_tile_order_is_bit_identical(M, N, K, cfg, g):
    """tile_config bits 8..15: the column tiles taken in groups of g (csrc/gemm_bf16_impl.h tile_coords).  Only the ORDER in which
    tiles are dealt to work-groups changes - every tile is computed once, by the same arithmetic: outputs equal the row-major run
    bit for bit, including group widths that do not divide the column-tile count, ragged edge tiles, and a residual epilogue."""
    ops = _ops()
    g0 = torch.Generator(device="cuda").manual_seed(M + N + K + g)
    A = (torch.randn(M, K, device="cuda", generator=g0) * 0.5).bfloat16()
    B = (torch.randn(N, K, device="cuda", generator=g0) * 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g0)
    res = torch.randn(M, N, device="cuda", generator=g0)
    outs = []
    for tc in (cfg, cfg + 256 * g):
        ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, bias=bias, act=1, out_bf16=ob, tile_config=tc)
        x = res.clone()
        ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, bias=bias, residual=x, out_f32=x, tile_config=tc)
        outs.append((ob, x))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    _report("grouped order", outs[1][1], A.float() @ B.float().t() + bias + res, 2e-3)
    with pytest.raises(RuntimeError):                     # configurations that walk tiles their own way refuse an order
        ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, out_bf16=outs[0][0], tile_config=10 + 256 * g)


def test_gemm_cfg8_refuses_what_it_does_not_implement():
    ops = _ops()
    A = torch.randn(256, 96, device="cuda").bfloat16()
    B = torch.randn(256, 96, device="cuda").bfloat16()
    out = torch.empty(256, 256, device="cuda")
    with pytest.raises(RuntimeError):        # K % 64 != 0
        ops.gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, out_f32=out, tile_config=8)
    Bt = torch.randn(128, 256, device="cuda").bfloat16()
    A2 = torch.randn(256, 128, device="cuda").bfloat16()
    with pytest.raises(RuntimeError):        # K-strided B
        ops.gemm_bf16(A2, Bt, a_kcontig=True, b_kcontig=False, out_f32=out, tile_config=8)


@pytest.mark.parametrize("M,N,K,split", [(256, 256, 256, 1), (768, 768, 1024, 2), (2304, 768, 6400, 5), (304, 264, 1280, 3), (3072, 768, 4096, 4)])
def test_gemm_cfg11_weight_gradient_layout(M, N, K, split):
    """Configuration 11 (four 128x128 waves, hand-scheduled asm K loop with transposing LDS reads): the weight-gradient layout -
    both operands K-strided, split-K slabs, the fused bias gradient (row sums of A).  Same summation order as the 8-wave 256x128
    configuration 2 at the same split -> bit-identical, bias gradient included."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((K, M), device="cuda", generator=g).bfloat16()       # dY  [tokens, out]
    B = torch.randn((K, N), device="cuda", generator=g).bfloat16()       # X   [tokens, in]
    ref = A.float().t() @ B.float()
    res = {}
    for cfg in (11, 2):           # (2 = the 256x128 8-wave kernel the training step used for its weight gradients)
        out = torch.full((M, N), float("nan"), device="cuda")
        cs = torch.full((M,), float("nan"), device="cuda")
        ws = torch.empty(split * (M * N + max(M, N)), device="cuda") if split > 1 else None
        ops.gemm_bf16(A, B, a_kcontig=False, b_kcontig=False, out_f32=out, tile_config=cfg, split_k=split, split_ws=ws, colsum_out=cs)
        res[cfg] = (out, cs)
    torch.cuda.synchronize()
    _report(f"wgrad {M}x{N}x{K} cfg11", res[11][0], ref, 2e-3)
    assert (res[11][1] - A.float().sum(0)).abs().max() < 2e-3 * A.float().abs().sum(0).max()
    assert torch.equal(res[11][0], res[2][0])
    assert torch.equal(res[11][1], res[2][1])
    # accumulate into an existing gradient (residual = out) and into an existing bias gradient
    out = ref.clone(); cs = torch.ones(M, device="cuda")
    ws = torch.empty(split * (M * N + max(M, N)), device="cuda") if split > 1 else None
    ops.gemm_bf16(A, B, a_kcontig=False, b_kcontig=False, residual=out, out_f32=out, tile_config=11, split_k=split, split_ws=ws,
                  colsum_out=cs, colsum_accumulate=True)
    _report("wgrad accumulate", out, 2 * ref, 2e-3)
    assert (cs - 1 - A.float().sum(0)).abs().max() < 2e-3 * A.float().abs().sum(0).max()


def test_gemm_cfg11_refuses_ragged_contraction():
    ops = _ops()
    A = torch.randn(200, 256, device="cuda").bfloat16(); B = torch.randn(200, 256, device="cuda").bfloat16()
    with pytest.raises(RuntimeError):        # K % 64 != 0
        ops.gemm_bf16(A, B, a_kcontig=False, b_kcontig=False, out_f32=torch.empty(256, 256, device="cuda"), tile_config=11)


def test_gemm_residual_inplace_and_ld():
    """out_f32 aliases residual (x += ...), outputs with a row stride larger than N."""
    ops = _ops()
    M, N, K = 200, 128, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    big = torch.randn(M, 3 * N, device="cuda", generator=g)
    x = big[:, N:2 * N]
    ref = x.clone() + A.float() @ B.float().t()
    ops.gemm_bf16(A, B, residual=x, out_f32=x)
    torch.cuda.synchronize()
    _report("inplace", x, ref, 2e-3)


@pytest.mark.parametrize("cfg", [1, 2, 3, 5])
@pytest.mark.parametrize("splits", [2, 5, 16])
def test_gemm_wgrad_splitk(splits, cfg):
    """wgrad layout (0,0) with ragged contraction (tokens) and split-K slabs + accumulate into grad."""
    ops = _ops()
    T, N, K = 1000 + 8, 256, 384      # contraction = tokens
    g = torch.Generator(device="cuda").manual_seed(splits)
    dY = torch.randn(T, N, device="cuda", generator=g).bfloat16()
    X = torch.randn(T, K, device="cuda", generator=g).bfloat16()
    grad = torch.randn(N, K, device="cuda", generator=g)
    ref = grad + dY.float().t() @ X.float()
    ws = torch.empty(splits * N * K, device="cuda")
    ops.gemm_bf16(dY, X, a_kcontig=False, b_kcontig=False, residual=grad, out_f32=grad, split_k=splits, split_ws=ws,
                  tile_config=cfg)
    torch.cuda.synchronize()
    _report(f"wgrad split{splits}", grad, ref, 2e-3)


def test_gemm_unaligned_n_and_k_with_padded_ld():
    """vocabulary-like shapes: N = 300 / K = 300 (not multiples of 8) on leading dimensions padded to 304."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(77)
    M, N, K = 130, 300, 128
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.full((M, 304), float("nan"), device="cuda")[:, :N]
    ops.gemm_bf16(A, W, bias=bias, out_f32=out)                      # lm_head: N = 300
    torch.cuda.synchronize()
    _report("N=300 fwd", out, A.float() @ W.float().t() + bias, 2e-3)
    dl = torch.zeros(M, 304, device="cuda", dtype=torch.bfloat16)[:, :N]
    dl.copy_(torch.randn(M, N, device="cuda", generator=g))
    dx = torch.empty(M, K, device="cuda")
    ops.gemm_bf16(dl, W, b_kcontig=False, out_f32=dx)                # dgrad: K = 300
    gw = torch.empty(N, K, device="cuda")
    ops.gemm_bf16(dl, A, a_kcontig=False, b_kcontig=False, out_f32=gw)   # wgrad: M' = 300
    torch.cuda.synchronize()
    _report("K=300 dgrad", dx, dl.float() @ W.float(), 2e-3)
    _report("M=300 wgrad", gw, dl.float().t() @ A.float(), 2e-3)


def test_gemm_bad_args_raise():
    ops = _ops()
    A = torch.zeros(64, 60, device="cuda", dtype=torch.bfloat16)   # row stride not a multiple of 8
    B = torch.zeros(64, 60, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(64, 64, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm_bf16(A, B, out_f32=out)


# ---- tile configuration 4: persistent workgroups, epilogue streamed under the next tile's K loop ----------------------
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,kind", [(512, 256, 768, "out16"), (2560, 384, 512, "gelu"), (1536, 256, 704, "gelu2"), (768, 128, 768, "res"),
                                         (25600, 768, 640, "res"), (12800, 2304, 1000, "out16"), (256, 128, 2048, "res")])
def test_gemm_streaming_config(M, N, K, kind, dt):
    """Several tiles per workgroup (more tiles than CUs in the big cases), ragged K, all three epilogue modes; against fp32 torch."""
    o = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(dt)
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    bias = torch.randn(N, device="cuda", generator=g)
    pre = A.float() @ B.float().t() + bias
    if kind == "res":
        x0 = torch.randn(M, N, device="cuda", generator=g)
        x = x0.clone()
        o.gemm_bf16(A, B, bias=bias, residual=x, out_f32=x, tile_config=4)
        _report("streamed fp32 residual", x, x0 + pre, 2e-3)
    elif kind == "gelu2":
        out, outp = torch.empty(M, N, device="cuda", dtype=dt), torch.empty(M, N, device="cuda", dtype=dt)
        o.gemm_bf16(A, B, bias=bias, act=1, out_bf16=out, out_pre=outp, tile_config=4)
        _report("streamed pre-activation", outp, pre, 1e-2)
        _report("streamed activation", out, _ref_act(pre, 1, None), 1e-2)
    else:
        out = torch.empty(M, N, device="cuda", dtype=dt)
        act = 1 if kind == "gelu" else 0
        o.gemm_bf16(A, B, bias=bias, act=act, out_bf16=out, tile_config=4)
        _report("streamed 16-bit out", out, _ref_act(pre, act, None), 1e-2)


def test_gemm_streaming_config_refuses_what_it_does_not_cover():
    from cclip_hip._lib import CclipError
    o = _ops()
    A = torch.zeros(300, 768, device="cuda", dtype=torch.bfloat16)       # M not a multiple of 256
    B = torch.zeros(128, 768, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(CclipError):
        o.gemm_bf16(A, B, out_bf16=torch.empty(300, 128, device="cuda", dtype=torch.bfloat16), tile_config=4)
    A = torch.zeros(256, 256, device="cuda", dtype=torch.bfloat16)       # K too short for the streaming window
    B = torch.zeros(128, 256, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(CclipError):
        o.gemm_bf16(A, B, out_bf16=torch.empty(256, 128, device="cuda", dtype=torch.bfloat16), tile_config=4)


# ---- skinny path: M <= 8 rows against K-strided (Conv1D) weights = the projections of a KV-cached decode step -----------
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,kind", [(3, 2304, 768, "out16"), (1, 768, 768, "res"), (3, 3072, 768, "gelu_new"), (5, 768, 3072, "res"),
                                         (8, 384, 128, "out16"), (2, 136, 200, "res"), (3, 384, 128, "pre")])
def test_gemm_skinny_decode_shapes(M, N, K, kind, dt):
    o = _ops()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(dt)
    W = (torch.randn(K, N, device="cuda", generator=g) * 0.05).to(dt)          # Conv1D layout [in, out]
    bias = torch.randn(N, device="cuda", generator=g)
    pre = A.float() @ W.float() + bias
    if kind == "res":
        x0 = torch.randn(M, N, device="cuda", generator=g)
        x = x0.clone()
        o.gemm_bf16(A, W, b_kcontig=False, bias=bias, residual=x, out_f32=x)
        _report("skinny residual", x, x0 + pre, 2e-3)
        x1 = x0.clone()
        o.gemm_bf16(A, W, b_kcontig=False, bias=bias, residual=x1, out_f32=x1, tile_config=1)       # the MFMA tile kernel on the same call
        assert (x - x1).abs().max() <= 2e-3 * (x0 + pre).abs().max()
    elif kind == "pre":
        out, outp = torch.empty(M, N, device="cuda", dtype=dt), torch.empty(M, N, device="cuda", dtype=dt)
        o.gemm_bf16(A, W, b_kcontig=False, bias=bias, act=3, out_bf16=out, out_pre=outp)
        _report("skinny pre", outp, pre, 1e-2)
        _report("skinny act", out, _ref_act(pre, 3, None), 1e-2)
    else:
        act = 3 if kind == "gelu_new" else 0
        out = torch.empty(M, N, device="cuda", dtype=dt)
        o.gemm_bf16(A, W, b_kcontig=False, bias=bias, act=act, out_bf16=out)
        _report("skinny 16-bit out", out, _ref_act(pre, act, None), 1e-2)


# ---- bias gradient fused into the weight-gradient GEMM (row sums of A over K) ------------------------------------------
@pytest.mark.parametrize("cfg,split", [(1, 1), (2, 1), (1, 8), (2, 5), (5, 1), (5, 7), (0, 0)])
@pytest.mark.parametrize("n_out,k_in,tokens", [(768, 256, 5000), (2304, 768, 3136), (200, 136, 1000)])
@pytest.mark.parametrize("conv1d", [False, True])
def test_wgrad_with_fused_bias_gradient(n_out, k_in, tokens, cfg, split, conv1d):
    """conv1d: GPT-2's [in, out] weight layout - the gradient is x^T dy and dy is the B operand (colsum_of_b)."""
    o = _ops()
    g = torch.Generator(device="cuda").manual_seed(n_out + k_in + tokens)
    dy = torch.randn(tokens, n_out, device="cuda", generator=g).bfloat16()
    x = torch.randn(tokens, k_in, device="cuda", generator=g).bfloat16()
    ref_w = x.float().t() @ dy.float() if conv1d else dy.float().t() @ x.float()
    ref_b = dy.float().sum(0)
    A_, B_ = (x, dy) if conv1d else (dy, x)
    M_, N_ = ref_w.shape
    for accumulate in (False, True):
        gw = torch.randn(M_, N_, device="cuda", generator=g)
        gb = torch.randn(n_out, device="cuda", generator=g)
        gw0, gb0 = gw.clone(), gb.clone()
        kw = {}
        if cfg == 0:        # the way the model calls it: autotuned (tile, split) candidates + scratch callback
            from cclip_hip.stack import Scratch, wgrad_candidates
            kw = dict(split_candidates=wgrad_candidates(M_, N_, tokens), scratch=Scratch(torch.device("cuda")).floats)
        elif split > 1:
            kw = dict(tile_config=cfg, split_k=split, split_ws=torch.empty(split * (M_ * N_ + max(M_, N_)), device="cuda"))
        else:
            kw = dict(tile_config=cfg)
        o.gemm_bf16(A_, B_, a_kcontig=False, b_kcontig=False, residual=gw if accumulate else None, out_f32=gw,
                    colsum_out=gb, colsum_accumulate=accumulate, colsum_of_b=conv1d, **kw)
        _report("fused wgrad", gw, ref_w + (gw0 if accumulate else 0), 2e-3)
        want_b = ref_b + (gb0 if accumulate else 0)
        assert (gb - want_b).abs().max() <= 2e-3 * want_b.abs().max(), (gb - want_b).abs().max()


def test_gemm_cfg8_layernorm_fold_and_rowstats():
    """LayerNorm folded into the projection (include/cclip_hip.h): the residual form emits the 16-bit copy of the new rows and
    per-64-column (sum, sum of squares) partials; cclip_rowstats_combine turns them into (mean, rstd); the folded projection
    multiplies the RAW rows by W' = gamma (.) W and applies rstd * (acc - mean * c1) + c2.  Checked against torch fp32 LayerNorm
    + matmul on the same 16-bit operands."""
    ops = _ops()
    M, D, N = 512, 768, 1024
    g = torch.Generator(device="cuda").manual_seed(11)
    a = (torch.randn(M, D, device="cuda", generator=g) * 0.2).bfloat16()
    wo = (torch.randn(D, D, device="cuda", generator=g) * 0.05).bfloat16()
    bo = torch.randn(D, device="cuda", generator=g) * 0.1
    x0 = torch.randn(M, D, device="cuda", generator=g) * 2.0 + 0.7          # residual stream with a non-zero mean
    # producer: x = x0 + a wo^T + bo, in place, + 16-bit copy + partial statistics
    x, xb = x0.clone(), torch.zeros(M, D, device="cuda", dtype=torch.bfloat16)
    part = torch.full((D // 64, M, 2), float("nan"), device="cuda")
    ops.gemm_bf16(a, wo, bias=bo, residual=x, out_f32=x, out_bf16=xb, rowstats_out=part)
    x_plain = x0.clone()
    ops.gemm_bf16(a, wo, bias=bo, residual=x_plain, out_f32=x_plain, tile_config=3)
    assert torch.equal(x, x_plain)                                       # the fp32 stream is bit-identical to the plain residual form
    assert torch.equal(xb, x.bfloat16())
    stats = torch.empty(M, 2, device="cuda")
    ops.rowstats_combine(part, stats, rows=M, D=D)
    mean, var = x.double().mean(1), x.double().var(1, unbiased=False)
    assert (stats[:, 0].double() - mean).abs().max() < 1e-5
    assert ((stats[:, 1].double() - (var + 1e-5).rsqrt()) / (var + 1e-5).rsqrt()).abs().max() < 1e-5
    # consumer: LayerNorm(x) W^T + b with the fold, against LayerNorm + plain GEMM
    gamma = 1.0 + 0.3 * torch.randn(D, device="cuda", generator=g)
    beta = 0.2 * torch.randn(D, device="cuda", generator=g)
    w = (torch.randn(N, D, device="cuda", generator=g) * 0.05).bfloat16()
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    ws = (w.float() * gamma[None, :]).bfloat16().contiguous()
    c1 = ws.float().sum(1).contiguous()
    c2 = (w.float() @ beta + b).contiguous()
    for act in (0, 1):
        y = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_bf16(xb, ws, bias=c2, act=act, out_bf16=y, ln_stats=stats, ln_c1=c1)
        ref = _ref_act(torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5) @ w.float().t() + b, act, None)
        _report(f"fold act{act}", y, ref, 1.2e-2)
        # and against the unfused pair on the device (LayerNorm kernel -> 16-bit -> GEMM): same size of 16-bit rounding error
        xn = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        ops.layernorm_fwd(x, gamma, beta, rows=M, out_bf16=xn)
        y2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_bf16(xn, w, bias=b, act=act, out_bf16=y2, tile_config=8)
        e_fold = (y.float() - ref).abs().max().item()
        e_pair = (y2.float() - ref).abs().max().item()
        assert e_fold < 2.5 * e_pair + 1e-3, (e_fold, e_pair)
    with pytest.raises(RuntimeError):          # whole 256-row tiles only
        ops.gemm_bf16(xb[:300], ws, bias=c2, out_bf16=torch.zeros(300, N, device="cuda", dtype=torch.bfloat16), ln_stats=stats, ln_c1=c1)
