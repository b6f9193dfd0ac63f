"""fp8 (OCP e4m3) projections - BASELINE.json configs[4].  No reference fp8 behaviour exists (parity unpinned): the quantiser
and the GEMM are checked exactly against torch arithmetic on the SAME quantised operands, and the model-level path is
bounded against the fp32 oracle with an fp8-sized tolerance."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ops():
    from cclip_hip import ops
    return ops


def _dequant(q8, scale):
    return q8.view(torch.float8_e4m3fn).float() * scale[:, None]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_quantize_rows_fp8(dt):
    o = _ops()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(300, 1024, device="cuda", generator=g) * torch.logspace(-3, 2, 300, device="cuda")[:, None]).to(dt)
    x[7] = 0
    q = torch.empty(300, 1024, device="cuda", dtype=torch.uint8)
    s = torch.empty(300, device="cuda")
    o.quantize_rows_fp8(x, q, s)
    amax = x.float().abs().amax(1)
    want_s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(s, want_s, rtol=1e-6)
    want_q = (x.float() * (1.0 / s)[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)   # same fp32 arithmetic as the kernel; RNE both sides
    assert torch.equal(q.view(torch.float8_e4m3fn).float(), want_q.float())
    rel = ((_dequant(q, s) - x.float()).norm(dim=1) / x.float().norm(dim=1).clamp_min(1e-30))
    assert rel[rel == rel].max() < 0.04                                                  # 3 mantissa bits


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,act", [(256, 256, 128, 0), (577, 1024, 1024, 0), (1000, 384, 400, 1), (2308, 4096, 1024, 1), (64, 72, 48, 0)])
def test_gemm_fp8_matches_torch_on_the_same_quantised_operands(M, N, K, act, dt):
    o = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).to(dt)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    bias = torch.randn(N, device="cuda", generator=g)
    A8, sa = torch.empty(M, K, device="cuda", dtype=torch.uint8), torch.empty(M, device="cuda")
    W8, sw = torch.empty(N, K, device="cuda", dtype=torch.uint8), torch.empty(N, device="cuda")
    o.quantize_rows_fp8(A, A8, sa)
    o.quantize_rows_fp8(W, W8, sw)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
    o.gemm_fp8(A8, sa, W8, sw, out, bias=bias, act=act)
    ref = (A8.view(torch.float8_e4m3fn).float() @ W8.view(torch.float8_e4m3fn).float().t()) * sa[:, None] * sw[None, :] + bias
    if act == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    err = (out.float() - ref).abs().max().item()
    assert err <= 1e-2 * ref.abs().max().item(), err                                     # 16-bit output rounding only
    full = A.float() @ W.float().t() + bias
    if act == 1:
        full = full * torch.sigmoid(1.702 * full)
    assert ((out.float() - full).norm() / full.norm()).item() < 0.06                     # the fp8 quantisation error itself


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_layernorm_fp8_matches_layernorm_then_quantise():
    o = _ops()
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(500, 1024, device="cuda", generator=g) * 3 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(1024, device="cuda", generator=g), 0.1 * torch.randn(1024, device="cuda", generator=g)
    q, s = torch.empty(500, 1024, device="cuda", dtype=torch.uint8), torch.empty(500, device="cuda")
    o.layernorm_fwd_fp8(x, gamma, beta, q, s, rows=500)
    y = torch.nn.functional.layer_norm(x, (1024,), gamma, beta, 1e-5)
    assert torch.allclose(s, y.abs().amax(1) / 448.0, rtol=1e-4)
    assert rel(_dequant(q, s), y) < 0.03


@pytest.mark.parametrize("fix", ["clip_test_small.pt", "clip_vit_b32.pt"])
def test_clip_fp8_projections_against_oracle(fix):
    """Model level: qkv and fc of the image tower in e4m3 (text tower stays 16-bit by default), everything else as before.
    Bound, not parity: image features within 4e-2 relative of the fp32 oracle (measured 2.3-2.6e-2), cosine > 0.999, logits
    within 0.5 (logit scale ~ 14), well-separated arg-maxes kept.  text=True is measured at ~7e-2 and bounded at 0.1."""
    import clip
    from clip.weights import MODELS, init_state_dict, synthetic_images
    gd = torch.load(os.path.join(GOLD, fix), weights_only=True)
    geo = MODELS[gd["model"]]
    model = clip.build_model(init_state_dict(geo, gd["seed"])).cuda().eval().fp8_projections()
    img = synthetic_images(gd["n"], geo, gd["seed"] + 1).cuda()
    with torch.no_grad():
        fi, ft = model.encode_image(img), model.encode_text(gd["text"].cuda())
        li, _ = model(img, gd["text"].cuda())
    assert rel(fi, gd["image_features"]) < 4e-2 and rel(ft, gd["text_features"]) < 1.2e-2, (rel(fi, gd["image_features"]), rel(ft, gd["text_features"]))
    cos = torch.nn.functional.cosine_similarity(fi.cpu().float(), gd["image_features"], dim=1)
    assert cos.min() > 0.999
    assert (li.cpu() - gd["logits_per_image"]).abs().max() < 0.5
    ref = gd["logits_per_image"]
    top2 = ref.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1.0
    assert torch.equal(li.argmax(1).cpu()[decided], ref.argmax(1)[decided])
    model.fp8_projections(True, text=True)
    with torch.no_grad():
        ft8 = model.encode_text(gd["text"].cuda())
    assert rel(ft8, gd["text_features"]) < 0.1
    model.fp8_projections(False)
    with torch.no_grad():
        fb = model.encode_image(img)
    assert rel(fb, gd["image_features"]) < 1.2e-2          # and back to the bf16 path


def test_vit_l14_336_fp8_encode_image():
    """BASELINE.json configs[4]: ViT-L/14@336px encode_image with the fp8 projections, 2 images, against the oracle's features."""
    import clip
    from clip.weights import MODELS, init_state_dict, synthetic_images
    gd = torch.load(os.path.join(GOLD, "clip_vit_l14_336.pt"), weights_only=True)
    geo = MODELS[gd["model"]]
    model = clip.build_model(init_state_dict(geo, gd["seed"])).cuda().eval().fp8_projections()
    img = synthetic_images(gd["n"], geo, gd["seed"] + 1).cuda()
    with torch.no_grad():
        fi = model.encode_image(img)
    r = rel(fi, gd["image_features"])
    assert r < 5e-2, r
