"""GPU parity of the prefix-caption path (mapper MLP + GPT-2 prefix forward/backward + fused LM loss) against the
CPU oracle's golden vectors.  Tolerances as in test_clip_parity_gpu.py (bf16 operands): logits abs <= 0.05 on a range
of ~+-1.3, loss <= 5e-3, gradients rel L2 <= 6e-2 (tied wte gets lm_head + embedding contributions)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def sample(t, keep=4096):
    f = t.detach().flatten()
    k = max(1, -(-f.numel() // keep))
    return f[::k].clone()


def _setup(fix="caption_test_tiny.pt"):
    from clip_caption import ClipCaptionModel, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
    g = torch.load(os.path.join(GOLD, fix), weights_only=True)
    geo = GPT2_MODELS[g["model"]]
    model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    model.load_state_dict(init_caption_state_dict(geo, g["seed"]))
    model = model.cuda().train()
    tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(g["b"], geo, g["lc"], g["seed"] + 1)]
    return g, geo, model, tokens, mask, prefix, attribute


def test_caption_forward_logits_and_mapper():
    g, geo, model, tokens, mask, prefix, attribute = _setup()
    with torch.no_grad():
        out = model(tokens, prefix, attribute, mask)
        mapped = model.clip_project(prefix)
    P, A = geo.prefix_length, geo.attribute_length
    assert out.logits.shape == (g["b"], P + A + g["lc"], geo.vocab_size)
    sl = out.logits[:, P + A - 1:-1]
    assert (sample(sl, 8192).cpu() - g["logits_slice"]).abs().max() < 0.05
    assert rel(sample(mapped, 8192), g["mapper_out"]) < 1.2e-2
    # the generate loops' entry: gpt(inputs_embeds=...) == forward() on the same embeddings
    with torch.no_grad():
        emb = torch.cat((mapped.view(-1, P, geo.n_embd), model.gpt.transformer.wte(torch.cat((attribute, tokens), 1))), 1)
        lg2 = model.gpt(inputs_embeds=emb, attention_mask=mask).logits
    assert torch.equal(lg2, out.logits)


def test_caption_fp16_mode_is_tighter():
    g, geo, model, tokens, mask, prefix, attribute = _setup()
    model.half()
    loss = model.caption_loss(tokens, prefix, attribute, mask)
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) < 6e-4
    params = dict(model.named_parameters())
    for k, ref in g["grads"].items():
        assert rel(sample(params[k].grad), ref) < 1e-2, (k, rel(sample(params[k].grad), ref))


def test_caption_loss_and_grads_fused():
    g, geo, model, tokens, mask, prefix, attribute = _setup()
    loss = model.caption_loss(tokens, prefix, attribute, mask)
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) < 5e-3
    params = dict(model.named_parameters())
    for k, ref in g["grads"].items():
        assert params[k].grad is not None, k
        assert rel(sample(params[k].grad), ref) < 6e-2, (k, rel(sample(params[k].grad), ref))
    for k, nrm in g["grad_norms"].items():
        if k == "model.lm_head.weight":
            continue
        assert abs(params[k].grad.norm().item() - nrm.item()) <= 0.04 * nrm.item() + 1e-7, k


@pytest.mark.parametrize("half", [False, True])
def test_caption_real_geometry_matches_golden(half):
    """BASELINE configs[3] at its real geometry: GPT-2-small (V = 21128, 12 layers, 12 heads), MLP mapper 512 -> 7680 ->
    15360, P = A = 20, Lc = 40 (S = 80), B = 2 (CLIP_prefix_caption/train.py:277-279, 354-357) against the oracle: mapper
    output, the logits slice the loss reads, the loss, 7 sampled gradients and every gradient norm."""
    g, geo, model, tokens, mask, prefix, attribute = _setup("caption_gpt2_base_chinese.pt")
    assert (geo.vocab_size, geo.n_layer, geo.n_embd, geo.prefix_length, geo.attribute_length) == (21128, 12, 768, 20, 20)
    sdm = model.state_dict()
    assert sdm["clip_project.model.0.weight"].shape == (7680, 512) and sdm["clip_project.model.2.weight"].shape == (15360, 7680)
    if half:
        model.half()
    tol = dict(logit=1e-2, feat=1.5e-3, loss=1e-3, grad=1.2e-2, norm=0.01) if half else \
        dict(logit=0.06, feat=1.2e-2, loss=5e-3, grad=6e-2, norm=0.04)
    P, A = geo.prefix_length, geo.attribute_length
    with torch.no_grad():
        out = model(tokens, prefix, attribute, mask)
        mapped = model.clip_project(prefix)
    assert out.logits.shape == (g["b"], P + A + g["lc"], geo.vocab_size)
    sl = out.logits[:, P + A - 1:-1]
    assert (sample(sl, 8192).cpu() - g["logits_slice"]).abs().max() < tol["logit"]
    assert rel(sample(mapped, 8192), g["mapper_out"]) < tol["feat"]
    loss = model.caption_loss(tokens, prefix, attribute, mask)
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) < tol["loss"]
    params = dict(model.named_parameters())
    for k, ref in g["grads"].items():
        assert rel(sample(params[k].grad), ref) < tol["grad"], (k, rel(sample(params[k].grad), ref))
    ratios = {k: params[k].grad.norm().item() / nrm.item() for k, nrm in g["grad_norms"].items() if k != "model.lm_head.weight"}
    worst = sorted(ratios.items(), key=lambda kv: -abs(kv[1] - 1.0))[:6]
    print("gradient-norm ratios (HIP / oracle), worst:", [(k, round(v, 4)) for k, v in worst])
    for k, r in ratios.items():
        assert abs(r - 1.0) <= tol["norm"], (k, r)


def test_caption_bs256_properties():
    """BASELINE configs[3] at its batch (bs 256, S = 80, real geometry): batch independence, key padding, and the fused LM
    loss against torch's cross_entropy on the materialised logits slice (train.py:354-357)."""
    from clip_caption import ClipCaptionModel, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
    geo = GPT2_MODELS["ckiplab/gpt2-base-chinese"]
    model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    model.load_state_dict(init_caption_state_dict(geo, 41))
    model = model.cuda().train()
    B, Lc = 256, 40
    tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(B, geo, Lc, 42)]
    P, A = geo.prefix_length, geo.attribute_length
    with torch.no_grad():
        full = model(tokens, prefix, attribute, mask).logits
        part = model(tokens[100:104], prefix[100:104], attribute[100:104], mask[100:104]).logits
        assert torch.equal(part, full[100:104])                                   # a sample does not see its batch
        m2 = mask.clone(); m2[:, -5:] = 0
        masked = model(tokens, prefix, attribute, m2).logits
        assert torch.equal(masked[:, :-5], full[:, :-5]) and not torch.equal(masked[:, -1], full[:, -1])
    assert torch.isfinite(full).all()
    loss = model.caption_loss(tokens, prefix, attribute, mask)
    sl = full[:, P + A - 1:-1].float()
    ref = torch.nn.functional.cross_entropy(sl.reshape(-1, sl.shape[-1]).double(), tokens.flatten(), ignore_index=0)
    assert abs(loss.item() - ref.item()) < 2e-4, (loss.item(), ref.item())
    loss.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    assert all(bool(torch.isfinite(v).all()) for v in g1.values())
    # torch's CE on the materialised slice drives the same kernels' backward (the reference loop's form)
    model.zero_grad(set_to_none=True)
    out = model(tokens, prefix, attribute, mask).logits[:, P + A - 1:-1]
    torch.nn.functional.cross_entropy(out.reshape(-1, out.shape[-1]), tokens.flatten(), ignore_index=0).backward()
    worst = max((rel(p.grad, g1[k]), k) for k, p in model.named_parameters())
    assert worst[0] < 2e-2, worst


def test_caption_loss_on_packed_rows_equals_dense_rows():
    """caption_loss runs GPT-2 on each sequence's rows up to the one that predicts its last non-zero token (default) - against
    the same call on all P + A + L rows (model.pack_rows = False): the same loss and gradients (only rows that are neither
    targets nor keys of a needed row are gone; sums are partitioned differently), a key-padding mask is carried along, and a zero
    token INSIDE a caption stays an ignored target without cutting the sequence."""
    from clip_caption import ClipCaptionModel, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
    geo = GPT2_MODELS["ckiplab/gpt2-base-chinese"]
    B, Lc = 24, 40
    tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(B, geo, Lc, 43)]
    tokens[3, 2] = 0                                   # interior zero: ignored target, sequence continues
    tokens[5, :] = 0                                   # an empty caption contributes nothing
    mask = mask.clone(); mask[7, 3:9] = 0              # some masked keys inside the attribute block
    outs = []
    for pack in (False, True):
        model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
        model.load_state_dict(init_caption_state_dict(geo, 41))
        model = model.cuda().train()
        model.pack_rows = pack
        loss = model.caption_loss(tokens, prefix, attribute, mask)
        loss.backward()
        outs.append((loss.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert abs(outs[0][0] - outs[1][0]) < 2e-6 * abs(outs[0][0]) + 1e-7, (outs[0][0], outs[1][0])
    assert set(outs[0][1]) == set(outs[1][1])
    worst = max((rel(outs[1][1][k], outs[0][1][k]), k) for k in outs[0][1])
    assert worst[0] < 2e-3, worst


def test_caption_reference_loop_with_torch_ce():
    """train.py:354-361 verbatim: outputs.logits slice -> nnf.cross_entropy(ignore_index=0) -> backward -> optimiser."""
    g, geo, model, tokens, mask, prefix, attribute = _setup()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    P, A = geo.prefix_length, geo.attribute_length
    losses = []
    for _ in range(3):
        model.zero_grad()
        outputs = model(tokens, prefix, attribute, mask)
        logits = outputs.logits[:, P + A - 1:-1]
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tokens.flatten(), ignore_index=0)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    assert abs(losses[0] - g["loss"].item()) < 5e-3 and losses[-1] < losses[0]


def test_prefix_only_training_freezes_gpt():
    from clip_caption import ClipCaptionPrefix, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
    geo = GPT2_MODELS["test-tiny"]
    model = ClipCaptionPrefix(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    model.load_state_dict(init_caption_state_dict(geo, 3))
    model = model.cuda().train()
    tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(2, geo, 10, 4)]
    model.caption_loss(tokens, prefix, attribute, mask).backward()
    assert all(p.grad is not None for p in model.clip_project.parameters())
    assert all(p.grad is None for p in model.model.parameters())
    assert len(list(model.parameters())) == 4


def test_key_padding_mask_is_honoured():
    g, geo, model, tokens, mask, prefix, attribute = _setup()
    m2 = mask.clone()
    m2[:, -3:] = 0
    with torch.no_grad():
        a = model(tokens, prefix, attribute, mask).logits
        b = model(tokens, prefix, attribute, m2).logits
    assert not torch.equal(a[:, -1], b[:, -1])            # last position no longer sees the masked keys... nor itself
    assert torch.equal(a[:, : a.shape[1] - 3], b[:, : a.shape[1] - 3])   # causal: earlier positions never saw them


# ---- --mapping_type transformer (train.py:233-248, 397) ----------------------------------------------------------------
def _setup_tmapper():
    from clip_caption import (ClipCaptionModel, GPT2_MODELS, MappingType, init_caption_state_dict, init_transformer_mapper_state_dict,
                              synthetic_caption_batch)
    g = torch.load(os.path.join(GOLD, "caption_tmapper_tiny.pt"), weights_only=True)
    geo = GPT2_MODELS[g["model"]]
    model = ClipCaptionModel(geo.prefix_length, clip_length=g["clip_length"], prefix_size=geo.prefix_size, num_layers=g["num_layers"],
                             mapping_type=MappingType.Transformer, gpt2_type=geo)
    sd = {k: v for k, v in init_caption_state_dict(geo, g["seed"]).items() if not k.startswith("clip_project.")}
    sd.update(init_transformer_mapper_state_dict(geo, g["clip_length"], g["num_layers"], g["seed"] + 7))
    assert set(sd) == set(model.state_dict())                 # the reference's checkpoint key layout, exactly
    model.load_state_dict(sd)
    model = model.cuda().train()
    tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(g["b"], geo, g["lc"], g["seed"] + 1)]
    return g, geo, model, tokens, mask, prefix, attribute


@pytest.mark.parametrize("half", [False, True])
def test_transformer_mapper_forward_loss_grads(half):
    g, geo, model, tokens, mask, prefix, attribute = _setup_tmapper()
    if half:
        model.half()
    ftol, ltol, gtol = (3e-3, 6e-4, 2e-2) if half else (1.5e-2, 5e-3, 6e-2)
    with torch.no_grad():
        mapped = model.clip_project(prefix)
    assert mapped.shape == (g["b"], geo.prefix_length, geo.n_embd)
    assert rel(sample(mapped, 8192), g["mapper_out"]) < ftol
    loss = model.caption_loss(tokens, prefix, attribute, mask)
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) < ltol
    params = dict(model.named_parameters())
    for k, ref in g["grads"].items():
        assert params[k].grad is not None, k
        assert rel(sample(params[k].grad), ref) < gtol, (k, rel(sample(params[k].grad), ref))
    for k, nrm in g["grad_norms"].items():
        assert abs(params[k].grad.norm().item() - nrm.item()) <= 0.04 * nrm.item() + 1e-7, k


def test_transformer_mapper_reference_shape_runs():
    """The reference's own configuration (train.py:391-399: prefix_length 20, clip_length 20 -> 40 tokens, 8 heads x 96, 8 layers) -
    a finite loss, gradients on every mapper tensor, and loss decreases under the fused AdamW."""
    from clip import optim as coptim
    from clip_caption import ClipCaptionPrefix, CaptionGeometry, MappingType, init_caption_state_dict, synthetic_caption_batch
    geo = CaptionGeometry(vocab_size=1000, n_layer=2)
    torch.manual_seed(5)
    model = ClipCaptionPrefix(20, clip_length=20, prefix_size=512, num_layers=8, mapping_type=MappingType.Transformer, gpt2_type=geo)
    model.model.load_state_dict({k[len("model."):]: v for k, v in init_caption_state_dict(geo, 5).items() if k.startswith("model.")})
    model = model.cuda().train()                 # the mapper keeps its constructor (nn.Linear-style) initialisation
    tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(4, geo, 16, 9)]
    opt = coptim.AdamW(model, lr=1e-3)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = model.caption_loss(tokens, prefix, attribute, mask)
        loss.backward()
        losses.append(loss.item())
        bad = [n for n, p in model.clip_project.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
        assert not bad, bad
        assert all(p.grad.abs().max() > 0 for p in model.clip_project.parameters())
        opt.step()
    assert losses[-1] < losses[0], losses


# ---- against fixtures produced by RUNNING the reference's own classes (tests/golden/make_reference_fixtures.py) ----------
def _ref_fixture(name):
    return torch.load(os.path.join(GOLD, name), weights_only=True)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("fix,name", [("ref_mlp_mapper_tiny.pt", "test-tiny"), ("ref_mlp_mapper_real.pt", "ckiplab/gpt2-base-chinese")])
def test_mlp_mapper_matches_reference_class_output(fix, name, half):
    """`clip_project` (MLP, train.py:110-123; real size 512 -> 7680 -> 15360) on the HIP path against what the reference's
    own MLP class computed from the same seeded weights: output and the gradients of sum(y * w)."""
    from clip_caption import ClipCaptionPrefix, GPT2_MODELS, init_caption_state_dict
    fx = _ref_fixture(fix)
    geo = GPT2_MODELS[name]
    model = ClipCaptionPrefix(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    model.load_state_dict(init_caption_state_dict(geo, fx["seed"]))
    model = model.cuda().train()
    if half:
        model.half()
    y = model.clip_project(fx["prefix"].cuda())
    assert list(y.shape) == fx["out_shape"].tolist()
    tol, gtol = (1.5e-3, 1.2e-2) if half else (1.2e-2, 6e-2)
    assert rel(sample(y, 8192), fx["out"]) < tol, rel(sample(y, 8192), fx["out"])
    w = (torch.randn(y.shape, generator=torch.Generator().manual_seed(fx["w_seed"])) / y.numel() ** 0.5).cuda()
    (y * w).sum().backward()
    params = dict(model.named_parameters())
    for k, ref in fx["grads"].items():
        assert rel(sample(params[k].grad), ref) < gtol, (k, rel(sample(params[k].grad), ref))


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("fix,name", [("ref_tmapper_tiny.pt", "test-tiny"), ("ref_tmapper_real.pt", "ckiplab/gpt2-base-chinese")])
def test_transformer_mapper_matches_reference_class_output(fix, name, half):
    """`--mapping_type transformer` (train.py:126-248; real size: 8 layers, 8 heads x 96, 40 tokens) against the reference's
    own TransformerMapper class on the same seeded weights."""
    from clip_caption import ClipCaptionPrefix, GPT2_MODELS, MappingType, init_transformer_mapper_state_dict
    fx = _ref_fixture(fix)
    geo = GPT2_MODELS[name]
    model = ClipCaptionPrefix(geo.prefix_length, clip_length=fx["clip_length"], prefix_size=geo.prefix_size,
                              num_layers=fx["num_layers"], mapping_type=MappingType.Transformer, gpt2_type=geo)
    sd = init_transformer_mapper_state_dict(geo, fx["clip_length"], fx["num_layers"], fx["seed"])
    model.clip_project.load_state_dict({k[len("clip_project."):]: v for k, v in sd.items()})
    model = model.cuda().train()
    if half:
        model.half()
    y = model.clip_project(fx["prefix"].cuda())
    assert y.shape == (fx["b"], geo.prefix_length, geo.n_embd)
    tol, gtol, ntol = (2e-3, 3e-2, 0.01) if half else (1.5e-2, 6e-2, 0.04)      # (measured fp16: 2.1e-2 worst, 8 layers deep)
    assert rel(sample(y, 8192), fx["out"]) < tol, rel(sample(y, 8192), fx["out"])
    w = (torch.randn(y.shape, generator=torch.Generator().manual_seed(fx["w_seed"])) / y.numel() ** 0.5).cuda()
    (y * w).sum().backward()
    params = dict(model.named_parameters())
    for k, ref in fx["grads"].items():
        assert rel(sample(params[k].grad, 2048), ref) < gtol, (k, rel(sample(params[k].grad, 2048), ref))
    for k, nrm in fx["grad_norms"].items():
        assert abs(params[k].grad.norm().item() - nrm.item()) <= ntol * nrm.item() + 1e-7, k


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("fix", ["ref_caption_forward_tiny.pt", "ref_caption_forward_real.pt"])
def test_caption_forward_matches_reference_forward(fix, half):
    """ClipCaptionModel.forward + the train.py:356-357 loss as the reference's own code computed them (its MLP class + the local
    transformers GPT-2, same seeded weights): logits slice, loss, sampled gradients, gradient norms."""
    from clip_caption import ClipCaptionModel, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
    fx = _ref_fixture(fix)
    geo = GPT2_MODELS[fx["model"]]
    model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    model.load_state_dict(init_caption_state_dict(geo, fx["seed"]))
    model = model.cuda().train()
    if half:
        model.half()
    tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(fx["b"], geo, fx["lc"], fx["seed"] + 1)]
    P, A = geo.prefix_length, geo.attribute_length
    outputs = model(tokens, prefix, attribute, mask)
    logits = outputs.logits[:, P + A - 1:-1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tokens.flatten(), ignore_index=0)
    loss.backward()
    t = dict(logit=1e-2, loss=1e-3, grad=1.2e-2, norm=0.01) if half else dict(logit=0.06, loss=5e-3, grad=6e-2, norm=0.04)
    assert (sample(logits, 8192).cpu() - fx["logits_slice"]).abs().max() < t["logit"]
    assert abs(loss.item() - fx["loss"].item()) < t["loss"]
    params = dict(model.named_parameters())
    for k, ref in fx["grads"].items():
        assert rel(sample(params[k].grad), ref) < t["grad"], (k, rel(sample(params[k].grad), ref))
    for k, nrm in fx["grad_norms"].items():
        if k in params and params[k].grad is not None and k != "model.lm_head.weight":
            assert abs(params[k].grad.norm().item() - nrm.item()) <= t["norm"] * nrm.item() + 1e-7, k
