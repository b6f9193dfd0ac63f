"""CPU: pin the oracle (oracle/*.py) - the checker every GPU parity test leans on.

1. against the only independent CLIP / GPT-2 arithmetic in the container, `transformers` (config-only
   models, random init) through oracle/hf_crosscheck.py: <= 1e-5 relative fp32;
2. against the committed golden vectors (tests/golden/*.pt, made by tests/golden/make_golden.py):
   the oracle + seeded weights must reproduce them bit-for-bit-ish (<= 1e-5), which also proves the
   seed -> weights path is deterministic on this machine.
The reference itself has no fixtures for this path: parity vs the reference is unpinned.
"""
import os

import pytest
import torch

from clip.weights import MODELS, init_state_dict, synthetic_images, synthetic_text
from clip_caption.weights import GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
from oracle import caption_oracle as CO
from oracle import clip_oracle as O
from oracle import hf_crosscheck as H
from oracle.optim_oracle import HFAdamW, linear_schedule

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
transformers = pytest.importorskip("transformers")


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("name", ["test-tiny", "test-small"])
def test_clip_oracle_matches_hf(name):
    geo = MODELS[name]
    sd = init_state_dict(geo, 567)
    img, txt = synthetic_images(5, geo, 1), synthetic_text(5, geo, 2)
    hf = H.build_hf_clip(sd)
    with torch.no_grad():
        li, lt = O.clip_forward(sd, img, txt)
        out = hf(input_ids=txt.long(), pixel_values=img)
        fi, ft = O.encode_image(sd, img), O.encode_text(sd, txt)
    assert rel(li, out.logits_per_image) < 1e-5
    assert rel(lt, out.logits_per_text) < 1e-5
    n = lambda x: x / x.norm(dim=1, keepdim=True)
    assert rel(n(fi), n(out.image_embeds)) < 1e-5
    assert rel(n(ft), n(out.text_embeds)) < 1e-5


def test_clip_oracle_gradients_match_hf():
    geo = MODELS["test-tiny"]
    sd = init_state_dict(geo, 5)
    img, txt = synthetic_images(4, geo, 1), synthetic_text(4, geo, 2)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, _ = O.contrastive_loss(*O.clip_forward(sdg, img, txt))
    loss.backward()
    hf = H.build_hf_clip(sd).train()
    out = hf(input_ids=txt.long(), pixel_values=img, return_loss=True)
    out.loss.backward()
    assert abs(out.loss.item() - loss.item()) < 1e-5
    g_hf = hf.visual_projection.weight.grad.t()
    assert rel(sdg["visual.proj"].grad, g_hf) < 1e-4
    assert rel(sdg["visual.conv1.weight"].grad, hf.vision_model.embeddings.patch_embedding.weight.grad) < 1e-4
    assert rel(sdg["logit_scale"].grad, hf.logit_scale.grad) < 1e-4


def test_text_pooling_ignores_tokens_after_eot():
    geo = MODELS["test-tiny"]
    sd = init_state_dict(geo, 7)
    txt = synthetic_text(3, geo, 3)
    txt2 = txt.clone()
    eot = txt.argmax(-1)
    for i in range(3):
        txt2[i, eot[i] + 1:] = 5          # junk after EOT (smaller ids than EOT)
    with torch.no_grad():
        assert torch.equal(O.encode_text(sd, txt), O.encode_text(sd, txt2))   # causal mask


def test_caption_oracle_matches_hf_gpt2():
    geo = GPT2_MODELS["test-tiny"]
    sd = init_caption_state_dict(geo, 5)
    tokens, mask, prefix, attribute = synthetic_caption_batch(3, geo, 10, 7)
    mask[1, -3:] = 0
    hf = H.build_hf_gpt2(sd, geo.n_head)
    with torch.no_grad():
        lg = CO.caption_forward(sd, tokens, prefix, attribute, mask, geo.prefix_length, geo.n_head)
        wte = sd["model.transformer.wte.weight"]
        emb = torch.cat((CO.mlp_mapper(sd, prefix).view(-1, geo.prefix_length, geo.n_embd),
                         wte[torch.cat((attribute, tokens), 1)]), 1)
        ref = hf(inputs_embeds=emb, attention_mask=mask).logits
    assert rel(lg, ref) < 1e-5


@pytest.mark.parametrize("fix", ["clip_test_tiny.pt", "clip_test_small.pt"])
def test_golden_clip_vectors_reproduce(fix):
    g = torch.load(os.path.join(GOLD, fix), weights_only=True)
    geo = MODELS[g["model"]]
    sd = init_state_dict(geo, g["seed"])
    img = synthetic_images(g["n"], geo, g["seed"] + 1)
    with torch.no_grad():
        li, lt = O.clip_forward(sd, img, g["text"])
        loss, acc = O.contrastive_loss(li, lt)
    assert rel(li, g["logits_per_image"]) < 1e-5
    assert abs(loss.item() - g["loss"].item()) < 1e-5
    with torch.no_grad():
        sim9, idx9 = O.zero_shot(sd, img[:1], g["text"][:9])
    assert torch.equal(idx9, g["zs9_idx"]) and rel(sim9, g["zs9_sim"]) < 1e-5


def test_golden_caption_vectors_reproduce():
    g = torch.load(os.path.join(GOLD, "caption_test_tiny.pt"), weights_only=True)
    geo = GPT2_MODELS[g["model"]]
    sd = init_caption_state_dict(geo, g["seed"])
    tokens, mask, prefix, attribute = synthetic_caption_batch(g["b"], geo, g["lc"], g["seed"] + 1)
    with torch.no_grad():
        lg = CO.caption_forward(sd, tokens, prefix, attribute, mask, geo.prefix_length, geo.n_head)
        loss = CO.caption_loss(lg, tokens, geo.prefix_length, geo.attribute_length)
    assert abs(loss.item() - g["loss"].item()) < 1e-5


def test_schedule_matches_transformers():
    from transformers import get_linear_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=7, num_training_steps=40)
    for step in range(45):
        assert abs(opt.param_groups[0]["lr"] - linear_schedule(step, 7, 40)) < 1e-12
        opt.step(); sch.step()


def test_hf_adamw_restatement_first_steps():
    # closed form of step 1 with correct_bias: p -= lr * g / (|g| + eps*sqrt(1-b2)) ... checked numerically
    p = {"w": torch.tensor([1.0, -2.0, 0.5])}
    g = {"w": torch.tensor([0.1, -0.3, 0.0])}
    opt = HFAdamW(p, lr=1e-2, eps=1e-6)
    opt.step(g)
    m = 0.1 * g["w"]; v = 0.001 * g["w"] ** 2
    step = 1e-2 * (1 - 0.999) ** 0.5 / (1 - 0.9)
    ref = torch.tensor([1.0, -2.0, 0.5]) - step * m / (v.sqrt() + 1e-6)
    assert torch.allclose(p["w"], ref, atol=1e-7)


# ---- 3. against fixtures produced by RUNNING the reference's own in-tree classes (tests/golden/make_reference_fixtures.py):
#         MLP / TransformerMapper (CLIP_prefix_caption/train.py:110-248) and ClipCaptionModel.forward (train.py:256-269) ----
def _sample(t, keep=8192):
    f = t.detach().flatten()
    k = max(1, -(-f.numel() // keep))
    return f[::k].clone()


@pytest.mark.parametrize("fix,name", [("ref_mlp_mapper_tiny.pt", "test-tiny"), ("ref_mlp_mapper_real.pt", "ckiplab/gpt2-base-chinese")])
def test_mlp_mapper_oracle_matches_reference_class(fix, name):
    from clip_caption.weights import init_caption_state_dict as init
    fx = torch.load(os.path.join(GOLD, fix), weights_only=True)
    geo = GPT2_MODELS[name]
    sd = {k: v.clone().requires_grad_(True) for k, v in init(geo, fx["seed"]).items() if k.startswith("clip_project.")}
    y = CO.mlp_mapper(sd, fx["prefix"])
    assert list(y.shape) == fx["out_shape"].tolist()
    assert rel(_sample(y), fx["out"]) < 1e-5
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(fx["w_seed"])) / y.numel() ** 0.5
    (y * w).sum().backward()
    for k, ref in fx["grads"].items():
        assert rel(_sample(sd[k].grad, 4096), ref) < 1e-4, k


@pytest.mark.parametrize("fix,name", [("ref_tmapper_tiny.pt", "test-tiny"), ("ref_tmapper_real.pt", "ckiplab/gpt2-base-chinese")])
def test_transformer_mapper_oracle_matches_reference_class(fix, name):
    from clip_caption.weights import init_transformer_mapper_state_dict as init
    fx = torch.load(os.path.join(GOLD, fix), weights_only=True)
    geo = GPT2_MODELS[name]
    sd = {k: v.clone().requires_grad_(True) for k, v in init(geo, fx["clip_length"], fx["num_layers"], fx["seed"]).items()}
    y = CO.transformer_mapper(sd, fx["prefix"], fx["clip_length"])
    assert y.shape == (fx["b"], geo.prefix_length, geo.n_embd)
    assert rel(_sample(y), fx["out"]) < 1e-5
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(fx["w_seed"])) / y.numel() ** 0.5
    (y * w).sum().backward()
    for k, ref in fx["grads"].items():
        assert rel(_sample(sd[k].grad, 2048), ref) < 1e-4, k
    for k, nrm in fx["grad_norms"].items():
        assert abs(sd[k].grad.double().norm().item() - nrm.item()) <= 1e-4 * nrm.item() + 1e-9, k


@pytest.mark.parametrize("fix", ["ref_caption_forward_tiny.pt", "ref_caption_forward_real.pt"])
def test_caption_oracle_matches_reference_forward(fix):
    """The reference's ClipCaptionModel.forward + train.py:356-357 loss, run on its MLP class and the local transformers GPT-2."""
    fx = torch.load(os.path.join(GOLD, fix), weights_only=True)
    geo = GPT2_MODELS[fx["model"]]
    sd = init_caption_state_dict(geo, fx["seed"])
    tokens, mask, prefix, attribute = synthetic_caption_batch(fx["b"], geo, fx["lc"], fx["seed"] + 1)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "model.lm_head.weight"}
    sdg["model.lm_head.weight"] = sdg["model.transformer.wte.weight"]
    logits = CO.caption_forward(sdg, tokens, prefix, attribute, mask, geo.prefix_length, geo.n_head)
    loss = CO.caption_loss(logits, tokens, geo.prefix_length, geo.attribute_length)
    loss.backward()
    P, A = geo.prefix_length, geo.attribute_length
    assert rel(_sample(logits[:, P + A - 1:-1]), fx["logits_slice"]) < 1e-5
    assert abs(loss.item() - fx["loss"].item()) < 1e-5
    for k, ref in fx["grads"].items():
        assert rel(_sample(sdg[k].grad, 4096), ref) < 2e-4, (k, rel(_sample(sdg[k].grad, 4096), ref))
    for k, nrm in fx["grad_norms"].items():
        if k in sdg and sdg[k].grad is not None:
            assert abs(sdg[k].grad.double().norm().item() - nrm.item()) <= 2e-4 * nrm.item() + 1e-9, k


def test_golden_vit_b32_gradients_reproduce():
    """the round-2 ViT-B/32 gradient golden (BASELINE configs[1] geometry) is what the oracle computes from the seed"""
    g = torch.load(os.path.join(GOLD, "clip_vit_b32.pt"), weights_only=True)
    geo = MODELS[g["model"]]
    sd = init_state_dict(geo, g["seed"])
    img = synthetic_images(g["n"], geo, g["seed"] + 1)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, _ = O.contrastive_loss(*O.clip_forward(sdg, img, g["text"]))
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) < 1e-5
    for k, ref in g["grads"].items():
        assert rel(_sample(sdg[k].grad, 4096), ref) < 1e-4, k
