"""KV-cached caption decoding (generate_beam / generate2, test.py:353-514) on the GPU against (a) the model's own full
forward - the cache must not change the arithmetic beyond kernel rounding - and (b) the CPU oracle, which re-runs GPT-2 on
the whole growing sequence every step exactly as the reference does."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Tok:                     # the reference passes a HF tokenizer; only encode / decode are used
    def encode(self, s):
        return [int(x) for x in s.split()]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def _model(seed=31, half=True):
    from clip_caption import ClipCaptionModel, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
    geo = GPT2_MODELS["test-tiny"]
    sd = init_caption_state_dict(geo, seed)
    model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    if half:
        model.half()
    tokens, mask, prefix, attribute = synthetic_caption_batch(1, geo, 6, seed + 1)
    return geo, sd, model, prefix, attribute


def _prefix_embed(model, geo, prefix, attribute):
    with torch.no_grad():
        pre = model.clip_project(prefix.cuda()).view(1, geo.prefix_length, geo.n_embd)
        return torch.cat((pre, model.gpt.transformer.wte(attribute.cuda())), dim=1)


@pytest.mark.parametrize("half", [False, True])
def test_kv_cache_equals_full_forward(half):
    from oracle import caption_oracle as CO
    geo, sd, model, prefix, attribute = _model(half=half)
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(2, 9, geo.n_embd, generator=g) * 0.1
    nxt = torch.randn(6, 2, 1, geo.n_embd, generator=g) * 0.1
    with torch.no_grad():
        out = model.gpt(inputs_embeds=emb.cuda(), use_cache=True)
        cache = out.past_key_values
        full = emb
        ref_full = model.gpt(inputs_embeds=emb.cuda()).logits
        assert (out.logits - ref_full).abs().max() < 1e-5           # prefill IS the ordinary forward
        for i in range(6):
            full = torch.cat((full, nxt[i]), dim=1)
            step = model.gpt(inputs_embeds=nxt[i].cuda(), past_key_values=cache)
            cache = step.past_key_values
            assert cache.length == full.shape[1]
            want = model.gpt(inputs_embeds=full.cuda()).logits[:, -1]
            oracle = CO.gpt2_forward(sd, full, None, geo.n_head)[:, -1]
            assert (step.logits[:, 0] - want).abs().max() < (2e-3 if half else 1.5e-2)
            assert (step.logits[:, 0].cpu() - oracle).abs().max() < (5e-3 if half else 5e-2)


def test_cache_reorder_and_expand():
    from clip_caption import KVCache
    c = KVCache(2, 3, 8, 128, "cuda", torch.float16)
    c.k.normal_(); c.v.normal_(); c.length = 5
    r = c.reorder(torch.tensor([2, 2, 0]))
    assert r.length == 5 and torch.equal(r.k[:, 0, :5], c.k[:, 2, :5]) and torch.equal(r.v[:, 2, :5], c.v[:, 0, :5])
    one = KVCache(2, 1, 8, 128, "cuda", torch.float16)
    one.k.normal_(); one.v.normal_(); one.length = 3
    e = one.expand(4)
    assert e.n_seq == 4 and all(torch.equal(e.k[:, i, :3], one.k[:, 0, :3]) for i in range(4))


def test_generate_beam_matches_oracle():
    from clip_caption import generate_beam
    from oracle import caption_oracle as CO
    geo, sd, model, prefix, attribute = _model()
    emb = _prefix_embed(model, geo, prefix, attribute)
    ref_emb = torch.cat((CO.mlp_mapper(sd, prefix).view(1, geo.prefix_length, geo.n_embd), sd["model.transformer.wte.weight"][attribute]), dim=1)
    assert (emb.cpu() - ref_emb).abs().max() < 2e-3
    texts, tokens, lengths, scores = generate_beam(model, _Tok(), beam_size=3, embed=emb, entry_length=12, stop_token=7, return_tokens=True)
    rt, rl, rs, trace = CO.generate_beam_tokens(sd, ref_emb, geo.n_head, beam_size=3, entry_length=12, stop_token=7)
    assert torch.equal(lengths.cpu(), rl)
    assert (scores.cpu() - rs).abs().max() < 2e-2
    # token-for-token wherever the oracle's own choice was decided by more than the fp16 logit tolerance
    order, rorder = scores.argsort(descending=True), rs.argsort(descending=True)
    assert torch.equal(tokens[order[0]].cpu(), rt[rorder[0]]), (tokens.cpu(), rt)
    assert texts[0] == " ".join(str(int(t)) for t in rt[rorder[0]][: int(rl[rorder[0]])])
    assert len(texts) == 3


def test_generate2_matches_oracle_and_stops():
    from clip_caption import generate2
    from oracle import caption_oracle as CO
    geo, sd, model, prefix, attribute = _model()
    emb = _prefix_embed(model, geo, prefix, attribute)
    ref_emb = torch.cat((CO.mlp_mapper(sd, prefix).view(1, geo.prefix_length, geo.n_embd), sd["model.transformer.wte.weight"][attribute]), dim=1)
    rt, trace = CO.generate2_tokens(sd, ref_emb, geo.n_head, entry_length=12, stop_token=26)
    text, tokens = generate2(model, _Tok(), embed=emb, entry_length=12, stop_token=26, return_tokens=True)
    assert torch.equal(tokens.cpu(), rt), (tokens.cpu(), rt)
    assert int(tokens[0, -1]) == 26 and tokens.shape[1] <= 12         # stopped on the stop token
    assert text == " ".join(str(int(t)) for t in rt[0])


def test_native_decode_driver_equals_python_launch_sequence(monkeypatch):
    """cclip_gpt2_decode_step (one native call per token; LayerNorm and the cache append fused into the skinny projections)
    against BlockStack.decode_step (stand-alone kernels launched one by one from Python): same arithmetic up to the last
    ulp of the LayerNorm statistics, so the 16-bit operands can differ in a rare rounding - compared at rounding level."""
    geo, sd, model, prefix, attribute = _model(half=False)
    g = torch.Generator().manual_seed(7)
    emb = torch.randn(3, 7, geo.n_embd, generator=g) * 0.1
    nxt = torch.randn(4, 3, 1, geo.n_embd, generator=g) * 0.1
    outs = {}
    for mode in ("python", "native"):
        monkeypatch.setenv("CCLIP_DECODE_DRIVER", mode)
        with torch.no_grad():
            cache = model.gpt(inputs_embeds=emb.cuda(), use_cache=True).past_key_values
            steps = []
            for i in range(4):
                o = model.gpt(inputs_embeds=nxt[i].cuda(), past_key_values=cache)
                cache = o.past_key_values
                steps.append(o.logits.clone())
        outs[mode] = (steps, cache.k[:, :, :cache.length].clone(), cache.v[:, :, :cache.length].clone())
    for a, b in zip(outs["python"][0], outs["native"][0]):
        assert (a - b).abs().max() < 5e-3, (a - b).abs().max()
    for i in (1, 2):
        assert (outs["python"][i].float() - outs["native"][i].float()).abs().max() < 2e-2


def test_generate_beam_and_generate2_at_gpt2_small_geometry():
    """The decode path at BASELINE configs[3]'s real geometry (GPT-2-small, V = 21128, 12 layers, prefix 20 + attribute 20):
    the KV-cached `generate_beam` (beam 3, temperature 0.5) and `generate2` (top-p 0.8) must produce the oracle's tokens,
    lengths and scores - the oracle re-runs the full forward on the growing sequence at every step, as the reference does
    (test.py:381, :489)."""
    from clip_caption import ClipCaptionModel, GPT2_MODELS, generate2, generate_beam, init_caption_state_dict, synthetic_caption_batch
    from oracle import caption_oracle as CO
    geo = GPT2_MODELS["ckiplab/gpt2-base-chinese"]
    sd = init_caption_state_dict(geo, 77)
    model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
    model.load_state_dict(sd)
    model = model.cuda().eval().half()
    _, _, prefix, attribute = synthetic_caption_batch(1, geo, 6, 78)
    emb = _prefix_embed(model, geo, prefix, attribute)
    ref_emb = torch.cat((CO.mlp_mapper(sd, prefix).view(1, geo.prefix_length, geo.n_embd), sd["model.transformer.wte.weight"][attribute]), dim=1)
    assert (emb.cpu() - ref_emb).abs().max() < 2e-3
    steps = 10
    texts, tokens, lengths, scores = generate_beam(model, _Tok(), beam_size=3, embed=emb, entry_length=steps, stop_token=102, return_tokens=True)
    rt, rl, rs, trace = CO.generate_beam_tokens(sd, ref_emb, geo.n_head, beam_size=3, entry_length=steps, stop_token=102)
    assert torch.equal(lengths.cpu(), rl)
    assert (scores.cpu() - rs).abs().max() < 2e-2
    order, rorder = scores.argsort(descending=True), rs.argsort(descending=True)
    assert torch.equal(tokens[order[0]].cpu(), rt[rorder[0]]), (tokens.cpu(), rt)
    rt2, _ = CO.generate2_tokens(sd, ref_emb, geo.n_head, entry_length=steps, stop_token=102)
    _, tok2 = generate2(model, _Tok(), embed=emb, entry_length=steps, stop_token=102, return_tokens=True)
    assert torch.equal(tok2.cpu(), rt2), (tok2.cpu(), rt2)


@pytest.mark.parametrize("half,beam,stop,grid_cap,use_prompt", [(True, 3, 7, 0, False), (False, 3, 7, 0, False), (True, 1, 7, 0, False),
                                                               (True, 5, 26, 0, False), (True, 3, -1, 8, False), (True, 8, 7, 3, False),
                                                               (True, 3, 7, 0, True)])
def test_persistent_beam_search_equals_host_loop(monkeypatch, half, beam, stop, grid_cap, use_prompt):
    """cclip_gpt2_beam_search (every decode step and every selection of a caption inside one persistent kernel; beam reorder
    through the cache-slot table) against the torch loop of clip_caption/generate.py, which is the reference's loop
    (test.py:380-434) op by op on the launch-by-launch decode step: same tokens inside every beam's length, same lengths, scores
    to rounding.  grid_cap exercises the strided phase loops with fewer workgroups than column blocks."""
    from clip_caption import generate_beam
    geo, sd, model, prefix, attribute = _model(half=half)
    emb = _prefix_embed(model, geo, prefix, attribute)
    kw = dict(beam_size=beam, entry_length=14, stop_token=stop, return_tokens=True)
    if use_prompt:
        kw["prompt"] = "5 9 11 3"
    else:
        kw["embed"] = emb
    monkeypatch.setenv("CCLIP_BEAM_NATIVE", "0")
    _, t0, l0, s0 = generate_beam(model, _Tok(), **kw)
    monkeypatch.setenv("CCLIP_BEAM_NATIVE", "1")
    assert model.beam_native_ok(beam)
    if grid_cap:
        gen = emb if not use_prompt else None
        t1, l1, s1 = model.beam_search_native(gen, beam, 14, 0.5, stop, grid_cap=grid_cap)
        s1 = s1 / l1
    else:
        _, t1, l1, s1 = generate_beam(model, _Tok(), **kw)
    assert torch.equal(l0.cpu(), l1.cpu()), (l0, l1)
    assert (s0 - s1).abs().max() < 2e-3, (s0, s1)
    n = min(t0.shape[1], t1.shape[1])
    npr = 4 if use_prompt else 0
    assert t1.shape[1] <= t0.shape[1]                     # the host loop looks for "all stopped" every 4th step only
    for b in range(beam):
        keep = npr + int(l1[b])
        assert torch.equal(t0[b, :min(keep, n)].cpu(), t1[b, :min(keep, n)].cpu()), (b, t0, t1)
    if stop == -1:
        assert t1.shape[1] == npr + 14                    # never stops: every selection made
