"""Host logic of the KV-cached decoders (clip_caption/generate.py) on the CPU: with a stub `model.gpt` whose logits come from the
oracle's GPT-2 (full forward on the concatenated sequence - the cache object only carries the embeddings seen so far), generate_beam
and generate2 must reproduce the oracle decoders, which restate the reference's loops (test.py:353-514), token for token."""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]


class _Tok:
    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


class _SeqCache:
    """stands in for KVCache: remembers every sequence's embeddings; reorder / expand as beam search needs them"""

    def __init__(self, emb):
        self.emb = emb

    def expand(self, n):
        return _SeqCache(self.emb.expand(n, *self.emb.shape[1:]).clone())

    def reorder(self, idx):
        return _SeqCache(self.emb[idx.long()])


class _StubModel:
    def __init__(self, sd, n_head):
        from oracle import caption_oracle as CO
        self.sd, self.n_head, self.CO = sd, n_head, CO
        wte = sd["model.transformer.wte.weight"]
        self.gpt = SimpleNamespace(transformer=SimpleNamespace(wte=lambda ids: wte[ids]))
        self.gpt.__call__ = None

        def call(inputs_embeds=None, past_key_values=None, use_cache=False, **kw):
            seq = inputs_embeds if past_key_values is None else torch.cat((past_key_values.emb, inputs_embeds), dim=1)
            logits = CO.gpt2_forward(self.sd, seq, None, self.n_head)
            return SimpleNamespace(logits=logits[:, -inputs_embeds.shape[1]:], past_key_values=_SeqCache(seq))
        self.gpt = type("G", (), {"__call__": staticmethod(call), "transformer": self.gpt.transformer})()

    def eval(self):
        return self

    def parameters(self):
        yield self.sd["model.transformer.wte.weight"]


def _setup():
    from clip_caption.weights import GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
    from oracle import caption_oracle as CO
    geo = GPT2_MODELS["test-tiny"]
    sd = init_caption_state_dict(geo, 31)
    _, _, prefix, attribute = synthetic_caption_batch(1, geo, 6, 32)
    emb = torch.cat((CO.mlp_mapper(sd, prefix).view(1, geo.prefix_length, geo.n_embd), sd["model.transformer.wte.weight"][attribute]), dim=1)
    return geo, sd, emb, CO


def test_generate_beam_host_logic_equals_oracle():
    from clip_caption.generate import generate_beam
    geo, sd, emb, CO = _setup()
    model = _StubModel(sd, geo.n_head)
    texts, tokens, lengths, scores = generate_beam(model, _Tok(), beam_size=3, embed=emb, entry_length=10, stop_token=7, return_tokens=True)
    rt, rl, rs, _ = CO.generate_beam_tokens(sd, emb, geo.n_head, beam_size=3, entry_length=10, stop_token=7)
    assert torch.equal(tokens, rt) and torch.equal(lengths, rl) and torch.allclose(scores, rs, atol=1e-6)
    order = rs.argsort(descending=True)
    assert texts == [" ".join(str(int(t)) for t in rt[i][: int(rl[i])]) for i in order]


def test_generate_beam_stops_early_and_pads_stopped_beams():
    from clip_caption.generate import generate_beam
    geo, sd, emb, CO = _setup()
    rt, rl, rs, _ = CO.generate_beam_tokens(sd, emb, geo.n_head, beam_size=3, entry_length=12, stop_token=251)   # 251 appears early for this seed
    model = _StubModel(sd, geo.n_head)
    texts, tokens, lengths, scores = generate_beam(model, _Tok(), beam_size=3, embed=emb, entry_length=12, stop_token=251, return_tokens=True)
    assert torch.equal(lengths, rl) and torch.allclose(scores, rs, atol=1e-6)
    n = min(tokens.shape[1], rt.shape[1])              # the all-stopped check runs every 4th step: extra columns are padding zeros
    assert torch.equal(tokens[:, :n], rt[:, :n]) and (tokens[:, n:] == 0).all()


def test_generate2_host_logic_equals_oracle():
    from clip_caption.generate import generate2
    geo, sd, emb, CO = _setup()
    model = _StubModel(sd, geo.n_head)
    rt, _ = CO.generate2_tokens(sd, emb, geo.n_head, entry_length=10, stop_token=26)
    text, tokens = generate2(model, _Tok(), embed=emb, entry_length=10, stop_token=26, return_tokens=True)
    assert torch.equal(tokens, rt) and text == " ".join(str(int(t)) for t in rt[0])


def test_kv_cache_reorder_expand_cpu():
    from clip_caption import KVCache
    c = KVCache(2, 3, 8, 16, "cpu", torch.float32)
    c.k.normal_(); c.v.normal_(); c.length = 5
    r = c.reorder(torch.tensor([2, 0, 0]))
    assert r.length == 5 and torch.equal(r.k[:, 1, :5], c.k[:, 0, :5]) and torch.equal(r.v[:, 0, :5], c.v[:, 2, :5])
    one = KVCache(2, 1, 8, 16, "cpu", torch.float32)
    one.k.normal_(); one.length = 3
    assert one.expand(4).n_seq == 4
