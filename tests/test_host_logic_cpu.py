"""CPU: host-side logic of the `clip` drop-in that needs no GPU - state_dict layout, geometry inference,
tokenizer mechanics, preprocess, schedule/optimizer bookkeeping, and the 'no CPU fallback' rule."""
import gzip
import os

import numpy as np
import pytest
import torch

import clip
from clip.weights import MODELS, init_state_dict, geometry_from_state_dict


def test_state_dict_layout_is_openai_clip():
    geo = MODELS["ViT-B/32"]
    model = clip.CLIP(geo)
    sd = model.state_dict()
    assert sum(v.numel() for v in sd.values()) == 151_277_313
    assert len(sd) == 302
    for k, shape in {"visual.conv1.weight": (768, 3, 32, 32), "visual.class_embedding": (768,),
                     "visual.positional_embedding": (50, 768), "visual.proj": (768, 512),
                     "visual.transformer.resblocks.11.attn.in_proj_weight": (2304, 768),
                     "visual.transformer.resblocks.0.attn.out_proj.bias": (768,),
                     "visual.transformer.resblocks.3.mlp.c_fc.weight": (3072, 768),
                     "visual.transformer.resblocks.3.mlp.c_proj.weight": (768, 3072),
                     "token_embedding.weight": (49408, 512), "positional_embedding": (77, 512),
                     "transformer.resblocks.11.ln_2.bias": (512,), "ln_final.weight": (512,),
                     "text_projection": (512, 512), "logit_scale": ()}.items():
        assert tuple(sd[k].shape) == shape, k
    assert set(sd) == set(init_state_dict(geo, 1))


def test_build_model_infers_geometry_and_roundtrips(tmp_path):
    for name in ("test-tiny", "test-small"):
        sd = init_state_dict(MODELS[name], 3)
        assert geometry_from_state_dict(sd) == MODELS[name]
        m = clip.build_model({k: v.half() for k, v in sd.items()})       # fp16 checkpoints (reference CUDA) load too
        assert m.dtype == torch.float32
        path = tmp_path / "ck.pt"
        torch.save(m.state_dict(), path)                                  # CLIP/train.py:213-217
        m2, pre = clip.load(str(path), device="cpu")
        for k, v in m.state_dict().items():
            assert torch.equal(v, m2.state_dict()[k])


def test_no_cpu_fallback():
    m = clip.build_model(init_state_dict(MODELS["test-tiny"], 3))
    with pytest.raises(RuntimeError, match="HIP"):
        m.encode_image(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 16, dtype=torch.int32))


def test_load_unknown_and_jit():
    with pytest.raises(RuntimeError, match="not found"):
        clip.load("RN50-nope", device="cpu")
    with pytest.raises(RuntimeError, match="jit"):
        clip.load("ViT-B/32", device="cpu", jit=True)
    assert "ViT-B/32" in clip.available_models()


def _toy_bpe(path):
    # a tiny merges file in the format of bpe_simple_vocab_16e6.txt.gz (header line + "a b" merges)
    merges = ["#version: toy", "h e", "l l", "he ll", "hell o</w>", "w o", "r l", "wo rl", "worl d</w>"]
    with gzip.open(path, "wb") as f:
        f.write("\n".join(merges).encode("utf-8"))


def test_tokenizer_mechanics(tmp_path, monkeypatch):
    from clip.simple_tokenizer import SimpleTokenizer
    p = str(tmp_path / "toy.txt.gz")
    _toy_bpe(p)
    tok = SimpleTokenizer(p)
    assert len(tok.encoder) == 512 + 8 + 2
    ids = tok.encode("Hello  WORLD")
    assert [tok.decoder[i] for i in ids] == ["hello</w>", "world</w>"]       # lower-cased, whitespace cleaned, merged
    assert tok.decode(ids) == "hello world "
    zh = tok.encode("墜落")                                                   # no merges -> 3 byte tokens per character
    assert len(zh) == 6 and tok.decode(zh).strip() == "墜落"
    monkeypatch.setenv("CCLIP_BPE_PATH", p)
    import clip.clip as cc
    monkeypatch.setattr(cc, "_tokenizer", None)
    t = clip.tokenize(["hello world", "墜落"], context_length=12)
    sot, eot = tok.encoder["<|startoftext|>"], tok.encoder["<|endoftext|>"]
    assert t.dtype == torch.int32 and t.shape == (2, 12)
    assert t[0, 0] == sot and t[0, 3] == eot and t[0, 4:].sum() == 0
    assert int(t[1].argmax()) == 7                                            # EOT is the largest id -> pooling index
    with pytest.raises(RuntimeError, match="too long"):
        clip.tokenize("墜落墜落墜落墜落", context_length=12)
    tt = clip.tokenize("墜落墜落墜落墜落", context_length=12, truncate=True)
    assert tt[0, -1] == eot


def test_tokenize_without_vocab_explains(monkeypatch, tmp_path):
    import clip.clip as cc
    monkeypatch.setattr(cc, "_tokenizer", None)
    monkeypatch.setenv("CCLIP_BPE_PATH", str(tmp_path / "missing.gz"))
    with pytest.raises(FileNotFoundError, match="CCLIP_BPE_PATH"):
        clip.tokenize("x")


def test_preprocess_matches_definition():
    from PIL import Image
    rng = np.random.RandomState(0)
    arr = rng.randint(0, 256, (300, 420, 3), dtype=np.uint8)
    img = Image.fromarray(arr)
    pre = clip._transform(224)
    out = pre(img)
    assert out.shape == (3, 224, 224) and out.dtype == torch.float32
    # manual: resize shorter side to 224 (bicubic), centre crop, /255, normalise
    r = img.resize((int(224 * 420 / 300), 224), Image.BICUBIC)
    left = int(round((r.size[0] - 224) / 2.0))
    c = np.asarray(r.crop((left, 0, left + 224, 224)).convert("RGB"), dtype=np.float32) / 255.0
    ref = (c - np.array(pre.MEAN, dtype=np.float32)) / np.array(pre.STD, dtype=np.float32)
    assert np.allclose(out.numpy(), ref.transpose(2, 0, 1), atol=1e-6)
    # an already-224 grey image goes through untouched
    g = Image.fromarray(np.full((224, 224), 128, dtype=np.uint8))
    o2 = pre(g)
    assert torch.allclose(o2[0], torch.full((224, 224), (128 / 255 - pre.MEAN[0]) / pre.STD[0]), atol=1e-6)


def test_linear_schedule_and_bucket_logic():
    from clip.optim import LinearWarmupSchedule
    from oracle.optim_oracle import linear_schedule

    class Dummy:
        param_groups = [dict(lr=2e-5)]
    s = LinearWarmupSchedule(Dummy(), 5, 30)
    for step in range(35):
        assert abs(Dummy.param_groups[0]["lr"] - 2e-5 * linear_schedule(step, 5, 30)) < 1e-15
        s.step()
    from cclip_hip.arena import ParamArena
    from clip.parallel import grad_buckets
    m = clip.CLIP(MODELS["test-tiny"]).initialize_parameters(1)
    ar = ParamArena(m, torch.device("cpu"))
    assert ar.intact() and ar.total % 64 == 0
    b = grad_buckets(ar, max_bucket_elems=200_000)
    assert b[0][0] == 0 and b[-1][1] == ar.total and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    # parameters are views of one flat buffer: an in-place change of the flat buffer is visible in the module
    ar.flat.add_(1.0)
    assert torch.equal(m.logit_scale.detach(), ar.flat[ar.offsets["logit_scale"]].reshape(()))


def test_caption_checkpoint_with_hf_attention_buffers_loads_strict():
    """A ClipCaptionModel state_dict as older transformers 4.x writes it (persistent `attn.bias` / `attn.masked_bias`
    causal-mask buffers, CLIP_prefix_caption/train.py:320 strict load) and one without the tied lm_head both load."""
    from clip_caption import ClipCaptionModel
    from clip_caption.weights import GPT2_MODELS, init_caption_state_dict
    geo = GPT2_MODELS["test-tiny"]
    sd = init_caption_state_dict(geo, 5)
    old = dict(sd)
    for i in range(geo.n_layer):
        old[f"model.transformer.h.{i}.attn.bias"] = torch.tril(torch.ones(geo.n_positions, geo.n_positions)).view(1, 1, geo.n_positions, -1)
        old[f"model.transformer.h.{i}.attn.masked_bias"] = torch.tensor(-1e4)
    new = {k: v for k, v in sd.items() if k != "model.lm_head.weight"}
    for variant in (old, new, sd):
        m = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
        res = m.load_state_dict(variant)                    # strict=True
        assert not res.missing_keys and not res.unexpected_keys
        got = m.state_dict()
        assert set(got) == set(sd)
        for k in sd:
            assert torch.equal(got[k].cpu(), sd[k]), k
    assert len(old) == len(sd) + 2 * geo.n_layer            # the caller's dict is left untouched


def test_gemm_tune_table_roundtrip_and_source_hash(tmp_path, monkeypatch):
    """The persisted autotune table: JSON round trip, ignored when made for other kernel sources, env file wins."""
    import json
    from cclip_hip import ops
    saved = dict(ops._TUNED)
    try:
        ops._TUNED.clear()
        k = ops._key_str(("bfloat16", 51200, 2304, 768, True, True, 0, False, True, False, False, True, 1, False, False))
        assert k == "bfloat16|51200|2304|768|1|1|0|0|1|0|0|1|1|0|0"
        ops._TUNED[k] = (3, 1)
        path = str(tmp_path / "tune.json")
        assert ops.save_tuned_table(path) == path
        blob = json.load(open(path))
        assert blob["kernel_source_hash"] == ops.kernel_source_hash() and blob["table"][k] == [3, 1]
        ops._TUNED.clear()
        assert ops.load_tuned_table(path) == 1 and ops._TUNED[k] == (3, 1)
        blob["kernel_source_hash"] = "0" * 16                       # a table for other kernels is not used
        json.dump(blob, open(path, "w"))
        ops._TUNED.clear()
        assert ops.load_tuned_table(path) == 0 and not ops._TUNED
    finally:
        ops._TUNED.clear()
        ops._TUNED.update(saved)


def test_gemm_tile_lookup_takes_the_nearest_token_count(monkeypatch):
    """A GEMM the table holds at another token count (a packed text batch, a last partial batch) takes that entry instead of being
    timed inside the step: nearest in the token dimension (rows of a forward-layout GEMM, the contraction length of a weight
    gradient, whose split-K count is scaled with it); other shapes, layouts or epilogues never match; the persistent streaming
    configuration is only handed to full 256-row tiles; CCLIP_TUNE_EXACT=1 switches the fallback off; derived entries are not saved."""
    from cclip_hip import ops
    saved, saved_d = dict(ops._TUNED), set(ops._DERIVED)
    try:
        ops._TUNED.clear(); ops._DERIVED.clear()
        fwd = "bfloat16|%d|1536|512|1|1|0|0|1|0|0|1|1|0|0"
        wg = "bfloat16|1536|512|%d|0|0|0|1|0|0|0|0|-1|1|0"
        ops._TUNED[fwd % 78848] = (7, 1)
        ops._TUNED[fwd % 40311] = (3, 1)
        ops._TUNED[wg % 78848] = (2, 10)
        ops._TUNED["bfloat16|51200|768|768|1|1|0|1|0|0|1|1|1|0|0"] = (4, 1)
        assert ops._nearest_tuned(fwd % 45000) == (3, 1)
        assert ops._nearest_tuned(fwd % 70000) == (7, 1)
        assert ops._nearest_tuned(wg % 39424) == (2, 5)                                   # same K-tiles per split
        assert ops._nearest_tuned(wg % 8000) == (2, 1)
        assert ops._nearest_tuned(wg % 4000) is None                                       # more than 16x away: tuned on its own
        assert ops._nearest_tuned("bfloat16|45000|1536|512|1|0|0|0|1|0|0|1|1|0|0") is None    # another layout
        assert ops._nearest_tuned("bfloat16|45000|1536|768|1|1|0|0|1|0|0|1|1|0|0") is None    # another K
        assert ops._nearest_tuned("bfloat16|51201|768|768|1|1|0|1|0|0|1|1|1|0|0") == (3, 1)   # configuration 4 needs M % 256 == 0
        assert ops._nearest_tuned("bfloat16|25600|768|768|1|1|0|1|0|0|1|1|1|0|0") == (4, 1)
        # the tile ORDER rides in bits 8..15 of the configuration: handed on with the entry, and kept when the hand-scheduled
        # configuration itself has to be replaced (a contraction that is not whole 64-deep K-tiles)
        fc = "bfloat16|%d|3072|%d|1|1|0|0|1|0|0|1|1|0|0"
        ops._TUNED[fc % (51200, 768)] = (8 + 256 * 4, 1)
        ops._TUNED[fc % (51200, 200)] = (8 + 256 * 6, 1)
        assert ops._nearest_tuned(fc % (25600, 768)) == (8 + 256 * 4, 1)
        assert ops._nearest_tuned(fc % (25600, 200)) == (3 + 256 * 6, 1)
        wg11 = "bfloat16|768|3072|%d|0|0|0|1|0|0|0|0|-1|1|0"
        ops._TUNED[wg11 % 51200] = (11, 7)
        assert ops._nearest_tuned(wg11 % 25600) == (11, 4)                                  # scaled split, still whole K-tiles
        assert ops._nearest_tuned(wg11 % 25000) == (2, 3)                                   # 25000 % 64 != 0: the 8-wave kernel
        monkeypatch.setenv("CCLIP_TUNE_EXACT", "1")
        assert ops._nearest_tuned(fwd % 45000) is None
    finally:
        ops._TUNED.clear(); ops._TUNED.update(saved)
        ops._DERIVED.clear(); ops._DERIVED.update(saved_d)


def test_gemm_tile_lookup_stays_flat_over_many_row_counts():
    """A run that meets a new packed row count every step (shuffled captions) must not grow the tuned table nor slow its lookups
    down: derived choices live in a bounded cache, the table holds timed / persisted entries only, and the nearest-token search is
    a bisect over a per-family index (no scan, no string split per table entry)."""
    import time
    from cclip_hip import ops
    saved, saved_d = dict(ops._TUNED), dict(ops._DERIVED)
    try:
        ops._TUNED.clear(); ops._DERIVED.clear()
        fams = []
        for n, k in ((1536, 512), (512, 512), (2048, 512), (512, 2048), (2304, 768), (768, 768), (3072, 768), (768, 3072)):
            fam = f"bfloat16|%d|{n}|{k}|1|1|0|0|1|0|0|1|1|0|0"
            fams.append(fam)
            for m in (20000, 40311, 78848):
                ops._TUNED[fam % m] = (8, 1)
        base = len(ops._TUNED)

        def lookups(ms):
            t0 = time.perf_counter()
            for m in ms:
                for fam in fams:
                    key = fam % m
                    c = ops._TUNED.get(key) or ops._DERIVED.get(key)
                    if c is None:
                        c = ops._nearest_tuned(key)
                        assert c == (8, 1)
                        ops._DERIVED.put(key, c)
            return (time.perf_counter() - t0) / (len(ms) * len(fams))

        first = lookups(range(30000, 31000))
        lookups(range(31000, 39000))
        last = lookups(range(39000, 40000))                       # 10 000 distinct row counts x 8 families later
        assert len(ops._TUNED) == base                            # the table did not grow
        assert len(ops._DERIVED) <= ops._DERIVED.cap              # the derived cache is bounded
        assert last < 3 * first + 2e-5, (first, last)             # per-call cost is flat (it was linear in the keys met)
        assert last < 2e-4
    finally:
        ops._TUNED.clear(); ops._TUNED.update(saved)
        ops._DERIVED.clear(); ops._DERIVED.update(saved_d)


def test_duet_takes_strict_turns_and_propagates_errors():
    """cclip_hip/duet.py: two launch sequences in two threads alternate exactly at their interleave points (the first one
    starts), a paused party lets the other run on, results come back in order and an exception of either side is re-raised."""
    from cclip_hip import duet
    log = []

    def seq(tag, n, pause_at=None):
        def f():
            for i in range(n):
                duet.interleave_point()
                if i == pause_at:
                    duet.pause()
                    import time
                    time.sleep(0.05)                # (a blocking read-back: the other side keeps going meanwhile)
                    duet.resume()
                log.append((tag, i))
            return tag
        return f

    assert duet.run(seq("a", 3), seq("b", 5)) == ("a", "b")
    assert log[:6] == [("a", 0), ("b", 0), ("a", 1), ("b", 1), ("a", 2), ("b", 2)] and log[6:] == [("b", 3), ("b", 4)]
    log.clear()
    duet.run(seq("a", 4), seq("b", 2, pause_at=0))
    assert [t for t in log if t[0] == "a"] == [("a", i) for i in range(4)] and log.index(("a", 3)) < log.index(("b", 0))
    duet.interleave_point(); duet.pause(); duet.resume()            # no-ops outside run()

    def boom():
        duet.interleave_point()
        raise ValueError("from the helper thread")
    import pytest
    with pytest.raises(ValueError):
        duet.run(seq("a", 3), boom)
    with pytest.raises(ValueError):
        duet.run(boom, seq("b", 3))
