"""The repo's own entry-point scripts (scripts/train_clip.py, predict_clip.py, train_caption.py = the loops of
/root/reference/CLIP/train.py:150-217, /root/reference/CLIP/predict.py:28-55, /root/reference/CLIP_prefix_caption/train.py:336-381
on the MI355X packages) run offline with --synthetic; the loss they print for their FIRST step is checked against the CPU oracle
evaluated on the same batch (same seeded weights, same generated images / captions, same stand-in tokenizer)."""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scripts")]


def _lines(capsys):
    return [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]


def test_train_clip_script_three_steps_and_first_loss_matches_oracle(tmp_path, capsys, monkeypatch):
    import _common as C
    import train_clip
    monkeypatch.setenv("CCLIP_COMPUTE_DTYPE", "bf16")
    n = train_clip.main(["--synthetic", "--model", "test-small", "--epochs", "1", "--max-steps", "3", "--out-dir", str(tmp_path),
                         "--save-every", "1", "--warmup-steps", "2"])
    assert n == 3
    lines = _lines(capsys)
    steps = [l for l in lines if "loss" in l]
    assert len(steps) == 3 and all(torch.isfinite(torch.tensor(l["loss"])) for l in steps)
    saved = [l["saved"] for l in lines if "saved" in l]
    assert saved and os.path.isfile(saved[0])
    sd = torch.load(saved[0], map_location="cpu", weights_only=True)
    assert "visual.conv1.weight" in sd and "logit_scale" in sd                     # OpenAI state_dict key layout
    # oracle on the first batch: rebuild the identical synthetic dataset (same seed -> same images, same shuffle is NOT needed:
    # one group of 9 classes per item and the loss of a group does not depend on which item - check the FIRST test-set group)
    import clip
    from clip.data import ClipPairDataset
    from clip.weights import MODELS, init_state_dict
    from oracle import clip_oracle as O
    d = tmp_path / "syn"
    js = C.make_synthetic_annotations(str(d))
    model, preprocess = clip.load("test-small", device="cuda:0")
    ds = ClipPairDataset(preprocess, js, str(d), 0.8, "violation_type", "train", 9, tokenize=C.get_tokenize(model))
    image, text = ds[0]
    with torch.no_grad():
        li, lt = model(image.cuda(), text.cuda())
    lab = torch.arange(9)
    got = (torch.nn.functional.cross_entropy(li.float().cpu(), lab) + torch.nn.functional.cross_entropy(lt.float().cpu(), lab)) / 2
    rli, rlt = O.clip_forward(init_state_dict(MODELS["test-small"], 567), image, text.long())
    ref, _ = O.contrastive_loss(rli, rlt)
    assert abs(float(got) - float(ref)) < 0.03, (float(got), float(ref))
    assert abs(steps[0]["loss"] - float(ref)) < 0.5       # the script's first (shuffled) group: same loss scale


def test_predict_clip_script(capsys, monkeypatch):
    import predict_clip
    monkeypatch.setenv("CCLIP_COMPUTE_DTYPE", "fp16")
    out = predict_clip.main(["--synthetic", "--model", "test-small"])
    assert len(out) == 16
    for r in out:
        assert r["label"] in ("violation", "status") and abs(sum(r["similarity"]) - 1.0) < 1e-3


def test_train_caption_script_three_steps(capsys):
    import train_caption
    n = train_caption.main(["--synthetic", "--gpt2", "test-tiny", "--epochs", "1", "--max-steps", "3", "--bs", "8", "--warmup_steps", "2"])
    assert n == 3
    steps = [l for l in _lines(capsys) if "loss" in l]
    assert len(steps) == 3 and steps[0]["loss"] > 0
    # the literal reference form (model(...) -> logits slice -> F.cross_entropy) gives the same first loss as the fused one
    n2 = train_caption.main(["--synthetic", "--gpt2", "test-tiny", "--epochs", "1", "--max-steps", "1", "--bs", "8", "--no-fused-loss"])
    assert n2 == 1
