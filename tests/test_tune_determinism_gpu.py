"""The GEMM tile autotuner is pinned by a persisted table (cclip_hip/ops.py): a second process that loads the table the
first one wrote tunes nothing and produces bitwise the same loss and gradients."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "construction-clip_amd")]
import torch
import clip
from cclip_hip import ops
from clip.weights import MODELS, init_state_dict, synthetic_text
geo = MODELS["ViT-B/32"]
model = clip.build_model(init_state_dict(geo, 567)).cuda().train()
B = 64
img = torch.randn(B, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
txt = synthetic_text(B, geo, 2).cuda()
li, lt = model(img, txt)
lab = torch.arange(B, device="cuda")
loss = (torch.nn.functional.cross_entropy(li, lab) + torch.nn.functional.cross_entropy(lt, lab)) / 2
loss.backward()
torch.cuda.synchronize()
ar = model.arena
o, n = ar.offsets["token_embedding.weight"], ar.params["token_embedding.weight"].numel()
gf = ar.gflat.clone(); gf[o:o + n] = 0      # token_embedding.weight is summed with fp32 atomics (hardware order): checked to rounding below
h = hashlib.sha256(gf.cpu().numpy().tobytes()).hexdigest()
te = float(ar.gflat[o:o + n].double().norm())
print(json.dumps(dict(loss=loss.item().hex(), grads=h, te=te, misses=ops._TUNE_STATE["misses"], entries=len(ops._TUNED))))
"""


def _run(env):
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_two_fresh_processes_share_the_tuned_table_and_agree_bitwise(tmp_path):
    env = dict(os.environ, CCLIP_TUNE_FILE=str(tmp_path / "tune.json"))
    first = _run(env)
    table = json.load(open(env["CCLIP_TUNE_FILE"]))
    assert len(table["table"]) >= first["misses"] > 0 or first["misses"] == 0     # (0: the committed table already covers it)
    second = _run(env)
    assert second["misses"] == 0, second                   # nothing left to time: launch-for-launch the same program
    assert second["loss"] == first["loss"] and second["grads"] == first["grads"], (first, second)
    assert abs(second["te"] - first["te"]) <= 1e-6 * first["te"]          # the atomically summed embedding gradient: to rounding
