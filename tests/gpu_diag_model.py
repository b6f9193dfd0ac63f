"""Diagnostic (not a pytest): print HIP-vs-golden error metrics for the CLIP path on a GPU box.
    python tests/gpu_diag_model.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]

import clip  # noqa: E402
from clip.weights import MODELS, init_state_dict, synthetic_images  # noqa: E402


def sample(t, keep=4096):
    f = t.detach().flatten()
    k = max(1, -(-f.numel() // keep))
    return f[::k].clone()


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item(), ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def run(fix, dtype=torch.bfloat16):
    g = torch.load(os.path.join(ROOT, "tests", "golden", fix), weights_only=True)
    geo = MODELS[g["model"]]
    sd = init_state_dict(geo, g["seed"])
    model = clip.build_model(sd, dtype).cuda()
    img = synthetic_images(g["n"], geo, g["seed"] + 1).cuda()
    txt = g["text"].cuda()
    with torch.no_grad():
        fi, ft = model.encode_image(img), model.encode_text(txt)
        li, lt = model(img, txt)
    torch.cuda.synchronize()
    print(f"== {fix} ({g['model']}, n={g['n']}) compute dtype {dtype}")
    print("  image_features rel(l2,max):", rel(fi, g["image_features"]))
    print("  text_features  rel(l2,max):", rel(ft, g["text_features"]))
    d = (li.cpu() - g["logits_per_image"]).abs()
    print("  logits abs err max %.4g  (logit range %.3g..%.3g)" % (d.max(), g["logits_per_image"].min(), g["logits_per_image"].max()))
    print("  argmax rows equal:", torch.equal(li.argmax(1).cpu(), g["logits_per_image"].argmax(1)),
          " cols equal:", torch.equal(li.argmax(0).cpu(), g["logits_per_image"].argmax(0)))
    with torch.no_grad():
        l2, _ = model(img, txt[:2])
    print("  zero-shot 2: idx equal", torch.equal(l2.softmax(-1).argmax(1).cpu(), g["zs2_idx"]),
          " sim err %.3g" % (l2.softmax(-1).cpu() - g["zs2_sim"]).abs().max())
    if "grads" in g:
        model.train()
        model.zero_grad()
        li, lt = model(img, txt)
        lab = torch.arange(li.shape[0], device="cuda")
        loss = (torch.nn.functional.cross_entropy(li, lab) + torch.nn.functional.cross_entropy(lt, lab)) / 2
        loss.backward()
        torch.cuda.synchronize()
        print("  loss %.6f golden %.6f" % (loss.item(), g["loss"].item()))
        params = dict(model.named_parameters())
        for k, ref in g["grads"].items():
            got = params[k].grad
            print("  grad %-55s rel(l2,max) %s" % (k, "MISSING" if got is None else "%.3e %.3e" % rel(sample(got), ref)))
        worst = 0
        for k, nrm in g["grad_norms"].items():
            got = params[k].grad
            r = abs(got.norm().item() - nrm.item()) / max(nrm.item(), 1e-12)
            worst = max(worst, r)
            if r > 0.05:
                print("  NORM MISMATCH", k, got.norm().item(), nrm.item())
        print("  worst grad-norm rel diff over all %d params: %.3e" % (len(g["grad_norms"]), worst))
        # fused loss path
        model.zero_grad()
        fi, ft = model.encode_image(img), model.encode_text(txt)
        loss2, stats = clip.contrastive_loss(fi, ft, model.logit_scale)
        loss2.backward()
        torch.cuda.synchronize()
        print("  fused loss %.6f correct %d (golden acc %.3f)" % (loss2.item(), int(stats[1].item()), g["acc"].item()))
        for k in ("logit_scale", "visual.proj", "transformer.resblocks.0.attn.out_proj.weight"):
            print("  fused grad %-49s rel(l2,max) %.3e %.3e" % ((k,) + rel(sample(params[k].grad), g["grads"][k])))


if __name__ == "__main__":
    t0 = time.time()
    for f, dt in [(f, dt) for dt in (torch.float16, torch.bfloat16) for f in ("clip_test_tiny.pt", "clip_test_small.pt", "clip_vit_b32.pt")]:
        try:
            run(f, dt)
        except Exception as e:  # keep going: one run should tell as much as possible
            import traceback
            traceback.print_exc()
            print("FAILED", f, e)
    print("done in %.1fs" % (time.time() - t0))
