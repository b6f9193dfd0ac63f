"""diagnostic (GPU box): where does the +1.7 % norm of the mapper's second-layer weight gradient (full caption step, real
geometry) come from?  Compares with the CPU oracle per prefix position."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from clip_caption import ClipCaptionModel, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
from oracle import caption_oracle as CO
geo = GPT2_MODELS["ckiplab/gpt2-base-chinese"]
sd = init_caption_state_dict(geo, 31)
tokens, mask, prefix, attribute = synthetic_caption_batch(2, geo, 40, 32)
sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "model.lm_head.weight"}
sdg["model.lm_head.weight"] = sdg["model.transformer.wte.weight"]
emb_hook = {}
logits = CO.caption_forward(sdg, tokens, prefix, attribute, mask, geo.prefix_length, geo.n_head)
loss = CO.caption_loss(logits, tokens, geo.prefix_length, geo.attribute_length)
loss.backward()
ref = sdg["clip_project.model.2.weight"].grad
refb = sdg["clip_project.model.2.bias"].grad
model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
model.load_state_dict(sd)
model = model.cuda().train()
l2 = model.caption_loss(*[t.cuda() for t in (tokens, prefix, attribute, mask)])
l2.backward()
got = dict(model.named_parameters())["clip_project.model.2.weight"].grad.cpu()
gotb = dict(model.named_parameters())["clip_project.model.2.bias"].grad.cpu()
print("loss", loss.item(), l2.item())
print("W2 rel", ((got - ref).norm() / ref.norm()).item(), "norm ratio", (got.norm() / ref.norm()).item())
print("b2 rel", ((gotb - refb).norm() / refb.norm()).item(), "norm ratio", (gotb.norm() / refb.norm()).item())
P, D = geo.prefix_length, geo.n_embd
for p in range(P):
    a, b = got[p * D:(p + 1) * D], ref[p * D:(p + 1) * D]
    ab, bb = gotb[p * D:(p + 1) * D], refb[p * D:(p + 1) * D]
    print(f"prefix pos {p:2d}: W2 ratio {(a.norm() / b.norm()).item():.4f} rel {((a - b).norm() / b.norm()).item():.4f} | "
          f"b2 ratio {(ab.norm() / bb.norm()).item():.4f} rel {((ab - bb).norm() / bb.norm()).item():.4f}")
cols = ((got - ref).norm(dim=0) / ref.norm(dim=0).clamp_min(1e-30))
print("worst hidden columns by rel err", cols.topk(5))
