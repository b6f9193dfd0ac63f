import os, sys, torch
ROOT="/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from clip_caption import ClipCaptionModel, GPT2_MODELS, init_caption_state_dict, synthetic_caption_batch
from cclip_hip import ops
geo = GPT2_MODELS["ckiplab/gpt2-base-chinese"]
model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
model.load_state_dict(init_caption_state_dict(geo, 567))
model = model.cuda().eval()
tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(1, geo, 40, 568)]
with torch.no_grad():
    emb = torch.cat((model.clip_project(prefix).view(1, geo.prefix_length, geo.n_embd), model.gpt.transformer.wte(attribute)), dim=1)
keep = {}
orig = ops.BeamState.__init__
def init(self, *a, **k):
    orig(self, *a, **k); keep['st'] = self
ops.BeamState.__init__ = init
for _ in range(2):
    model.beam_search_native(emb, 3, 20, 0.5, -1)
torch.cuda.synchronize()
ws = keep['st'].select_ws
st = ws[45000:45000+256].view(torch.int64).cpu().tolist()
for wg, off in (("WG0", 0), ("WG20", 64)):
    v = st[off:off+64]
    ev = sorted((t, i) for i, t in enumerate(v) if t)
    print(wg)
    for (t, i), (t0, i0) in zip(ev[1:], ev[:-1]):
        print(f"   stamp {i0:2d} -> {i:2d}: {t - t0:8d} ticks")
