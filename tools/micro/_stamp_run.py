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
st = ws[45000:45000+64].view(torch.int64).cpu().tolist()
names = ["weights issued","wait done","A staged","sync","FMA done","reduce done","stores issued","arrive done (sync+atomic)"]
for ph, off in (("P3 (A from 16-bit rows, K=768)", 0), ("P4 (LayerNorm, K=768, N=3072)", 16)):
    v = st[off:off+8]
    print(ph)
    for i in range(1, 8):
        print(f"   {names[i]:28s} +{v[i]-v[i-1]:7d} ticks")
