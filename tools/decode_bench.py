"""Caption decoding speed: KV-cached generate_beam (this repo) vs the reference's loop shape (full GPT-2 forward on the growing
sequence every step, test.py:381) run on the SAME kernels.  GPT-2-small geometry of ckiplab/gpt2-base-chinese, prefix 20 +
attribute 20 tokens, beam 3, 67 new tokens, synthetic weights (never emits the stop token)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from clip_caption import ClipCaptionModel, GPT2_MODELS, generate_beam, init_caption_state_dict, synthetic_caption_batch


class Tok:
    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


@torch.no_grad()
def full_recompute_beam(model, embed, beam, steps):
    """the reference's loop shape: whole sequence through GPT-2 at every step; greedy per beam (selection cost is identical)"""
    gen = embed.expand(beam, *embed.shape[1:]).contiguous()
    for _ in range(steps):
        logits = model.gpt(inputs_embeds=gen).logits[:, -1]
        nxt = logits.argmax(-1)
        gen = torch.cat((gen, model.gpt.transformer.wte(nxt).view(beam, 1, -1)), dim=1)
    return gen


geo = GPT2_MODELS["ckiplab/gpt2-base-chinese"]
model = ClipCaptionModel(geo.prefix_length, prefix_size=geo.prefix_size, gpt2_type=geo)
model.load_state_dict(init_caption_state_dict(geo, 567))
model = model.cuda().eval()
tokens, mask, prefix, attribute = [t.cuda() for t in synthetic_caption_batch(1, geo, 40, 568)]
with torch.no_grad():
    emb = torch.cat((model.clip_project(prefix).view(1, geo.prefix_length, geo.n_embd), model.gpt.transformer.wte(attribute)), dim=1)
steps, beam = 67, 3
for name, fn in (("kv-cached generate_beam", lambda: generate_beam(model, Tok(), beam_size=beam, embed=emb, entry_length=steps, stop_token=-1)),
                 ("full recompute per step", lambda: full_recompute_beam(model, emb, beam, steps))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name:26s}: {dt * 1e3:8.1f} ms per caption ({steps} steps x {beam} beams) = {steps / dt:7.1f} steps/s", flush=True)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# the decode steps alone: a 67-selection caption minus a 1-selection caption (prefill + first selection), per step
t_full = timed(lambda: generate_beam(model, Tok(), beam_size=beam, embed=emb, entry_length=steps, stop_token=-1))
t_one = timed(lambda: generate_beam(model, Tok(), beam_size=beam, embed=emb, entry_length=1, stop_token=-1))
print(f"prefill + first selection  : {t_one * 1e3:8.2f} ms;  decode step (step + selection, {beam} beams): {(t_full - t_one) / (steps - 1) * 1e3:6.3f} ms", flush=True)
