"""Probe: encode_image on one 1024-image batch vs. two 512-image halves on two HIP streams (inference).  If independent halves
running side by side fill each other's tile tails / epilogue bubbles, the split run is faster - the case for splitting the image
tower of the train step across two streams now that the (packed) text tower only overlaps a quarter of it."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
import clip
from clip.weights import MODELS, init_state_dict
geo = MODELS["ViT-B/32"]
model = clip.build_model(init_state_dict(geo, 567)).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(1)
img = torch.randn(1024, 3, 224, 224, device="cuda", generator=g)
s = [torch.cuda.Stream() for _ in range(4)]


def full():
    with torch.no_grad():
        return model.encode_image(img)


def split(n):
    cur = torch.cuda.current_stream()
    outs = []
    with torch.no_grad():
        for i in range(n):
            s[i].wait_stream(cur)
            with torch.cuda.stream(s[i]):
                outs.append(model.encode_image(img[i * 1024 // n:(i + 1) * 1024 // n]))
        for i in range(n):
            cur.wait_stream(s[i])
    return torch.cat(outs)


def timeit(fn, it=15):
    for _ in range(4):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


ref = full()
for n in (2, 4):
    o = split(n); torch.cuda.synchronize()
    print(f"{n} streams: max |diff| vs one batch {(o - ref).abs().max().item():.3e}")
for _ in range(2):
    print(f"one batch of 1024: {timeit(full):.3f} ms | 2 x 512 on two streams: {timeit(lambda: split(2)):.3f} ms | 4 x 256 on four streams: {timeit(lambda: split(4)):.3f} ms", flush=True)
