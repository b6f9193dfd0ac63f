"""Step-level refinement of the persisted GEMM tile table.

The table (cclip_hip/gemm_tune.json) is made by timing every tile configuration of a shape ALONE on the GPU.  In the train step the
two tower streams and the weight-gradient side stream run kernels side by side, so the configuration that wins alone (one big
workgroup per CU) is not always the one that wins the step.  This tool does a coordinate descent over the forward-layout keys the
train step actually looks up: switch one key to another configuration, time the whole step, keep the change when it is a
confirmed improvement.  Output: gpurun_out/gemm_tune_step.json (copy over cclip_hip/gemm_tune.json to adopt it).

    gpurun -- 'python tools/tune_step.py'            (about two GPU-minutes)
"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
import clip  # noqa: E402
from clip import optim as coptim, parallel  # noqa: E402
from clip.weights import MODELS, init_state_dict, synthetic_text  # noqa: E402
from cclip_hip import ops  # noqa: E402

dev = torch.device("cuda", 0)
geo = MODELS["ViT-B/32"]
B = 1024
model = clip.build_model(init_state_dict(geo, 567), torch.bfloat16).to(dev)
model.train()
opt = coptim.AdamW(model, lr=1e-5)
g = torch.Generator(device=dev).manual_seed(567)
image = torch.randn(B, 3, geo.image_resolution, geo.image_resolution, device=dev, generator=g)
text = synthetic_text(B, geo, 567).to(dev)
reducer = parallel.GradReducer(model, None)


def step():
    opt.zero_grad()
    fi, ft = model.encode_image_text(image, text)
    loss, stats = clip.contrastive_loss(fi, ft, model.logit_scale, None)
    reducer.begin()
    loss.backward()
    opt.step(pending=reducer.finish())


def timed(n=8):
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


class Logged(ops._GenDict):
    seen = []

    def get(self, k, d=None):
        if k not in Logged.seen:
            Logged.seen.append(k)
        return dict.get(self, k, d)


for _ in range(3):
    step()
torch.cuda.synchronize()
ops.load_tuned_table()
logged = Logged(ops._TUNED)
ops._TUNED = logged
step(); torch.cuda.synchronize()
keys = [k for k in Logged.seen if k in logged and logged[k][1] == 1 and k.split("|")[12] == "1"]      # unsplit (forward-layout) keys
print(f"{len(keys)} forward-layout keys in the step; baseline {timed():.3f} ms", flush=True)
base = min(timed(), timed())
changed = {}
for k in keys:
    cur = logged[k]
    best, best_t = cur, base
    ncol = (int(k.split("|")[2]) + 255) // 256
    for cfg in (1, 2, 3, 4, 5, 7, 8) + tuple(c + 256 * g for c in (3, 8) for g in (3, 4, 6) if ncol > g):   # (+ grouped tile orders)
        if cfg == cur[0]:
            continue
        logged[k] = (cfg, 1)
        try:
            t = timed()
        except Exception:
            logged[k] = cur
            torch.cuda.synchronize()
            continue
        if t < best_t - 0.12:
            t2 = timed()                                    # confirm: the pool's step-to-step noise is ~0.1 ms
            if t2 < best_t - 0.12:
                best, best_t = (cfg, 1), max(t, t2)
    logged[k] = best
    if best != cur:
        changed[k] = (cur, best, base, best_t)
        base = min(best_t, timed())
    print(f"{k}: {cur} -> {best}   step {base:.3f} ms", flush=True)
# weight-gradient keys (split candidates): the same descent over the (tile, split) pairs the isolated tuner chooses from
from cclip_hip.stack import wgrad_candidates  # noqa: E402
wkeys = [k for k in Logged.seen if k in logged and k.split("|")[12] == "-1"]
print(f"{len(wkeys)} weight-gradient keys", flush=True)
for k in wkeys:
    f = k.split("|")
    M, N, K = int(f[1]), int(f[2]), int(f[3])
    cur = logged[k]
    best, best_t = cur, base
    for cand in wgrad_candidates(M, N, K):
        if tuple(cand) == tuple(cur) or (int(f[13]) and cand[0] == 3):      # (the 256x256 tile has no fused bias gradient)
            continue
        logged[k] = tuple(cand)
        try:
            t = timed()
        except Exception:
            logged[k] = cur
            torch.cuda.synchronize()
            continue
        if t < best_t - 0.12:
            t2 = timed()
            if t2 < best_t - 0.12:
                best, best_t = tuple(cand), max(t, t2)
    logged[k] = best
    if best != cur:
        changed[k] = (cur, best, base, best_t)
        base = min(best_t, timed())
    print(f"{k}: {cur} -> {best}   step {base:.3f} ms", flush=True)
print("changed:", json.dumps({k: [list(v[0]), list(v[1]), round(v[2], 3), round(v[3], 3)] for k, v in changed.items()}, indent=1))
print(f"final {min(timed(), timed()):.3f} ms")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
ops._TUNED = ops._GenDict(logged)
ops.save_tuned_table(os.path.join(ROOT, "gpurun_out", "gemm_tune_step.json"))
