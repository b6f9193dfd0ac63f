"""AdamW launch timing over a ViT-B/32-sized flat arena (151 M parameters, 30 B per parameter with the 16-bit shadow)."""
import os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops  # noqa: E402
n = 151_277_312
p = torch.randn(n, device="cuda") * 0.02
g = torch.randn(n, device="cuda") * 1e-3
m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
f = lambda: ops.adamw_step(p, g, m, v, lr=1e-5, step=3, bf16_shadow=sh)
import ctypes
OLD = os.path.join(ROOT, "tools/micro/ab_old/liboptim_old.so")      # optional: a build of the previous kernel to A/B against
tag = "new"
if len(sys.argv) > 1 and sys.argv[1] == "old" and os.path.isfile(OLD):
    ops.lib = ctypes.CDLL(OLD); tag = "old"
for _ in range(3):
    f()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 5 * 1e3)
t = statistics.median(ts)
print(f"{tag} adamw {n / 1e6:.0f} M parameters: {t:7.1f} us = {n * 30 / t * 1e-6:5.2f} TB/s", flush=True)
