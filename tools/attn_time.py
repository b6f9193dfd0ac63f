import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops
for (B, T, H) in ((256, 577, 16), (256, 257, 16), (1024, 197, 12)):
    D = H * 64
    qkv = torch.randn(B * T, 3 * D, device="cuda").bfloat16()
    out = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, B=B, T=T, H=H)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 4.0 * B * H * T * T * 64
    by = B * T * 4 * D * 2
    print(f"B={B} T={T} H={H}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF  min-traffic {by/1e6:.0f} MB -> {by/ms/1e9:.2f} TB/s")
