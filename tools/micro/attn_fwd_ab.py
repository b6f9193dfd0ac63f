"""Interleaved A/B of the short-sequence attention forward of two library builds (alternating groups of launches in one process):
    python tools/micro/attn_fwd_ab.py old.so"""
import ctypes, os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops, _lib
new = _lib.load_library()
old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for (B, T, H, causal) in ((1024, 50, 12, False), (512, 50, 12, False), (1024, 77, 8, True)):
    D = H * 64
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).bfloat16()
    outs = {}
    ts = {"new": [], "old": []}
    for rep in range(9):
        for tag, lib in ((("new", new), ("old", old)) if rep % 2 == 0 else (("old", old), ("new", new))):
            _lib._lib = lib
            out = torch.zeros(B * T, D, device="cuda", dtype=torch.bfloat16)
            lse = torch.empty(B, H, T, device="cuda")
            f = lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, lse=lse, B=B, T=T, H=H, causal=causal)
            f(); f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); e1.synchronize()
            ts[tag].append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[tag] = out
    print(f"B={B} T={T} H={H} causal={causal}: new {statistics.median(ts['new']):6.1f} us  old {statistics.median(ts['old']):6.1f} us  equal {torch.equal(outs['new'], outs['old'])}", flush=True)
