"""Short-sequence attention (T <= 128, the CLIP towers' shapes) forward / backward timing, optionally against a second
build of the library: python tools/attn_short_ab.py [baseline.so].  HIP events on the launch stream, 20 launches each."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops, _lib

def run(tag):
    for (B, T, H, causal) in ((1024, 50, 12, False), (1024, 77, 8, True), (256, 50, 12, False), (2048, 50, 12, False)):
        D = H * 64
        g = torch.Generator(device="cuda").manual_seed(1)
        qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).bfloat16()
        dout = torch.randn(B * T, D, device="cuda", generator=g).bfloat16()
        out = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, T, device="cuda")
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        f = lambda: ops.attention_fwd(q, k, v, out, lse=lse, B=B, T=T, H=H, causal=causal)
        b = lambda: ops.attention_bwd(q, k, v, out, lse, dout, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B=B, T=T, H=H, causal=causal)
        res = []
        for fn, by in ((f, B * T * 4 * D * 2), (b, B * T * 8 * D * 2)):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            res.append(f"{us:7.1f} us ({by / us / 1e6:.2f} TB/s min-traffic)")
        print(f"{tag} B={B} T={T} H={H} causal={causal}: fwd {res[0]}  bwd {res[1]}  chk {out.float().abs().sum().item():.6e} {dqkv.float().abs().sum().item():.6e}")

run("new ")
if len(sys.argv) > 1:
    base = ctypes.CDLL(os.path.abspath(sys.argv[1]))
    _lib.load_library()
    _lib._lib = base                      # every later entry-point lookup resolves in the baseline build
    run("base")
