"""Long-sequence attention (T > 128: ViT-B/16, ViT-L/14, ViT-L/14@336 token counts) forward / backward timing, optionally
against a second build of the library: python tools/attn_time.py [baseline.so]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops, _lib


def run(tag):
    for (B, T, H) in ((256, 577, 16), (256, 257, 16), (1024, 197, 12)):
        D = H * 64
        g = torch.Generator(device="cuda").manual_seed(1)
        qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).bfloat16()
        dout = torch.randn(B * T, D, device="cuda", generator=g).bfloat16()
        out = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, T, device="cuda")
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        f = lambda: ops.attention_fwd(q, k, v, out, lse=lse, B=B, T=T, H=H)
        bw = lambda: ops.attention_bwd(q, k, v, out, lse, dout, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B=B, T=T, H=H)
        res = []
        for fn, fl in ((f, 4.0 * B * H * T * T * 64), (bw, 14.0 * B * H * T * T * 64)):   # bwd: S, dP twice + dV, dK, dQ
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res.append(f"{ms * 1e3:8.1f} us {fl / ms / 1e9:5.0f} TF")
        print(f"{tag} B={B} T={T} H={H}: fwd {res[0]}   bwd {res[1]}   chk {out.float().abs().sum().item():.6e} {dqkv.float().abs().sum().item():.6e}")


run("new ")
if len(sys.argv) > 1:
    base = ctypes.CDLL(os.path.abspath(sys.argv[1]))
    _lib.load_library()
    _lib._lib = base                      # every later entry-point lookup resolves in the baseline build
    run("base")
