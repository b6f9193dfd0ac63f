"""fp32 head GEMM (cclip_gemm_f32) on the train step's shapes, in-tree build vs a baseline library:
python tools/gemm_f32_ab.py [baseline.so]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops, _lib


def run(tag):
    g = torch.Generator(device="cuda").manual_seed(1)
    I, T = torch.randn(1024, 512, device="cuda", generator=g), torch.randn(1024, 512, device="cuda", generator=g)
    dl = torch.randn(1024, 1024, device="cuda", generator=g)
    px, pw = torch.randn(1024, 768, device="cuda", generator=g), torch.randn(768, 512, device="cuda", generator=g)
    cases = [("logits  I @ T^T        ", I, T, torch.empty(1024, 1024, device="cuda"), 0.0),
             ("dI      dL @ T         ", dl, T.t(), torch.empty(1024, 512, device="cuda"), 0.0),
             ("proj    x @ W          ", px, pw.t(), torch.empty(1024, 512, device="cuda"), 0.0),
             ("dW      x^T @ dy (acc) ", px.t(), I.t(), torch.zeros(768, 512, device="cuda"), 1.0),
             ("dx      dy @ W^T       ", I, pw, torch.empty(1024, 768, device="cuda"), 0.0)]
    out = []
    for name, A, B, C, beta in cases:
        f = lambda: ops.gemm_f32(A, B, C, beta=beta)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        out.append(f"{name.strip().split()[0]} {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
    print(tag, " | ".join(out))


run("new ")
if len(sys.argv) > 1:
    _lib.load_library()
    _lib._lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
    run("base")
