"""diagnostic (GPU box): the mapper's second-layer weight gradient at the real geometry - W2 grad = dproj^T h1 with a 2-row
contraction and a strided dproj view - against float64, per region of the output"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops
for dt in (torch.bfloat16, torch.float16):
    for B in (2, 3, 8, 64):
        g = torch.Generator(device="cuda").manual_seed(B)
        S, D, P, NH = 80, 768, 20, 7680
        dxb = (torch.randn(B, S * D, device="cuda", generator=g) * 1e-2).to(dt)
        h1 = torch.tanh(torch.randn(B, NH, device="cuda", generator=g)).to(dt)
        dproj = dxb[:, :P * D]
        out = torch.full((P * D, NH), float("nan"), device="cuda")
        ops.gemm_bf16(dproj, h1, a_kcontig=False, b_kcontig=False, out_f32=out)
        ref = dproj.double().t() @ h1.double()
        err = (out.double() - ref)
        print(dt, "B", B, "rel", (err.norm() / ref.norm()).item(), "norm ratio", (out.double().norm() / ref.norm()).item(),
              "nan", int(torch.isnan(out).sum()), "max|err| rows", err.abs().amax(dim=1).topk(3).indices.tolist())
