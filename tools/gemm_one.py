"""Run ONE GEMM shape / tile configuration a few times (for rocprofv3 --pmc passes on the GPU box):

    python tools/gemm_one.py "<shape name>" <cfg> [iters] [split]

shape names are those of tools/gemm_ab.py (e.g. "img qkv", "img wgrad qkv", "img fc train")."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import gemm_ab as G

name, cfg = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
split = int(sys.argv[4]) if len(sys.argv) > 4 else 8
Mi, Mt = 51200, 78848
SH = {"img qkv": (Mi, 2304, 768, 1, 1, "bf16"), "img out": (Mi, 768, 768, 1, 1, "res"), "img fc train": (Mi, 3072, 768, 1, 1, "gelu"),
      "img proj": (Mi, 768, 3072, 1, 1, "res"), "txt qkv": (Mt, 1536, 512, 1, 1, "bf16"),
      "img dgrad proj": (Mi, 3072, 768, 1, 0, "dact"), "img wgrad qkv": (2304, 768, Mi, 0, 0, "split"),
      "img wgrad fc": (3072, 768, Mi, 0, 0, "split"), "square 4096": (4096, 4096, 4096, 1, 1, "bf16")}
M, N, K, akc, bkc, kind = SH[name]
d, keep = G.desc(M, N, K, akc, bkc, kind, cfg, split)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(iters):
    assert G.LB.cclip_gemm_bf16(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
print("ran", name, "cfg", cfg, "x", iters)
