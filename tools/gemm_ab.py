"""A/B of two builds of libcclip_hip.so on the step's GEMM shapes, interleaved in ONE process (run-to-run and
box-to-box variance on the pool is +-10 %, larger than most kernel changes).

    python tools/gemm_ab.py [A.so] [B.so] [shape-substring ...]
defaults: A = tools/micro/_bin/libcclip_hip_base.so (built from the last commit), B = the in-tree library.
"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops
from cclip_hip.ops import GemmDesc

args = sys.argv[1:]
libs = [a for a in args if a.endswith(".so")]
only = [a for a in args if not a.endswith(".so")]
A_SO = libs[0] if libs else os.path.join(ROOT, "tools/micro/_bin/libcclip_hip_base.so")
B_SO = libs[1] if len(libs) > 1 else os.path.join(ROOT, "construction-clip_amd/cclip_hip/libcclip_hip.so")
LA, LB = ctypes.CDLL(A_SO), ctypes.CDLL(B_SO)


def desc(M, N, K, akc, bkc, kind, cfg, split):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn((M, K) if akc else (K, M), device="cuda", generator=g).bfloat16()
    B = torch.randn((N, K) if bkc else (K, N), device="cuda", generator=g).bfloat16()
    keep = [A, B]
    d = GemmDesc()
    d.A, d.B, d.a_kcontig, d.b_kcontig, d.lda, d.ldb = A.data_ptr(), B.data_ptr(), akc, bkc, A.stride(0), B.stride(0)
    d.M, d.N, d.K, d.alpha, d.ldc, d.split_k, d.tile_config = M, N, K, 1.0, N, 1, cfg
    bias = torch.randn(N, device="cuda"); keep.append(bias)
    if kind in ("bf16", "gelu", "geluinf", "res", "dact"):
        d.bias = bias.data_ptr()
    if kind == "geluinf":                                  # inference fc: QuickGELU, 16-bit output only
        d.act = 1
    if kind in ("bf16", "gelu", "geluinf", "dact"):
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); keep.append(o); d.out_bf16 = o.data_ptr()
    if kind == "gelu":
        o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); keep.append(o2); d.out_pre_bf16 = o2.data_ptr(); d.act = 1
    if kind == "dact":
        ax = torch.randn(M, N, device="cuda").bfloat16(); keep.append(ax); d.aux, d.ldaux, d.act = ax.data_ptr(), N, 16
    if kind == "res":
        x = torch.randn(M, N, device="cuda"); keep.append(x); d.out_f32, d.residual, d.ldr = x.data_ptr(), x.data_ptr(), N
    if kind == "split":
        o = torch.empty(M, N, device="cuda"); ws = torch.empty(split * M * N, device="cuda"); keep += [o, ws]
        d.out_f32, d.split_k, d.split_ws, d.bias = o.data_ptr(), split, ws.data_ptr(), 0
    return d, keep


def time_lib(lib, d, iters=10):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        assert lib.cclip_gemm_bf16(ctypes.byref(d), st) == 0
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


if __name__ == "__main__":
    Mi, Mt = 51200, 78848
    shapes = [("img qkv", Mi, 2304, 768, 1, 1, "bf16"), ("img out", Mi, 768, 768, 1, 1, "res"), ("img fc train", Mi, 3072, 768, 1, 1, "gelu"),
              ("img fc infer", Mi, 3072, 768, 1, 1, "bf16"), ("img proj", Mi, 768, 3072, 1, 1, "res"),
              ("txt qkv", Mt, 1536, 512, 1, 1, "bf16"), ("txt out", Mt, 512, 512, 1, 1, "res"), ("txt fc train", Mt, 2048, 512, 1, 1, "gelu"),
              ("txt proj", Mt, 512, 2048, 1, 1, "res"),
              ("img dgradT fc", Mi, 768, 3072, 1, 1, "bf16"), ("img dgradT proj", Mi, 3072, 768, 1, 1, "dact"), ("img dgradT qkv", Mi, 768, 2304, 1, 1, "bf16"),
              ("img dgrad fc", Mi, 768, 3072, 1, 0, "bf16"), ("img dgrad proj", Mi, 3072, 768, 1, 0, "dact"), ("img dgrad qkv", Mi, 768, 2304, 1, 0, "bf16"),
              ("img wgrad qkv", 2304, 768, Mi, 0, 0, "split"), ("img wgrad proj", 768, 3072, Mi, 0, 0, "split"), ("img wgrad fc", 3072, 768, Mi, 0, 0, "split"),
              ("square 4096", 4096, 4096, 4096, 1, 1, "bf16")]
    print(f"A = {os.path.relpath(A_SO, ROOT)}\nB = {os.path.relpath(B_SO, ROOT)}\n(us: A -> B per tile config; best-of-configs last)")
    totA = totB = 0.0
    for name, M, N, K, akc, bkc, kind in shapes:
        if only and not any(o in name for o in only):
            continue
        cells, bestA, bestB = [], 1e9, 1e9
        for cfg in (1, 2, 3, 4, 5, 7):                  # (5 against a library that predates it: that library runs 1)
            d, keep = desc(M, N, K, akc, bkc, kind, cfg, 8)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if any(lib.cclip_gemm_bf16(ctypes.byref(d), st) != 0 for lib in (LA, LB)):
                continue                             # configuration 4: forward layout, unsplit, full tiles only
            for lib in (LA, LB):
                time_lib(lib, d, 3)
            ta = tb = 1e9
            for _ in range(3):                       # interleaved repeats, keep the minimum
                ta = min(ta, time_lib(LA, d)); tb = min(tb, time_lib(LB, d))
            cells.append(f"cfg{cfg} {ta * 1e3:6.1f} -> {tb * 1e3:6.1f} ({(tb / ta - 1) * 100:+5.1f}%)")
            bestA, bestB = min(bestA, ta), min(bestB, tb)
            del keep
        totA += bestA; totB += bestB
        print(f"{name:15s} | " + " | ".join(cells) + f" | best {bestA * 1e3:6.1f} -> {bestB * 1e3:6.1f} ({(bestB / bestA - 1) * 100:+5.1f}%)", flush=True)
    print(f"sum of best: {totA * 1e3:.1f} -> {totB * 1e3:.1f} us ({(totB / totA - 1) * 100:+.1f}%)")
