"""A/B of cclip_gemm_fp8 between two builds of libcclip_hip.so, interleaved in ONE process, on the ViT-L/14@336px (bs 256) and
ViT-B/32 (bs 1024) projection shapes; also checks that both builds give bit-identical outputs.

    python tools/fp8_ab.py [A.so] [B.so]
defaults: A = tools/micro/_bin/libcclip_hip_base.so, B = the in-tree library.
"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]
from cclip_hip import ops

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
A_SO = libs[0] if libs else os.path.join(ROOT, "tools/micro/_bin/libcclip_hip_base.so")
B_SO = libs[1] if len(libs) > 1 else os.path.join(ROOT, "construction-clip_amd/cclip_hip/libcclip_hip.so")
LA, LB = ctypes.CDLL(A_SO), ctypes.CDLL(B_SO)
c_long, c_int, c_void_p = ctypes.c_long, ctypes.c_int, ctypes.c_void_p


def call(lib, A8, sa, W8, sw, bias, act, out):
    M, K = A8.shape
    N = W8.shape[0]
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.cclip_gemm_fp8(c_void_p(A8.data_ptr()), c_long(A8.stride(0)), c_void_p(sa.data_ptr()), c_void_p(W8.data_ptr()),
                            c_long(W8.stride(0)), c_void_p(sw.data_ptr()), c_int(M), c_int(N), c_int(K), c_void_p(bias.data_ptr()),
                            c_int(act), c_void_p(out.data_ptr()), c_long(out.stride(0)), st)
    assert rc == 0, rc


def timeit(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


print(f"A = {os.path.relpath(A_SO, ROOT)}\nB = {os.path.relpath(B_SO, ROOT)}")
ML, MB = 256 * 577, 1024 * 50
shapes = [("L14 qkv", ML, 3072, 1024, 0), ("L14 fc", ML, 4096, 1024, 1), ("L14 out", ML, 1024, 1024, 0), ("L14 proj", ML, 1024, 4096, 0),
          ("B32 qkv", MB, 2304, 768, 0), ("B32 fc", MB, 3072, 768, 1), ("ragged", 1000, 520, 400, 0)]
for name, M, N, K, act in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    A8, sa = torch.empty(M, K, device="cuda", dtype=torch.uint8), torch.empty(M, device="cuda")
    W8, sw = torch.empty(N, K, device="cuda", dtype=torch.uint8), torch.empty(N, device="cuda")
    ops.quantize_rows_fp8(W, W8, sw)
    ops.quantize_rows_fp8(A, A8, sa)
    oa = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    call(LA, A8, sa, W8, sw, bias, act, oa)
    call(LB, A8, sa, W8, sw, bias, act, ob)
    torch.cuda.synchronize()
    same = torch.equal(oa, ob)
    ta, tb = [], []
    for _ in range(5):
        ta.append(timeit(lambda: call(LA, A8, sa, W8, sw, bias, act, oa)))
        tb.append(timeit(lambda: call(LB, A8, sa, W8, sw, bias, act, ob)))
    a, b = min(ta), min(tb)
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M} N={N} K={K}: A {a * 1e3:7.1f} us ({fl / a / 1e9:6.0f} TF) -> B {b * 1e3:7.1f} us ({fl / b / 1e9:6.0f} TF) "
          f"({(b / a - 1) * 100:+.1f}%)  bit-identical: {same}", flush=True)

# ---- in-tree library only: what the block-scaled A operand and the fp32 residual epilogue cost on the out-proj / c_proj shapes
print("in-tree library: row-scaled A, 16-bit out | block-scaled A, 16-bit out | block-scaled A, fp32 residual stream | bf16 GEMM + residual")
for name, M, N, K in (("L14 out", ML, 1024, 1024), ("L14 proj", ML, 1024, 4096), ("B32 out", MB, 768, 768), ("B32 proj", MB, 768, 3072)):
    g = torch.Generator(device="cuda").manual_seed(2)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    A8, sa = torch.empty(M, K, device="cuda", dtype=torch.uint8), torch.empty(M, device="cuda")
    Am, ae = torch.empty(M, K, device="cuda", dtype=torch.uint8), ops.mx_scale_buffer(M, K, "cuda")
    W8, sw = torch.empty(N, K, device="cuda", dtype=torch.uint8), torch.empty(N, device="cuda")
    ops.quantize_rows_fp8(W, W8, sw); ops.quantize_rows_fp8(A, A8, sa); ops.quantize_mx_fp8(A, Am, ae)
    o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, N, device="cuda", generator=g)
    fns = [lambda: ops.gemm_fp8(A8, sa, W8, sw, o16, bias=bias),
           lambda: ops.gemm_fp8(Am, None, W8, sw, o16, bias=bias, block_scale_a=ae),
           lambda: ops.gemm_fp8(Am, None, W8, sw, bias=bias, block_scale_a=ae, out_f32=x, residual=x, half=torch.bfloat16),
           lambda: ops.gemm_bf16(A, W, bias=bias, residual=x, out_f32=x)]
    ts = [[] for _ in fns]
    for _ in range(4):
        for i, f in enumerate(fns):
            ts[i].append(timeit(f))
    print(f"{name:9s} " + " | ".join(f"{min(t) * 1e3:7.1f} us" for t in ts) + f"   quantise (MX) pass {timeit(lambda: ops.quantize_mx_fp8(A, Am, ae)) * 1e3:6.1f} us", flush=True)
