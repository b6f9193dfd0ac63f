"""Which vendor-library kernels torch.matmul picks on the image-tower shapes (run under rocprofv3 --kernel-trace: the kernel names encode
macro tile / wave layout / prefetch depth, and the trace carries grid, work-group size, LDS bytes and register counts).  Yardstick only."""
import torch
for name, M, N, K in [("img qkv", 51200, 2304, 768), ("img out", 51200, 768, 768), ("img fc", 51200, 3072, 768), ("img proj", 51200, 768, 3072),
                      ("sq 4096", 4096, 4096, 4096)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(A, W.t(), out=o)
    torch.cuda.synchronize()
print("done")
