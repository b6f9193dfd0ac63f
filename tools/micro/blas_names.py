"""Which vendor-BLAS kernels torch picks for the step's shapes (run under rocprofv3 --kernel-trace --stats; the Tensile kernel
names spell out macro-tile, wave layout and pipelining).  Measurement tool only."""
import torch
Mi, Mt = 51200, 78848
for M, N, K in [(Mi, 2304, 768), (Mi, 3072, 768), (Mi, 768, 3072), (8192, 8192, 8192), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
