import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
Mi = 51200
for name, M, N, K, kind in [("out K=768 res", Mi, 768, 768, "res"), ("out K=64 res", Mi, 768, 64, "res"), ("out K=128 res", Mi, 768, 128, "res"),
                            ("out K=768 bf16", Mi, 768, 768, "bf16"), ("out K=64 bf16", Mi, 768, 64, "bf16"),
                            ("out K=1536 res", Mi, 768, 1536, "res"), ("out K=3072 res", Mi, 768, 3072, "res"),
                            ("qkv K=768", Mi, 2304, 768, "bf16"), ("qkv K=64", Mi, 2304, 64, "bf16"), ("qkv K=1536", Mi, 2304, 1536, "bf16"),
                            ("fc K=768", Mi, 3072, 768, "gelu"), ("fc K=64", Mi, 3072, 64, "gelu")]:
    row = []
    for cfg in (1, 3, 4):
        ms, tf = bench(M, N, K, True, True, cfg, kind)
        row.append(f"cfg{cfg}: {ms*1000:7.1f} us {tf:7.1f} TF")
    print(f"{name:16s} | " + " | ".join(row), flush=True)
