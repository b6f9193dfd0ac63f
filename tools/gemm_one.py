"""Run one GEMM shape a few times (for rocprofv3 --pmc): python tools/gemm_one.py M N K akc bkc cfg kind [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
M, N, K, akc, bkc, cfg = [int(x) for x in sys.argv[1:7]]
kind = sys.argv[7]
it = int(sys.argv[8]) if len(sys.argv) > 8 else 5
print(bench(M, N, K, bool(akc), bool(bkc), cfg, kind, iters=it, split=8))
