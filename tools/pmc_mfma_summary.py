"""MFMA-busy per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass.

    python tools/pmc_mfma_summary.py <dir with *_counter_collection.csv> out.json

Normalisation (checked on this data): SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of the cycles their matrix pipe is busy -
it equals 16 cycles x the number of v_mfma_f32_16x16x32 instructions a kernel issues (fc forward launches: 1.77e13 FLOP / 16384 FLOP
per MFMA x 16 = 1.73e10 vs 1.749e10 counted).  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs.  So
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
= the fraction of SIMD-cycles with the matrix pipe busy while the kernel ran (clock-independent, unlike TFLOP/s vs the 2.4 GHz peak).
"""
import os
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import provenance  # noqa: E402

d, out = sys.argv[1:3]
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k] += r["Counter_Name"] == "GRBM_GUI_ACTIVE"
rows = []
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    gui, mf = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0:
        continue
    rows.append(dict(kernel=k, launches=n[k], gui_active_sum=gui, mfma_busy_cycles=mf, mfma_busy=round(mf / (gui / 8 * 1024), 4)))
g = [r for r in rows if "gemm_bf16_kernel" in r["kernel"] or "gemm_stream_kernel" in r["kernel"] or "gemm_a4" in r["kernel"]]
fam = sum(r["mfma_busy_cycles"] for r in g) / (sum(r["gui_active_sum"] for r in g) / 8 * 1024)
json.dump(dict(note=__doc__.strip().split("\n\n")[1] if False else "see tools/pmc_mfma_summary.py for the normalisation and its check",
               command="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- " + os.environ.get("PMC_BENCH_COMMAND", "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --tower-streams 1"),
               **provenance(),
               gemm_family_mfma_busy=round(fam, 4), kernels=rows[:30]), open(out, "w"), indent=1)
print("GEMM family MFMA-busy:", round(fam, 4))
for r in rows[:12]:
    print(f"{r['kernel'][:80]:80s} {r['launches']:5d}  mfma_busy {r['mfma_busy']:.3f}")
