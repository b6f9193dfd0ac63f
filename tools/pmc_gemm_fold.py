"""fold tools/pmc_gemm.sh output: per tag, counters of the GEMM kernel dispatches (mean over dispatches)"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_gemm"
tags = sorted({os.path.basename(d).rsplit("_p", 1)[0] for d in glob.glob(os.path.join(root, "*_p[12]")) if os.path.isdir(d)})
for tag in tags:
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, tag + "_p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_" not in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    v = {k: acc[k] / max(n[k], 1) for k in acc}
    if not v:
        print(tag, "no data"); continue
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    gui = v.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    print(f"== {tag}: GUI_ACTIVE/8 = {gui:.3e} clk")
    for k in sorted(v):
        extra = f"  ({v[k] / wc:.3f} of WAVE_CYCLES)" if k.startswith("SQ_") and k not in ("SQ_WAVE_CYCLES",) else ""
        print(f"   {k:28s} {v[k]:.4e}{extra}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
        pass
