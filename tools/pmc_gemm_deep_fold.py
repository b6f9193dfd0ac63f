"""Fold tools/pmc_gemm_deep.sh passes: per-launch counter means of the GEMM kernel (last 3 of the 6 launches)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(d, "p*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d))
    for k in sorted(vals):
        v = vals[k][-3:]
        print(f"   {k:34s} {sum(v) / len(v):16.0f}")
