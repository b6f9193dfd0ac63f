"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command, no tracing domains)
into the per-kernel HBM-traffic summary committed under profiles/.

    python tools/pmc_summary.py <dir with *_counter_collection.csv of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> out.json

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are reported in KB;
on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads -> x2 (calibration: colsum_partial reads its input
exactly once and reports half of it raw).  The counters sit at the L2 <-> fabric boundary: Infinity-Cache hits are included.
"""
import os
import csv, glob, json, os, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "construction-clip_amd")]


def provenance():
    """what the summary is valid for: the GEMM kernel-source hash bench.py compares against, and the commit it was taken at"""
    from cclip_hip import ops
    try:
        git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        git = None
    return dict(kernel_source_hash=ops.kernel_source_hash(), git=git)


def fold(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = fold(fdir, "FETCH_SIZE"), fold(wdir, "WRITE_SIZE")
    rows = []
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        n, kb = fe[k]
        wn, wkb = wr.get(k, (0, 0.0))
        rows.append(dict(kernel=k, launches=n, fetch_kb_per_launch_raw=kb / n, write_kb_per_launch=(wkb / wn if wn else None)))
    g = [r for r in rows if "gemm_bf16_kernel" in r["kernel"] or "gemm_stream_kernel" in r["kernel"] or "gemm_a4" in r["kernel"]]
    gl = sum(r["launches"] for r in g)
    gf = sum(r["fetch_kb_per_launch_raw"] * r["launches"] for r in g) / gl * 1e3 / 1e6              # MB / launch, raw
    gw = sum((r["write_kb_per_launch"] or 0) * r["launches"] for r in g) / gl * 1e3 / 1e6
    json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 --no-cpu-baseline; "
                        "FETCH_SIZE raw is in KB; gfx950 reports half the bytes of wide coalesced reads -> x2 (MI355X_MICROARCH.md)",
                   command="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- " + os.environ.get("PMC_BENCH_COMMAND", "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --tower-streams 1"),
                   **provenance(), kernels=rows[:40],
                   gemm_family=dict(launches=gl, fetch_mb_per_launch_raw=gf, fetch_mb_per_launch_corrected=2 * gf, write_mb_per_launch=gw)),
              open(out, "w"), indent=1)
    print(json.dumps(dict(gemm_launches=gl, fetch_mb_corrected=2 * gf, write_mb=gw)))


if __name__ == "__main__":
    main()
