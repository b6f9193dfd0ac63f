"""Join the committed round-2 rocprofv3 summaries into one table per workload: per kernel the average duration (kernel-trace stats),
HBM-side bytes per launch (FETCH_SIZE x 2 per MI355X_MICROARCH.md + WRITE_SIZE, separate --pmc passes), the resulting GB/s against
the 8 TB/s HBM3E peak, and the MFMA-busy fraction.   python tools/profile_table.py  ->  markdown on stdout"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def short(k):
    k = k.replace("void ", "").replace("cclip_bf16::", "")
    return k.split("(")[0][:70]


def table(tag, stats_csv, traffic_json, mfma_json, steps):
    st = {}
    for r in csv.DictReader(open(os.path.join(P, stats_csv))):
        st[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]))
    tr = {short(k["kernel"]): k for k in json.load(open(os.path.join(P, traffic_json)))["kernels"]}
    mf = {short(k["kernel"]): k["mfma_busy"] for k in json.load(open(os.path.join(P, mfma_json)))["kernels"]}
    print(f"\n#### {tag}\n")
    print("| kernel | launches/step | avg µs | % of GPU time | HBM MB / launch (fetch×2 + write) | HBM GB/s | of 8 TB/s | MFMA busy |")
    print("|---|---|---|---|---|---|---|---|")
    for k, (calls, avg, tot, pct) in sorted(st.items(), key=lambda kv: -kv[1][2])[:14]:
        t = tr.get(k)
        mb = (2 * t["fetch_kb_per_launch_raw"] + (t["write_kb_per_launch"] or 0)) / 1e3 if t else None
        gbs = mb / 1e3 / (avg * 1e-6) if mb is not None else None
        print(f"| `{k}` | {calls / steps:.0f} | {avg:.1f} | {pct:.1f} | {mb:.0f} | {gbs:.0f} | {gbs / 8000:.2f} | {mf.get(k, float('nan')):.3f} |"
              if mb is not None else f"| `{k}` | {calls / steps:.0f} | {avg:.1f} | {pct:.1f} | - | - | - | {mf.get(k, float('nan')):.3f} |")


import sys
if len(sys.argv) > 1 and sys.argv[1] == "r03":
    table("Round 3 - encode_image, ViT-B/32 bs 1024 bf16, batch whole on ONE stream (CCLIP_IMAGE_LANES=1; 25 passes under the profiler)",
          "r03_image_single_kernel_stats.csv", "r03_image_bs1024_hbm_traffic_pmc.json", "r03_image_bs1024_mfma_busy_pmc.json", 25)
    table("Round 3 - train step, ViT-B/32 bs 1024 bf16, packed text rows, single stream (10 steps under the profiler: 2 warm-up + 6 timed + 2 roofline-leg)",
          "r03_train_single_kernel_stats.csv", "r03_train_bs1024_hbm_traffic_pmc.json", "r03_train_bs1024_mfma_busy_pmc.json", 10)
    sys.exit(0)
table("Train step, ViT-B/32 bs 1024 bf16, single stream (10 steps under the profiler: 2 warm-up + 6 timed + 2 roofline-leg)",
      "r02_train_single_kernel_stats.csv", "r02_train_bs1024_hbm_traffic_pmc.json", "r02_train_bs1024_mfma_busy_pmc.json", 10)
table("BASELINE configs[3]: caption train step (MLP mapper + GPT-2-small, V = 21128, bs 256, S = 80), 8 steps under the profiler",
      "r02_caption_kernel_stats.csv", "r02_caption_bs256_hbm_traffic_pmc.json", "r02_caption_bs256_mfma_busy_pmc.json", 8)
table("BASELINE configs[4]: ViT-L/14@336px encode_image, bs 256, all four block projections in e4m3 (qkv / fc row-scaled, out-proj / c_proj block-scaled), 6 steps under the profiler",
      "r02_l14_fp8_kernel_stats.csv", "r02_l14_336_fp8_hbm_traffic_pmc.json", "r02_l14_336_fp8_mfma_busy_pmc.json", 6)
