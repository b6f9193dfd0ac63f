#!/bin/bash
# Fold what tools/profile_r3.sh (image) and tools/profile_r3_train.sh (train) left under gpurun_out/ into profiles/r03_*.
#   bash tools/profile_fold_r3.sh <image tag> <train tag>
set -e
cd "$(dirname "$0")/.."
OI=gpurun_out/prof_r3_${1:-final}
OT=gpurun_out/prof_r3t_${2:-final}
P=profiles
for d in $OI/*/ $OT/*/; do
  newest=$(ls -t $d*/* 2>/dev/null | head -1); [ -z "$newest" ] && continue
  pfx=$(basename "$newest" | sed 's/_.*//')
  for f in $d*/*; do case "$(basename "$f")" in ${pfx}_*) ;; *) rm -f "$f" ;; esac; done
done
stats() { cp "$(ls -t $1/*/*kernel_stats.csv | head -1)" $P/$2; }
line() { grep '^{"metric"' $1 | tail -1 > $P/$2; }
stats $OI/image_lanes r03_image_lanes_kernel_stats.csv;   line $OI/image_lanes.log r03_bench_image_lanes_under_rocprof.json
stats $OI/image_single r03_image_single_kernel_stats.csv; line $OI/image_single.log r03_bench_image_single_under_rocprof.json
export PMC_BENCH_COMMAND='CCLIP_IMAGE_LANES=1 python3 bench.py --no-cpu-baseline --no-extras --mode image --steps 2 --warmup 1'
python tools/pmc_summary.py $OI/image_fetch $OI/image_write $P/r03_image_bs1024_hbm_traffic_pmc.json
python tools/pmc_mfma_summary.py $OI/image_mfma $P/r03_image_bs1024_mfma_busy_pmc.json
unset PMC_BENCH_COMMAND
stats $OT/train_default r03_train_default_kernel_stats.csv; line $OT/train_default.log r03_bench_train_default_under_rocprof.json
stats $OT/train_single r03_train_single_kernel_stats.csv;   line $OT/train_single.log r03_bench_train_single_under_rocprof.json
python tools/pmc_summary.py $OT/train_fetch $OT/train_write $P/r03_train_bs1024_hbm_traffic_pmc.json
python tools/pmc_mfma_summary.py $OT/train_mfma $P/r03_train_bs1024_mfma_busy_pmc.json
python tools/profile_table.py r03 > gpurun_out/profile_tables_r03.md
echo "tables: gpurun_out/profile_tables_r03.md"
