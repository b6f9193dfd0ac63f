#!/bin/bash
# Fold what tools/profile_r2.sh left under gpurun_out/prof_r2 into the summaries committed under profiles/ (run in the build
# container after the gpurun call; the PMC summaries record the GEMM kernel-source hash and the commit they are folded at).
#   bash tools/profile_fold.sh [train] [caption] [l14]        (default: all three)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r2
P=profiles
what=${@:-train caption l14}
# gpurun MERGES a call's gpurun_out/ into the local one: files of earlier calls of the same pass linger (other process-id prefix);
# keep only the newest run's files per pass directory, or the fold would average two runs
for d in $O/*/; do
  newest=$(ls -t $d*/* 2>/dev/null | head -1); [ -z "$newest" ] && continue
  pfx=$(basename "$newest" | sed 's/_.*//')
  for f in $d*/*; do case "$(basename "$f")" in ${pfx}_*) ;; *) rm -f "$f" ;; esac; done
done
stats() { cp "$(ls -t $O/$1/*/*kernel_stats.csv | head -1)" $P/$2; }      # newest (gpurun merges: older calls' files may linger locally)
line() { grep '^{"metric"' $O/$1.log | tail -1 > $P/$2; }
for w in $what; do
  case $w in
    train)
      stats train_default r02_train_default_kernel_stats.csv; line train_default r02_bench_train_default_under_rocprof.json
      stats train_single r02_train_single_kernel_stats.csv; line train_single r02_bench_train_single_under_rocprof.json
      python tools/pmc_summary.py $O/train_fetch $O/train_write $P/r02_train_bs1024_hbm_traffic_pmc.json
      python tools/pmc_mfma_summary.py $O/train_mfma $P/r02_train_bs1024_mfma_busy_pmc.json ;;
    caption)
      stats caption r02_caption_kernel_stats.csv; line caption r02_bench_caption_under_rocprof.json
      python tools/pmc_summary.py $O/caption_fetch $O/caption_write $P/r02_caption_bs256_hbm_traffic_pmc.json
      python tools/pmc_mfma_summary.py $O/caption_mfma $P/r02_caption_bs256_mfma_busy_pmc.json ;;
    l14)
      stats l14_fp8 r02_l14_fp8_kernel_stats.csv; line l14_fp8 r02_bench_l14_fp8_under_rocprof.json
      python tools/pmc_summary.py $O/l14_fp8_fetch $O/l14_fp8_write $P/r02_l14_336_fp8_hbm_traffic_pmc.json
      python tools/pmc_mfma_summary.py $O/l14_fp8_mfma $P/r02_l14_336_fp8_mfma_busy_pmc.json ;;
  esac
done
python tools/profile_table.py > gpurun_out/profile_tables.md
echo "tables: gpurun_out/profile_tables.md"
