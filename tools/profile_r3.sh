#!/bin/bash
# Round-3 rocprofv3 evidence for the workload the 0.40 target is quoted on: ViT-B/32 encode_image, bs 1024, bf16
# (bench.py --mode image).  Run on the GPU box through gpurun; tools/profile_fold_r3.sh copies the summaries to profiles/.
# Kernel-trace passes and PMC passes are separate runs (gpurun refuses --pmc together with tracing domains); the program
# comes directly after `--`.
#   PROF_TAG=base bash tools/profile_r3.sh      (the tag names the output directory: gpurun_out/prof_r3_<tag>)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r3_${PROF_TAG:-final}
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras --mode image"
kt() { rm -rf $O/$1; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1 || echo "FAILED $1"; }
pmc() { rm -rf $O/$1; rocprofv3 --pmc $3 --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1 || echo "FAILED $1"; }
# 1. kernel trace: two half-batch lanes (the default; kernels of the two lanes overlap, durations inflated by co-residency)
kt image_lanes "$B --steps 20 --warmup 5"
# 2. kernel trace: the batch whole on one stream (clean per-kernel durations)
export CCLIP_IMAGE_LANES=1
kt image_single "$B --steps 20 --warmup 5"
# 3. HBM traffic + MFMA busy, single stream
pmc image_fetch "$B --steps 2 --warmup 1" FETCH_SIZE
pmc image_write "$B --steps 2 --warmup 1" WRITE_SIZE
pmc image_mfma "$B --steps 2 --warmup 1" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
unset CCLIP_IMAGE_LANES
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
echo profiles done
