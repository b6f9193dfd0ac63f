#!/bin/bash
# Round-2 rocprofv3 evidence (run on the GPU box through gpurun; summaries are copied to profiles/ afterwards).
# Kernel-trace passes and PMC passes are separate runs (gpurun refuses --pmc together with tracing domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r2
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
# (gpurun MERGES gpurun_out/ back: drop an earlier call's files of the same pass first, the fold would mix them)
kt() { rm -rf $O/$1; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1 || echo "FAILED $1"; }
pmc() { rm -rf $O/$1; rocprofv3 --pmc $3 --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1 || echo "FAILED $1"; }
# PROF_ONLY=l14 (or train / caption) restricts the run to one workload
if [ -z "$PROF_ONLY" ] || [ "$PROF_ONLY" = train ]; then
# 1. the headline train step: default (two tower streams + wgrad side stream) and serialised on one stream
kt train_default "$B --steps 4 --warmup 2"
CCLIP_WGRAD_STREAM=0 kt train_single "$B --steps 6 --warmup 2 --tower-streams 1"
# 2. HBM traffic + MFMA busy of the train step (single stream)
export CCLIP_WGRAD_STREAM=0
pmc train_fetch "$B --steps 1 --warmup 1 --tower-streams 1" FETCH_SIZE
pmc train_write "$B --steps 1 --warmup 1 --tower-streams 1" WRITE_SIZE
pmc train_mfma "$B --steps 1 --warmup 1 --tower-streams 1" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
fi
if [ -z "$PROF_ONLY" ] || [ "$PROF_ONLY" = caption ]; then
# 3. BASELINE configs[3]: caption train step
kt caption "$B --mode caption --steps 4 --warmup 2"
pmc caption_fetch "$B --mode caption --steps 1 --warmup 1" FETCH_SIZE
pmc caption_write "$B --mode caption --steps 1 --warmup 1" WRITE_SIZE
pmc caption_mfma "$B --mode caption --steps 1 --warmup 1" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
fi
if [ -z "$PROF_ONLY" ] || [ "$PROF_ONLY" = l14 ]; then
# 4. BASELINE configs[4]: ViT-L/14@336px encode_image, bs 256, fp8 projections
L="python3 $R/bench.py --no-cpu-baseline --mode image --model ViT-L/14@336px --batch 256 --dtype fp8"
kt l14_fp8 "$L --steps 4 --warmup 2"
pmc l14_fp8_fetch "$L --steps 1 --warmup 1" FETCH_SIZE
pmc l14_fp8_write "$L --steps 1 --warmup 1" WRITE_SIZE
pmc l14_fp8_mfma "$L --steps 1 --warmup 1" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
fi
# keep only the small summaries (the raw traces are large)
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
echo profiles done
