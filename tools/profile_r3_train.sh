#!/bin/bash
# Round-3 rocprofv3 evidence for the headline train step (bench.py default): kernel traces (default streams / one stream) and the
# two --pmc passes, as tools/profile_r2.sh did.   PROF_TAG=x bash tools/profile_r3_train.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r3t_${PROF_TAG:-final}
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
kt() { rm -rf $O/$1; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1 || echo "FAILED $1"; }
pmc() { rm -rf $O/$1; rocprofv3 --pmc $3 --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1 || echo "FAILED $1"; }
kt train_default "$B --steps 4 --warmup 2"
CCLIP_WGRAD_STREAM=0 kt train_single "$B --steps 6 --warmup 2 --tower-streams 1"
export CCLIP_WGRAD_STREAM=0
pmc train_fetch "$B --steps 1 --warmup 1 --tower-streams 1" FETCH_SIZE
pmc train_write "$B --steps 1 --warmup 1 --tower-streams 1" WRITE_SIZE
pmc train_mfma "$B --steps 1 --warmup 1 --tower-streams 1" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
echo profiles done
