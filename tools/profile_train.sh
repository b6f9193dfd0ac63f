# rocprofv3 kernel-trace summaries of the default and the single-stream train step (run on the GPU box via gpurun; copies go to profiles/)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_default.log 2>&1
export CCLIP_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_single -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --tower-streams 1 > $R/gpurun_out/prof_single.log 2>&1
