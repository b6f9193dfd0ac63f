#!/bin/bash
# Refresh the persisted GEMM tile table (construction-clip_amd/cclip_hip/gemm_tune.json) on a GPU box:
#   gpurun -- 'bash tools/tune_gemm.sh'      then      cp gpurun_out/gemm_tune.json construction-clip_amd/cclip_hip/
# Every workload bench.py knows is run once with CCLIP_TUNE_FILE set, so every (shape, layout, epilogue) key the hot path
# issues is timed once and written; keys are tied to the kernel-source hash (cclip_hip/ops.py:kernel_source_hash).
set -e
mkdir -p gpurun_out
export CCLIP_TUNE_FILE=$PWD/gpurun_out/gemm_tune.json
export CCLIP_TUNE_EXACT=1      # time every shape these runs meet (no nearest-shape fallback)
export CCLIP_PACK_TEXT=0       # the table holds the dense text shapes; packed batches take the nearest entry
cp -f construction-clip_amd/cclip_hip/gemm_tune.json "$CCLIP_TUNE_FILE" 2>/dev/null || true
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/tune_train_bf16.log 2>&1
# the packed text tower's shapes for the default synthetic batch (about half the dense rows): other packed batches take the nearest
CCLIP_PACK_TEXT=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/tune_train_bf16_packed.log 2>&1
CCLIP_PACK_TEXT=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --dtype fp16 > gpurun_out/tune_train_fp16_packed.log 2>&1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --dtype fp16 > gpurun_out/tune_train_fp16.log 2>&1
python bench.py --mode image --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tune_image.log 2>&1
python bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tune_fwd.log 2>&1
python bench.py --mode caption --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tune_caption.log 2>&1
CCLIP_PACK_TEXT=1 python bench.py --mode caption --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tune_caption_packed.log 2>&1
python bench.py --mode image --model ViT-L/14@336px --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tune_l14.log 2>&1
python - <<'PY'
import json, os
t = json.load(open(os.environ["CCLIP_TUNE_FILE"]))
print("entries:", len(t["table"]), "hash:", t["kernel_source_hash"])
PY
