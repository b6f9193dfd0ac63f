#!/bin/bash
# PMC passes over isolated GEMM launches (GPU box): where do the cycles of the K loop go?  Output: gpurun_out/pmc_gemm/<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"
run() {  # tag, shape, cfg
  for pass in 1 2; do
    if [ $pass = 1 ]; then C="$P1"; else C="$P2"; fi
    rocprofv3 --pmc $C --output-format csv -d $OUT/$1_p$pass -- python3 $R/tools/gemm_one.py "$2" $3 3 > $OUT/$1_p$pass.log 2>&1 || echo "pass failed: $1 $pass"
  done
}

IFS=$'\n'
for spec in $(echo "${PMC_RUNS:-qkv_cfg5|img qkv|5;qkv_cfg6|img qkv|6;fctrain_cfg6|img fc train|6}" | tr ';' '\n'); do
  tag=$(echo "$spec" | cut -d'|' -f1); shape=$(echo "$spec" | cut -d'|' -f2); cfg=$(echo "$spec" | cut -d'|' -f3)
  run "$tag" "$shape" "$cfg"
done
echo done
