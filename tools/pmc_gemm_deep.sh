#!/bin/bash
# LDS / L1 / wait counters of ONE GEMM launch configuration (tools/gemm_one.py), several --pmc passes (no tracing domains):
#   bash tools/pmc_gemm_deep.sh "img qkv" 8        -> gpurun_out/pmc_deep/<shape>_<cfg>/<pass>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S="$1"; C="$2"
O=$R/gpurun_out/pmc_deep/$(echo "$S" | tr ' ' '_')_$C
mkdir -p $O
i=0
for P in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_WAVE_CYCLES" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $O/p$i -- python3 $R/tools/gemm_one.py "$S" $C 6 > $O/p$i.log 2>&1 || echo "FAILED pass $i"
done
echo done "$S" $C
