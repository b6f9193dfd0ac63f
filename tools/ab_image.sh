#!/bin/bash
# A/B of bench.py --mode image on ONE box: the tuned table as is against the same table with configurations 8 / 10 mapped to 3
# (interleaved runs).   gpurun -- 'bash tools/ab_image.sh'
for r in 1 2 3; do
  for v in "" "8:3,10:3" "10:8"; do
    CCLIP_GEMM_REMAP=$v python bench.py --mode image --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('remap=[$v]', d['value'], d['ms_per_step'])"
  done
done
