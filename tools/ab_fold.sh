#!/bin/bash
# A/B of the folded-LayerNorm inference path on ONE box (bench.py --mode image; lanes and single stream)
for r in 1 2; do
  for lanes in 2 1; do
    for fold in 1 0; do
      CCLIP_IMAGE_LANES=$lanes CCLIP_LN_FOLD=$fold python bench.py --mode image --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes=$lanes fold=$fold', d['value'], d['ms_per_step'])"
    done
  done
done
