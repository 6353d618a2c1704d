#!/bin/bash
# A/B of host-schedule settings, interleaved, on the GPU box
for rep in 1 2 3; do
  for cfg in "SPEC=1 T=4" "SPEC=0 T=4" "SPEC=1 T=2" "SPEC=0 T=1"; do
    eval "$cfg"
    ML2048_HOSTRNG_SPEC=$SPEC ML2048_HOST_THREADS=$T python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', '%.3e'%d['value'], '%.3f'%d['ms_per_step'])"
  done
done
