#!/bin/bash
# the round driver's command (python bench.py --steps 20 --warmup 5), repeated, under a few settings
for rep in 1 2 3; do
  for depth in 3 8 16; do
    ML2048_STREAM_DEPTH=$depth python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('depth $depth', '%.3e'%d['value'], '%.4f ms/step'%d['ms_per_step'], 'kernel %.1f us'%(r['avg_launch_ms']*1e3))"
  done
done
