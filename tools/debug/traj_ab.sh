#!/bin/bash
# same-box A/B of the trajectory-writing rollout: tools/debug/traj_ab.sh variant.so ...
for rep in 1 2; do
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset ML2048_HIP_LIB; else export ML2048_HIP_LIB=$PWD/$lib; fi
  for M in 65536 262144 1048576; do
    python bench.py --mode fused-traj --boards $M --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib'.split('/')[-1].ljust(18), 'M=$M', '%.3e'%d['value'], 'kernel %.1f us'%(r['avg_launch_ms']*1e3), 'frac %.3f'%r['frac'])"
  done
done
done
