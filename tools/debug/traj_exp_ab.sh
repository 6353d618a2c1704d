#!/bin/bash
# the trajectory-writing rollout on expanded tables + LDS look-ups (default) vs plain tables (ML2048_TRAJ_EXPANDED=0), same box
for rep in 1 2; do
for e in 1 0; do
  for M in 65536 262144 1048576; do
    ML2048_TRAJ_EXPANDED=$e python bench.py --mode fused-traj --boards $M --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('expanded=$e', 'M=$M', '%.3e'%d['value'], 'kernel %.1f us'%(r['avg_launch_ms']*1e3), 'frac %.3f'%r['frac'])"
  done
done
done
