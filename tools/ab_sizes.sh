#!/bin/bash
# same-box comparison of library builds across sizes/modes: tools/ab_sizes.sh lib1.so ...
for cfg in "--boards 1048576 --steps 40 --warmup 5" "--boards 262144 --steps 100" "--mode fused-traj --boards 1048576 --steps 20 --warmup 5" "--mode fused-traj"; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset ML2048_HIP_LIB; else export ML2048_HIP_LIB=$lib; fi
    python bench.py --no-cpu-baseline --no-extras $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg |', '$lib'.split('/')[-1].ljust(28), '%.3e'%d['value'], '%.3f ms'%d['ms_per_step'], 'kernel %.1f us'%(r['avg_launch_ms']*1e3))"
  done
done
