#!/bin/bash
# The secondary bench lines of profiles/README.md (run through gpurun from the repo root): tools/collect_variants.sh <tag>
TAG=${1:-r01_x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || echo "$name failed"; }
run 262144 --boards 262144 --steps 300
run 1048576 --boards 1048576 --steps 100
run fused_traj_c2 --mode fused-traj
run fused_traj_1m --mode fused-traj --boards 1048576 --steps 30 --warmup 5
run masked --masked 1
run ids_inline --ids inline
run sched_sync --schedule host --steps 100
run sched_device --schedule device
run stepwise --mode stepwise --steps 20 --warmup 3
run stepwise_4m --mode stepwise --boards 4194304 --env-steps 16 --steps 3 --warmup 1
run obs --mode obs --masked 1 --boards 262144 --env-steps 32 --steps 3 --warmup 1
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d.get('roofline') or {}
    print(sys.argv[1].split('/')[-1].ljust(32), '%.3e'%d['value'], '%.4f ms'%d['ms_per_step'], 'kernel', r.get('kernel'), '%.1f us'%(1e3*r.get('avg_launch_ms',0) if r.get('avg_launch_ms') else 0), 'frac %.2f'%r.get('frac',0))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
