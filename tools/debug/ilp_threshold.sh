# same-box A/B of the two builds of the plain rollout kernels per board count (ML2048_PLAIN_ILP_MAX)
for m in 65536 131072 196608 262144; do
  for thr in 0 100000000; do
    for rep in 1 2; do
    ML2048_PLAIN_ILP_MAX=$thr python bench.py --no-cpu-baseline --no-extras --boards $m --steps 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('M=$m', 'ilp' if $thr else 'dflt', '%.3e'%d['value'], '%.4f ms'%d['ms_per_step'], 'kernel %.1f us'%(r['avg_launch_ms']*1e3))"
    done
  done
done
