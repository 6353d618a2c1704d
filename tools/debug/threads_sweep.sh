# e2e rate of the default bench vs the number of schedule-stream workers (same box); optional ML2048_HIP_LIB
for n in 6 8 10 12 14; do
  for rep in 1 2; do
  ML2048_HOST_THREADS=$n python bench.py --no-cpu-baseline --no-extras --steps 1000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('workers $n', '%.3e'%d['value'], '%.4f ms'%d['ms_per_step'], 'kernel %.1f us'%(r['avg_launch_ms']*1e3))"
  done
done
