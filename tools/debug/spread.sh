for s in 200 2000; do
  for rep in 1 2 3 4 5; do
    python bench.py --no-cpu-baseline --no-extras --steps $s --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('steps $s', '%.3e'%d['value'], '%.4f ms'%d['ms_per_step'], 'kernel %.1f us'%(r['avg_launch_ms']*1e3))"
  done
done
