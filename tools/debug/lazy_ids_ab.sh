# the driver's short run (and the long one) with the newest rollout's id scan issued lazily (default) or at once (ML2048_LAZY_IDS=0)
mkdir -p gpurun_out/lazy
for rep in 1 2 3; do for lazy in 1 0; do for args in "--steps 20 --warmup 5" "--steps 600"; do
  ML2048_LAZY_IDS=$lazy python bench.py --no-extras --no-cpu-baseline $args > gpurun_out/lazy/b.json 2>gpurun_out/lazy/b.err || { tail -5 gpurun_out/lazy/b.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/lazy/b.json')); print('lazy $lazy  $args ', '%.4g'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel us %.2f'%(d['roofline']['avg_launch_ms']*1e3))"
done; done; done
