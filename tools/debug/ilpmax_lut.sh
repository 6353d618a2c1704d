# where should the small-M / large-M switch sit now that the large-M pass has the line table
mkdir -p gpurun_out/lut
for m in 98304 131072; do for mx in 196608 65536; do for masked in 0 1; do
  ML2048_PLAIN_ILP_MAX=$mx python bench.py --no-extras --no-cpu-baseline --boards $m --masked $masked --steps 200 --warmup 20 > gpurun_out/lut/b.json 2>gpurun_out/lut/b.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/lut/b.json')); print('M $m masked $masked ilpmax $mx', '%.4g'%d['value'], 'kernel us %.2f'%(d['roofline']['avg_launch_ms']*1e3))"
done; done; done
