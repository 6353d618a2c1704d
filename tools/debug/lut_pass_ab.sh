# which compiled pass serves M=262144 best with / without the LDS line table
mkdir -p gpurun_out/lut
for lut in 1 0; do for mx in 196608 300000; do
  ML2048_PLAIN_ILP_MAX=$mx ML2048_ROW_LUT=$lut python bench.py --no-extras --no-cpu-baseline --boards ${1:-262144} --steps 120 --warmup 20 > gpurun_out/lut/b.json 2>gpurun_out/lut/b.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/lut/b.json')); print('lut $lut ilpmax $mx', '%.4g'%d['value'], 'kernel us %.2f'%(d['roofline']['avg_launch_ms']*1e3))"
done; done
