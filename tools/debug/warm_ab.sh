# does the driver's short run (W=5, K=20) time kernels that are slower than steady state because the GPU is not warm?
set -o pipefail
mkdir -p gpurun_out/warm
for rep in 1 2; do
 for w in 5 50 400; do
  python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup $w > gpurun_out/warm/b.json || exit 1
  python - $w <<'PY'
import json, sys
d=json.load(open("gpurun_out/warm/b.json"))
print("warmup", sys.argv[1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
 done
done
python tools/debug/driver_timeline.py | tail -12
