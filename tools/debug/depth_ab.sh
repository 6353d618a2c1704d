# end-to-end C2 throughput against the schedule stream's depth / hold (how far producer and launch thread run ahead)
set -o pipefail
mkdir -p gpurun_out/depth
for rep in 1 2 3 4; do
  for cfg in "8 5" "16 8" "16 12" "32 16" "32 24"; do
    set -- $cfg
    ML2048_STREAM_DEPTH=$1 ML2048_STREAM_HOLD=$2 python bench.py --no-extras --no-cpu-baseline --steps 600 > gpurun_out/depth/b.json || exit 1
    python - $1 $2 <<'PY'
import json, sys
d=json.load(open("gpurun_out/depth/b.json"))
print("depth", sys.argv[1], "hold", sys.argv[2], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
done
