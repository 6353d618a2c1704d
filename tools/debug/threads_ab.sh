# end-to-end C2 throughput against the number of producer workers of the schedule stream
set -o pipefail
mkdir -p gpurun_out/thr
for rep in $(seq 1 ${AB_REPS:-3}); do
  for t in ${AB_THREADS:-6 8 10 12 14}; do
    ML2048_HOST_THREADS=$t python bench.py --no-extras --no-cpu-baseline --steps ${AB_STEPS:-600} ${AB_ARGS:-} > gpurun_out/thr/b.json || exit 1
    python - $t <<'PY'
import json, sys
d=json.load(open("gpurun_out/thr/b.json"))
print("threads", sys.argv[1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
done
