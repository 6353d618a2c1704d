# what does attaching the kernel timer to every n-th timed launch cost end to end?
set -o pipefail
mkdir -p gpurun_out/ev
for rep in 1 2 3 4; do
  for ev in 8 64 4; do
    python bench.py --no-extras --no-cpu-baseline --steps 600 --event-every $ev > gpurun_out/ev/b.json || exit 1
    python - $ev long <<'PY'
import json, sys
d=json.load(open("gpurun_out/ev/b.json"))
print(sys.argv[2], "event-every", sys.argv[1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
    python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --event-every $ev > gpurun_out/ev/b.json || exit 1
    python - $ev short <<'PY'
import json, sys
d=json.load(open("gpurun_out/ev/b.json"))
print(sys.argv[2], "event-every", sys.argv[1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
done
