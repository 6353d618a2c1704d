# the driver's short run (20 steps after 5 warm-up) repeated: how often does a run land on a slow stretch?
set -o pipefail
for rep in $(seq 1 ${REPS:-12}); do
  python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"%.4g ms/step %.4f kernel %.2f\" % (d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"avg_launch_ms\"]*1e3))"
done
