# driver-style run (20 steps after 5 warm-up) with the runtime's completion signals polled vs interrupt-driven
set -o pipefail
mkdir -p gpurun_out/intr
for rep in 1 2 3; do
  for v in default poll; do
    if [ $v = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
    python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/intr/b_${v}_$rep.json 2>> gpurun_out/intr/err.log || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/intr/b_${v}_$rep.json"))
print("$v $rep", "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
done
unset HSA_ENABLE_INTERRUPT
for v in default poll; do
  if [ $v = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
  python bench.py --no-extras --no-cpu-baseline > gpurun_out/intr/long_$v.json 2>> gpurun_out/intr/err.log || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/intr/long_$v.json"))
print("long $v", "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
done
