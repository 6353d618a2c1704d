# every bench mode / schedule once, short: a smoke run after host-path or layout changes
set -o pipefail
mkdir -p gpurun_out/modes
for args in "--schedule host" "--schedule device" "--schedule auto" "--mode stepwise --steps 2 --warmup 1" "--mode stepwise3 --steps 2 --warmup 1" \
            "--mode obs --boards 262144 --steps 2 --warmup 1" "--mode fused-traj --boards 262144" "--spawn rocrand --boards 262144" "--masked 1 --schedule device"; do
  python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 $args > gpurun_out/modes/b.json 2>> gpurun_out/modes/err.log || { echo "FAILED: $args"; tail -5 gpurun_out/modes/err.log; exit 1; }
  python - "$args" <<'PY'
import json, sys
d = json.load(open("gpurun_out/modes/b.json"))
r = d.get("roofline") or {}
print("%-46s value %.4g  ms/step %.4f  frac %s" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("frac")), flush=True)
PY
done
