set -o pipefail
mkdir -p gpurun_out/quick
for args in "--masked 1" "" "--masked 1" "--masked 1 --boards 262144" "--spawn rocrand"; do
  python bench.py --no-extras --no-cpu-baseline --steps 300 $args > gpurun_out/quick/b.json 2>> gpurun_out/quick/bench.err || exit 1
  python - "$args" <<'PY'
import json, sys
d = json.load(open("gpurun_out/quick/b.json"))
r = d.get("roofline") or {}
print("%-34s value %.4g  ms/step %.4f  kernel us %.2f" % (sys.argv[1] or "(default C2)", d["value"], d["ms_per_step"], (r.get("avg_launch_ms") or 0) * 1e3), flush=True)
PY
done
