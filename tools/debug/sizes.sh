# the plain fused rollout over board counts (both sides of the small-M / large-M pass switch), unmasked / masked / rocRAND
mkdir -p gpurun_out/sizes
for args in "--boards 65536" "--boards 131072 --steps 300" "--boards 196608 --steps 300" "--boards 262144 --steps 200" "--boards 524288 --steps 100" "--boards 1048576 --steps 60" "--boards 65536 --masked 1" "--boards 262144 --masked 1 --steps 200" "--boards 65536 --spawn rocrand" "--boards 262144 --spawn rocrand --steps 200" "--steps 20 --warmup 5"; do
  python bench.py --no-extras --no-cpu-baseline $args > gpurun_out/sizes/b.json 2>> gpurun_out/sizes/bench.err || exit 1
  python - "$args" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sizes/b.json"))
r = d.get("roofline") or {}
print("%-40s value %.4g  ms/step %.4f  kernel us %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], (r.get("avg_launch_ms") or 0) * 1e3), flush=True)
PY
done
