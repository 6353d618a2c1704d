# parity of the plain rollout instantiations + the C2 bench line (default and driver-style), for kernel A/B work
set -o pipefail
mkdir -p gpurun_out/quick
timeout -k 10 900 python -m pytest tests/test_gpu_plain_rollout.py tests/test_gpu_parity.py tests/test_gpu_step_prepare.py -x -q -m gpu > gpurun_out/quick/tests.log 2>&1 || { tail -30 gpurun_out/quick/tests.log; exit 1; }
tail -2 gpurun_out/quick/tests.log
for args in "" "--boards 262144" "--boards 1048576 --steps 100" "--masked 1" "--spawn rocrand" "--mode fused-traj" "--steps 20 --warmup 5"; do
  python bench.py --no-extras --no-cpu-baseline $args > gpurun_out/quick/b.json 2>> gpurun_out/quick/bench.err || exit 1
  python - "$args" <<'PY'
import json, sys
d = json.load(open("gpurun_out/quick/b.json"))
r = d.get("roofline") or {}
print("%-34s value %.4g  ms/step %.4f  kernel us %.2f  frac %s" % (sys.argv[1] or "(default C2)", d["value"], d["ms_per_step"],
      (r.get("avg_launch_ms") or 0) * 1e3, r.get("frac")), flush=True)
PY
done
python tools/debug/profile_rollout_host.py > gpurun_out/quick/host.log 2>&1; sed -n 2,3p gpurun_out/quick/host.log
