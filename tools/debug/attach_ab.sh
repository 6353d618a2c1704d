# the rollout's marker event attached to the kernel's dispatch (ML2048_ATTACH_EVENT=1) vs recorded behind it
set -o pipefail
mkdir -p gpurun_out/att
ML2048_ATTACH_EVENT=1 timeout -k 10 300 python -m pytest tests/test_gpu_plain_rollout.py -x -q -m gpu -k "c2_twice or chain_matrix" 2>&1 | tail -2
for rep in 1 2 3 4; do
  for a in 0 1; do
    ML2048_ATTACH_EVENT=$a python bench.py --no-extras --no-cpu-baseline --steps 600 > gpurun_out/att/b.json || exit 1
    python - $a long <<'PY'
import json, sys
d=json.load(open("gpurun_out/att/b.json"))
print(sys.argv[2], "attach", sys.argv[1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
    ML2048_ATTACH_EVENT=$a python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/att/b.json || exit 1
    python - $a short <<'PY'
import json, sys
d=json.load(open("gpurun_out/att/b.json"))
print(sys.argv[2], "attach", sys.argv[1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
done
