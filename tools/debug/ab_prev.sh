# same-box A/B of the default build against a variant library (ML2048_HIP_LIB), plain C2 rollout, alternating
set -o pipefail
mkdir -p gpurun_out/abprev
V=${1:-ml2048_amd/csrc/libml2048_hip_prev.so}
for rep in 1 2 3 4; do
  for v in default variant; do
    if [ $v = default ]; then unset ML2048_HIP_LIB; else export ML2048_HIP_LIB=$PWD/$V; fi
    python bench.py --no-extras --no-cpu-baseline --steps 300 ${AB_ARGS:-} > gpurun_out/abprev/b_${v}_$rep.json || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/abprev/b_${v}_$rep.json"))
print("$v $rep", "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
done
