# same-box A/B of the default build against several variant libraries, plain C2 rollout, alternating
set -o pipefail
mkdir -p gpurun_out/abm
for rep in $(seq 1 ${AB_REPS:-3}); do
  for v in default "$@"; do
    if [ $v = default ]; then unset ML2048_HIP_LIB; else export ML2048_HIP_LIB=$PWD/ml2048_amd/csrc/variants/$v.so; fi
    python bench.py --no-extras --no-cpu-baseline --steps 300 ${AB_ARGS:-} > gpurun_out/abm/b.json || exit 1
    python - $v $rep <<'PY'
import json, sys
d=json.load(open("gpurun_out/abm/b.json"))
print(sys.argv[1], sys.argv[2], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
done
