# A/B of the small-M pass's table-row prefetch distance (2 steps ahead = default build, 1 = ML2048_ROW_AHEAD=1 variant):
#   python -c "from ml2048_amd import build; build.build(extra_hipcc_flags=['-DML2048_ROW_AHEAD=1'], out='ml2048_amd/csrc/libml2048_hip_ahead1.so')"
set -o pipefail
mkdir -p gpurun_out/ahead
timeout -k 10 600 python -m pytest tests/test_gpu_plain_rollout.py -x -q -m gpu > gpurun_out/ahead/tests.log 2>&1 || { tail -30 gpurun_out/ahead/tests.log; exit 1; }
tail -2 gpurun_out/ahead/tests.log
for rep in 1 2 3; do
  for v in default ahead1; do
    if [ $v = default ]; then unset ML2048_HIP_LIB; else export ML2048_HIP_LIB=$PWD/ml2048_amd/csrc/libml2048_hip_ahead1.so; fi
    python bench.py --no-extras --no-cpu-baseline --steps 300 > gpurun_out/ahead/b_${v}_$rep.json || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/ahead/b_${v}_$rep.json"))
print("$v $rep", "%.4g"%d["value"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
done
unset ML2048_HIP_LIB
python tools/debug/driver_timeline.py > gpurun_out/ahead/timeline.log 2>&1; tail -8 gpurun_out/ahead/timeline.log
