# the plain fused rollout with the LDS line table (ML2048_ROW_LUT=1, default) against the arithmetic move (=0), same box
set -o pipefail
mkdir -p gpurun_out/lut
for rep in 1 2; do
 for lut in 1 0; do
  for m in 65536 262144 1048576; do
   steps=400; [ $m -gt 65536 ] && steps=120; [ $m -gt 262144 ] && steps=40
   ML2048_ROW_LUT=$lut python bench.py --no-extras --no-cpu-baseline --boards $m --steps $steps --warmup 20 > gpurun_out/lut/b.json 2> gpurun_out/lut/b.err || { tail -5 gpurun_out/lut/b.err; exit 1; }
   python - $lut $m <<'PY'
import json, sys
d=json.load(open("gpurun_out/lut/b.json"))
print("lut", sys.argv[1], "M", sys.argv[2], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
  done
 done
done
