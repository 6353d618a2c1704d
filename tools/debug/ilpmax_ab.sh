# which small-M / large-M build is faster at a given board count (ML2048_PLAIN_ILP_MAX moves the switch)
set -o pipefail
mkdir -p gpurun_out/ilpmax
for m in 131072 196608 262144 524288; do
  for rep in 1 2; do
    for lim in 131072 1048576; do
      ML2048_PLAIN_ILP_MAX=$lim python bench.py --no-extras --no-cpu-baseline --steps 150 --boards $m > gpurun_out/ilpmax/b.json || exit 1
      python - $m $lim <<'PY'
import json, sys
d=json.load(open("gpurun_out/ilpmax/b.json"))
print("M", sys.argv[1], "small-M build up to", sys.argv[2], "%.4g"%d["value"], "kernel us %.2f"%(d["roofline"]["avg_launch_ms"]*1e3), flush=True)
PY
    done
  done
done
