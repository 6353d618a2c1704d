# The shader clock the plain rollout kernel actually runs at: GRBM_GUI_ACTIVE (GPU clocks while busy) and SQ_BUSY_CYCLES
# over the kernel's duration, per board count.   clock.sh [M ...]
set -o pipefail
OUT=gpurun_out/clock; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for M in ${@:-65536 131072 262144 1048576}; do
 timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/m$M -- python3 bench.py --no-cpu-baseline --no-extras --boards $M --steps 12 --warmup 6 > $OUT/m$M.json 2> $OUT/m$M.log || { echo "M=$M failed"; tail -3 $OUT/m$M.log; }
 python - $OUT/m$M $M <<'PY'
import csv, glob, statistics as st, sys
d, M = sys.argv[1], int(sys.argv[2])
acc, dur = {}, []
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rollout" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if not dur:
    print("M=%d: no k_rollout rows" % M); sys.exit(0)
us = st.median(dur) / 1e3
line = "M=%8d  kernel %7.1f us" % (M, us)
for k, v in sorted(acc.items()):
    m = st.median(v)
    line += "  %s=%.0f (%.3f GHz if per instance; /8: %.3f)" % (k, m, m / us / 1e3, m / 8 / us / 1e3)
print(line)
PY
done
