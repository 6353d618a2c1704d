# SQ counters of the plain rollout kernel with / without the LDS line table: sq_lut.sh <M> [bench args]
set -o pipefail
M=${1:-262144}; shift
OUT=gpurun_out/sqlut_$M; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for lut in 1 0; do
 i=0
 for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES"; do
  i=$((i+1))
  ML2048_ROW_LUT=$lut timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/l${lut}_p$i -- python3 bench.py --no-cpu-baseline --no-extras --boards $M --steps 10 --warmup 5 "$@" > $OUT/l${lut}_p$i.json 2> $OUT/l${lut}_p$i.log || { echo "pass $i failed: $set"; tail -3 $OUT/l${lut}_p$i.log; }
 done
 python tools/sq_summary.py --M $M $OUT/l${lut}_p? > $OUT/summary_lut${lut}_$M.json
 echo "== lut=$lut M=$M"; python - $OUT/summary_lut${lut}_$M.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d["counters"].items()):
    print("%-28s per wave-step %10.2f" % (k, v["per_wave_step"]))
print("launch us", d.get("launch_us_under_pmc"))
PY
done
