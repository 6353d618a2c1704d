# average latency of the rollout kernel's LDS / vector-memory instructions (SQ_INST_LEVEL_* / SQ_INSTS_*): sq_levels.sh <variant|default> [M]
set -o pipefail
V=$1; M=${2:-65536}
OUT=gpurun_out/sqlev_$V; rm -rf $OUT; mkdir -p $OUT
R=$PWD
if [ $V != default ]; then export ML2048_HIP_LIB=$R/ml2048_amd/csrc/variants/$V.so; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --no-cpu-baseline --no-extras --boards $M --steps 10 --warmup 5 > $OUT/p$i.json 2> $OUT/p$i.log || { echo "pass $i failed: $set"; tail -3 $OUT/p$i.log; }
done
python tools/sq_summary.py --M $M $OUT/p? > $OUT/summary.json
echo "== $V M=$M"; python - $OUT/summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = {k: v["per_wave_step"] for k, v in d["counters"].items()}
for k, v in sorted(c.items()):
    print("%-28s per wave-step %10.2f" % (k, v))
if c.get("SQ_INSTS_LDS"): print("avg LDS latency (LEVEL/INSTS)  %.1f" % (c["SQ_INST_LEVEL_LDS"] / c["SQ_INSTS_LDS"]))
if c.get("SQ_INSTS_VMEM_RD"): print("avg VMEM latency (LEVEL/INSTS) %.1f" % (c["SQ_INST_LEVEL_VMEM"] / c["SQ_INSTS_VMEM_RD"]))
print("launch us", d.get("launch_us_under_pmc"))
PY
