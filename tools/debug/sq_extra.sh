# extra SQ counters of the plain C2 rollout kernel: instruction fetch, branches, VMEM latency, VALU cycles
set -o pipefail
OUT=gpurun_out/sqx; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VALU" "SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_MISC" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 5 "$@" > $OUT/p$i.json 2> $OUT/p$i.log || { echo "pass $i failed: $set"; tail -3 $OUT/p$i.log; }
done
python tools/sq_summary.py --M ${SQX_M:-65536} $OUT/p? > $OUT/summary.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/sqx/summary.json"))
for k, v in sorted(d["counters"].items()):
    print("%-28s per wave-step %10.2f" % (k, v["per_wave_step"]))
print("launch us", d.get("launch_us_under_pmc"))
PY
