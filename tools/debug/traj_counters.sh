#!/bin/bash
# SQ / memory-pipeline counters of the trajectory-writing rollout at 1 M boards (separate --pmc passes)
OUT=gpurun_out/traj_cnt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC|TCP|TA|TD|SQ|GRBM)_[A-Z0-9_]+" | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
grep -E "STALL|WRREQ|BUSY|WRITE|WR_" $OUT/counter_names.txt | head -80
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --mode fused-traj --boards 1048576 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/p$i.json 2> $OUT/p$i.log || { echo "pass $i failed ($set)"; tail -2 $OUT/p$i.log; }
done
python tools/sq_summary.py --M 1048576 $OUT/p? $OUT/p?? 2>/dev/null > $OUT/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/traj_cnt/summary.json"))
print("launch us", d["launch_us_under_pmc"])
for k,v in d["counters"].items(): print(k, "per launch %.4g"%v["per_launch"], "per wave-step %.2f"%v["per_wave_step"])
PY
