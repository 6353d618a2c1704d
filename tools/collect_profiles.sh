#!/bin/bash
# Collect the round's judged evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>      e.g.  r01_i
# 1. default bench line; 2. same command under rocprofv3 --kernel-trace --stats; 3. HBM traffic of the
# dominant kernel (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md prescribes);
# 4. SQ issue counters of the rollout kernel (separate passes, --kernel-trace only).
set -o pipefail
TAG=${1:-r01_x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline --steps 100 \
    > $OUT/bench_default_under_rocprof.json 2> $OUT/stats.log || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 5 \
      > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log || exit 1
done
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 5 \
      > $OUT/sq$i.json 2> $OUT/sq$i.log || { echo "sq pass $i failed (counter set: $set)"; tail -3 $OUT/sq$i.log; }
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_summary.json || true
python tools/sq_summary.py $OUT/sq? > $OUT/sq_counters.json || true
cat $OUT/sq_counters.json
