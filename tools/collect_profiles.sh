#!/bin/bash
# Collect the round's judged evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag> [part ...]      e.g.  tools/collect_profiles.sh r02_d bench stats pmc sq
# parts:
#   bench   default bench line (C2 + extras: stepwise, C3 obs, fused 262144, rocRAND)
#   stats   the same command under rocprofv3 --kernel-trace --stats (its kernel averages must agree with the line's)
#   pmc     HBM traffic of the dominant kernels: FETCH_SIZE / WRITE_SIZE in separate --pmc passes, as
#           MI355X_MICROARCH.md prescribes (fused C2, fused 262144, C3 obs)
#   sq      SQ issue counters of the rollout kernel (separate passes, --kernel-trace only) for the variants
#           bench.py prices: plain small-M pass (C2), plain large-M pass (262144), masked small / large
set -o pipefail
TAG=${1:-r02_x}; shift
PARTS=${*:-bench stats pmc sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has bench; then
  python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
  python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench_default.err || exit 1
fi
if has stats; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 100 \
      > $OUT/bench_default_under_rocprof.json 2> $OUT/stats.log || exit 1
fi
pmc_pass() { # name, bench args...
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${name}_$c -- python3 bench.py --no-cpu-baseline --no-extras "$@" \
        > $OUT/pmc_${name}_$c.json 2> $OUT/pmc_${name}_$c.log || return 1
  done
  python tools/pmc_summary.py $OUT/pmc_${name}_FETCH_SIZE $OUT/pmc_${name}_WRITE_SIZE $OUT/pmc_${name}_summary.json || true
}
if has pmc; then
  pmc_pass fused_c2 --steps 10 --warmup 5 || exit 1
  pmc_pass fused_262144 --boards 262144 --steps 10 --warmup 5 || exit 1
  pmc_pass obs_262144 --mode obs --boards 262144 --steps 2 --warmup 1 || exit 1
  pmc_pass stepwise_c2 --mode stepwise --steps 2 --warmup 1 || exit 1
fi
sq_pass() { # name, M, bench args...
  local name=$1 M=$2; shift 2
  local i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq_${name}_$i -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 5 "$@" \
        > $OUT/sq_${name}_$i.json 2> $OUT/sq_${name}_$i.log || { echo "sq pass $name/$i failed (counter set: $set)"; tail -3 $OUT/sq_${name}_$i.log; }
  done
  python tools/sq_summary.py --M $M $OUT/sq_${name}_? > $OUT/sq_counters_$name.json || true
}
if has sq; then
  sq_pass plain_small 65536
  sq_pass plain_large 262144 --boards 262144
  sq_pass masked_small 65536 --masked 1
  sq_pass masked_large 262144 --boards 262144 --masked 1
  sq_pass rocrand_small 65536 --spawn rocrand
  sq_pass rocrand_large 262144 --boards 262144 --spawn rocrand
fi
ls $OUT | head -50
