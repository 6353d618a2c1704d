#!/bin/bash
# Real-DRAM roofline points (working set >> the 256 MB Infinity Cache): the per-step kernels at 16 M boards (1.6 GB of
# state), kernel time by rocprofv3 --kernel-trace --stats and HBM bytes by FETCH_SIZE / WRITE_SIZE in separate passes.
OUT=gpurun_out/dram; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
M=${1:-16777216}
for mode in stepwise3 stepwise; do
  ARGS="--mode $mode --boards $M --env-steps 6 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$mode -- python3 bench.py $ARGS > $OUT/$mode.json 2> $OUT/$mode.log || { tail -3 $OUT/$mode.log; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${mode}_$c -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_${mode}_$c.log || { tail -3 $OUT/pmc_${mode}_$c.log; exit 1; }
  done
  python tools/pmc_summary.py $OUT/pmc_${mode}_FETCH_SIZE $OUT/pmc_${mode}_WRITE_SIZE $OUT/pmc_$mode.json
  head -5 $OUT/stats_$mode/*/*kernel_stats.csv | cut -c1-150
done
