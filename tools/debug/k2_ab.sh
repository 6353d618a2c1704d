#!/bin/bash
# A/B of the k2_rows unroll x graph-cache behaviour under tools/debug/fuzz_device2.py (background rollouts on the same GPU).
# usage: tools/debug/k2_ab.sh <seconds per variant> <variant.so> ...   ("default" = the in-tree library)
SECS=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  if [ "$lib" = default ]; then unset ML2048_HIP_LIB; else export ML2048_HIP_LIB=$PWD/$lib; fi
  for seed in 3 4; do
    echo "== $lib seed $seed" | tee -a gpurun_out/k2_ab.log
    timeout -k 10 $((SECS + 120)) python tools/debug/fuzz_device2.py $seed $SECS >> gpurun_out/k2_ab.log 2>&1
    rc=$?
    echo "rc=$rc" | tee -a gpurun_out/k2_ab.log
    if [ $rc -ge 124 ]; then echo "timeout/kill: stopping" | tee -a gpurun_out/k2_ab.log; exit 1; fi
  done
done
exit 0
