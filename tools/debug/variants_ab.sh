# parity (plain rollout tests) + same-box timing of variant libraries: variants_ab.sh v1 v2 ...   (AB_ARGS: bench args)
set -o pipefail
mkdir -p gpurun_out/abm
for v in "$@"; do
  ML2048_HIP_LIB=$PWD/ml2048_amd/csrc/variants/$v.so timeout -k 10 600 python -m pytest tests/test_gpu_plain_rollout.py -m gpu -x -q -k "${AB_TESTS:-c2_twice or chain_matrix or masked or rocrand}" > gpurun_out/abm/tests_$v.log 2>&1 || { echo "PARITY FAILED: $v"; tail -20 gpurun_out/abm/tests_$v.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/abm/tests_$v.log)"
done
bash tools/debug/ab_multi.sh "$@"
