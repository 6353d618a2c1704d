set -o pipefail
mkdir -p gpurun_out/zc
timeout -k 10 900 python -m pytest tests/test_gpu_plain_rollout.py tests/test_gpu_parity.py tests/test_gpu_step_prepare.py -x -q -m gpu > gpurun_out/zc/tests.log 2>&1 || { tail -30 gpurun_out/zc/tests.log; exit 1; }
tail -3 gpurun_out/zc/tests.log
python tools/debug/profile_rollout_host.py > gpurun_out/zc/host.log 2>&1; tail -15 gpurun_out/zc/host.log
python bench.py --no-extras > gpurun_out/zc/bench_default.json 2> gpurun_out/zc/bench.err && cat gpurun_out/zc/bench_default.json
for i in 1 2 3; do python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 | tee -a gpurun_out/zc/bench_driver.json; done
ML2048_STREAM_PINNED=0 python bench.py --no-extras --no-cpu-baseline > gpurun_out/zc/bench_copy.json && cat gpurun_out/zc/bench_copy.json
ML2048_STREAM_PINNED=0 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 | tee -a gpurun_out/zc/bench_copy_driver.json
