cd tools/microbench && g++ -O2 -mssse3 -fopenmp -pthread -std=c++17 -o /tmp/stream_bench stream_bench.cpp 2>/dev/null || exit 1
nproc; for T in 8 9 10 11 12 14; do echo -n "T=$T: "; ML2048_HOST_THREADS=$T /tmp/stream_bench 128 600 2>&1 | tail -1; done
for T in 9 12; do echo -n "taskset16 T=$T: "; ML2048_HOST_THREADS=$T taskset -c 0-15 /tmp/stream_bench 128 600 2>&1 | tail -1; done
