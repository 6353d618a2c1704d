#!/bin/bash
# throughput of the host schedule stream alone on this box, over worker-thread counts
cd "$(dirname "$0")"
g++ -O2 -mssse3 -fopenmp -pthread -std=c++17 $1 -o /tmp/stream_bench stream_bench.cpp 2>/dev/null || exit 1
nproc
for T in 1 2 4 6 8 12; do echo -n "T=$T: "; ML2048_HOST_THREADS=$T /tmp/stream_bench 128 400 2>&1 | tail -${2:-1}; done
