#!/bin/bash
# throughput of the host schedule stream alone under different compiler flags (8 workers)
cd "$(dirname "$0")"
for flags in "-O2 -mssse3" "-O3 -mssse3" "-O2 -mavx2 -mbmi2" "-O3 -mavx2 -mbmi2 -mtune=znver4" "-O3 -march=x86-64-v3"; do
  g++ $flags -fopenmp -pthread -std=c++17 -o /tmp/stream_bench_f stream_bench.cpp 2>/dev/null || { echo "build failed: $flags"; continue; }
  for T in 8; do echo -n "[$flags] T=$T: "; ML2048_HOST_THREADS=$T /tmp/stream_bench_f 128 600 2>&1 | tail -1; done
  echo -n "[$flags] T=1: "; ML2048_HOST_THREADS=1 /tmp/stream_bench_f 128 100 2>&1 | tail -1
done
