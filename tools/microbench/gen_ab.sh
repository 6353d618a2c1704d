# PCG64 output generation (scalar 4-lane vs AVX-512) and the schedule stream with either, on this host
cd "$(dirname "$0")"
g++ -O2 -mssse3 -fopenmp -pthread -std=c++17 -o /tmp/genbench genbench.cpp 2>/dev/null && /tmp/genbench
g++ -O2 -mssse3 -fopenmp -pthread -std=c++17 -o /tmp/stream_bench stream_bench.cpp 2>/dev/null || exit 1
for T in 7 8 9 10; do for a in 0 1; do echo -n "T=$T avx512=$a: "; ML2048_HOSTRNG_AVX512=$a ML2048_HOST_THREADS=$T /tmp/stream_bench 128 600 | tail -1; done; done
