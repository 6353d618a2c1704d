// throughput of the asynchronous schedule stream alone (no GPU): g++ -O2 -mssse3 -fopenmp -pthread stream_bench.cpp
#include <chrono>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "../../ml2048_amd/csrc/hostrng.cpp"
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 128, reps = argc > 2 ? atoi(argv[2]) : 300;
    ml2048_pcg64 r{123, 456, 789, 1011 | 1, 0, 0};
    std::vector<uint8_t> perm(16384);
    for (int i = 0; i < 16384; i++) perm[i] = i & 15;
    std::vector<float> flt(1024, 0.5f);
    ml2048_sched_stream *s = ml2048_sched_stream_create(&r, perm.data(), flt.data(), 0, n, 0.8, 3);
    const int32_t *sched; const uint8_t *ep; const uint32_t *masks; int32_t ne;
    for (int w = 0; w < 20; w++) { ml2048_sched_stream_next(s, &sched, &ep, &masks, &ne, 0, 0, 0, 0); ml2048_sched_stream_release(s); }
    auto t0 = std::chrono::steady_clock::now();
    long tot = 0; unsigned long long sum = 0;
    for (int rep = 0; rep < reps; rep++) {
        ml2048_sched_stream_next(s, &sched, &ep, &masks, &ne, 0, 0, 0, 0);
        tot += ne - 1; sum += ep[(size_t)(ne - 1) * 16384 + 77] + sched[5];
        ml2048_sched_stream_release(s);
    }
    auto t1 = std::chrono::steady_clock::now();
    printf("%.3f ms per schedule, %.1f regens (chk %llu)\n", std::chrono::duration<double, std::milli>(t1 - t0).count() / reps, (double)tot / reps, sum);
    ml2048_sched_stream_destroy(s);
}
