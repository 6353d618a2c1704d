// halfwave.hip - does a wave64 whose upper 32 lanes are inactive (exec_hi = 0) issue its VALU instructions in one
// pass on gfx950's SIMD-32 (which issues a full wave64 instruction over 2 cycles)?  And how does one wave per SIMD
// with all 64 lanes compare with two half-active waves per SIMD doing the same per-lane work - the shape a
// 65 536-board rollout would take with 32 boards per wave.
//
//   hipcc --offload-arch=gfx950 -O3 -o halfwave halfwave.hip && ./halfwave
//
// Each wave runs ITER iterations of CHAINS independent dependent-chains of integer VALU work (v_mad_u32_u24 /
// v_xor): a lone wave is latency/issue-bound exactly like the rollout kernel at C2.  Reported: wall time by HIP
// events, and the wave's own cycle count (s_memtime) per VALU instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 20000;

template <int CHAINS, bool HALF>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *cycles, uint32_t seed) {
    const int lane = threadIdx.x & 63;
    if (HALF && lane >= 32) return; // exec_hi = 0 for the rest of the kernel
    uint32_t v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = seed + threadIdx.x * 31u + c * 977u + blockIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) { // 4 dependent VALU instructions per chain per iteration
            v[c] = v[c] * 0x9E3779B1u + 0x7F4A7C15u;
            v[c] ^= v[c] >> 15;
            v[c] += (v[c] << 3);
            v[c] ^= seed;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc ^= v[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0) cycles[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int CHAINS, bool HALF>
void run(const char *name, int waves_per_simd, uint32_t *out, unsigned long long *cyc) {
    const int blocks = 256 * waves_per_simd; // 256-thread workgroups: one wave per SIMD each
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<CHAINS, HALF>), dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u); // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<CHAINS, HALF>), dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c0 = 0;
    CHECK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
    // instruction count per wave: the multiply-add may be 1-2 instructions; report per source-level op (4 per chain)
    const double ops = (double)ITER * CHAINS * 4;
    printf("%-34s chains=%d waves/SIMD=%d  %8.3f ms   %6.2f counter ticks per source op per wave   (%.3f ns/op)\n", name, CHAINS,
           waves_per_simd, ms, (double)c0 / ops, ms * 1e6 / ops);
}

int main() {
    uint32_t *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHECK(hipMalloc(&cyc, 256 * 8 * 4 * 8));
    printf("ITER=%d; every line = the same per-lane work per wave\n", ITER);
    run<1, false>("full wave, 1 dependent chain", 1, out, cyc);
    run<1, true>("half wave (exec_hi=0), 1 chain", 1, out, cyc);
    run<1, true>("half waves, 1 chain", 2, out, cyc);
    run<1, false>("full waves, 1 chain", 2, out, cyc);
    run<4, false>("full wave, 4 independent chains", 1, out, cyc);
    run<4, true>("half wave (exec_hi=0), 4 chains", 1, out, cyc);
    run<4, true>("half waves, 4 chains", 2, out, cyc);
    run<4, false>("full waves, 4 chains", 2, out, cyc);
    run<4, true>("half waves, 4 chains", 4, out, cyc);
    run<4, false>("full waves, 4 chains", 4, out, cyc);
    return 0;
}
