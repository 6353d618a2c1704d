// valu_rate.hip - what does one VALU instruction of the rollout kernels' mix cost a gfx950 SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
//
// MI355X_MICROARCH.md prices a wave64 VALU instruction at 2 cycles on the SIMD-32 (and at 4 for a wave alone on its
// SIMD).  That is the FP32 FMA rate; the rollout kernels issue integer / bit / byte-permute instructions, and their
// VALU-issue roofline needs the rate THOSE sustain.  Every wave runs ITER iterations of 64 instructions of one opcode
// on 8 independent registers (no dependent pair closer than 8 instructions); 1, 2, 4 and 8 waves per SIMD on all
// 256 CUs.  Reported: ns per wave-instruction per SIMD (wall time by HIP events / instructions issued on one SIMD) and
// the same in shader-clock ticks of the wave's own s_memtime counter.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4000;

// 8 instructions, one per accumulator; OP3 ops take (dst, a, b, c)
#define REP8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

#define DEFINE_KERNEL(NAME, ASM_LINE)                                                                            \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, unsigned long long *ticks, uint32_t seed) {    \
        uint32_t v[8];                                                                                            \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) v[c] = seed + threadIdx.x * 31u + c * 977u + blockIdx.x;    \
        uint32_t x = seed ^ 0x01020304u, y = 0x0c050100u, z = 0;                                                  \
        asm volatile("s_mov_b64 vcc, 0x55555555\n\ts_mov_b64 s[22:23], 0x33333333\n\ts_mov_b32 s26, 7" ::: "vcc", "s22", "s23", "s26"); \
        const unsigned long long t0 = __builtin_readcyclecounter();                                               \
        for (int it = 0; it < ITER; ++it) {                                                                       \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                       \
                REP8(ASM_LINE)                                                                                    \
            }                                                                                                     \
        }                                                                                                         \
        const unsigned long long t1 = __builtin_readcyclecounter();                                               \
        uint32_t acc = 0;                                                                                         \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) acc ^= v[c];                                                \
        acc ^= z;                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                                \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;                     \
    }

#define A_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(x));
#define A_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(x));
#define A_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(x));
#define A_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[i]) : "v"(x));
#define A_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[i]));
#define A_NOT(i) asm volatile("v_not_b32 %0, %0" : "+v"(v[i]));
#define A_BFEI(i) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(v[i]));
#define A_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(w[i & 3]) : "v"(w[(i + 1) & 3]));
#define A_LSHR(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(v[i]));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(v[i]) : "v"(x));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(v[i]));
#define A_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(x) : );
#define A_CNDMASK64V(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(x) : );
#define A_CNDMASK_K(i) asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc" : "+v"(v[i]) : : );
#define A_CNDMASK_2V(i) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(v[i]) : "v"(x), "v"(y) : );
#define A_ADDC(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(v[i]) : "v"(x) : "vcc");
#define A_CNDMASK64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(v[i]) : "v"(x) : );
#define A_CMP64(i) asm volatile("v_cmp_eq_u32_e64 s[24:25], %0, %1" : : "v"(v[i]), "v"(x) : "s24", "s25");
// a compare and the select that consumes it (what `c ? a : b` on per-lane data compiles to), against the same select as
// mask arithmetic: v_bfe_i32 (mask from a bit) + v_bitop3_b32 (bitwise m ? a : b)
#define A_CMPSEL(i) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(v[i]) : "v"(x), "v"(y) : "vcc");
#define A_MASKSEL(i) asm volatile("v_bfe_i32 %1, %0, 3, 1\n\tv_bitop3_b32 %0, %1, %0, %2 bitop3:0xca" : "+v"(v[i]), "+v"(z) : "v"(y));
#define A_WRITELANE(i) asm volatile("v_writelane_b32 %0, s26, 5" : "+v"(v[i]) : : );
#define A_SAVEEXEC(i) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n\ts_and_saveexec_b64 s[28:29], vcc\n\tv_add_u32 %0, %0, %1\n\ts_or_b64 exec, exec, s[28:29]" : "+v"(v[i]) : "v"(x) : "vcc", "s28", "s29", "scc");
#define A_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, 48, %1" : "+v"(v[i]) : "v"(x));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(x));
#define A_SUB(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(v[i]) : "v"(x));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(x));
#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_CMP(i) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(v[i]), "v"(x) : "vcc");
#define A_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i & 3]));
#define A_SDWA(i) asm volatile("v_lshlrev_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(v[i]) : "v"(x));
#define A_XAD(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x), "v"(y));
#define A_FFBL(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(v[i]));
#define A_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(v[i]) : "v"(x));
#define A_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(v[i]) : "s20");

DEFINE_KERNEL(add_u32, A_ADD)
DEFINE_KERNEL(sub_u32, A_SUB)
DEFINE_KERNEL(xor_b32, A_XOR)
DEFINE_KERNEL(and_b32, A_AND)
DEFINE_KERNEL(lshrrev_b32, A_LSHR)
DEFINE_KERNEL(or_b32, A_OR)
DEFINE_KERNEL(lshlrev_b32, A_LSHL)
DEFINE_KERNEL(not_b32, A_NOT)
DEFINE_KERNEL(bfe_i32, A_BFEI)
DEFINE_KERNEL(lshl_or_b32, A_LSHLOR)
DEFINE_KERNEL(and_or_b32, A_ANDOR)
DEFINE_KERNEL(or3_b32, A_OR3)
DEFINE_KERNEL(add3_u32, A_ADD3)
DEFINE_KERNEL(xad_u32, A_XAD)
DEFINE_KERNEL(bitop3_b32, A_BITOP3)
DEFINE_KERNEL(perm_b32, A_PERM)
DEFINE_KERNEL(bfe_u32, A_BFE)
DEFINE_KERNEL(bfi_b32, A_BFI)
DEFINE_KERNEL(cndmask_b32, A_CNDMASK)
DEFINE_KERNEL(cmp_eq_u32, A_CMP)
DEFINE_KERNEL(cmp_eq_u32_e64, A_CMP64)
DEFINE_KERNEL(cndmask_b32_e64, A_CNDMASK64)
DEFINE_KERNEL(cndmask_b32_e64_vcc, A_CNDMASK64V)
DEFINE_KERNEL(cndmask_b32_e32_const, A_CNDMASK_K)
DEFINE_KERNEL(cndmask_b32_e32_2src, A_CNDMASK_2V)
DEFINE_KERNEL(addc_co_u32, A_ADDC)
DEFINE_KERNEL(cmp_then_cndmask_x2, A_CMPSEL)
DEFINE_KERNEL(bfe_then_bitop3_x2, A_MASKSEL)
DEFINE_KERNEL(writelane_b32, A_WRITELANE)
DEFINE_KERNEL(cmp_saveexec_add_or_x4, A_SAVEEXEC)
DEFINE_KERNEL(dot4_u32_u8, A_DOT4)
DEFINE_KERNEL(mad_u32_u24, A_MAD24)
DEFINE_KERNEL(mul_lo_u32, A_MULLO)
DEFINE_KERNEL(lshlrev_b32_sdwa, A_SDWA)
DEFINE_KERNEL(ffbl_b32, A_FFBL)
DEFINE_KERNEL(bcnt_u32_b32, A_BCNT)
DEFINE_KERNEL(mov_b32, A_MOV)
DEFINE_KERNEL(readlane_b32, A_READLANE)
DEFINE_KERNEL(fma_f32, A_FMA)

__global__ __launch_bounds__(256) void k_lshlrev_b64(uint32_t *out, unsigned long long *ticks, uint32_t seed) {
    uint64_t w[4];
    for (int c = 0; c < 4; ++c) w[c] = seed + threadIdx.x * 31u + c * 977u + blockIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            REP8(A_LSHL64)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint64_t acc = 0;
    for (int c = 0; c < 4; ++c) acc ^= w[c];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_mov_b64(uint32_t *out, unsigned long long *ticks, uint32_t seed) {
    uint64_t w[4];
    for (int c = 0; c < 4; ++c) w[c] = seed + threadIdx.x * 31u + c * 977u + blockIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            REP8(A_MOV64)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint64_t acc = 0;
    for (int c = 0; c < 4; ++c) acc ^= w[c];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

// LDS look-up latency as the rollout's line table sees it: a wave gathers 16-bit entries of a 128 KiB LDS table at
// per-lane pseudo-random addresses.  CHAIN = 1: each look-up's address depends on the previous result (pure latency);
// CHAIN = 0: four independent look-ups, then their sum (what move_board_lut does).
template <int CHAIN>
__global__ __launch_bounds__(256) void k_lds_lookup(uint32_t *out, unsigned long long *ticks, uint32_t seed) {
    __shared__ uint16_t tab[65536];
    for (int i = threadIdx.x; i < 65536; i += 256) tab[i] = (uint16_t)((i * 40503u + seed) >> 3);
    __syncthreads();
    uint32_t a = (seed + threadIdx.x * 977u) & 0xFFFFu, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if (CHAIN) {
#pragma unroll
            for (int r = 0; r < 8; ++r) a = tab[a];
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t r0 = tab[a], r1 = tab[a ^ 0x1234u], r2 = tab[(a + 0x4321u) & 0xFFFFu], r3 = tab[a ^ 0x00FFu];
                a = (r0 + r1 + r2 + r3) & 0xFFFFu;
            }
        }
        acc += a;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

static void run_lds(uint32_t *out, unsigned long long *tk) {
    for (int chain = 1; chain >= 0; --chain) {
        for (int wpc : {4, 16}) { // waves per CU: one workgroup of 256 or 1024 threads... (LDS-bound: one WG per CU)
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            const int blocks = 256 * (wpc / 4);
            for (int rep = 0; rep < 2; ++rep) {
                if (rep) CHECK(hipEventRecord(e0));
                if (chain) hipLaunchKernelGGL(k_lds_lookup<1>, dim3(blocks), dim3(256), 0, 0, out, tk, 12345u);
                else hipLaunchKernelGGL(k_lds_lookup<0>, dim3(blocks), dim3(256), 0, 0, out, tk, 12345u);
                if (rep) CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
            }
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long t = 0;
            CHECK(hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost));
            const double units = (double)ITER * (chain ? 8 : 2);
            printf("lds_lookup %-22s %2d WGs/CU in turn (256-thread WGs, 128 KiB each): %7.1f ticks per %s   (%.3f ms)\n",
                   chain ? "dependent chain" : "4 independent + sum", wpc / 4, (double)t / units,
                   chain ? "look-up" : "group of four look-ups and their combine", ms);
            fflush(stdout);
        }
    }
}

typedef void (*kern_t)(uint32_t *, unsigned long long *, uint32_t);

static void run(const char *name, kern_t k, uint32_t *out, unsigned long long *tk) {
    printf("%-24s", name);
    fflush(stdout);
    double base = 0;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps; // 256-thread workgroups: one wave on each of a CU's four SIMDs
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, tk, 12345u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, tk, 12345u);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t = 0;
        CHECK(hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost));
        const double per_wave = (double)ITER * 64;
        const double ns = ms * 1e6 / (per_wave * wps); // ns per wave-instruction issued on one SIMD
        if (wps == 1) base = ns;
        printf("  %dw: %6.3f ns (%5.2f ticks/inst/wave)", wps, ns, (double)t / per_wave);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
    printf("\n");
    fflush(stdout);
    (void)base;
}

int main() {
    uint32_t *out;
    unsigned long long *tk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHECK(hipMalloc(&tk, 256 * 8 * 4 * 8));
    printf("ITER=%d x 64 instructions per wave; ns = wall time per wave-instruction per SIMD (all SIMDs of the chip busy)\n", ITER);
#define RUN(NAME) run(#NAME, k_##NAME, out, tk);
    RUN(fma_f32) RUN(add_u32) RUN(sub_u32) RUN(xor_b32) RUN(and_b32) RUN(lshrrev_b32) RUN(or_b32) RUN(lshlrev_b32) RUN(not_b32) RUN(bfe_i32) RUN(lshl_or_b32) RUN(and_or_b32)
    RUN(or3_b32) RUN(add3_u32) RUN(xad_u32) RUN(bitop3_b32) RUN(perm_b32) RUN(bfe_u32) RUN(bfi_b32) RUN(cndmask_b32)
    RUN(cmp_eq_u32) RUN(cmp_eq_u32_e64) RUN(cndmask_b32_e64) RUN(cndmask_b32_e64_vcc) RUN(cndmask_b32_e32_const) RUN(cndmask_b32_e32_2src) RUN(addc_co_u32) RUN(cmp_then_cndmask_x2) RUN(bfe_then_bitop3_x2) RUN(writelane_b32)
    RUN(cmp_saveexec_add_or_x4) RUN(dot4_u32_u8) RUN(mad_u32_u24) RUN(mul_lo_u32) RUN(lshlrev_b32_sdwa) RUN(ffbl_b32)
    RUN(bcnt_u32_b32) RUN(mov_b32) RUN(readlane_b32) RUN(lshlrev_b64) RUN(mov_b64)
    run_lds(out, tk);
    return 0;
}
