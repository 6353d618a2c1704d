// devrng.hip - the host PCG64 draw schedule of the bit-exact spawn mode, computed ON THE GPU.
//
// What hostrng.cpp does on the host (the numpy Generator(PCG64) draw order of VecGame.prepare /
// VecGame.step, src/ml2048/game_numba.py:589-591, 622-626, 670, including the in-place
// Generator.permuted table regeneration) as ONE single-workgroup kernel, so that a fused rollout
// in bit-exact mode needs no host work at all and the schedule of rollout k+1 is produced on one
// CU while the other 255 run rollout k.  Checked word for word against the host implementation
// (itself checked against numpy) in tests/test_gpu_parity.py::test_device_schedule_*.
//
// Per table regeneration (~20.5 k rejection-sampled 32-bit draws), 1024 threads:
//   1. every thread jumps the 128-bit LCG to its own position (state_t = A^(12t) s + G_(12t) c from a
//      precomputed table; the LCG is position-addressable) and emits 12 outputs = 24 words; only the
//      low 4 bits of a word decide acceptance, so one byte per word goes to LDS (24.6 KB);
//   2. the Fisher-Yates scan is a 15-state automaton over the word stream (state = index i; a word
//      accepted in state i moves to i-1; 1 wraps to 15 = next row).  Thread t owns words
//      [24t, 24t+24); the state at each segment start is found by relaxation: every thread walks its
//      segment from its current guess and hands the end state to its neighbour until nothing changes.
//      Thread 0's start is known (15), trajectories coalesce within a row or two, so a few sweeps
//      suffice; the fixed point is unique and exact;
//   3. a block scan of the per-segment row completions numbers the rows; a second walk records where
//      each of the 1024 rows starts;
//   4. thread r replays row r from its start and applies the swaps to its row of the table (kept in
//      LDS, column-major so that lanes hit consecutive bytes);
//   5. the 1024 float draws that follow and the cursor update (jump-ahead by the consumed count).
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/ml2048_hip.h"

namespace {

typedef unsigned __int128 u128;

__host__ __device__ inline u128 lcg_mul() { return ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull; }

__device__ __forceinline__ uint64_t xsl_rr(u128 st) {
    const uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64 - rot) & 63));
}

constexpr int kThreads = 1024;
constexpr int kStepsPerThread = 12;                       // LCG steps generated per thread
constexpr int kWords = kThreads * kStepsPerThread * 2;    // 24576 words per regeneration window
constexpr int kSeg = kWords / kThreads;                   // 24 words per thread segment

// workspace layout (u128 pairs {P, G}: state after k steps = P s + G c)
//   [0, 1024)      k = 12 t       (per-thread jump)
//   [1024, 1040)   k = 2^j, j<16  (binary jump-ahead)
struct Jump {
    u128 P, G;
};

__global__ __launch_bounds__(kThreads) void k_devrng_init(Jump *ws) {
    const int t = threadIdx.x;
    const u128 A = lcg_mul();
    // (P,G) for k steps by the binary method on (A, 1)
    auto jump_of = [&](uint32_t k) {
        u128 accP = 1, accG = 0, curP = A, curG = 1;
        while (k) {
            if (k & 1u) {
                accG = curP * accG + curG;
                accP = curP * accP;
            }
            curG = curP * curG + curG;
            curP = curP * curP;
            k >>= 1;
        }
        Jump j;
        j.P = accP;
        j.G = accG;
        return j;
    };
    ws[t] = jump_of((uint32_t)t * kStepsPerThread);
    if (t < 16) ws[kThreads + t] = jump_of(1u << t);
}

__device__ __forceinline__ u128 jump_ahead(u128 s, u128 c, uint32_t k, const Jump *__restrict__ pow2) {
    for (int j = 0; j < 16; ++j)
        if ((k >> j) & 1u) s = pow2[j].P * s + pow2[j].G * c;
    return s;
}

__device__ __forceinline__ uint32_t accept_bits(uint32_t low) {
    // bit i (1..15) <=> a word with low nibble `low` is accepted in state i:
    // (low & mask(i)) <= i with mask = 15 (i>=8), 7 (i>=4), 3 (i>=2), 1 (i==1)
    uint32_t m = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const uint32_t mask = i >= 8 ? 15u : i >= 4 ? 7u : i >= 2 ? 3u : 1u;
        m |= ((low & mask) <= (uint32_t)i ? 1u : 0u) << i;
    }
    return m;
}

__global__ __launch_bounds__(kThreads) void k_devrng_schedule(const Jump *__restrict__ ws, ml2048_devrng *rng,
                                                              uint8_t *randperm, float *randfloat, int N,
                                                              double two_prob, int32_t *__restrict__ schedule,
                                                              uint8_t *__restrict__ perm_epochs,
                                                              uint32_t *__restrict__ two_masks, int max_epochs,
                                                              int32_t *__restrict__ n_epochs) {
    __shared__ uint8_t nib[kWords + 8];          // low nibble of every word of the current window
    __shared__ uint8_t rows[16][kThreads];       // the permutation table, rows[i][r] = table[r][i]
    __shared__ uint16_t acc_tab[16];
    __shared__ uint8_t seg_state[kThreads + 1];  // automaton state at each segment start
    __shared__ int scan_tmp[kThreads / 64];
    __shared__ uint32_t row_start[kThreads + 1];
    __shared__ uint32_t s_float_bits[16];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const Jump *pow2 = ws + kThreads;
    const u128 A = lcg_mul();
    u128 cur = ((u128)rng->state_hi << 64) | rng->state_lo;
    const u128 inc = ((u128)rng->inc_hi << 64) | rng->inc_lo;
    int has32 = rng->has_uint32 != 0;
    uint32_t buf32 = rng->uinteger;
    int rs = rng->rand_step;
    int err = 0;

    if (t < 16) acc_tab[t] = (uint16_t)accept_bits((uint32_t)t);
    {   // my row of the table -> LDS (column-major) and epoch 0 of the output
        const uint4 v = reinterpret_cast<const uint4 *>(randperm)[t];
        reinterpret_cast<uint4 *>(perm_epochs)[t] = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) rows[i][t] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    }
    if (wave == 0) {
        const float f = randfloat[lane & 15];
        const unsigned long long m = __ballot((double)f < two_prob);
        if (lane == 0) two_masks[0] = (uint32_t)(m & 0xFFFFull);
    }
    int E = 1;
    __syncthreads();
#ifdef ML2048_DEVRNG_STATS
    long long t_gen = 0, t_relax = 0, t_rows = 0, t_replay = 0, t_float = 0, sweeps = 0, t_all = clock64();
#define ML_T0 long long _t0 = clock64();
#define ML_T(acc) { long long _t1 = clock64(); acc += _t1 - _t0; _t0 = _t1; }
#else
#define ML_T0
#define ML_T(acc)
#endif

    const uint64_t kNine = 8106479329266893ull; // u >= 0.9  <=>  (out >> 11) >= 0.9 * 2^53 (exact)

    for (int step = 0; step < N; ++step) {
        // ---- VecGame.prepare's draws (game_numba.py:622-626); every thread tracks the cursor
        cur = cur * A + inc;
        const uint64_t o = xsl_rr(cur);
        const bool regen = ((o >> 11) >= kNine) || rs >= ML2048_RAND_SIZE;
        if (regen) {
            rs = 0;
            if (E >= max_epochs) {
                err = 1;
                break;
            }
            ML_T0
            // (1) the word window: [pending half] + halves of the next kThreads*12 outputs
            const int p0 = has32 ? 1 : 0;
            if (p0 && t == 0) nib[0] = (uint8_t)(buf32 & 15u);
            {
                u128 st = ws[t].P * cur + ws[t].G * inc;
                uint8_t *dst = nib + p0 + 2 * kStepsPerThread * t;
#pragma unroll 4
                for (int k = 0; k < kStepsPerThread; ++k) {
                    st = st * A + inc;
                    const uint64_t out = xsl_rr(st);
                    dst[2 * k] = (uint8_t)(out & 15u);
                    dst[2 * k + 1] = (uint8_t)((out >> 32) & 15u);
                }
            }
            seg_state[t] = 15;
            __syncthreads();
            ML_T(t_gen)
            // (2) relaxation: state at the start of my segment.  The acceptance bitmaps of my 24
            //     words are loaded once (they do not depend on the state), so a sweep is a pure ALU
            //     chain: i -= (bits_k >> i) & 1 - no LDS latency inside it.
            uint32_t accw[kSeg / 2]; // two 16-bit bitmaps per register
            {
                const uint8_t *seg = nib + kSeg * t;
#pragma unroll
                for (int k = 0; k < kSeg; k += 2)
                    accw[k / 2] = (uint32_t)acc_tab[seg[k]] | ((uint32_t)acc_tab[seg[k + 1]] << 16);
            }
            for (;;) {
                int i = seg_state[t];
#pragma unroll
                for (int k = 0; k < kSeg; ++k) {
                    i -= (accw[k / 2] >> (16 * (k & 1) + i)) & 1;
                    i = i == 0 ? 15 : i;
                }
                const int nxt = t + 1 < kThreads ? seg_state[t + 1] : i;
                const int changed = (t + 1 < kThreads) && nxt != i;
                __syncthreads();
                if (changed) seg_state[t + 1] = (uint8_t)i;
#ifdef ML2048_DEVRNG_STATS
                ++sweeps;
#endif
                if (!__syncthreads_or(changed)) break;
            }
            ML_T(t_relax)
            // (3) number the rows: completions per segment -> exclusive scan -> row starts
            int cnt = 0;
            {
                int i = seg_state[t];
#pragma unroll
                for (int k = 0; k < kSeg; ++k) {
                    i -= (accw[k / 2] >> (16 * (k & 1) + i)) & 1;
                    if (i == 0) {
                        i = 15;
                        ++cnt;
                    }
                }
            }
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int n = __shfl_up(incl, off, 64);
                if (lane >= off) incl += n;
            }
            if (lane == 63) scan_tmp[wave] = incl;
            __syncthreads();
            int before = incl - cnt;
            for (int w = 0; w < wave; ++w) before += scan_tmp[w];
            if (t == 0) row_start[0] = 0;
            {
                int i = seg_state[t], done = before;
#pragma unroll
                for (int k = 0; k < kSeg; ++k) {
                    i -= (accw[k / 2] >> (16 * (k & 1) + i)) & 1;
                    if (i == 0) {
                        i = 15;
                        ++done; // row (done-1) just completed; row `done` starts at the next word
                        if (done <= kThreads) row_start[done] = (uint32_t)(kSeg * t + k + 1);
                    }
                }
            }
            __syncthreads();
            int total_rows = 0;
            for (int w = 0; w < kThreads / 64; ++w) total_rows += scan_tmp[w];
            // LDS values are per-lane to the compiler; these are wave-uniform by construction, and
            // saying so keeps the whole generator cursor (128-bit math) on the scalar unit
            total_rows = __builtin_amdgcn_readfirstlane(total_rows);
            if (total_rows < kThreads) { // the window did not cover 1024 rows (never in practice)
                err = 2;
                break;
            }
            const uint32_t used = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_start[kThreads]); // words consumed by the 1024 rows
            if (used + ML2048_RAND_SIZE > (uint32_t)kWords) {
                err = 2;
                break;
            }
            ML_T(t_rows)
            // (4) replay my row and apply its swaps (Generator.permuted, i = 15..1)
            {
                uint32_t pos = row_start[t];
                int i = 15;
                while (i >= 1) {
                    const uint32_t low = nib[pos++];
                    if ((acc_tab[low] >> i) & 1) {
                        const uint32_t mask = i >= 8 ? 15u : i >= 4 ? 7u : i >= 2 ? 3u : 1u;
                        const uint32_t j = low & mask;
                        const uint8_t a = rows[i][t], b = rows[j][t];
                        rows[j][t] = a;
                        rows[i][t] = b;
                        --i;
                    }
                }
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q >> 2] |= (uint32_t)rows[q][t] << (8 * (q & 3));
                reinterpret_cast<uint4 *>(perm_epochs)[(size_t)E * kThreads + t] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            __syncthreads();
            ML_T(t_replay)
            // (5) Generator.random(float32): word used+t of the window; needs the full 32 bits, so
            //     the output is recomputed: W[k] (k >= p0) = half (k-p0)&1 of LCG step (k-p0)/2 + 1
            {
                const uint32_t k = used + (uint32_t)t;
                uint32_t word;
                if (k < (uint32_t)p0) {
                    word = buf32;
                } else {
                    const uint32_t stp = (k - p0) / 2;           // 0-based step inside the window
                    const uint32_t blk = stp / kStepsPerThread;  // table jump + a few single steps
                    u128 st = ws[blk].P * cur + ws[blk].G * inc;
                    for (uint32_t q = 0; q <= stp - blk * kStepsPerThread; ++q) st = st * A + inc;
                    const uint64_t out = xsl_rr(st);
                    word = ((k - p0) & 1u) ? (uint32_t)(out >> 32) : (uint32_t)out;
                }
                const float f = (float)(word >> 8) * (1.0f / 16777216.0f);
                randfloat[t] = f;
                if (t < 16) s_float_bits[t] = ((double)f < two_prob) ? 1u : 0u;
            }
            __syncthreads();
            if (t == 0) {
                uint32_t m = 0;
                for (int c = 0; c < 16; ++c) m |= s_float_bits[c] << c;
                two_masks[E] = m;
            }
            // cursor: used + 1024 words of the window are gone
            {
                const uint32_t wc = used + ML2048_RAND_SIZE - (uint32_t)p0; // words taken from fresh outputs
                const uint32_t steps = (wc + 1) / 2;
                has32 = (int)(wc & 1u);
                const u128 nxt = jump_ahead(cur, inc, steps, pow2);
                if (has32) buf32 = (uint32_t)(xsl_rr(nxt) >> 32);
                cur = nxt;
            }
            ++E;
            __syncthreads();
            ML_T(t_float)
        }
        // rand_offset of prepare (:626) and of step (:670): integers(0, 1024) = next32 >> 22
        uint32_t w1, w2;
        if (has32) {
            w1 = buf32;
            has32 = 0;
        } else {
            cur = cur * A + inc;
            const uint64_t oo = xsl_rr(cur);
            w1 = (uint32_t)oo;
            buf32 = (uint32_t)(oo >> 32);
            has32 = 1;
        }
        if (has32) {
            w2 = buf32;
            has32 = 0;
        } else {
            cur = cur * A + inc;
            const uint64_t oo = xsl_rr(cur);
            w2 = (uint32_t)oo;
            buf32 = (uint32_t)(oo >> 32);
            has32 = 1;
        }
        if (t == 0) {
            schedule[3 * step + 0] = E - 1;
            schedule[3 * step + 1] = rs + (int32_t)(w1 >> 22);
            schedule[3 * step + 2] = rs + (int32_t)(w2 >> 22);
        }
        rs += 1;
    }

    // final state: table rows, generator, rand_step
    {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q >> 2] |= (uint32_t)rows[q][t] << (8 * (q & 3));
        reinterpret_cast<uint4 *>(randperm)[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (t == 0) {
        rng->state_hi = (uint64_t)(cur >> 64);
        rng->state_lo = (uint64_t)cur;
        rng->has_uint32 = has32;
        rng->uinteger = buf32;
        rng->rand_step = rs;
        rng->error = err;
        *n_epochs = E;
#ifdef ML2048_DEVRNG_STATS
        long long *dbg = reinterpret_cast<long long *>(n_epochs + 2);
        dbg[0] = sweeps; dbg[1] = t_gen; dbg[2] = t_relax; dbg[3] = t_rows; dbg[4] = t_replay; dbg[5] = t_float;
        dbg[6] = clock64() - t_all;
#endif
    }
}

thread_local char g_err2[256] = "";

} // namespace

extern "C" {

int64_t ml2048_devrng_workspace_bytes(void) { return (int64_t)sizeof(Jump) * (kThreads + 16); }

int ml2048_devrng_init(void *stream, void *workspace) {
    if (!workspace || (uintptr_t)workspace % 16) return ML2048_E_ALIGN;
    hipLaunchKernelGGL(k_devrng_init, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, (Jump *)workspace);
    return (int)hipGetLastError();
}

int ml2048_devrng_schedule(void *stream, const void *workspace, ml2048_devrng *rng, uint8_t *randperm,
                           float *randfloat, int32_t n_steps, double two_prob, int32_t *schedule,
                           uint8_t *perm_epochs, uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs) {
    if (!workspace || !rng || !randperm || !randfloat || !schedule || !perm_epochs || !two_masks || !n_epochs ||
        n_steps <= 0 || max_epochs <= 0)
        return ML2048_E_INVALID;
    if (((uintptr_t)workspace | (uintptr_t)randperm | (uintptr_t)perm_epochs) % 16) return ML2048_E_ALIGN;
    hipLaunchKernelGGL(k_devrng_schedule, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, (const Jump *)workspace, rng,
                       randperm, randfloat, n_steps, two_prob, schedule, perm_epochs, two_masks, max_epochs, n_epochs);
    return (int)hipGetLastError();
}

} // extern "C"
