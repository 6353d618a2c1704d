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
#include "board_ops.hpp"

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
constexpr int kRelax = 512;                               // threads that walk the automaton (one wave per SIMD:
                                                          // the walk is a dependent chain, more waves only slow it)
constexpr int kSeg = kWords / kRelax;                     // 48 words per relaxation segment

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

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would
// wait for the epoch-table and float-table global stores issued just before it (several us each)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
}

// `cont` != NULL: CONTINUATION of a data-parallel schedule (csrc/devrng2.hip) whose window of generator outputs
// ran out before step N (more than ~23 regenerations in 128 steps: rare, but it happens): cont[0] = epochs so
// far, cont[1] = error, cont[4] = steps already done, cont[5] = "ran out" flag.  This kernel then finishes steps
// cont[4]..N-1 from the generator state the other one left; when nothing is left it returns at once.
// EXPANDED: 48-byte epoch rows (board_ops.hpp expand_perm_row) as ml2048_rollout_chain consumes them.
template <bool EXPANDED>
__device__ __forceinline__ void store_epoch_row1(uint8_t *perm_epochs, size_t epoch, int t, uint4 v) {
    uint4 *out = reinterpret_cast<uint4 *>(perm_epochs);
    if (EXPANDED) {
        const uint32_t perm[4] = {v.x, v.y, v.z, v.w};
        uint32_t x[ml2048::kExpandedRowWords];
        ml2048::expand_perm_row(perm, x);
        uint4 *row = out + 3 * (epoch * kThreads + t);
        row[0] = make_uint4(x[0], x[1], x[2], x[3]);
        row[1] = make_uint4(x[4], x[5], x[6], x[7]);
        row[2] = make_uint4(x[8], x[9], x[10], x[11]);
    } else {
        out[epoch * kThreads + t] = v;
    }
}

template <bool EXPANDED>
__global__ __launch_bounds__(kThreads) void k_devrng_schedule(const Jump *__restrict__ ws, ml2048_devrng *rng,
                                                              uint8_t *randperm, float *randfloat, int N,
                                                              double two_prob, int32_t *__restrict__ schedule,
                                                              uint8_t *__restrict__ perm_epochs,
                                                              uint32_t *__restrict__ two_masks, int max_epochs,
                                                              int32_t *__restrict__ n_epochs,
                                                              const int32_t *__restrict__ cont) {
    if (cont && (!cont[5] || cont[4] >= N || cont[1])) return; // nothing left for the continuation
    __shared__ uint32_t words[kWords + 8];       // the 32-bit words of the current window (96 KiB)
    __shared__ uint16_t acc_tab[16];
    __shared__ uint8_t seg_state[2][kRelax + 1]; // automaton state (i-1) at each segment start, ping-pong
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
    const int step_begin = cont ? cont[4] : 0;

    if (t < 16) acc_tab[t] = (uint16_t)accept_bits((uint32_t)t);
    // my row of the table lives in two 64-bit registers for the whole kernel; epoch 0 of the output
    uint64_t row_lo, row_hi;
    {
        const uint4 v = reinterpret_cast<const uint4 *>(randperm)[t];
        if (!cont) store_epoch_row1<EXPANDED>(perm_epochs, 0, t, v);
        row_lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
        row_hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    if (wave == 0 && !cont) {
        const float f = randfloat[lane & 15];
        const unsigned long long m = __ballot((double)f < two_prob);
        if (lane == 0) two_masks[0] = (uint32_t)(m & 0xFFFFull);
    }
    int E = cont ? cont[0] : 1;
    const u128 myP = ws[t].P, myG = ws[t].G; // my jump to LCG step 12 t (loaded once)
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

    for (int step = step_begin; step < N; ++step) {
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
            if (p0 && t == 0) words[0] = buf32;
            {
                u128 st = myP * cur + myG * inc;
                uint32_t *dst = words + p0 + 2 * kStepsPerThread * t;
#pragma unroll 4
                for (int k = 0; k < kStepsPerThread; ++k) {
                    st = st * A + inc;
                    const uint64_t out = xsl_rr(st);
                    dst[2 * k] = (uint32_t)out;
                    dst[2 * k + 1] = (uint32_t)(out >> 32);
                }
            }
            if (t < kRelax) seg_state[0][t] = seg_state[1][t] = 14; // i = 15
            __syncthreads();
            ML_T(t_gen)
            // (2) relaxation over kRelax segments of kSeg words (threads 0..kRelax-1, one wave per
            //     SIMD).  The acceptance bitmaps of my words are loaded once (they do not depend on the
            //     state), so a sweep is a pure ALU chain: i -= (bits_k >> i) & 1.
            //     State is kept as s = i-1 (0..14) and the bitmaps pre-shifted, so one word is
            //     lshr/and/sub/min: s' = min_u32(s - accepted, 14) (0 - 1 wraps to 0xFFFFFFFF -> 14 = next row).
            //     Sweeps ping-pong between two state arrays (one barrier each); convergence is only
            //     tested every 4th sweep.
            uint32_t accw[kSeg / 2]; // two 15-bit bitmaps per register (bit s <=> accepted in state s+1)
            const bool walker = t < kRelax;
            if (walker) {
                const uint32_t *seg = words + kSeg * t;
#pragma unroll
                for (int k = 0; k < kSeg; k += 2)
                    accw[k / 2] = ((uint32_t)acc_tab[seg[k] & 15u] >> 1) | (((uint32_t)acc_tab[seg[k + 1] & 15u] >> 1) << 16);
            }
            int buf = 0;
            for (int sweep = 1;; ++sweep) {
                int changed = 0;
                if (walker) {
                    uint32_t sst = seg_state[buf][t];
#pragma unroll
                    for (int k = 0; k < kSeg; ++k) {
                        const uint32_t acc = (accw[k / 2] >> (16 * (k & 1) + sst)) & 1u;
                        sst = min(sst - acc, 14u);
                    }
                    if (t + 1 < kRelax) {
                        changed = seg_state[buf][t + 1] != (uint8_t)sst;
                        seg_state[buf ^ 1][t + 1] = (uint8_t)sst;
                    }
                }
                buf ^= 1;
#ifdef ML2048_DEVRNG_STATS
                ++sweeps;
#endif
                if ((sweep & 3) == 0) {
                    if (!__syncthreads_or(changed)) break; // nobody changed in THIS sweep: fixed point
                } else {
                    __syncthreads();
                }
            }
            ML_T(t_relax)
            // (3) number the rows: completions per segment -> exclusive scan -> row starts
            int cnt = 0;
            if (walker) {
                uint32_t sst = seg_state[buf][t];
#pragma unroll
                for (int k = 0; k < kSeg; ++k) {
                    const uint32_t acc = (accw[k / 2] >> (16 * (k & 1) + sst)) & 1u;
                    cnt += (sst == 0u) & acc; // accepted in state i = 1: the row is complete
                    sst = min(sst - acc, 14u);
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
            if (walker) {
                uint32_t sst = seg_state[buf][t];
                int done = before;
#pragma unroll
                for (int k = 0; k < kSeg; ++k) {
                    const uint32_t acc = (accw[k / 2] >> (16 * (k & 1) + sst)) & 1u;
                    if ((sst == 0u) & acc) {
                        ++done; // row (done-1) just completed; row `done` starts at the next word
                        if (done <= kThreads) row_start[done] = (uint32_t)(kSeg * t + k + 1);
                    }
                    sst = min(sst - acc, 14u);
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
            // (4) replay my row and apply its swaps (Generator.permuted, i = 15..1) to the row held
            //     in two 64-bit registers: swapping bytes i and j = xor both with (a ^ b)
            {
                uint32_t pos = row_start[t];
                int i = 15;
                uint32_t nextw = words[pos];
                while (i >= 1) {
                    const uint32_t low = nextw & 15u;
                    nextw = words[++pos]; // one word ahead: hides the LDS latency behind the swap
                    const uint32_t mask = i >= 8 ? 15u : i >= 4 ? 7u : i >= 2 ? 3u : 1u;
                    const uint32_t j = low & mask;
                    if (j <= (uint32_t)i) {
                        const uint64_t hi_i = (uint32_t)i >= 8u ? row_hi : row_lo, hi_j = j >= 8u ? row_hi : row_lo;
                        const uint32_t si = 8u * ((uint32_t)i & 7u), sj = 8u * (j & 7u);
                        const uint64_t x = ((hi_i >> si) ^ (hi_j >> sj)) & 0xFFull;
                        const uint64_t xi = x << si, xj = x << sj;
                        row_lo ^= ((uint32_t)i < 8u ? xi : 0ull) ^ (j < 8u ? xj : 0ull);
                        row_hi ^= ((uint32_t)i >= 8u ? xi : 0ull) ^ (j >= 8u ? xj : 0ull);
                        --i;
                    }
                }
                store_epoch_row1<EXPANDED>(perm_epochs, (size_t)E, t,
                                           make_uint4((uint32_t)row_lo, (uint32_t)(row_lo >> 32), (uint32_t)row_hi, (uint32_t)(row_hi >> 32)));
            }
            ML_T(t_replay)
            // (5) Generator.random(float32): word used+t of the window
            {
                const uint32_t word = words[used + (uint32_t)t];
                const float f = (float)(word >> 8) * (1.0f / 16777216.0f);
                randfloat[t] = f;
                if (wave == 0) { // cells 0..15 are lanes 0..15 of wave 0
                    const unsigned long long m = __ballot((double)f < two_prob);
                    if (lane == 0) two_masks[E] = (uint32_t)(m & 0xFFFFull);
                }
            }
            // cursor: used + 1024 words of the window are gone
            {
                const uint32_t wc = used + ML2048_RAND_SIZE - (uint32_t)p0; // words taken from fresh outputs
                const uint32_t steps = (wc + 1) / 2;
                has32 = (int)(wc & 1u);
                // steps = 12 q + r: one table jump (2 multiplies) + r < 12 single steps
                const uint32_t q = steps / kStepsPerThread, r = steps - q * kStepsPerThread;
                u128 nxt = q < (uint32_t)kThreads ? ws[q].P * cur + ws[q].G * inc : jump_ahead(cur, inc, steps - r, pow2);
                for (uint32_t z = 0; z < r; ++z) nxt = nxt * A + inc;
                if (has32) buf32 = (uint32_t)(xsl_rr(nxt) >> 32);
                cur = nxt;
            }
            ++E;
            lds_barrier(); // the next regeneration overwrites `words`
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
    reinterpret_cast<uint4 *>(randperm)[t] =
        make_uint4((uint32_t)row_lo, (uint32_t)(row_lo >> 32), (uint32_t)row_hi, (uint32_t)(row_hi >> 32));
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
    hipLaunchKernelGGL((k_devrng_schedule<false>), dim3(1), dim3(kThreads), 0, (hipStream_t)stream, (const Jump *)workspace, rng,
                       randperm, randfloat, n_steps, two_prob, schedule, perm_epochs, two_masks, max_epochs, n_epochs,
                       (const int32_t *)nullptr);
    return (int)hipGetLastError();
}

// internal: the continuation launch of ml2048_devrng2_schedule (see k_devrng_schedule); `jump_ws` is a workspace
// prepared by ml2048_devrng_init
int ml2048_internal_devrng_continue(void *stream, const void *jump_ws, ml2048_devrng *rng, uint8_t *randperm,
                                    float *randfloat, int32_t n_steps, double two_prob, int32_t *schedule,
                                    uint8_t *perm_epochs, uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs,
                                    const int32_t *cont, int expanded) {
    if (expanded)
        hipLaunchKernelGGL((k_devrng_schedule<true>), dim3(1), dim3(kThreads), 0, (hipStream_t)stream, (const Jump *)jump_ws,
                           rng, randperm, randfloat, n_steps, two_prob, schedule, perm_epochs, two_masks, max_epochs,
                           n_epochs, cont);
    else
        hipLaunchKernelGGL((k_devrng_schedule<false>), dim3(1), dim3(kThreads), 0, (hipStream_t)stream, (const Jump *)jump_ws,
                           rng, randperm, randfloat, n_steps, two_prob, schedule, perm_epochs, two_masks, max_epochs,
                           n_epochs, cont);
    return (int)hipGetLastError();
}

} // extern "C"
