// devrng2.hip - the host PCG64 draw schedule of the bit-exact spawn mode on the GPU, data-parallel.
//
// Same contract as devrng.hip / hostrng.cpp (the numpy Generator(PCG64) draw order of VecGame.prepare /
// VecGame.step, src/ml2048/game_numba.py:589-591, 622-626, 670, including Generator.permuted), but built
// from data-parallel primitives instead of one workgroup following the stream:
//
//   gen     the next 2^18 LCG outputs after the generator's current state, by jump-ahead (the LCG is
//           position-addressable), as a flat array F of 2^19 32-bit words (low half first);
//   succ    for EVERY word position p: where a Fisher-Yates row that starts at p (index i = 15) ends
//           (L0[p], one thread per position, ~19 words each).  A row that starts with a pending buffered
//           word and continues at p ends where a row starting at p-1 would (index 15 accepts any word),
//           so the same table serves both cases;
//   levels  L1..L5[p] = 4, 16, 64, 256, 1024 rows ahead (each level = four look-ups in the previous one):
//           L5 is "the end of a whole table regeneration that starts at p";
//   walk    ONE wavefront follows the draw order: the steps between regenerations consume exactly four
//           words each, so 64 lanes test the next 64 steps' float64 draws at once; a regeneration is one
//           look-up in L5, sixteen float words and a cursor update;
//   rows    one thread per (regeneration, row): the row's start from the level tables (<= 15 look-ups),
//           then its accepted words -> 15 swap targets -> the row's shuffle as a permutation of 16 bytes;
//   apply   sixteen lanes per table row: a scan of permutation compositions over the regenerations
//           (v_perm_b32), every epoch's table out at once.
//
// Long schedules run as segments of <= 128 steps, each on a fresh window at the generator's then-current
// state.  Everything is checked bit for bit against the host schedule (tests/test_gpu_parity.py
// test_device_schedule_*).  The serial part is the ~130 dependent look-ups of `walk`.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/ml2048_hip.h"
#include "board_ops.hpp"

namespace {

typedef unsigned __int128 u128;

__host__ __device__ inline u128 lcg_mul2() { return ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull; }

__device__ __forceinline__ uint64_t xsl_rr2(u128 st) {
    const uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64 - rot) & 63));
}

constexpr uint32_t kSteps = 1u << 18;       // LCG outputs per window
constexpr uint32_t kWordsW = 2 * kSteps;    // flat 32-bit words
constexpr uint32_t kInv = 0xFFFFFFFFu;      // "runs out of the window"
constexpr int kSegSteps = 128;              // env steps per window
constexpr int kGenPerThread = 16;
constexpr int kLevels = 6;                  // 1, 4, 16, 64, 256, 1024 rows ahead
constexpr uint64_t kNine2 = 8106479329266893ull; // u >= 0.9  <=>  (out >> 11) >= 0.9 * 2^53 (exact)

struct Jump2 {
    u128 P, G; // state after k steps = P s + G c
};

// workspace: [Jump2 pow2[24]] [Jump2 jt[2^18/16]] [control: int32 x 16] [F: kWordsW + 64 words] [L: kLevels x kWordsW]
//            [regens: max 256 x 4 words] [sig: 256 x 1024 x 16 bytes]
constexpr int kMaxRegens = 256;
struct Layout {
    Jump2 *pow2;
    Jump2 *jt;   // jt[t] = jump by kGenPerThread * t steps, one entry per k2_gen thread
    int32_t *ctl; // [0] E (running epoch count), [1] error, [2] regens in this segment, [3] first epoch of the segment,
                  // [4] steps done so far, [5] "the window ran out": the remaining steps go to the continuation
    uint32_t *F;
    uint32_t *L[kLevels];
    uint32_t *regens; // per regeneration of the segment: vstart, pending flag, pending word, float position
    uint4 *sig;      // per (regeneration, row): Generator.permuted's shuffle of that row as a permutation
    void *v1ws;       // jump tables of the single-workgroup kernel (csrc/devrng.hip), the continuation
};
__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
__host__ __device__ inline Layout layout_of(void *ws) {
    uint8_t *p = (uint8_t *)ws;
    Layout l;
    l.pow2 = (Jump2 *)p;
    p += align256(sizeof(Jump2) * 24);
    l.jt = (Jump2 *)p;
    p += align256(sizeof(Jump2) * (kSteps / kGenPerThread));
    l.ctl = (int32_t *)p;
    p += 256;
    l.F = (uint32_t *)p;
    p += align256((size_t)(kWordsW + 64) * 4);
    for (int k = 0; k < kLevels; ++k) {
        l.L[k] = (uint32_t *)p;
        p += align256((size_t)kWordsW * 4);
    }
    l.regens = (uint32_t *)p;
    p += align256((size_t)kMaxRegens * 16);
    l.sig = (uint4 *)p;
    p += (size_t)kMaxRegens * ML2048_RAND_SIZE * 16;
    l.v1ws = p;
    return l;
}
inline size_t workspace_bytes() {
    return align256(sizeof(Jump2) * 24) + align256(sizeof(Jump2) * (kSteps / kGenPerThread)) + 256 + align256((size_t)(kWordsW + 64) * 4) + kLevels * align256((size_t)kWordsW * 4) +
           align256((size_t)kMaxRegens * 16) + (size_t)kMaxRegens * ML2048_RAND_SIZE * 16 +
           align256((size_t)ml2048_devrng_workspace_bytes());
}

__device__ __forceinline__ Jump2 jump_of(uint32_t k) {
    const u128 A = lcg_mul2();
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
    Jump2 j;
    j.P = accP;
    j.G = accG;
    return j;
}

__global__ __launch_bounds__(256) void k2_init(Jump2 *pow2, Jump2 *jt) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < 24) pow2[t] = jump_of(1u << t);
    if (t < kSteps / kGenPerThread) jt[t] = jump_of(t * kGenPerThread);
}

__device__ __forceinline__ u128 jump2(u128 s, u128 c, uint32_t k, const Jump2 *__restrict__ pow2) {
    for (int j = 0; j < 24; ++j)
        if ((k >> j) & 1u) s = pow2[j].P * s + pow2[j].G * c;
    return s;
}

// ---- gen: F[2k], F[2k+1] = halves of output k+1 after the generator's current state
__global__ __launch_bounds__(256) void k2_gen(Layout l, const ml2048_devrng *__restrict__ rng, int first_segment) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t k0 = gid * kGenPerThread;
    if (k0 >= kSteps || (!first_segment && l.ctl[5])) return;
    const u128 cur = ((u128)rng->state_hi << 64) | rng->state_lo;
    const u128 inc = ((u128)rng->inc_hi << 64) | rng->inc_lo;
    const u128 A = lcg_mul2();
    const Jump2 jmp = l.jt[gid];
    u128 st = jmp.P * cur + jmp.G * inc;
    uint2 *dst = reinterpret_cast<uint2 *>(l.F) + k0;
#pragma unroll 4
    for (int j = 0; j < kGenPerThread; ++j) {
        st = st * A + inc;
        const uint64_t out = xsl_rr2(st);
        dst[j] = make_uint2((uint32_t)out, (uint32_t)(out >> 32));
    }
    if (gid == 0)
        for (int j = 0; j < 64; ++j) l.F[kWordsW + j] = 0; // readable padding behind the window
}

__device__ __forceinline__ uint32_t accept_bits2(uint32_t low) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const uint32_t mask = i >= 8 ? 15u : i >= 4 ? 7u : i >= 2 ? 3u : 1u;
        m |= ((low & mask) <= (uint32_t)i ? 1u : 0u) << i;
    }
    return m;
}

// ---- succ: end of the row that starts at p
__global__ __launch_bounds__(256) void k2_succ(Layout l) {
    __shared__ uint16_t acc[16];
    if (threadIdx.x < 16) acc[threadIdx.x] = (uint16_t)accept_bits2(threadIdx.x);
    __syncthreads();
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    uint32_t pos = p, i = 15;
    while (i >= 1u && pos < kWordsW) {
        const uint32_t low = l.F[pos++] & 15u;
        i -= ((uint32_t)acc[low] >> i) & 1u;
    }
    l.L[0][p] = i == 0u ? pos : kInv;
}

// ---- levels: four look-ups in the previous level
__global__ __launch_bounds__(256) void k2_level(const uint32_t *__restrict__ prev, uint32_t *__restrict__ next) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    uint32_t v = p;
#pragma unroll
    for (int q = 0; q < 4; ++q) v = v < kWordsW ? prev[v] : kInv;
    next[p] = v;
}

// ---- walk: the draw order of up to 128 steps, one wavefront
__global__ __launch_bounds__(64) void k2_walk(Layout l, ml2048_devrng *rng, const float *__restrict__ randfloat_in,
                                              float *__restrict__ randfloat, int n_seg, int step0, double two_prob,
                                              int32_t *__restrict__ schedule, uint32_t *__restrict__ two_masks,
                                              int max_epochs, int first_segment, int32_t *__restrict__ n_epochs) {
    const int lane = threadIdx.x;
    const uint32_t *F = l.F;
    uint32_t n = 0; // LCG outputs consumed from the window
    int h = rng->has_uint32 != 0;
    uint32_t buf = rng->uinteger;
    int rs = rng->rand_step;
    if (!first_segment && l.ctl[5]) { // an earlier segment ran out of window: everything left is the continuation's
        if (lane == 0) {
            l.ctl[2] = 0;        // no regenerations of its own: k2_rows / k2_apply have nothing to do, and
            l.ctl[3] = l.ctl[0]; // what k2_apply rewrites as "the table at entry" is the current last epoch
        }
        return;
    }
    int E = first_segment ? 1 : l.ctl[0];
    int err = first_segment ? 0 : l.ctl[1];
    int overflow = 0;
    const int E_in = E;
    int n_reg = 0;
    uint32_t last_float = kInv;
    if (first_segment) { // epoch 0 = the tables at entry
        const float f = randfloat_in[lane & 15];
        const unsigned long long m = __ballot((double)f < two_prob);
        if (lane == 0) two_masks[0] = (uint32_t)(m & 0xFFFFull);
    }
    int t = 0;
    while (t < n_seg && !err) {
        if (2 * n + 4 * 64 + 8 >= kWordsW) { // (state is at a step boundary here)
            overflow = 1;
            break;
        }
        // the next (up to) 64 steps, assuming none of them regenerates: step t+s draws its float64 from
        // output n + 1 + 2 s (words 2n + 4s, +1) and its two offsets from the following output
        const int s = lane;
        const uint32_t w0 = 2 * n + 4 * (uint32_t)s;
        const uint64_t u = (uint64_t)F[w0] | ((uint64_t)F[w0 + 1] << 32);
        const bool valid = t + s < n_seg;
        const bool reg = valid && (((u >> 11) >= kNine2) || (rs + s >= ML2048_RAND_SIZE));
        const unsigned long long rmask = __ballot(reg);
        int first = rmask ? __builtin_ctzll(rmask) : 64;
        if (first > n_seg - t) first = n_seg - t;
        if (s < first) {
            uint32_t w1, w2;
            if (!h) {
                w1 = F[w0 + 2];
                w2 = F[w0 + 3];
            } else {
                w1 = s == 0 ? buf : F[w0 - 1]; // the high half left pending by the previous step
                w2 = F[w0 + 2];
            }
            int32_t *o = schedule + 3 * (size_t)(step0 + t + s);
            o[0] = E - 1;
            o[1] = rs + s + (int32_t)(w1 >> 22);
            o[2] = rs + s + (int32_t)(w2 >> 22);
        }
        if (first > 0) {
            n += 2 * (uint32_t)first;
            rs += first;
            t += first;
            buf = F[2 * n - 1]; // numpy keeps the last buffered high half in `uinteger` whether it is pending or not
        }
        if (t < n_seg && first < 64 && ((rmask >> first) & 1ull)) { // step t regenerates the tables
            if (E >= max_epochs) {
                err = 1;
                break;
            }
            const uint32_t q = 2 * (n + 1); // after the step's float64 draw
            const uint32_t vstart = h ? q - 1 : q; // see the header: a pending first word = a row from q-1
            const uint32_t end = l.L[kLevels - 1][vstart];
            if (end == kInv || end + ML2048_RAND_SIZE + 8 >= kWordsW || n_reg >= kMaxRegens) {
                overflow = 1; // nothing of step t has been committed: the continuation redoes it from its start
                break;
            }
            n += 1;
            rs = 0;
            if (lane == 0) {
                uint32_t *r = l.regens + 4 * n_reg;
                r[0] = vstart;
                r[1] = (uint32_t)h;
                r[2] = buf;
                r[3] = end;
            }
            { // Generator.random(float32): words end .. end+1023; cells 0..15 decide the two-mask (H1)
                const float f = (float)(F[end + (uint32_t)(lane & 15)] >> 8) * (1.0f / 16777216.0f);
                const unsigned long long m = __ballot((double)f < two_prob);
                if (lane == 0) two_masks[E] = (uint32_t)(m & 0xFFFFull);
            }
            last_float = end;
            const uint32_t pe2 = end + ML2048_RAND_SIZE;
            n = (pe2 + 1) >> 1;
            h = (int)(pe2 & 1u);
            buf = h ? F[pe2] : F[pe2 - 1];
            ++E;
            ++n_reg;
            // this step's two offsets
            uint32_t w1, w2;
            if (!h) {
                w1 = F[2 * n];
                w2 = F[2 * n + 1];
            } else {
                w1 = buf;
                w2 = F[2 * n];
            }
            buf = F[2 * n + 1];
            n += 1;
            if (lane == 0) {
                int32_t *o = schedule + 3 * (size_t)(step0 + t);
                o[0] = E - 1;
                o[1] = (int32_t)(w1 >> 22);
                o[2] = (int32_t)(w2 >> 22);
            }
            rs = 1;
            t += 1;
        }
    }
    // the float table of the last regeneration, in full (per-step API continues from it)
    if (!err && last_float != kInv)
        for (int k = lane; k < ML2048_RAND_SIZE; k += 64) randfloat[k] = (float)(F[last_float + (uint32_t)k] >> 8) * (1.0f / 16777216.0f);
    if (lane == 0) {
        const u128 cur = ((u128)rng->state_hi << 64) | rng->state_lo;
        const u128 inc = ((u128)rng->inc_hi << 64) | rng->inc_lo;
        const u128 nxt = jump2(cur, inc, n, l.pow2);
        rng->state_hi = (uint64_t)(nxt >> 64);
        rng->state_lo = (uint64_t)nxt;
        rng->has_uint32 = h;
        rng->uinteger = buf;
        rng->rand_step = rs;
        rng->error = err;
        l.ctl[0] = E;
        l.ctl[1] = err;
        l.ctl[2] = n_reg;
        l.ctl[3] = E_in;
        l.ctl[4] = step0 + t;
        l.ctl[5] = overflow;
        *n_epochs = E;
    }
}

// ---- rows: the 15 swap targets of row r of regeneration e, packed 4 bits each (bit 4i = target of index i)
__global__ __launch_bounds__(256) void k2_rows(Layout l) {
    __shared__ uint16_t acc[16];
    if (threadIdx.x < 16) acc[threadIdx.x] = (uint16_t)accept_bits2(threadIdx.x);
    __syncthreads();
    const int n_reg = l.ctl[2];
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t e = gid >> 10, r = gid & 1023u;
    if ((int)e >= n_reg || l.ctl[1]) return;
    const uint32_t *rec = l.regens + 4 * e;
    uint32_t p = rec[0];
    // start of row r = r rows ahead of the (virtual) start: base-4 digits of r through the level tables
#pragma unroll
    for (int lev = kLevels - 2; lev >= 0; --lev) {
        const uint32_t d = (r >> (2 * lev)) & 3u;
        for (uint32_t z = 0; z < d; ++z) p = l.L[lev][p];
    }
    uint64_t js = 0;
    uint32_t i = 15, pos = p;
    bool first = true;
    while (i >= 1u) {
        uint32_t w = l.F[pos++];
        if (first && r == 0 && rec[1]) w = rec[2]; // the pending buffered word stands in for word vstart
        first = false;
        const uint32_t low = w & 15u;
        if (((uint32_t)acc[low] >> i) & 1u) {
            const uint32_t mask = i >= 8 ? 15u : i >= 4 ? 7u : i >= 2 ? 3u : 1u;
            js |= (uint64_t)(low & mask) << (4 * i);
            --i;
        }
    }
    // Generator.permuted's swaps (i = 15..1: d[i] <-> d[j]) applied to the identity row: the row's permutation sigma,
    // so that "table row after this regeneration" = (row before) o sigma, a composition k2_apply can scan
    uint64_t row_lo = 0x0706050403020100ull, row_hi = 0x0F0E0D0C0B0A0908ull;
    // NOT unrolled on purpose: the fully unrolled form (constant shifts) of this 64-bit xor-swap chain gave wrong
    // rows in part of one wavefront, only while other kernels shared the GPU (the swap targets and the input rows
    // were verified correct, the non-unrolled loop never fails: tools/debug/fuzz_device2.py); cause not isolated
#ifdef ML2048_K2_ROWS_UNROLL // A/B build of the unrolled form (tools/debug/fuzz_device2.py)
#pragma unroll
#else
#pragma unroll 1
#endif
    for (uint32_t i2 = 15; i2 >= 1; --i2) {
        const uint32_t j = (uint32_t)(js >> (4 * i2)) & 15u;
        const uint64_t hi_i = i2 >= 8u ? row_hi : row_lo, hi_j = j >= 8u ? row_hi : row_lo;
        const uint32_t si = 8u * (i2 & 7u), sj = 8u * (j & 7u);
        const uint64_t x = ((hi_i >> si) ^ (hi_j >> sj)) & 0xFFull;
        const uint64_t xi = x << si, xj = x << sj;
        row_lo ^= (i2 < 8u ? xi : 0ull) ^ (j < 8u ? xj : 0ull);
        row_hi ^= (i2 >= 8u ? xi : 0ull) ^ (j >= 8u ? xj : 0ull);
    }
    l.sig[(size_t)e * ML2048_RAND_SIZE + r] =
        make_uint4((uint32_t)row_lo, (uint32_t)(row_lo >> 32), (uint32_t)row_hi, (uint32_t)(row_hi >> 32));
}

// ---- apply: every table row through all regenerations of the segment, epoch tables out.
// Sixteen lanes per row: lane c holds the permutation of regeneration 16*chunk + c, an inclusive scan of
// compositions over the sixteen lanes (four rounds) gives every epoch's row at once.
// EXPANDED: epoch tables as 48-byte expanded rows (what ml2048_rollout_chain consumes, board_ops.hpp
// expand_perm_row / spawn_table_dot)
template <bool EXPANDED>
__device__ __forceinline__ void store_epoch_row(uint8_t *perm_epochs, size_t epoch, uint32_t r, uint4 v) {
    uint4 *out = reinterpret_cast<uint4 *>(perm_epochs);
    if (EXPANDED) {
        const uint32_t perm[4] = {v.x, v.y, v.z, v.w};
        uint32_t x[ml2048::kExpandedRowWords];
        ml2048::expand_perm_row(perm, x);
        uint4 *row = out + 3 * (epoch * ML2048_RAND_SIZE + r);
        row[0] = make_uint4(x[0], x[1], x[2], x[3]);
        row[1] = make_uint4(x[4], x[5], x[6], x[7]);
        row[2] = make_uint4(x[8], x[9], x[10], x[11]);
    } else {
        out[epoch * ML2048_RAND_SIZE + r] = v;
    }
}

// out[k] = left[sel[k]] on 16-byte rows of values 0..15 (two v_perm_b32 + a blend per dword)
__device__ __forceinline__ uint4 compose_rows(uint4 left, uint4 sel) {
    const uint32_t s[4] = {sel.x, sel.y, sel.z, sel.w};
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t s7 = s[q] & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(left.y, left.x, s7); // bytes 0..7 of left
        const uint32_t hi = __builtin_amdgcn_perm(left.w, left.z, s7); // bytes 8..15
        const uint32_t m = ((s[q] >> 3) & 0x01010101u) * 0xFFu;
        o[q] = (hi & m) | (lo & ~m);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

constexpr int kApplyLanes = 16;
template <bool EXPANDED>
__global__ __launch_bounds__(256) void k2_apply(Layout l, uint8_t *__restrict__ randperm, uint8_t *__restrict__ perm_epochs) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t r = gid / kApplyLanes, c = gid % kApplyLanes;
    const int n_reg = l.ctl[2], E_in = l.ctl[3];
    uint4 row = reinterpret_cast<const uint4 *>(randperm)[r];
    if (c == 0) store_epoch_row<EXPANDED>(perm_epochs, (size_t)(E_in - 1), r, row); // the table at entry
    if (l.ctl[1]) return;
    const uint4 ident = make_uint4(0x03020100u, 0x07060504u, 0x0B0A0908u, 0x0F0E0D0Cu);
    for (int e0 = 0; e0 < n_reg; e0 += kApplyLanes) {
        const int e = e0 + (int)c;
        uint4 P = e < n_reg ? l.sig[(size_t)e * ML2048_RAND_SIZE + r] : ident;
#pragma unroll
        for (int d = 1; d < kApplyLanes; d *= 2) {
            uint4 left;
            left.x = __shfl_up(P.x, d, kApplyLanes);
            left.y = __shfl_up(P.y, d, kApplyLanes);
            left.z = __shfl_up(P.z, d, kApplyLanes);
            left.w = __shfl_up(P.w, d, kApplyLanes);
            if ((int)c >= d) P = compose_rows(left, P);
        }
        const uint4 mine = compose_rows(row, P);
        if (e < n_reg) store_epoch_row<EXPANDED>(perm_epochs, (size_t)(E_in + e), r, mine);
        row.x = __shfl(mine.x, kApplyLanes - 1, kApplyLanes); // lanes past n_reg hold the identity: lane 15 = last
        row.y = __shfl(mine.y, kApplyLanes - 1, kApplyLanes);
        row.z = __shfl(mine.z, kApplyLanes - 1, kApplyLanes);
        row.w = __shfl(mine.w, kApplyLanes - 1, kApplyLanes);
    }
    if (c == 0) reinterpret_cast<uint4 *>(randperm)[r] = row;
}

} // namespace

extern "C" {

int64_t ml2048_devrng2_workspace_bytes(void) { return (int64_t)workspace_bytes(); }

int ml2048_internal_devrng_continue(void *stream, const void *jump_ws, ml2048_devrng *rng, uint8_t *randperm,
                                    float *randfloat, int32_t n_steps, double two_prob, int32_t *schedule,
                                    uint8_t *perm_epochs, uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs,
                                    const int32_t *cont, int expanded);

int ml2048_devrng2_init(void *stream, void *workspace) {
    if (!workspace || (uintptr_t)workspace % 256) return ML2048_E_ALIGN;
    const Layout l = layout_of(workspace);
    hipLaunchKernelGGL(k2_init, dim3(kSteps / kGenPerThread / 256), dim3(256), 0, (hipStream_t)stream, l.pow2, l.jt);
    if (int rc = ml2048_devrng_init(stream, l.v1ws)) return rc;
    return (int)hipGetLastError();
}

// The launches of one schedule (gen, succ, 5 levels, walk, rows, apply per segment + the continuation).
static int enqueue_schedule(hipStream_t s, void *workspace, ml2048_devrng *rng, uint8_t *randperm, float *randfloat,
                            int32_t n_steps, double two_prob, int32_t *schedule, uint8_t *perm_epochs,
                            uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs, int expanded_tables) {
    const Layout l = layout_of(workspace);
    for (int step0 = 0; step0 < n_steps; step0 += kSegSteps) {
        const int n_seg = n_steps - step0 < kSegSteps ? n_steps - step0 : kSegSteps;
        hipLaunchKernelGGL(k2_gen, dim3(kSteps / kGenPerThread / 256), dim3(256), 0, s, l, rng, step0 == 0 ? 1 : 0);
        hipLaunchKernelGGL(k2_succ, dim3(kWordsW / 256), dim3(256), 0, s, l);
        for (int k = 1; k < kLevels; ++k)
            hipLaunchKernelGGL(k2_level, dim3(kWordsW / 256), dim3(256), 0, s, l.L[k - 1], l.L[k]);
        hipLaunchKernelGGL(k2_walk, dim3(1), dim3(64), 0, s, l, rng, randfloat, randfloat, n_seg, step0, two_prob, schedule,
                           two_masks, max_epochs, step0 == 0 ? 1 : 0, n_epochs);
        hipLaunchKernelGGL(k2_rows, dim3(kMaxRegens * ML2048_RAND_SIZE / 256), dim3(256), 0, s, l);
        if (expanded_tables)
            hipLaunchKernelGGL((k2_apply<true>), dim3(ML2048_RAND_SIZE * kApplyLanes / 256), dim3(256), 0, s, l, randperm, perm_epochs);
        else
            hipLaunchKernelGGL((k2_apply<false>), dim3(ML2048_RAND_SIZE * kApplyLanes / 256), dim3(256), 0, s, l, randperm, perm_epochs);
    }
    // If a window ran out (an unusual number of regenerations), the single-workgroup kernel finishes the remaining
    // steps from the state reached; otherwise it returns at once.
    return ml2048_internal_devrng_continue((void *)s, l.v1ws, rng, randperm, randfloat, n_steps, two_prob, schedule,
                                           perm_epochs, two_masks, max_epochs, n_epochs, l.ctl, expanded_tables);
}

// One schedule is 11+ small dependent launches whose arguments repeat from call to call (two output slots in
// game.py): the host cost of enqueueing them one by one (~170 us per 128-step schedule, measured) exceeded the
// rollout kernel they feed.  The chain is captured once per argument set into a hipGraph and replayed with a
// single hipGraphLaunch.  ML2048_DEVRNG_GRAPH=0 keeps the direct launches.
struct GraphKey {
    void *ws, *rng, *randperm, *randfloat, *schedule, *perm_epochs, *two_masks, *n_epochs;
    double two_prob;
    int32_t n_steps, max_epochs, expanded;
    bool operator==(const GraphKey &o) const { return memcmp(this, &o, sizeof(GraphKey)) == 0; }
};
struct GraphEntry {
    GraphKey key;
    hipGraphExec_t exec;
    uint64_t last_use;
    hipStream_t last_stream; // where the exec was launched last: drained before the exec is destroyed
};
// One env keeps 3 output slots x 2 table layouts alive per rollout length; with fewer entries than that times the
// number of live envs every call recaptured (and destroyed an exec that could still be executing).
#ifdef ML2048_GRAPH_CACHE_R1 // A/B only: round 1's cache (8 entries, evicted execs destroyed without draining)
constexpr size_t kGraphCache = 8;
#else
constexpr size_t kGraphCache = 96;
#endif
static std::mutex g_graph_mu;
static std::vector<GraphEntry> g_graphs;
static uint64_t g_graph_clock = 0;

static bool graphs_enabled() {
    static const bool on = [] {
        const char *e = getenv("ML2048_DEVRNG_GRAPH");
        return !(e && e[0] == '0');
    }();
    return on;
}

int ml2048_devrng2_schedule(void *stream, void *workspace, ml2048_devrng *rng, uint8_t *randperm, float *randfloat,
                            int32_t n_steps, double two_prob, int32_t *schedule, uint8_t *perm_epochs,
                            uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs, int expanded_tables) {
    if (!workspace || !rng || !randperm || !randfloat || !schedule || !perm_epochs || !two_masks || !n_epochs ||
        n_steps <= 0 || max_epochs <= 0)
        return ML2048_E_INVALID;
    if ((uintptr_t)workspace % 256 || ((uintptr_t)randperm | (uintptr_t)perm_epochs) % 16) return ML2048_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (s != nullptr && graphs_enabled()) { // the NULL stream cannot be captured
        GraphKey key;
        memset(&key, 0, sizeof(key));
        key.ws = workspace, key.rng = rng, key.randperm = randperm, key.randfloat = randfloat, key.schedule = schedule;
        key.perm_epochs = perm_epochs, key.two_masks = two_masks, key.n_epochs = n_epochs, key.two_prob = two_prob;
        key.n_steps = n_steps, key.max_epochs = max_epochs, key.expanded = expanded_tables ? 1 : 0;
        std::lock_guard<std::mutex> lock(g_graph_mu);
        for (GraphEntry &g : g_graphs)
            if (g.key == key) {
                g.last_use = ++g_graph_clock;
                g.last_stream = s;
                return (int)hipGraphLaunch(g.exec, s);
            }
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int rc = enqueue_schedule(s, workspace, rng, randperm, randfloat, n_steps, two_prob, schedule, perm_epochs,
                                            two_masks, max_epochs, n_epochs, expanded_tables);
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(s, &graph);
            hipGraphExec_t exec = nullptr;
            if (e == hipSuccess && rc == 0) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (rc) return rc;
            if (e == hipSuccess) {
                if (g_graphs.size() >= kGraphCache) {
                    size_t oldest = 0;
                    for (size_t i = 1; i < g_graphs.size(); ++i)
                        if (g_graphs[i].last_use < g_graphs[oldest].last_use) oldest = i;
                    // An exec owns the kernel-argument memory of its nodes: destroying one that is still executing
                    // lets a later instantiation reuse that memory under waves that have not read their arguments
                    // yet.  Its last stream is drained first (rare: only when the cache is full).
#ifndef ML2048_GRAPH_CACHE_R1
                    (void)hipStreamSynchronize(g_graphs[oldest].last_stream);
                    (void)hipGetLastError();
#endif
                    (void)hipGraphExecDestroy(g_graphs[oldest].exec);
                    g_graphs.erase(g_graphs.begin() + (long)oldest);
                }
                g_graphs.push_back(GraphEntry{key, exec, ++g_graph_clock, s});
                return (int)hipGraphLaunch(exec, s);
            }
            (void)hipGetLastError(); // capture or instantiation refused: fall through to the direct launches
        } else {
            (void)hipGetLastError();
        }
    }
    if (int rc = enqueue_schedule(s, workspace, rng, randperm, randfloat, n_steps, two_prob, schedule, perm_epochs,
                                  two_masks, max_epochs, n_epochs, expanded_tables))
        return rc;
    return (int)hipGetLastError();
}

} // extern "C"
