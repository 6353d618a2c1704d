// device_helpers.hpp - small device-side helpers shared by the kernel files of libml2048_hip.so
// (ml2048_hip.hip, step_prepare.hip): 16-byte board loads/stores, the action loader, the spawn-table row loaders,
// the rocRAND Philox words of the sampler / the counter-based spawn, wave reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdint.h>

#include "../../include/ml2048_hip.h"
#include "board_ops.hpp"

namespace ml2048 {
namespace dev {

__device__ __forceinline__ Board load_board(const uint8_t *base, int64_t i) {
    uint4 v = reinterpret_cast<const uint4 *>(base)[i];
    Board b;
    b.r[0] = v.x; b.r[1] = v.y; b.r[2] = v.z; b.r[3] = v.w;
    return b;
}

__device__ __forceinline__ void store16(uint8_t *base, int64_t i, const uint32_t w[4]) {
    reinterpret_cast<uint4 *>(base)[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t load_action(const void *actions, int dtype, int64_t i) {
    if (dtype == ML2048_ACT_U8) return reinterpret_cast<const uint8_t *>(actions)[i];
    if (dtype == ML2048_ACT_I32) return (uint32_t) reinterpret_cast<const int32_t *>(actions)[i];
    int64_t a = reinterpret_cast<const int64_t *>(actions)[i];
    return (a < 0 || a > 3) ? 0xFFFFFFFFu : (uint32_t)a;
}

// bit c <=> float64(randfloat[c]) < two_prob for the 16 cells (game_numba.py:207; H1, H2)
__device__ __forceinline__ uint32_t two_mask_from_table(const float *randfloat, double two_prob) {
    const int lane = threadIdx.x & 63;
    float f = randfloat[lane & 15];
    unsigned long long m = __ballot((double)f < two_prob);
    return (uint32_t)(m & 0xFFFFull);
}

__device__ __forceinline__ void load_perm_row(const uint8_t *randperm, int64_t row, uint32_t perm[4]) {
    uint4 v = reinterpret_cast<const uint4 *>(randperm)[row & (ML2048_RAND_SIZE - 1)];
    perm[0] = v.x; perm[1] = v.y; perm[2] = v.z; perm[3] = v.w;
}

// first four cells of a permutation row (the re-deal reads two: board_ops.hpp deal_table)
__device__ __forceinline__ uint32_t load_perm_head(const uint8_t *randperm, int64_t row) {
    return reinterpret_cast<const uint32_t *>(randperm)[(row & (ML2048_RAND_SIZE - 1)) * 4];
}

// INV tables hold EXPANDED rows (board_ops.hpp expand_perm_row / spawn_table_dot): 48 bytes = three uint4 units
// {permutation as nibbles, 0, 0}, {low-byte weights}, {high-byte weights}.  With INV, perm[0..1] receive the nibbles and
// wlo / whi the weights; without, perm[] is the plain 16-byte row.
constexpr int kExpandedRowUnits = 3;
template <bool INV>
__device__ __forceinline__ void load_table_row(const uint8_t *tab, int64_t row, uint32_t perm[4], uint32_t wlo[4],
                                               uint32_t whi[4]) {
    const uint4 *t = reinterpret_cast<const uint4 *>(tab);
    const int64_t r = row & (ML2048_RAND_SIZE - 1);
    if (INV) {
        const uint2 p = *reinterpret_cast<const uint2 *>(t + kExpandedRowUnits * r);
        const uint4 v = t[kExpandedRowUnits * r + 1], w = t[kExpandedRowUnits * r + 2];
        perm[0] = p.x; perm[1] = p.y;
        wlo[0] = v.x; wlo[1] = v.y; wlo[2] = v.z; wlo[3] = v.w;
        whi[0] = w.x; whi[1] = w.y; whi[2] = w.z; whi[3] = w.w;
    } else {
        const uint4 v = t[r];
        perm[0] = v.x; perm[1] = v.y; perm[2] = v.z; perm[3] = v.w;
    }
}
template <bool INV>
__device__ __forceinline__ uint32_t load_table_head(const uint8_t *tab, int64_t row) {
    return reinterpret_cast<const uint32_t *>(tab)[(row & (ML2048_RAND_SIZE - 1)) * (INV ? 4 * kExpandedRowUnits : 4)];
}

__device__ __forceinline__ void store_expanded_row(uint4 *dst, const uint32_t perm[4]) {
    uint32_t x[kExpandedRowWords];
    expand_perm_row(perm, x);
    dst[0] = make_uint4(x[0], x[1], x[2], x[3]);
    dst[1] = make_uint4(x[4], x[5], x[6], x[7]);
    dst[2] = make_uint4(x[8], x[9], x[10], x[11]);
}

// spawn_table_dot (board_ops.hpp) with the tile looked up instead of shifted into place: `tiles` = 32 boards of 16 bytes in
// LDS, entry c | bit << 4 = an empty board with the tile 2 - bit (a 2 where the epoch's mask bit of cell c is set, else a
// 4) in cell c.  One ds_read_b128 instead of the value / 64-bit shift / half-of-the-board selects (~10 VALU), in the
// passes whose other waves cover the look-up's round trip.
template <bool HAS_EMPTY>
__device__ __forceinline__ uint32_t spawn_table_dot_tiles(Board &b, uint32_t pn_lo, uint32_t pn_hi, const uint32_t wlo[4],
                                                          const uint32_t whi[4], uint32_t two_mask, const uint4 *tiles) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t z = z_bytes(b.r[k]);
        lo = udot4(z, wlo[k], lo);
        hi = udot4(z, whi[k], hi);
    }
    const uint32_t hits = ((hi << 8) | lo) >> 7;
    if (!HAS_EMPTY && hits == 0) return 0u; // a full board is left alone, as _spawn2 leaves it
    const uint32_t i = (uint32_t)__builtin_ctz(hits);
    const uint32_t c = ubfe((i & 8u) ? pn_hi : pn_lo, 4u * i, 4);
    const uint4 t = tiles[c | (ubfe(two_mask, c, 1) << 4)];
    b.r[0] |= t.x;
    b.r[1] |= t.y;
    b.r[2] |= t.z;
    b.r[3] |= t.w;
    return hits;
}

// One 16-byte unit of a schedule upload.  The source is a schedule-stream slot: [schedule | masks | epoch tables];
// the units from `tab16` on are permutation rows: each goes out as a 48-byte expanded row, the header in front of
// them is copied unchanged (tab16 >= the number of units: a plain copy).
__device__ __forceinline__ void upload_unit(uint4 *dst, const uint4 *src, int64_t j, int64_t tab16) {
    const uint4 v = src[j];
    if (j >= tab16) {
        const uint32_t perm[4] = {v.x, v.y, v.z, v.w};
        store_expanded_row(dst + tab16 + kExpandedRowUnits * (j - tab16), perm);
    } else {
        dst[j] = v;
    }
}

// rocRAND Philox4x32-10 device generator: words of (seed, subsequence = idx, offset = 4*counter)
__device__ __forceinline__ void rocrand_words(uint64_t seed, uint64_t counter, uint64_t idx, uint32_t w[4]) {
    rocrand_state_philox4x32_10 s;
    rocrand_init(seed, idx, counter * 4ull, &s);
    uint4 v = rocrand4(&s);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}

// W(j) of the action sampler's stream (board_ops.hpp sampler_word): rocRAND's sequence of subsequence idx at
// offset j (= word j&3 of Philox block j>>2)
// (initialised at the block boundary j & ~3 and selected with a wave-uniform compare: starting
// rocRAND at a non-multiple-of-4 offset makes its `result[substate]` a runtime-indexed private
// array, which hipcc promotes to 16 KiB of LDS per workgroup - 17 us instead of 2.4 us per launch)
__device__ __forceinline__ uint32_t rocrand_seq_word(uint64_t seed, uint64_t j, uint64_t idx) {
    rocrand_state_philox4x32_10 s;
    rocrand_init(seed, idx, j & ~3ull, &s);
    const uint4 v = rocrand4(&s);
    const uint32_t k = (uint32_t)(j & 3ull); // wave-uniform
    return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
}
// the two words of re-deal call c (board_ops.hpp spawn_word): words 2c, 2c+1 of the deal stream - one Philox block
__device__ __forceinline__ void rocrand_deal_words(uint64_t seed, uint64_t c, uint64_t idx, uint32_t &w0, uint32_t &w1) {
    rocrand_state_philox4x32_10 s;
    rocrand_init(seed ^ kDealSeedXor, idx, (2ull * c) & ~3ull, &s);
    const uint4 v = rocrand4(&s);
    const bool hi = ((2ull * c) & 2ull) != 0ull; // wave-uniform
    w0 = hi ? v.z : v.x;
    w1 = hi ? v.w : v.y;
}

// the draw of call-counter t, its bits at the bottom of the word: a masked draw takes W(t), a plain uniform
// draw two bits of W(t >> 4)
__device__ __forceinline__ uint32_t rocrand_sampler_word(uint64_t seed, uint64_t t, uint64_t idx, int masked) {
    return masked ? rocrand_seq_word(seed, t, idx) : rocrand_seq_word(seed, t >> 4, idx) >> (2u * (uint32_t)(t & 15ull));
}

// lane `lane` (wave-uniform) of the pair (lo, hi) replaced by the wave-uniform 64-bit `value`: two v_writelane_b32
// (one scalar operand per VALU instruction on gfx9: the lane select goes through m0)
__device__ __forceinline__ void write_lane64(uint32_t &lo, uint32_t &hi, unsigned long long value, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
        : "+v"(lo), "+v"(hi)
        : "s"((uint32_t)value), "s"((uint32_t)(value >> 32)), "s"(lane)
        : "m0");
#else
    (void)lo, (void)hi, (void)value, (void)lane;
#endif
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v; // lane 0 holds the sum
}


} // namespace dev
} // namespace ml2048
