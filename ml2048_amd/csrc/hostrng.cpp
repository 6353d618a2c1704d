// hostrng.cpp - host side of the bit-exact ("table") spawn mode, native.
//
// The reference draws its spawn randomness from ONE numpy Generator(PCG64) on the host
// (src/ml2048/game_numba.py:589-591, 606-611, 622-626, 670): per prepare() a float64, on ~10 % of
// prepares a regeneration of the 1024x16 permutation table (Generator.permuted, in place) and the
// 1024 float32 table (Generator.random), then one integers(0, 1024); per step() one
// integers(0, 1024).  That draw order never looks at game state, so a fused N-step rollout needs
// it pre-drawn - and 3 numpy calls per env step (~2 ms per 128 steps) would then dominate the
// 0.25 ms kernel.  This file restates exactly those numpy routines for PCG64 so the schedule for
// N steps is one C call:
//   PCG64 = PCG XSL-RR 128/64 (O'Neill), numpy's default multiplier; next_uint32 hands out the
//   low then the high half of one 64-bit draw (numpy/random/src/pcg64/pcg64.h);
//   random()            = (next64 >> 11) * 2^-53
//   random(float32)     = (next32 >> 8) * 2^-24
//   integers(0, 1024)   = Lemire bounded uint32 = (next32 * 1024) >> 32  (threshold 0 for 2^k)
//   permuted(axis=1)    = per row Fisher-Yates, i = n-1..1, j = random_interval(i) (masked
//                         rejection on next32)
// tests/test_hostrng_native.py checks it word for word against numpy itself over long runs, so a
// numpy upgrade that changed any of these algorithms would be caught, not silently absorbed.
#include <stdint.h>
#include <string.h>

#include "../../include/ml2048_hip.h"

namespace {

typedef unsigned __int128 u128;

struct Pcg {
    u128 state, inc;
    int has32;
    uint32_t buf32;
};

inline uint64_t next64(Pcg &g) {
    const u128 mult = ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
    g.state = g.state * mult + g.inc;
    const uint64_t hi = (uint64_t)(g.state >> 64), lo = (uint64_t)g.state;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64 - rot) & 63));
}

inline uint32_t next32(Pcg &g) {
    if (g.has32) {
        g.has32 = 0;
        return g.buf32;
    }
    const uint64_t v = next64(g);
    g.has32 = 1;
    g.buf32 = (uint32_t)(v >> 32);
    return (uint32_t)v;
}

inline double next_double(Pcg &g) { return (double)(next64(g) >> 11) * (1.0 / 9007199254740992.0); }
inline float next_float(Pcg &g) { return (float)(next32(g) >> 8) * (1.0f / 16777216.0f); }
inline uint32_t bounded_1024(Pcg &g) { return (uint32_t)(((uint64_t)next32(g) * 1024u) >> 32); }

// Buffered view of the generator's 32-bit output stream.  Producing the raw words in a tight loop
// (bounded by the 128-bit multiply latency only) and consuming them from an array decouples the
// LCG's dependency chain from the rejection-sampling chain; the exact generator state after an
// arbitrary number of consumed words is recovered by re-stepping from the chunk's start state.
struct Stream {
    static const int kWords = 512;
    Pcg &g;
    Pcg chunk_start;   // generator state when the current chunk was produced
    uint32_t buf[kWords + 2];
    int n, pos;

    explicit Stream(Pcg &gen) : g(gen), n(0), pos(0) { refill(); }

    void refill() {
        chunk_start = g;
        n = 0;
        pos = 0;
        Pcg t = g;
        if (t.has32) {
            buf[n++] = t.buf32;
            t.has32 = 0;
        }
        while (n < kWords) {
            const uint64_t v = next64(t);
            buf[n++] = (uint32_t)v;
            buf[n++] = (uint32_t)(v >> 32);
        }
        // t is now the state after the whole chunk has been consumed (no pending half)
        g = t;
    }

    inline uint32_t next() {
        if (pos == n) refill();
        return buf[pos++];
    }

    // put the generator where it would be had exactly the consumed words been drawn one by one
    void finish() {
        Pcg t = chunk_start;
        for (int k = 0; k < pos; ++k) (void)next32(t);
        g = t;
    }
};

void regen_tables(Pcg &g, uint8_t *randperm, float *randfloat) {
    // Generator.permuted(axis=1, out=self): per row, i = 15..1, j = random_interval(i) (masked
    // rejection on 32-bit words: j = w & mask(i) accepted iff j <= i), swap(d[i], d[j]).
    // Two passes so that the only loop-carried dependency of the serial scan is `i`:
    //   1. scan the word stream; whether state i accepts a word depends on its low 4 bits only,
    //      so a 16-entry bitmap table (bit i = accepted in state i) turns the rejection test into
    //      shift/and/sub.  The candidate is stored unconditionally: a rejected one is overwritten
    //      by the next word for the same i.
    //   2. apply the 15 swaps per row (rows are independent: the CPU overlaps them).
    static const uint8_t kMask[16] = {0, 1, 3, 3, 7, 7, 7, 7, 15, 15, 15, 15, 15, 15, 15, 15};
    uint16_t acc_bits[16];
    for (uint32_t low = 0; low < 16; ++low) {
        uint16_t m = 0;
        for (int i = 1; i < 16; ++i) m |= (uint16_t)(((low & kMask[i]) <= (uint32_t)i) ? 1u : 0u) << i;
        acc_bits[low] = m;
    }
    uint8_t local_js[ML2048_RAND_SIZE * 16];
    Stream st(g);
    for (int row = 0; row < ML2048_RAND_SIZE; ++row) {
        uint8_t *jrow = local_js + row * 16;
        int i = 15;
        while (i >= 1) {
            const uint32_t low = st.next() & 15u;
            jrow[i] = (uint8_t)low;
            i -= (acc_bits[low] >> i) & 1;
        }
    }
    for (int row = 0; row < ML2048_RAND_SIZE; ++row) {
        uint8_t *d = randperm + row * 16;
        const uint8_t *jrow = local_js + row * 16;
#pragma GCC unroll 15
        for (int i = 15; i >= 1; --i) {
            const uint32_t j = jrow[i] & kMask[i];
            const uint8_t a = d[i], b = d[j];
            d[j] = a;
            d[i] = b;
        }
    }
    for (int i = 0; i < ML2048_RAND_SIZE; ++i) // Generator.random(dtype=float32)
        randfloat[i] = (float)(st.next() >> 8) * (1.0f / 16777216.0f);
    st.finish();
}

uint32_t two_mask_of(const float *randfloat, double two_prob) {
    uint32_t m = 0;
    for (int c = 0; c < 16; ++c) m |= ((double)randfloat[c] < two_prob ? 1u : 0u) << c;
    return m;
}

} // namespace

extern "C" {

int ml2048_hostrng_schedule(ml2048_pcg64 *rng, uint8_t *randperm, float *randfloat, int32_t *rand_step,
                            int32_t n_steps, double two_prob, int32_t *schedule, uint8_t *perm_epochs,
                            uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs) {
    if (!rng || !randperm || !randfloat || !rand_step || !schedule || !perm_epochs || !two_masks || !n_epochs ||
        n_steps <= 0 || max_epochs <= 0)
        return ML2048_E_INVALID;
    Pcg g;
    g.state = ((u128)rng->state_hi << 64) | rng->state_lo;
    g.inc = ((u128)rng->inc_hi << 64) | rng->inc_lo;
    g.has32 = rng->has_uint32 != 0;
    g.buf32 = rng->uinteger;
    int32_t rs = *rand_step;
    int32_t ne = 1;
    memcpy(perm_epochs, randperm, ML2048_RAND_SIZE * 16);
    two_masks[0] = two_mask_of(randfloat, two_prob);
    for (int32_t t = 0; t < n_steps; ++t) {
        // VecGame.prepare, game_numba.py:622-626
        const double u = next_double(g);
        if (u >= 0.9 || rs >= ML2048_RAND_SIZE) {
            rs = 0;
            regen_tables(g, randperm, randfloat);
            if (ne >= max_epochs) return ML2048_E_UNSUPPORTED; // caller sized the epoch buffer too small
            memcpy(perm_epochs + (size_t)ne * ML2048_RAND_SIZE * 16, randperm, ML2048_RAND_SIZE * 16);
            two_masks[ne] = two_mask_of(randfloat, two_prob);
            ++ne;
        }
        const int32_t off_p = (int32_t)bounded_1024(g);
        // VecGame.step, :670, :681, :685
        const int32_t off_s = (int32_t)bounded_1024(g);
        schedule[3 * t + 0] = ne - 1;
        schedule[3 * t + 1] = rs + off_p;
        schedule[3 * t + 2] = rs + off_s;
        rs += 1;
    }
    *rand_step = rs;
    *n_epochs = ne;
    rng->state_hi = (uint64_t)(g.state >> 64);
    rng->state_lo = (uint64_t)g.state;
    rng->has_uint32 = g.has32;
    rng->uinteger = g.buf32;
    return 0;
}

} // extern "C"
