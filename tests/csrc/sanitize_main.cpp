// sanitize_main.cpp - the oracle (C) and the device arithmetic (board_ops.hpp, host-compiled) run
// against each other under AddressSanitizer + UndefinedBehaviorSanitizer.  Built and executed by
// tests/test_sanitizers.py; exits non-zero on any mismatch, and the sanitizers abort on any
// out-of-bounds access, misaligned access, shift overflow or signed overflow in either of them.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

extern "C" {
#include "../../oracle/ml2048_oracle.h"
}
#include "host_ops_harness.cpp" // h_* wrappers over ml2048_amd/csrc/board_ops.hpp

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main() {
    // 1. every 18^4 line: push result and merge exponents
    {
        const int n = 18 * 18 * 18 * 18;
        std::vector<uint8_t> r1(n * 4), b1(n * 18), mv(n * 2), r2(n * 4), mexp(n * 2);
        orc_lines18_table(r1.data(), b1.data(), mv.data());
        h_lines18(r2.data(), mexp.data());
        if (memcmp(r1.data(), r2.data(), r1.size())) return puts("lines differ"), 1;
        for (int i = 0; i < n; ++i) {
            uint8_t b[18] = {0};
            for (int j = 0; j < 2; ++j)
                if (mexp[2 * i + j]) b[mexp[2 * i + j]]++;
            if (memcmp(b, &b1[i * 18], 18)) return printf("buckets differ at %d\n", i), 1;
        }
    }
    // 2. seeded rollouts, all reward kinds, tiles up to 2^17, invalid and out-of-range actions
    const int64_t M = 777;
    std::vector<uint8_t> perm(1024 * 16);
    std::vector<float> flt(1024);
    for (int kind = 0; kind < 4; ++kind) {
        std::vector<orc_record> a(M), b(M);
        std::vector<uint8_t> pa(M * 16), va(M * 4), pb(M * 16), vb(M * 4), act(M);
        orc_reset_records(a.data(), M, pa.data(), va.data());
        orc_reset_records(b.data(), M, pb.data(), vb.data());
        int64_t gca = 0, gcb = 0;
        for (int step = 0; step < 400; ++step) {
            if (step % 9 == 0) { // new tables: a random permutation per row, floats around two_prob
                for (int r = 0; r < 1024; ++r) {
                    uint8_t *p = &perm[r * 16];
                    for (int i = 0; i < 16; ++i) p[i] = (uint8_t)i;
                    for (int i = 15; i > 0; --i) {
                        int j = rnd() % (i + 1);
                        uint8_t t = p[i]; p[i] = p[j]; p[j] = t;
                    }
                    flt[r] = (float)(rnd() % 1000) / 1000.0f;
                }
            }
            const int64_t rowp = rnd() % 2048, rows = rnd() % 2048;
            const int64_t na = orc_prepare(a.data(), M, 5, rowp, perm.data(), flt.data(), 0.8, &gca, nullptr);
            const int64_t nb = h_prepare((Rec *)b.data(), M, 5, rowp, perm.data(), flt.data(), 0.8, &gcb);
            if (na != nb) return puts("prepare count differs"), 1;
            if (step == 50) // inject big tiles so that merges of 2^16 / 2^17 happen (H9)
                for (int64_t i = 0; i < M; i += 3) {
                    for (int c = 0; c < 16; ++c) a[i].board[c] = b[i].board[c] = (uint8_t)(14 + rnd() % 4) * (rnd() % 4 != 0);
                    orc_compute_valid_actions(a[i].board, a[i].valid_actions);
                    memcpy(b[i].valid_actions, a[i].valid_actions, 4);
                }
            for (int64_t i = 0; i < M; ++i) act[i] = (uint8_t)(rnd() % 5 == 4 ? 200 : rnd() % 4);
            orc_step(a.data(), M, 5, act.data(), ORC_ACT_U8, pa.data(), va.data(), kind, 0.8, rows, perm.data(), flt.data());
            h_step((Rec *)b.data(), M, 5, act.data(), pb.data(), vb.data(), kind, 0.8, rows, perm.data(), flt.data());
            if (memcmp(a.data(), b.data(), M * sizeof(orc_record)) || memcmp(pa.data(), pb.data(), pa.size()) ||
                memcmp(va.data(), vb.data(), va.size()))
                return printf("records differ: kind %d step %d\n", kind, step), 1;
            if (orc_digest(a.data(), M, 5, 1) != h_digest((Rec *)b.data(), M, 5, 1)) return puts("digest differs"), 1;
        }
        if (gca != gcb || gca <= M) return puts("game count"), 1;
    }
    puts("sanitize ok");
    return 0;
}
