// host_ops_harness.cpp - TEST HARNESS: compiles ml2048_amd/csrc/board_ops.hpp (the device
// arithmetic) with g++ so the bit tricks can be checked exhaustively on the CPU, without a GPU.
// Not part of the product; built and loaded only by tests/test_host_ops.py.
#include <stdint.h>
#include <string.h>

#include "../../ml2048_amd/csrc/board_ops.hpp"

using namespace ml2048;

extern "C" {

// every 18^4 line (itertools.product order) through push_line; mexp[n][2] = merge exponents
void h_lines18(uint8_t *result, uint8_t *mexp) {
    int64_t i = 0;
    for (uint32_t a = 0; a < 18; ++a)
        for (uint32_t b = 0; b < 18; ++b)
            for (uint32_t c = 0; c < 18; ++c)
                for (uint32_t d = 0; d < 18; ++d, ++i) {
                    uint32_t x = a | (b << 8) | (c << 16) | (d << 24), m;
                    uint32_t r = push_line(x, m);
                    memcpy(result + 4 * i, &r, 4);
                    mexp[2 * i] = (uint8_t)(m & 0xFF);
                    mexp[2 * i + 1] = (uint8_t)(m >> 8);
                }
}

void h_move(const uint8_t *boards, int64_t n, const uint8_t *actions, uint8_t *out, uint8_t *merged,
            uint32_t *stats /* n x 3 */) {
    for (int64_t i = 0; i < n; ++i) {
        Board b;
        memcpy(b.r, boards + 16 * i, 16);
        MoveResult mv = move_board(b, actions[i]);
        MergeStats st = merge_stats(mv.mexp_first, mv.mexp_second);
        memcpy(out + 16 * i, mv.board.r, 16);
        memcpy(merged + 16 * i, st.merged, 16);
        stats[3 * i] = st.sum_value;
        stats[3 * i + 1] = st.sum_rank;
        stats[3 * i + 2] = st.count;
    }
}

void h_valid(const uint8_t *boards, int64_t n, uint8_t *valid, uint16_t *empties, uint8_t *maxcell) {
    for (int64_t i = 0; i < n; ++i) {
        Board b;
        memcpy(b.r, boards + 16 * i, 16);
        uint32_t v = valid_actions(b);
        memcpy(valid + 4 * i, &v, 4);
        empties[i] = (uint16_t)empty_mask(b);
        maxcell[i] = (uint8_t)max_cell(b);
    }
}

// the lazy-mask forms against the eager ones: no_move_left == (valid_actions == 0); a move changes the
// board exactly when its valid bit is set.  Returns the number of disagreements.
int64_t h_lazy_mask_check(const uint8_t *boards, int64_t n) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        Board b;
        memcpy(b.r, boards + 16 * i, 16);
        const uint32_t v = valid_actions(b);
        const bool has_tile = (b.r[0] | b.r[1] | b.r[2] | b.r[3]) != 0;
        if (has_tile && no_move_left(b) != (v == 0 ? 1u : 0u)) ++bad;
        for (uint32_t a = 0; a < 4; ++a) {
            const MoveResult mv = move_board(b, a);
            const bool valid_bit = ((v >> (8 * a)) & 0xFFu) != 0;
            if (board_differs(mv.board, b) != valid_bit) ++bad;
        }
        if (!valid_consistent(b, v)) ++bad;
    }
    return bad;
}

// the four-lines-at-once move against four separate line pushes; returns the number of differences
int64_t h_move_forms_check(const uint8_t *boards, int64_t n) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        Board b;
        memcpy(b.r, boards + 16 * i, 16);
        for (uint32_t a = 0; a < 4; ++a) {
            const MoveResult x = move_board(b, a), y = move_board_by_lines(b, a);
            if (memcmp(x.board.r, y.board.r, 16) != 0 || x.mexp_first != y.mexp_first || x.mexp_second != y.mexp_second)
                ++bad;
        }
    }
    return bad;
}

// the move by table look-up (move_board_lut over the 65 536 entries of row_lut_entry) against move_board, on boards
// inside the table's domain (exponents <= 15, no merge of two 2^15 tiles); boards outside it are skipped and counted
// in *skipped.  Also checks lut_guard: a board it accepts must be in the domain for every direction.
int64_t h_move_lut_check(const uint8_t *boards, int64_t n, int64_t *skipped) {
    static uint16_t lut[kRowLutEntries], lut_swz[kRowLutEntries]; // plain and bank-swizzled layout (row_lut_slot)
    static uint32_t sel[32];                                       // move_selectors(a) at sel[8 a ..] (the kernel's LDS table)
    static bool filled = false;
    if (!filled) {
        for (uint32_t i = 0; i < (uint32_t)kRowLutEntries; ++i) lut[i] = lut_swz[row_lut_slot(i)] = row_lut_entry(i);
        for (uint32_t a = 0; a < 4; ++a) move_selectors(a, sel + 8 * a);
        filled = true;
    }
    int64_t bad = 0, skip = 0;
    for (int64_t i = 0; i < n; ++i) {
        Board b;
        memcpy(b.r, boards + 16 * i, 16);
        const bool guarded = lut_guard(b);
        for (uint32_t a = 0; a < 4; ++a) {
            const MoveResult x = move_board(b, a);
            const bool in_domain = ((b.r[0] | b.r[1] | b.r[2] | b.r[3]) & 0xF0F0F0F0u) == 0u &&
                                   ((x.board.r[0] | x.board.r[1] | x.board.r[2] | x.board.r[3]) & 0xF0F0F0F0u) == 0u;
            if (!in_domain) {
                if (guarded) ++bad;
                ++skip;
                continue;
            }
            const Board y = move_board_lut<false>(b, a, lut), z = move_board_lut<true>(b, a, lut_swz);
            if (memcmp(x.board.r, y.r, 16) != 0 || memcmp(x.board.r, z.r, 16) != 0) ++bad;
            // the same with the orientation folded into looked-up byte selectors
            const Board ys = move_board_lut_sel<false>(b, a, lut, sel), zs = move_board_lut_sel<true>(b, a, lut_swz, sel);
            if (memcmp(x.board.r, ys.r, 16) != 0 || memcmp(x.board.r, zs.r, 16) != 0) ++bad;
        }
    }
    *skipped = skip;
    return bad;
}

// the counter-based re-deal as a table index (deal_words_index + deal_table_nibbles' table) == spawn_word twice on an
// empty board, on n word pairs; returns the number of differences
int64_t h_deal_words_check(const uint32_t *words, int64_t n, uint32_t thr16) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t w0 = words[2 * i], w1 = words[2 * i + 1];
        Board a;
        a.r[0] = a.r[1] = a.r[2] = a.r[3] = 0;
        spawn_word(a, w0, thr16);
        spawn_word(a, w1, thr16);
        const uint32_t idx = deal_words_index(w0, w1, thr16);
        const uint32_t c1 = idx & 15u, c2 = (idx >> 4) & 15u;
        const uint32_t m = (((idx >> 8) & 1u) << c1) | (((idx >> 9) & 1u) << c2);
        Board b;
        deal_table_nibbles(b, idx & 0xFFu, m); // what the kernel's LDS table holds at `idx`
        if (c1 == c2 || idx >= 1024u || memcmp(a.r, b.r, 16) != 0) ++bad;
    }
    return bad;
}

// stable compaction by position masks (compact_cells) == the six-step bubble it replaced, on n x 4 cell words
int64_t h_compact_check(const uint32_t *words, int64_t n) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t a = words[4 * i], b = words[4 * i + 1], c = words[4 * i + 2], d = words[4 * i + 3];
        uint32_t p = a, q = b, r = c, t = d;
        compact_cells(a, b, c, d);
        pull_into_empty(p, q);
        pull_into_empty(q, r);
        pull_into_empty(r, t);
        pull_into_empty(p, q);
        pull_into_empty(q, r);
        pull_into_empty(p, q);
        if (a != p || b != q || c != r || d != t) ++bad;
    }
    return bad;
}

// spawn from the inverse permutation / by dot products == spawn probing the permutation in order; returns differences
int64_t h_spawn_forms_check(const uint8_t *boards, const uint8_t *perms, const uint32_t *two_masks, int64_t n) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        Board a, b;
        memcpy(a.r, boards + 16 * i, 16);
        b = a;
        uint32_t perm[4], inv[4];
        memcpy(perm, perms + 16 * i, 16);
        invert_perm_row(perm, inv);
        for (int c = 0; c < 16; ++c) { // inv really is the inverse
            const uint32_t pos = (inv[c >> 2] >> (8 * (c & 3))) & 0xFF;
            if (((perm[pos >> 2] >> (8 * (pos & 3))) & 0xFF) != (uint32_t)c) ++bad;
        }
        bool had_empty = false;
        for (int c = 0; c < 16; ++c) had_empty |= ((b.r[c >> 2] >> (8 * (c & 3))) & 0xFF) == 0;
        // ... and so is the expanded row of the upload kernels: nibble i = perm[i]; the weight of cell c = 2^inv[c],
        // in the low vector when inv[c] < 8, in the high one (as 2^(inv[c]-8)) otherwise
        uint32_t x[kExpandedRowWords];
        expand_perm_row(perm, x);
        if (x[2] != 0 || x[3] != 0) ++bad;
        for (int k = 0; k < 16; ++k)
            if (((x[k >> 3] >> (4 * (k & 7))) & 0xFu) != ((perm[k >> 2] >> (8 * (k & 3))) & 0xFFu)) ++bad;
        for (int c = 0; c < 16; ++c) {
            const uint32_t pos = (inv[c >> 2] >> (8 * (c & 3))) & 0xFF;
            const uint32_t wl = (x[4 + (c >> 2)] >> (8 * (c & 3))) & 0xFF, wh = (x[8 + (c >> 2)] >> (8 * (c & 3))) & 0xFF;
            if (wl != (pos < 8 ? 1u << pos : 0u) || wh != (pos >= 8 ? 1u << (pos - 8) : 0u)) ++bad;
        }
        Board d = a, e = a, g = a;
        spawn_table(a, perm, two_masks[i], 1);
        const uint32_t empties = spawn_table_inv(b, perm, inv, two_masks[i]);
        if (memcmp(a.r, b.r, 16) != 0) ++bad;
        const uint32_t empties_dot = spawn_table_dot(d, x[0], x[1], x + 4, x + 8, two_masks[i]);
        if (memcmp(a.r, d.r, 16) != 0 || empties_dot != empties) ++bad;
        // the re-deal from the row's head, byte form and nibble form
        deal_table(e, perm, two_masks[i]);
        Board f;
        deal_table_nibbles(f, x[0], two_masks[i]);
        if (memcmp(e.r, f.r, 16) != 0) ++bad;
        if (had_empty) { // the form for callers that know of an empty cell (the lazy loop, after a move that changed the board)
            const uint32_t e2 = spawn_table_dot<true>(g, x[0], x[1], x + 4, x + 8, two_masks[i]);
            if (memcmp(a.r, g.r, 16) != 0 || e2 != empties) ++bad;
        }
        if (had_empty) { // "terminated" from the spawn's emptiness mask == the full test on the board after the spawn
            StepRegs s1 = {}, s2 = {};
            s1.board = b, s2.board = b;
            step_post_lazy(s1);
            step_post_lazy(s2, empties);
            if (s1.terminated != s2.terminated) ++bad;
        }
    }
    return bad;
}

// merged_value_sum on n word pairs: the SDWA-shift form of "sum of 2^(num+1) over the bytes 1 <= num <= 15"
void h_merged_value_sum(const uint32_t *words, int64_t n, uint32_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = ml2048::merged_value_sum(words[2 * i], words[2 * i + 1]);
}

// score_potential / phi_guard on n boards (16 bytes each)
void h_score_potential(const uint8_t *boards, const int32_t *step, const float *score, int64_t n, int32_t n_steps,
                       int32_t *pot, uint8_t *guard) {
    for (int64_t i = 0; i < n; ++i) {
        ml2048::Board b;
        memcpy(b.r, boards + 16 * i, 16);
        pot[i] = ml2048::score_potential(b, step[i]);
        guard[i] = ml2048::phi_guard(b, score[i], n_steps) ? 1 : 0;
    }
}

void h_transpose(const uint8_t *in, uint8_t *out) {
    Board b;
    memcpy(b.r, in, 16);
    Board t = transpose(b);
    memcpy(out, t.r, 16);
}

static uint32_t two_mask_of(const float *randfloat, double two_prob) {
    uint32_t m = 0;
    for (int c = 0; c < 16; ++c) m |= ((double)randfloat[c] < two_prob ? 1u : 0u) << c;
    return m;
}

// records: reference 64-byte layout (id, step, score, reward, board, merged, valid, term, invalid)
struct Rec {
    int32_t id, step;
    float score, reward;
    uint8_t board[16], merged[16], valid[4], terminated, invalid, pad[10];
};

static void load(const Rec &e, StepRegs &s) {
    memcpy(s.board.r, e.board, 16);
    memcpy(&s.valid, e.valid, 4);
    memcpy(s.merged, e.merged, 16);
    s.step = e.step;
    s.score = e.score;
    s.reward = e.reward;
    s.terminated = e.terminated;
}

static void store(Rec &e, const StepRegs &s) {
    memcpy(e.board, s.board.r, 16);
    memcpy(e.valid, &s.valid, 4);
    memcpy(e.merged, s.merged, 16);
    e.step = s.step;
    e.score = s.score;
    e.reward = s.reward;
    e.terminated = (uint8_t)s.terminated;
}

void h_step(Rec *data, int64_t M, int64_t idx_base, const uint8_t *actions, uint8_t *prev_state,
            uint8_t *prev_valid, int reward_kind, double two_prob, int64_t rand_seed,
            const uint8_t *randperm, const float *randfloat) {
    uint32_t two_mask = two_mask_of(randfloat, two_prob);
    for (int64_t i = 0; i < M; ++i) {
        Rec &e = data[i];
        memcpy(prev_state + 16 * i, e.board, 16);
        memcpy(prev_valid + 4 * i, e.valid, 4);
        StepRegs s;
        load(e, s);
        uint32_t mf, ms;
        bool ok = (i & 1) ? step_pre(s, actions[i], reward_kind)
                          : step_pre_lite(s, actions[i], reward_kind, mf, ms); // both forms must agree
        if (!ok) {
            e.invalid = 1;
            continue;
        }
        if (!(i & 1)) merge_buckets(mf, ms, s.merged);
        uint32_t perm[4];
        memcpy(perm, randperm + 16 * ((rand_seed + idx_base + i) & 1023), 16);
        spawn_table(s.board, perm, two_mask, 1);
        step_post(s);
        store(e, s);
        e.invalid = 0;
    }
}

int64_t h_prepare(Rec *data, int64_t M, int64_t idx_base, int64_t row_base, const uint8_t *randperm,
                  const float *randfloat, double two_prob, int64_t *game_count) {
    uint32_t two_mask = two_mask_of(randfloat, two_prob);
    int64_t n = 0;
    for (int64_t i = 0; i < M; ++i) {
        Rec &e = data[i];
        if (!e.terminated) continue;
        StepRegs s;
        clear_regs(s);
        uint32_t perm[4];
        memcpy(perm, randperm + 16 * ((row_base + idx_base + i) & 1023), 16);
        if (i & 1)
            spawn_table(s.board, perm, two_mask, 2);
        else
            deal_table(s.board, perm, two_mask); // specialised form for an empty board
        s.valid = valid_actions(s.board);
        memset(&e, 0, sizeof(e));
        store(e, s);
        e.id = (int32_t)(*game_count)++;
        ++n;
    }
    return n;
}

static uint32_t thr_of(double two_prob) { // ceil(two_prob * 2^16), clamped (board_ops.hpp spawn_word)
    double t = two_prob * 65536.0;
    double c = (double)(uint64_t)t;
    if (c < t) c += 1.0;
    if (c <= 0.0) return 0;
    if (c >= 65536.0) return 65536u;
    return (uint32_t)c;
}

void h_step_fast(Rec *data, int64_t M, int64_t idx_base, const uint8_t *actions, uint8_t *prev_state,
                 uint8_t *prev_valid, int reward_kind, double two_prob, uint64_t seed, uint64_t counter) {
    const uint32_t thr = thr_of(two_prob);
    for (int64_t i = 0; i < M; ++i) {
        Rec &e = data[i];
        memcpy(prev_state + 16 * i, e.board, 16);
        memcpy(prev_valid + 4 * i, e.valid, 4);
        StepRegs s;
        load(e, s);
        if (!step_pre(s, actions[i], reward_kind)) {
            e.invalid = 1;
            continue;
        }
        spawn_word(s.board, philox_seq_word(seed, counter, (uint64_t)(idx_base + i)), thr);
        step_post(s);
        store(e, s);
        e.invalid = 0;
    }
}

int64_t h_prepare_fast(Rec *data, int64_t M, int64_t idx_base, double two_prob, uint64_t seed,
                       uint64_t counter, int64_t *game_count) {
    const uint32_t thr = thr_of(two_prob);
    int64_t n = 0;
    for (int64_t i = 0; i < M; ++i) {
        Rec &e = data[i];
        if (!e.terminated) continue;
        StepRegs s;
        clear_regs(s);
        spawn_word(s.board, philox_seq_word(seed ^ kDealSeedXor, 2 * counter, (uint64_t)(idx_base + i)), thr);
        spawn_word(s.board, philox_seq_word(seed ^ kDealSeedXor, 2 * counter + 1, (uint64_t)(idx_base + i)), thr);
        s.valid = valid_actions(s.board);
        memset(&e, 0, sizeof(e));
        store(e, s);
        e.id = (int32_t)(*game_count)++;
        ++n;
    }
    return n;
}

void h_sample(const uint8_t *valid, int64_t M, int64_t idx_base, uint64_t seed, uint64_t counter,
              int masked, uint8_t *actions, float *logp) {
    for (int64_t i = 0; i < M; ++i) {
        uint32_t v;
        memcpy(&v, valid + 4 * i, 4);
        uint32_t cnt;
        actions[i] = (uint8_t)pick_action(v, sampler_word(seed, counter, (uint64_t)(idx_base + i), masked), masked, cnt);
        logp[i] = neg_log_count(cnt);
    }
}

uint64_t h_digest(const Rec *data, int64_t M, int64_t idx_base, int with_id) {
    uint64_t total = 0;
    for (int64_t i = 0; i < M; ++i) {
        const Rec &e = data[i];
        Board b;
        uint32_t merged[4], valid, sb, rb;
        memcpy(b.r, e.board, 16);
        memcpy(merged, e.merged, 16);
        memcpy(&valid, e.valid, 4);
        memcpy(&sb, &e.score, 4);
        memcpy(&rb, &e.reward, 4);
        total += board_digest((uint64_t)(idx_base + i), b, merged, valid, e.terminated, e.invalid, e.step, e.id, sb, rb, with_id);
    }
    return total;
}

void h_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    U4 c = {ctr[0], ctr[1], ctr[2], ctr[3]};
    U4 r = philox4x32_10(c, key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
}
