/*
 * ml2048_oracle.c - CPU restatement of the reference hot path.  See ml2048_oracle.h.
 * TEST INFRASTRUCTURE ONLY (checker + reported CPU baseline).  Parity status: PINNED
 * (tests/test_oracle_golden.py, fixtures from tools/gen_golden.py).
 *
 * Written from the behaviour of /root/reference/src/ml2048/game_numba.py; each function names the
 * lines it follows.  Control flow is deliberately the scalar, per-board form of the reference so
 * that it is an independent check on the bit-parallel HIP kernels.
 */
#include "ml2048_oracle.h"

#include <math.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_abi_version(void) { return 1; }

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* game_numba.py:48-90 */
void orc_push_row(uint8_t *board, int offset, int stride, uint8_t *buckets) {
    uint8_t memo = 0; /* held tile */
    int w = 0;        /* write cursor */
    for (int r = 0; r < 4; ++r) {
        uint8_t num = board[offset + r * stride];
        if (num == 0) continue;
        if (memo == 0) {
            memo = num;
        } else if (memo == num) {
            memo = 0;
            board[offset + w * stride] = (uint8_t)(num + 1);
            ++w;
            if (buckets) buckets[num] += 1; /* :76 - index is the pre-merge exponent */
        } else {
            board[offset + w * stride] = memo;
            ++w;
            memo = num;
        }
    }
    if (memo != 0) {
        board[offset + w * stride] = memo;
        ++w;
    }
    for (; w < 4; ++w) board[offset + w * stride] = 0;
}

/* game_numba.py:93-134 : (offset, stride) per direction */
void orc_step_kernel(uint8_t *board, uint8_t *merged, int64_t action) {
    static const int off[4][4] = {{0, 4, 8, 12}, {3, 7, 11, 15}, {0, 1, 2, 3}, {12, 13, 14, 15}};
    static const int str[4] = {1, -1, 4, -4};
    if (action < 0 || action > 3) return; /* :127-134 falls through every branch */
    for (int j = 0; j < 4; ++j) orc_push_row(board, off[action][j], str[action], merged);
}

/* game_numba.py:172-212.  Note :207 indexes randfloat by the *cell* index (H1). */
int orc_spawn2(const uint8_t *randperm, const float *randfloat, int64_t rand_idx, uint8_t *board,
               double two_prob, int count) {
    while (rand_idx >= ORC_RAND_SIZE) rand_idx -= ORC_RAND_SIZE; /* :195-196 */
    const uint8_t *perm = randperm + rand_idx * 16;
    int i = 0;
    while (count > 0 && i < 16) {
        int idx = perm[i];
        ++i;
        if (board[idx] != 0) continue;
        --count;
        /* float32 < float64 compares in float64 under numba (H2) */
        board[idx] = ((double)randfloat[idx] < two_prob) ? 1 : 2;
    }
    return count == 0;
}

/* game_numba.py:215-256 */
void orc_line_movable(int n1, int n2, int n3, int n4, int *forward, int *backward) {
    int z1 = n1 == 0, z2 = n2 == 0, z3 = n3 == 0, z4 = n4 == 0;
    int m41 = z3 & z2 & (z1 | (n1 == n4));
    int m14 = z2 & z3 & (z4 | (n1 == n4));
    int m31 = z2 & (z1 | (n1 == n3));
    int m13 = z2 & (z3 | (n1 == n3));
    int m21 = z1 | (n1 == n2);
    int m12 = z2 | (n1 == n2);
    int m42 = z3 & (z2 | (n2 == n4));
    int m24 = z3 & (z4 | (n2 == n4));
    int m32 = z2 | (n2 == n3);
    int m23 = z3 | (n2 == n3);
    int m43 = z3 | (n3 == n4);
    int m34 = z4 | (n3 == n4);
    int f2 = (n2 >= 1) && m21 != 0;
    int f3 = (n3 >= 1) && (m31 | m32) != 0;
    int f4 = (n4 >= 1) && (m41 | m42 | m43) != 0;
    int b3 = (n3 >= 1) && m34 != 0;
    int b2 = (n2 >= 1) && (m23 | m24) != 0;
    int b1 = (n1 >= 1) && (m12 | m13 | m14) != 0;
    *forward = (f2 | f3 | f4) != 0;
    *backward = (b3 | b2 | b1) != 0;
}

/* game_numba.py:259-289 */
int orc_compute_valid_actions(const uint8_t *b, uint8_t *result) {
    int l[4], r[4], u[4], d[4];
    for (int j = 0; j < 4; ++j) {
        orc_line_movable(b[4 * j], b[4 * j + 1], b[4 * j + 2], b[4 * j + 3], &l[j], &r[j]);
        orc_line_movable(b[j], b[4 + j], b[8 + j], b[12 + j], &u[j], &d[j]);
    }
    int left = l[0] | l[1] | l[2] | l[3];
    int right = r[0] | r[1] | r[2] | r[3];
    int up = u[0] | u[1] | u[2] | u[3];
    int down = d[0] | d[1] | d[2] | d[3];
    result[0] = (uint8_t)left;
    result[1] = (uint8_t)right;
    result[2] = (uint8_t)up;
    result[3] = (uint8_t)down;
    return left | right | up | down;
}

/* exhaustive table over all 18^4 lines (itertools.product order): orc_push_row + orc_line_movable */
void orc_lines18_table(uint8_t *result, uint8_t *buckets, uint8_t *movable) {
    int64_t i = 0;
    for (int a = 0; a < 18; ++a)
        for (int b = 0; b < 18; ++b)
            for (int c = 0; c < 18; ++c)
                for (int d = 0; d < 18; ++d, ++i) {
                    uint8_t *r = result + i * 4;
                    r[0] = (uint8_t)a; r[1] = (uint8_t)b; r[2] = (uint8_t)c; r[3] = (uint8_t)d;
                    memset(buckets + i * 18, 0, 18);
                    orc_push_row(r, 0, 1, buckets + i * 18);
                    int f, bk;
                    orc_line_movable(a, b, c, d, &f, &bk);
                    movable[i * 2] = (uint8_t)f;
                    movable[i * 2 + 1] = (uint8_t)bk;
                }
}

/* game_numba.py:408-438 : int64 accumulation, then float() */
static int64_t reward_normal_i64(const uint8_t *merged) {
    int64_t reward = 0;
    for (int k = 0; k < 16; ++k) reward += (int64_t)merged[k] * ((int64_t)2 << k);
    return reward;
}

double orc_reward(int kind, const uint8_t *state, const uint8_t *prev, const uint8_t *merged) {
    switch (kind) {
    default:
    case ORC_REWARD_NORMAL:
        return (double)reward_normal_i64(merged);
    case ORC_REWARD_IMPROVED: { /* :441-466 */
        double reward = (double)reward_normal_i64(merged);
        int64_t extra = 0;
        const int64_t factor = 64;
        if (prev[0] != 0) extra -= factor * ((int64_t)1 << prev[0]);
        if (state[0] != 0) extra += factor * ((int64_t)1 << state[0]);
        return reward + (double)extra;
    }
    case ORC_REWARD_RANK: { /* :469-484 */
        double reward = 0.0;
        for (int idx = 0; idx < 16; ++idx) reward += (double)((int64_t)(idx + 1) * merged[idx]);
        return reward;
    }
    case ORC_REWARD_MAXCELL: { /* :487-504 */
        double reward = 0.0;
        uint8_t cur = 0, prv = 0;
        int64_t sum = 0;
        for (int i = 0; i < 16; ++i) {
            if (state[i] > cur) cur = state[i];
            if (prev[i] > prv) prv = prev[i];
            sum += merged[i];
        }
        if (cur > prv) reward += (double)((int64_t)1 << cur);
        reward += (double)sum;
        return reward;
    }
    }
}

/* game_numba.py:613-617 */
void orc_reset_records(orc_record *data, int64_t M, uint8_t *prev_state, uint8_t *prev_valid) {
    memset(data, 0, (size_t)M * sizeof(orc_record));
    for (int64_t i = 0; i < M; ++i) data[i].terminated = 1;
    if (prev_state) memset(prev_state, 0, (size_t)M * 16);
    if (prev_valid) memset(prev_valid, 0, (size_t)M * 4);
}

/* game_numba.py:629-658 : ascending slot order, id = running game_count (H7) */
int64_t orc_prepare(orc_record *data, int64_t M, int64_t idx_base, int64_t row_base,
                    const uint8_t *randperm, const float *randfloat, double two_prob,
                    int64_t *game_count, int64_t *indices) {
    int64_t n = 0;
    for (int64_t idx = 0; idx < M; ++idx) {
        if (!data[idx].terminated) continue;
        orc_record *e = &data[idx];
        memset(e, 0, sizeof(*e)); /* entry.fill(0) :639 */
        e->id = (int32_t)(*game_count);
        *game_count += 1;
        orc_spawn2(randperm, randfloat, (row_base + idx_base + idx) % ORC_RAND_SIZE, e->board,
                   two_prob, 2);
        orc_compute_valid_actions(e->board, e->valid_actions);
        if (indices) indices[n] = idx;
        ++n;
    }
    return n;
}

static inline int64_t load_action(const void *actions, int dtype, int64_t i) {
    switch (dtype) {
    case ORC_ACT_U8:
        return ((const uint8_t *)actions)[i];
    case ORC_ACT_I32:
        return ((const int32_t *)actions)[i];
    default:
        return ((const int64_t *)actions)[i];
    }
}

/* one board of _vec_step's loop body, game_numba.py:716-738, up to (not including) the spawn.
 * returns 1 when the move was valid. */
static inline int step_one_pre(orc_record *e, int64_t action, const uint8_t *prev_board,
                               int reward_kind) {
    if (action < 0 || action > 3 || !e->valid_actions[action]) { /* H11: out-of-range = invalid */
        e->invalid = 1; /* :738 - nothing else changes (H3) */
        return 0;
    }
    e->step += 1;
    /* numba does no bounds check: buckets[16]/[17] land on valid_actions[0..1] (H9), which
     * _compute_valid_actions rewrites below before anything reads them. */
    uint8_t *raw = (uint8_t *)e;
    memset(raw + 32, 0, 16); /* merged[0:16] = 0 :723 */
    orc_step_kernel(raw + 16, raw + 32, action);
    double reward = orc_reward(reward_kind, e->board, prev_board, e->merged);
    double ds = orc_reward(ORC_REWARD_NORMAL, e->board, prev_board, e->merged);
    e->reward = (float)reward;
    e->score = (float)((double)e->score + ds); /* f32 + f64 -> f64 -> f32 (H8) */
    return 1;
}

static inline void step_one_post(orc_record *e) {
    int playable = orc_compute_valid_actions(e->board, e->valid_actions);
    e->terminated = (uint8_t)!playable;
    e->invalid = 0;
}

void orc_step(orc_record *data, int64_t M, int64_t idx_base, const void *actions, int action_dtype,
              uint8_t *prev_state, uint8_t *prev_valid, int reward_kind, double two_prob,
              int64_t rand_seed, const uint8_t *randperm, const float *randfloat) {
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < M; ++idx) {
        orc_record *e = &data[idx];
        memcpy(prev_state + idx * 16, e->board, 16);      /* :672 */
        memcpy(prev_valid + idx * 4, e->valid_actions, 4); /* :673 */
        int64_t a = load_action(actions, action_dtype, idx);
        if (!step_one_pre(e, a, prev_state + idx * 16, reward_kind)) continue;
        orc_spawn2(randperm, randfloat, (rand_seed + idx_base + idx) % ORC_RAND_SIZE, e->board,
                   two_prob, 1); /* :733 */
        step_one_post(e);
    }
}

/* policy/_network.py:91-95,105 */
int64_t orc_encode_onehot(const uint8_t *board, int64_t M, float *out) {
    int64_t bad = 0;
    memset(out, 0, (size_t)M * 256 * sizeof(float));
    for (int64_t n = 0; n < M; ++n)
        for (int cell = 0; cell < 16; ++cell) {
            int k = board[n * 16 + cell];
            if (k < 16)
                out[n * 256 + k * 16 + cell] = 1.0f;
            else
                ++bad;
        }
    return bad;
}

/* runner.py:120-136 / game_numba.py:593-604 */
void orc_hist_maxcell(const uint8_t *board, int64_t board_stride, const uint8_t *terminated,
                      int64_t term_stride, int64_t M, int32_t *counts20) {
    for (int64_t i = 0; i < M; ++i) {
        if (terminated && !terminated[i * term_stride]) continue;
        const uint8_t *b = board + i * board_stride;
        uint8_t mx = 0;
        for (int c = 0; c < 16; ++c)
            if (b[c] > mx) mx = b[c];
        if (mx < 20) counts20[mx] += 1;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10, Salmon/Moraes/Dror/Shaw SC'11.  Multipliers 0xD2511F53, 0xCD9E8D57; Weyl key
 * increments 0x9E3779B9, 0xBB67AE85.                                                          */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void philox_at(uint64_t seed, uint64_t counter, uint64_t idx, uint32_t out[4]) {
    uint32_t ctr[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)idx,
                       (uint32_t)(idx >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

/* -ln(cnt) as float32 literals, shared with the HIP kernels so both sides agree bit for bit */
static inline float neg_log_count(int cnt) {
    return cnt == 4 ? -1.38629436111989f : cnt == 3 ? -1.09861228866811f
         : cnt == 2 ? -0.693147180559945f : 0.0f;
}

void orc_sample_actions(const uint8_t *valid, int64_t valid_stride, int64_t M, int64_t idx_base,
                        uint64_t seed, uint64_t counter, int masked, uint8_t *actions,
                        float *logp) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        /* The stream of board i is rocRAND's Philox sequence W(0), W(1), ...: W(j) = word (j&3) of block (j>>2).
           A masked draw uses the whole word W(t); a plain uniform draw the two bits 2(t&15).. of W(t>>4). */
        const uint64_t j = masked ? counter : (counter >> 4);
        uint32_t blk[4];
        philox_at(seed, j >> 2, (uint64_t)(idx_base + i), blk);
        uint32_t r[1] = {blk[j & 3u]};
        if (!masked) {
            actions[i] = (uint8_t)((r[0] >> (2u * (unsigned)(counter & 15u))) & 3u);
            if (logp) logp[i] = neg_log_count(4);
            continue;
        }
        const uint8_t *v = valid + i * valid_stride;
        int cnt = (v[0] != 0) + (v[1] != 0) + (v[2] != 0) + (v[3] != 0);
        if (cnt == 0) {
            actions[i] = 0;
            if (logp) logp[i] = 0.0f;
            continue;
        }
        int k = (int)(((uint64_t)r[0] * (uint64_t)cnt) >> 32);
        int a = 0;
        for (int j = 0; j < 4; ++j)
            if (v[j]) {
                if (k == 0) {
                    a = j;
                    break;
                }
                --k;
            }
        actions[i] = (uint8_t)a;
        if (logp) logp[i] = neg_log_count(cnt);
    }
}

static inline uint64_t mix_word(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    return h;
}

uint64_t orc_digest(const orc_record *data, int64_t M, int64_t idx_base, int with_id) {
    uint64_t total = 0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t i = 0; i < M; ++i) {
        const orc_record *e = &data[i];
        uint64_t w[7];
        memcpy(&w[0], e->board, 8);
        memcpy(&w[1], e->board + 8, 8);
        memcpy(&w[2], e->merged, 8);
        memcpy(&w[3], e->merged + 8, 8);
        uint32_t v;
        memcpy(&v, e->valid_actions, 4);
        w[4] = (uint64_t)v | ((uint64_t)e->terminated << 32) | ((uint64_t)e->invalid << 40);
        w[5] = (uint64_t)(uint32_t)e->step;
        uint32_t sb, rb;
        memcpy(&sb, &e->score, 4);
        memcpy(&rb, &e->reward, 4);
        w[6] = (uint64_t)sb | ((uint64_t)rb << 32);
        uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(idx_base + i + 1);
        for (int k = 0; k < 7; ++k) h = mix_word(h, w[k]);
        if (with_id) h = mix_word(h, (uint64_t)(uint32_t)e->id);
        total += h;
    }
    return total;
}

/* ------------------------------------------------------------------------------------------ */
/* Fast-mode spawn (spawn="rocrand"): on-device counter-based RNG instead of the host PCG64 tables.  Build-defined
 * (the reference has no such mode; it is the same GAME as _spawn2, game_numba.py:172-212: a uniformly random empty cell,
 * a 2 with probability two_prob); the HIP kernels' fast mode must equal this bit for bit.
 *   one 32-bit word w of the board's Philox4x32-10 sequence per spawned tile:
 *       cell = k-th empty cell (ascending), k = ((w >> 16) * n_empty) >> 16;
 *       tile = ((w & 0xFFFF) < ceil(two_prob * 2^16)) ? 1 : 2.
 *   step call c:    word c of the sequence keyed (seed; subsequence = board index)   -> block c >> 2, word c & 3
 *   re-deal call c: words 2c, 2c+1 of the sequence keyed (seed ^ 0x9E3779B97F4A7C15; board index)              */
#define ORC_DEAL_SEED_XOR 0x9E3779B97F4A7C15ull
static inline uint32_t two_threshold16(double two_prob) {
    double t = ceil(two_prob * 65536.0);
    if (t <= 0.0) return 0;
    if (t >= 65536.0) return 65536u;
    return (uint32_t)t;
}

static inline uint32_t philox_seq_word(uint64_t seed, uint64_t j, uint64_t idx) {
    uint32_t blk[4];
    philox_at(seed, j >> 2, idx, blk);
    return blk[j & 3u];
}

static inline void spawn_fast_one(uint8_t *board, uint32_t w, uint32_t thr16) {
    int n_empty = 0;
    for (int c = 0; c < 16; ++c) n_empty += board[c] == 0;
    if (n_empty == 0) return;
    int k = (int)(((w >> 16) * (uint32_t)n_empty) >> 16);
    for (int c = 0; c < 16; ++c)
        if (board[c] == 0) {
            if (k == 0) {
                board[c] = ((w & 0xFFFFu) < thr16) ? 1 : 2;
                return;
            }
            --k;
        }
}

void orc_step_fast(orc_record *data, int64_t M, int64_t idx_base, const void *actions,
                   int action_dtype, uint8_t *prev_state, uint8_t *prev_valid, int reward_kind,
                   double two_prob, uint64_t seed, uint64_t counter) {
    const uint32_t thr = two_threshold16(two_prob);
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < M; ++idx) {
        orc_record *e = &data[idx];
        memcpy(prev_state + idx * 16, e->board, 16);
        memcpy(prev_valid + idx * 4, e->valid_actions, 4);
        int64_t a = load_action(actions, action_dtype, idx);
        if (!step_one_pre(e, a, prev_state + idx * 16, reward_kind)) continue;
        spawn_fast_one(e->board, philox_seq_word(seed, counter, (uint64_t)(idx_base + idx)), thr);
        step_one_post(e);
    }
}

int64_t orc_prepare_fast(orc_record *data, int64_t M, int64_t idx_base, double two_prob,
                         uint64_t seed, uint64_t counter, int64_t *game_count, int64_t *indices) {
    const uint32_t thr = two_threshold16(two_prob);
    int64_t n = 0;
    for (int64_t idx = 0; idx < M; ++idx) {
        if (!data[idx].terminated) continue;
        orc_record *e = &data[idx];
        memset(e, 0, sizeof(*e));
        e->id = (int32_t)(*game_count);
        *game_count += 1;
        spawn_fast_one(e->board, philox_seq_word(seed ^ ORC_DEAL_SEED_XOR, 2 * counter, (uint64_t)(idx_base + idx)), thr);
        spawn_fast_one(e->board, philox_seq_word(seed ^ ORC_DEAL_SEED_XOR, 2 * counter + 1, (uint64_t)(idx_base + idx)), thr);
        orc_compute_valid_actions(e->board, e->valid_actions);
        if (indices) indices[n] = idx;
        ++n;
    }
    return n;
}

/* gae.py:50, :58-68.  gamma and the python-float product gamma*lambda are each rounded once to
 * float32 (torch's treatment of python scalars); every operation is a separate float32 op. */
void orc_gae(int64_t U, int64_t S, int64_t G, const float *v0, const float *v1, const float *reward,
             const uint8_t *terminated, double gamma, double lambda_, float *adv, float *delta_out) {
    const float g = (float)gamma, coef = (float)(gamma * lambda_);
    for (int64_t u = 0; u < U; ++u)
        for (int64_t j = 0; j < G; ++j) {
            float tmp = 0.0f;
            for (int64_t s = S - 1; s >= 0; --s) {
                const int64_t k = (u * S + s) * G + j;
                const float mask = terminated[k] ? 0.0f : 1.0f;
                volatile float a = g * v1[k];
                volatile float b = a * mask;
                volatile float c = b + reward[k];
                const float d = c - v0[k];
                volatile float e = tmp * coef;
                volatile float f = e * mask;
                tmp = d + f;
                adv[k] = tmp;
                if (delta_out) delta_out[k] = d;
            }
        }
}
