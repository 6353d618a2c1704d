/*
 * ml2048_oracle.h - CPU restatement (plain C) of the reference's vectorized 2048 environment.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package (ml2048_amd/) may include, link,
 * import or execute anything under oracle/.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py use it - as the checker / reported CPU baseline, never as the
 * thing shipped.
 *
 * Parity status: PINNED.  Checked (tests/test_oracle_golden.py) against
 *   - the reference's only known-answer vector (playground.ipynb cell 11), and
 *   - fixtures produced by executing the reference's own source (src/ml2048/game_numba.py) in
 *     this build container with tools/gen_golden.py: exhaustive 18^4 line table, host RNG
 *     tables/schedule, and per-step sha256 of the 64-byte records for seeded rollouts
 *     (tests/golden/).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * repository root).
 */
#ifndef ML2048_ORACLE_H
#define ML2048_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/ml2048/game.py:14-17 */
enum { ORC_STEP_LEFT = 0, ORC_STEP_RIGHT = 1, ORC_STEP_UP = 2, ORC_STEP_DOWN = 3 };

/* reward function selector: src/ml2048/game_numba.py:408-504 */
enum { ORC_REWARD_NORMAL = 0, ORC_REWARD_IMPROVED = 1, ORC_REWARD_RANK = 2, ORC_REWARD_MAXCELL = 3 };

/* action array element type */
enum { ORC_ACT_U8 = 0, ORC_ACT_I32 = 1, ORC_ACT_I64 = 2 };

#define ORC_RAND_SIZE 1024 /* VecGame._RAND_SIZE, game_numba.py:533 */

/* VecGame._DATA_SPEC, game_numba.py:537-551 (align=True, itemsize asserted 64) */
typedef struct orc_record {
    int32_t id;               /* @0  */
    int32_t step;             /* @4  */
    float score;              /* @8  */
    float reward;             /* @12 */
    uint8_t board[16];        /* @16 */
    uint8_t merged[16];       /* @32 */
    uint8_t valid_actions[4]; /* @48 */
    uint8_t terminated;       /* @52 */
    uint8_t invalid;          /* @53 */
    uint8_t _padding[10];     /* @54 */
} orc_record;

int orc_abi_version(void);
int orc_num_threads(void);
void orc_set_num_threads(int n);

/* game_numba.py:48-90.  `buckets` may be NULL; otherwise it must have room for index `num`
 * (up to 17 for an 18-symbol line; the reference's 16-slot `merged` overflows for num==16, H9). */
void orc_push_row(uint8_t *board, int offset, int stride, uint8_t *buckets);
/* all 18^4 lines in itertools.product order: result[n][4], buckets[n][18], movable[n][2] */
void orc_lines18_table(uint8_t *result, uint8_t *buckets, uint8_t *movable);
/* game_numba.py:93-134 */
void orc_step_kernel(uint8_t *board, uint8_t *merged, int64_t action);
/* game_numba.py:172-212 */
int orc_spawn2(const uint8_t *randperm, const float *randfloat, int64_t rand_idx, uint8_t *board,
               double two_prob, int count);
/* game_numba.py:215-256 */
void orc_line_movable(int n1, int n2, int n3, int n4, int *forward, int *backward);
/* game_numba.py:259-289 */
int orc_compute_valid_actions(const uint8_t *board, uint8_t *result);
/* game_numba.py:408-504; numba integer typing (int64) then float() */
double orc_reward(int kind, const uint8_t *state, const uint8_t *prev_state, const uint8_t *merged);

/* VecGame.reset's record part, game_numba.py:613-617 */
void orc_reset_records(orc_record *data, int64_t M, uint8_t *prev_state, uint8_t *prev_valid);

/* VecGame.prepare's reset loop, game_numba.py:629-658.  row_base = rand_step + rand_offset.
 * `indices` (may be NULL) receives the ascending reset slots; returns their count and advances
 * *game_count.  idx_base offsets the table row for a shard of a larger global game. */
int64_t orc_prepare(orc_record *data, int64_t M, int64_t idx_base, int64_t row_base,
                    const uint8_t *randperm, const float *randfloat, double two_prob,
                    int64_t *game_count, int64_t *indices);

/* VecGame.step = the two copies (game_numba.py:672-673) + _vec_step (:701-738).
 * rand_seed = rand_step + rand_offset.  OpenMP static schedule over boards mirrors numba.prange. */
void orc_step(orc_record *data, int64_t M, int64_t idx_base, const void *actions, int action_dtype,
              uint8_t *prev_state, uint8_t *prev_valid, int reward_kind, double two_prob,
              int64_t rand_seed, const uint8_t *randperm, const float *randfloat);

/* policy/_network.py:91-95,105: out[n][k][r][c] = (board[n][4r+c]==k), k<16, float32.
 * returns the number of cells with exponent >= 16 (un-encodable, H10). */
int64_t orc_encode_onehot(const uint8_t *board, int64_t M, float *out);

/* runner.py:120-136 (_update_count): counts[max(board)] += 1 for terminated boards
 * (terminated==NULL: all boards, i.e. the histogram behind VecGame.summary, game_numba.py:593). */
void orc_hist_maxcell(const uint8_t *board, int64_t board_stride, const uint8_t *terminated,
                      int64_t term_stride, int64_t M, int32_t *counts20);

/* ---- build-defined pieces (no reference counterpart); the oracle side of their parity ---- */

/* Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14 constants).  ctr/key/out: 4/2/4 words. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Action sampler.  word = philox(ctr={(counter>>2)_lo,_hi,idx_lo,idx_hi}, key=seed)[counter & 3]
 * (= rocRAND's Philox sequence of subsequence idx at offset `counter`).
 *  masked=0: action = word & 3, logp = -log(4).
 *  masked=1: cnt = #valid; k = (word * cnt) >> 32; action = k-th set bit in L,R,U,D order;
 *            logp = -logf(cnt); cnt==0 -> action 0, logp 0.
 * (semantics of policy/random.py:17-27 - uniform over valid - with a reproducible stream) */
void orc_sample_actions(const uint8_t *valid, int64_t valid_stride, int64_t M, int64_t idx_base,
                        uint64_t seed, uint64_t counter, int masked, uint8_t *actions, float *logp);

/* order-independent 64-bit digest over all record fields of all boards (with_id=0: all but id) */
uint64_t orc_digest(const orc_record *data, int64_t M, int64_t idx_base, int with_id);

/* Fast-mode spawn (not reference-exact; statistically equivalent): uses philox word pair. */
void orc_step_fast(orc_record *data, int64_t M, int64_t idx_base, const void *actions,
                   int action_dtype, uint8_t *prev_state, uint8_t *prev_valid, int reward_kind,
                   double two_prob, uint64_t seed, uint64_t counter);
int64_t orc_prepare_fast(orc_record *data, int64_t M, int64_t idx_base, double two_prob,
                         uint64_t seed, uint64_t counter, int64_t *game_count, int64_t *indices);

/* compute_gae's arithmetic, src/ml2048/gae.py:50 and :58-68, on [U][S][G] float32 tensors, in the
 * reference's float32 operation order (compile with -ffp-contract=off). */
void orc_gae(int64_t U, int64_t S, int64_t G, const float *v0, const float *v1, const float *reward,
             const uint8_t *terminated, double gamma, double lambda_, float *adv, float *delta_out);

#ifdef __cplusplus
}
#endif
#endif
