/*
 * ml2048_hip.h - C ABI of libml2048_hip.so: the MI355X (gfx950) vectorized 2048 environment.
 *
 * The reference (tsangwpx/ml2048 @ 2024_10_08) has no FFI: its hot path is a Python class
 * (`VecGame`, src/ml2048/game_numba.py:522-698) whose per-board work numba JIT-compiles
 * (`_vec_step` :701-738 and the helpers it inlines).  This header is the boundary a drop-in
 * replacement of that path binds instead of numba: every entry point below names the reference
 * lines it replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no C++ / torch types.
 *  - Every buffer is CALLER-OWNED DEVICE memory (e.g. torch.empty(..., device="cuda").data_ptr());
 *    the library allocates nothing and keeps no state between calls (re-entrant; one host
 *    thread per GPU is fine).
 *  - `stream` is a hipStream_t (NULL = the null stream).  All work is enqueued asynchronously;
 *    no entry point synchronizes.  All launches are hipGraph-capturable.
 *  - Return 0 on success; < 0 library error (ML2048_E_*), > 0 a hipError_t.  Text of the last
 *    error of the calling thread: ml2048_last_error().
 *  - Board cells are tile EXPONENTS (0 empty, k = tile 2^k), one uint8 each, cell 4*row+col
 *    (game_numba.py:13-20).  Actions: 0 left, 1 right, 2 up, 3 down (src/ml2048/game.py:14-17).
 */
#ifndef ML2048_HIP_H
#define ML2048_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ML2048_ABI_VERSION 1
#define ML2048_RAND_SIZE 1024 /* VecGame._RAND_SIZE, game_numba.py:533 */
#define ML2048_EXPANDED_ROW_BYTES 48 /* a permutation row as ml2048_rollout_chain reads it */

enum { ML2048_E_INVALID = -1, ML2048_E_ALIGN = -2, ML2048_E_UNSUPPORTED = -3 };
enum { ML2048_REWARD_NORMAL = 0, ML2048_REWARD_IMPROVED = 1, ML2048_REWARD_RANK = 2, ML2048_REWARD_MAXCELL = 3 };
enum { ML2048_ACT_U8 = 0, ML2048_ACT_I32 = 1, ML2048_ACT_I64 = 2 };

/*
 * Device-resident game state, structure-of-arrays.  Field for field the reference's 64-byte
 * record `VecGame._DATA_SPEC` (game_numba.py:537-551) plus its two persistent copies
 * `_prev_state` / `_prev_valid_actions` (:574-575), split per field so that one wavefront's
 * 64 boards are one contiguous, fully coalesced segment of every array.
 * `board`, `merged`, `prev_board` must be 16-byte aligned; the 4-byte fields 4-byte aligned.
 */
typedef struct ml2048_state {
    uint8_t *board;      /* [M][16]  data["board"]          */
    uint8_t *merged;     /* [M][16]  data["merged"]         */
    uint8_t *valid;      /* [M][4]   data["valid_actions"]  (L,R,U,D as 0/1) */
    int32_t *step;       /* [M]      data["step"]           */
    float *score;        /* [M]      data["score"]          */
    float *reward;       /* [M]      data["reward"]         */
    uint8_t *terminated; /* [M]      data["terminated"]     */
    uint8_t *invalid;    /* [M]      data["invalid"]        */
    int32_t *id;         /* [M]      data["id"]             */
    uint8_t *prev_board; /* [M][16]  VecGame._prev_state         */
    uint8_t *prev_valid; /* [M][4]   VecGame._prev_valid_actions */
} ml2048_state;

int ml2048_abi_version(void);
const char *ml2048_last_error(void);
/* number of int32 words of device scratch ml2048_prepare* needs for M boards */
int64_t ml2048_scratch_ints(int64_t M);

/* VecGame.reset's record part (game_numba.py:613-617): zero every field, terminated = 1. */
int ml2048_reset(void *stream, int64_t M, const ml2048_state *st);

/*
 * VecGame.prepare's reset loop (game_numba.py:629-658): every terminated board is zeroed, gets
 * id = *game_count + (its rank among the terminated boards, ascending slot order), two tiles from
 * table row (row_base + idx_base + idx) mod 1024 (`_spawn2(..., 2)`, :648-655) and its
 * valid-action mask (:656).  row_base = rand_step + rand_offset drawn by the host (:622-626).
 *   game_count    device int32, advanced by the number of resets (VecGame._game_count)
 *   n_reset       device int32 out: number of boards reset
 *   reset_indices device int64[M] out or NULL: the ascending slots (the tuple prepare() returns)
 *   scratch       device int32[ml2048_scratch_ints(M)], 8-byte aligned, ZEROED ONCE by the caller before its
 *                 first use and then left alone: it holds the 64-bit status words of the single-pass ordered
 *                 scan, which are tagged with a per-launch epoch and never need resetting between calls
 * randperm: device uint8[1024][16]; randfloat: device float[1024] (VecGame._randperm/_randfloat).
 */
int ml2048_prepare(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st,
                   const uint8_t *randperm, const float *randfloat, double two_prob,
                   int64_t row_base, int32_t *game_count, int32_t *n_reset,
                   int64_t *reset_indices, int32_t *scratch);

/*
 * VecGame.step's device part: the two copies (game_numba.py:672-673) and `_vec_step`
 * (:701-738) - slide/merge (`_step_kernel` :125-134, `_push_row` :48-90), reward_fn (:408-504,
 * selected by reward_kind), score, one-tile spawn from table row (row_base + idx_base + idx) mod
 * 1024 (`_spawn2`, :172-212), valid mask (`_compute_valid_actions` :259-289), terminated/invalid.
 * An action that is not currently valid only sets invalid = 1 (:737-738).  Actions outside 0..3
 * are treated as invalid.  row_base = rand_step + rand_offset drawn by the host (:670, :681).
 */
int ml2048_step(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st,
                const void *actions, int action_dtype, int reward_kind, double two_prob,
                int64_t row_base, const uint8_t *randperm, const float *randfloat);

/*
 * ml2048_step with the one-wavefront-per-board mapping (board staged in LDS, 4 lanes own the 4
 * lines, 16 lanes probe the spawn row, 8 lanes the movability tests).  Identical semantics and
 * results; exists so that the two mappings can be measured against each other.
 */
int ml2048_step_wave(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st,
                     const void *actions, int action_dtype, int reward_kind, double two_prob,
                     int64_t row_base, const uint8_t *randperm, const float *randfloat);

/*
 * rocRAND spawn mode (no reference counterpart; NOT bit-exact with the numba path, the same game
 * statistically - tests/test_gpu_rocrand_stats.py): the spawn uses rocRAND's Philox4x32-10 device
 * generator instead of the host PCG64 tables, ONE 32-bit word w per spawned tile:
 *     cell = k-th empty cell in ascending order, k = ((w >> 16) * n_empty) >> 16;
 *     tile "2" iff (w & 0xFFFF) < ceil(two_prob * 2^16).
 * `counter` is the env's call counter c (0 for the first prepare / step after a reset, +1 per pair):
 *     ml2048_step_fast     word c of the sequence keyed (seed; subsequence = idx_base + idx), so one Philox
 *                          block serves four consecutive steps (the fused rollout computes it once per four);
 *     ml2048_prepare_fast  words 2c and 2c+1 of the sequence keyed (seed ^ 0x9E3779B97F4A7C15; same subsequence).
 * Bit-exact against oracle/ml2048_oracle.c orc_*_fast.
 */
int ml2048_prepare_fast(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st,
                        double two_prob, uint64_t seed, uint64_t counter, int32_t *game_count,
                        int32_t *n_reset, int64_t *reset_indices, int32_t *scratch);
int ml2048_step_fast(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st,
                     const void *actions, int action_dtype, int reward_kind, double two_prob,
                     uint64_t seed, uint64_t counter);

/*
 * Uniform action sampling (RandomPolicy.sample_actions, src/ml2048/policy/random.py:17-27 with
 * categorical_sample, src/ml2048/stats.py:32-47: uniform over the valid actions, log-prob
 * -log(#valid)).  torch.multinomial's stream is not reproducible across devices, so the stream is
 * defined here.  W(j) = word (j & 3) of Philox4x32-10(key = seed, counter = {j >> 2, idx_base + idx}) =
 * rocRAND's sequence rocrand_init(seed, idx_base + idx, offset = j);
 * masked: word = W(counter), action = the ((word * cnt) >> 32)-th valid action in L,R,U,D order
 * (cnt == 0 -> 0);
 * unmasked: two bits per draw, action = (W(counter >> 4) >> 2 (counter & 15)) & 3 - sixteen draws per word,
 * so that a fused rollout evaluates Philox once per 64 steps (7.5 % of the rollout kernel at 1 M boards).
 * actions: uint8[M]; logp: float[M] or NULL.
 */
int ml2048_sample_actions(void *stream, int64_t M, int64_t idx_base, const uint8_t *valid,
                          uint64_t seed, uint64_t counter, int masked, uint8_t *actions,
                          float *logp);

/*
 * Observation encode (CNNEncoder.forward's first lines, src/ml2048/policy/_network.py:91-95,105):
 * onehot[n][k][r][c] = (board[n][4r+c] == k), k < 16, float32, contiguous (M,16,4,4).
 * board_i64 (or NULL): the (M,16) int64 tensor the reference network asserts as input (:87-88).
 * bad_count (or NULL): device int32, incremented by the number of cells with exponent >= 16,
 * which the reference cannot encode (F.one_hot raises; NUM_CLASSES = 16, _network.py:8).
 */
int ml2048_encode_onehot(void *stream, int64_t M, const uint8_t *board, float *onehot,
                         int64_t *board_i64, int32_t *bad_count);

/*
 * Max-tile histogram: counts[max(board[i])] += 1 for every board with terminated[i] != 0
 * (`_update_count`, src/ml2048/runner.py:120-136), or for every board when terminated is NULL
 * (the histogram behind VecGame.summary, game_numba.py:593-604).  counts: device int32[20],
 * accumulated into (not cleared).
 */
int ml2048_hist_maxcell(void *stream, int64_t M, const uint8_t *board, const uint8_t *terminated,
                        int32_t *counts);

/* SoA <-> the reference's 64-byte array-of-records (`VecGame._data`, game_numba.py:537-551,
 * :571) for callers that read `env._data` (src/ml2048/replay.py:147-151).  records: [M][64]. */
int ml2048_pack_records(void *stream, int64_t M, const ml2048_state *st, void *records);
int ml2048_unpack_records(void *stream, int64_t M, const ml2048_state *st, const void *records);

/* Order-independent 64-bit digest of all record fields of all boards (with_id = 0: all but
 * `id`), ADDED to *out (device uint64).  Same function as oracle orc_digest; used for
 * full-size parity checks. */
int ml2048_digest(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int with_id,
                  uint64_t *out);

/* Test hook: out[4*i..] = the 4 words rocRAND's Philox generator yields for
 * (seed, subsequence = idx_base + i, offset = 4 * counter); must equal Philox4x32-10 of
 * counter {counter, idx_base + i}, key seed. */
int ml2048_debug_philox(void *stream, int64_t n, int64_t idx_base, uint64_t seed, uint64_t counter,
                        uint32_t *out);

/*
 * Host RNG schedule for the bit-exact spawn mode, drawn ahead for n_steps env steps (HOST memory
 * only; no GPU work).  Restates the numpy Generator(PCG64) calls VecGame makes per prepare()/step()
 * (game_numba.py:589-591, 622-626, 670) so that N steps cost one C call instead of 3N numpy calls.
 *   rng           PCG64 state as numpy exposes it (bit_generator.state), updated in place
 *   randperm/randfloat  the host tables (uint8[1024][16], float[1024]), regenerated in place
 *   rand_step     VecGame._rand_step, updated in place
 *   schedule      int32[n_steps][3] out: {epoch index, row_base of prepare, row_base of step}
 *   perm_epochs   uint8[max_epochs][1024][16] out: epoch 0 = the tables at entry, then one per
 *                 regeneration;  two_masks uint32[max_epochs] out;  n_epochs out
 * Returns ML2048_E_UNSUPPORTED if more than max_epochs tables would be needed (state is then
 * unspecified: size max_epochs as n_steps + 1 to make that impossible).
 */
typedef struct ml2048_pcg64 {
    uint64_t state_hi, state_lo, inc_hi, inc_lo;
    int32_t has_uint32;
    uint32_t uinteger;
} ml2048_pcg64;
int ml2048_hostrng_schedule(ml2048_pcg64 *rng, uint8_t *randperm, float *randfloat, int32_t *rand_step,
                            int32_t n_steps, double two_prob, int32_t *schedule, uint8_t *perm_epochs,
                            uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs);

/*
 * Asynchronous form of ml2048_hostrng_schedule: a producer thread owned by the stream keeps `depth`
 * n_steps-schedules drawn ahead (the only entry points that own memory / a thread; explicit
 * create/destroy).  next() blocks until the oldest undelivered schedule is ready and returns pointers
 * into its slot, valid until release(); rng_after / rand_step_after / randperm_after /
 * randfloat_after are the generator, counter and tables exactly after that schedule, so the caller
 * can re-synchronise its own generator to what it has consumed and discard the rest by destroy().
 */
typedef struct ml2048_sched_stream ml2048_sched_stream;
ml2048_sched_stream *ml2048_sched_stream_create(const ml2048_pcg64 *rng, const uint8_t *randperm,
                                                const float *randfloat, int32_t rand_step, int32_t n_steps,
                                                double two_prob, int32_t depth);
int ml2048_sched_stream_next(ml2048_sched_stream *s, const int32_t **schedule, const uint8_t **perm_epochs,
                             const uint32_t **two_masks, int32_t *n_epochs, ml2048_pcg64 *rng_after,
                             int32_t *rand_step_after, const uint8_t **randperm_after,
                             const float **randfloat_after);
int ml2048_sched_stream_release(ml2048_sched_stream *s);
/*
 * A slot of the stream holds one schedule in the layout the upload kernels read:
 *     [ int32[n_steps][3] schedule, padded to 16 B | uint32[n_steps + 1] two-masks, padded | n_epochs x 1024 x 16 tables ]
 * (ml2048_sched_stream_slot_bytes(n_steps) bytes at most).  create_pinned lets the CALLER provide the `depth` slot
 * buffers - pinned host memory that a GPU copy kernel (ml2048_upload / ml2048_rollout_chain) reads in place, so that
 * handing a schedule to the device copies nothing on the host (the buffers must stay valid until
 * ml2048_sched_stream_destroy has returned; take_slot on a stream without caller buffers is ML2048_E_UNSUPPORTED):
 *   take_slot   the next schedule in draw order, NOT released: *slot = which buffer, *n_epochs, *o_mask / *o_tab = byte
 *               offsets of the masks / tables, *nbytes = bytes in use; the 20 KB of after-state (generator, counter,
 *               tables after the schedule) are copied out (randperm_after / randfloat_after may be NULL).  Up to
 *               depth schedules may be held at once (ML2048_E_UNSUPPORTED, nothing consumed, when every slot is held -
 *               leave the producer a slot or two to work ahead in);
 *   release     gives back the OLDEST held schedule, once nothing reads its slot any more.
 * take = next() + a copy of the whole slot layout into dst + release(), for callers without pinned slots
 * (ML2048_E_UNSUPPORTED, nothing consumed, when dst_capacity < *nbytes).
 */
int64_t ml2048_sched_stream_slot_bytes(int32_t n_steps);
ml2048_sched_stream *ml2048_sched_stream_create_pinned(const ml2048_pcg64 *rng, const uint8_t *randperm,
                                                       const float *randfloat, int32_t rand_step, int32_t n_steps,
                                                       double two_prob, int32_t depth, void *const *slot_mem,
                                                       int64_t slot_bytes);
int ml2048_sched_stream_take_slot(ml2048_sched_stream *s, int32_t *slot, int32_t *n_epochs, int64_t *o_mask, int64_t *o_tab,
                                  int64_t *nbytes, ml2048_pcg64 *rng_after, int32_t *rand_step_after,
                                  uint8_t *randperm_after, float *randfloat_after);
int ml2048_sched_stream_take(ml2048_sched_stream *s, void *dst, int64_t dst_capacity, int32_t *n_epochs,
                             int64_t *o_mask, int64_t *o_tab, int64_t *nbytes, ml2048_pcg64 *rng_after,
                             int32_t *rand_step_after, uint8_t *randperm_after, float *randfloat_after);
void ml2048_sched_stream_destroy(ml2048_sched_stream *s);

/* One table regeneration (`VecGame._reset_rand`, game_numba.py:589-591: permuted(axis=1) in place +
 * random(float32)) on the given generator state and HOST tables, as numpy would perform it. */
int ml2048_hostrng_regen(ml2048_pcg64 *rng, uint8_t *randperm, float *randfloat);

/*
 * The same schedule computed ON THE GPU (one 1024-thread workgroup; see csrc/devrng.hip), so that
 * the bit-exact fused rollout needs no host work.  All pointers are DEVICE memory.
 *   workspace  ml2048_devrng_workspace_bytes() bytes, filled once by ml2048_devrng_init
 *   rng        the PCG64 state + VecGame._rand_step on the device, updated in place; rng->error is
 *              set non-zero if max_epochs was too small or a regeneration window overflowed
 *              (outputs are then invalid; the host path is the fallback)
 *   randperm / randfloat  the device copies of the tables, regenerated in place
 *   schedule, perm_epochs, two_masks, n_epochs  as ml2048_hostrng_schedule, in device memory
 */
typedef struct ml2048_devrng {
    uint64_t state_hi, state_lo, inc_hi, inc_lo;
    int32_t has_uint32;
    uint32_t uinteger;
    int32_t rand_step;
    int32_t error;
} ml2048_devrng;
int64_t ml2048_devrng_workspace_bytes(void);
int ml2048_devrng_init(void *stream, void *workspace);
int ml2048_devrng_schedule(void *stream, const void *workspace, ml2048_devrng *rng, uint8_t *randperm,
                           float *randfloat, int32_t n_steps, double two_prob, int32_t *schedule,
                           uint8_t *perm_epochs, uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs);

/*
 * The same schedule from data-parallel primitives (csrc/devrng2.hip): a window of 2^18 generator outputs by
 * jump-ahead, the row-successor function of EVERY word position and its 4/16/64/256/1024-fold powers, one
 * wavefront that follows the draw order with one table look-up per regeneration, and row-parallel swaps.
 * Same arguments and results as ml2048_devrng_schedule; workspace: ml2048_devrng2_workspace_bytes() bytes,
 * 256-byte aligned, prepared once by ml2048_devrng2_init (the schedule call reuses it as scratch).
 * expanded_tables != 0: perm_epochs is uint8[max_epochs][1024][48] - ML2048_EXPANDED_ROW_BYTES per row, the table
 * form ml2048_rollout_chain consumes (described there).
 * On a non-NULL stream the dozen dependent launches of one call are captured once per argument set into a
 * hipGraph (the library keeps the last 96) and replayed with one hipGraphLaunch; calls with repeating
 * arguments (a ring of output buffers) therefore cost one enqueue.  ML2048_DEVRNG_GRAPH=0 in the environment
 * keeps the direct launches.  The workspace must stay allocated while such calls may repeat.
 */
int64_t ml2048_devrng2_workspace_bytes(void);
int ml2048_devrng2_init(void *stream, void *workspace);
int ml2048_devrng2_schedule(void *stream, void *workspace, ml2048_devrng *rng, uint8_t *randperm, float *randfloat,
                            int32_t n_steps, double two_prob, int32_t *schedule, uint8_t *perm_epochs,
                            uint32_t *two_masks, int32_t max_epochs, int32_t *n_epochs, int expanded_tables);

/*
 * Fused N-step rollout with on-device action sampling: the random-action loop
 *     for t in range(N): env.prepare(); a = policy.sample_actions(...); env.step(a)
 * (VecRunner.step_many with RandomPolicy, src/ml2048/runner.py:74-117, policy/random.py:17-27)
 * as ONE launch, every board resident in registers for all N steps; bit-exact with N calls of
 * ml2048_prepare / ml2048_sample_actions / ml2048_step on the same host RNG schedule.
 *   perm_epochs   device uint8[n_epochs][1024][16]: the successive `_randperm` tables (n_epochs <= 65536 per launch:
 *                 rows are addressed by 32-bit byte offsets from this base; more is ML2048_E_INVALID)
 *   two_masks     device uint32[n_epochs]: bit c <=> float64(randfloat[c]) < two_prob (H1: only
 *                 randfloat[0..15] are ever read, game_numba.py:207)
 *   schedule      device int32[N][3]: {table epoch, row_base of prepare, row_base of step} - the
 *                 host PCG64 draw order (:622-626, :670) does not depend on game state, so the
 *                 host pre-draws it for N steps
 *   sampler       as ml2048_sample_actions with counter = act_counter0 + t
 *   actions_out   device uint8[N][M] or NULL: the sampled actions (to replay on an oracle)
 *   digests_out   device uint64[N] or NULL: per-step digest of every field except `id`, ADDED to
 *   reset_flags   device uint64[N][ml2048_rollout_flag_words(M)]: bit = board was re-dealt by
 *                 step t's prepare;  last_reset: device int32[M] out, last t a board was dealt (-1)
 * `id` is left untouched by this launch: the reference numbers new games in ascending
 * (step, slot) order (game_numba.py:641-644), a device-wide ordered scan per step.  That scan is
 * done once, after the fact, by ml2048_rollout_finish_ids from reset_flags/last_reset.
 */
int64_t ml2048_rollout_flag_words(int64_t M);
int64_t ml2048_rollout_scratch_ints(int64_t M, int32_t N);

/*
 * Measurement only: a HIP event pair attached to the dispatch of the next ml2048_rollout* / ml2048_step_prepare launch made
 * by the calling thread (hipExtLaunchKernel start/stop events), i.e. the kernel's own duration as
 * rocprofv3 reports it, without the marker packets that events recorded around a launch add.
 * create -> arm (one-shot) -> launch -> elapsed_ms (synchronises on the stop event) -> destroy.
 * The thread's one-shots (this one, ml2048_rollout_attach_event, ml2048_rollout_attach_row_lut) are spent by the next
 * ml2048_rollout* call whether it launches or fails an argument check: none stays armed for a later, unrelated launch
 * (elapsed_ms of a timer that never rode on a launch is an error).
 */
int ml2048_timer_create(void **timer);
int ml2048_timer_arm(void *timer);
int ml2048_timer_elapsed_ms(void *timer, float *ms);
void ml2048_timer_destroy(void *timer);

/*
 * The pipelined form of ml2048_rollout: expanded tables + chained upload.
 *  - perm_epochs is uint8[n_epochs][1024][ML2048_EXPANDED_ROW_BYTES = 48]: an EXPANDED row is twelve 32-bit words
 *    {the permutation as sixteen nibbles (2 words), 0, 0, wlo[4], whi[4]}, where byte c of the weight vectors
 *    (board layout) is 2^p for the position p of cell c in the row - in wlo when p < 8, as 2^(p-8) in whi otherwise.
 *    "Which positions of the row hold an empty cell" is then a dot product of the board's emptiness flags with the
 *    weights (eight v_dot4_u32_u8 instead of probing sixteen positions; csrc/board_ops.hpp spawn_table_dot /
 *    expand_perm_row; same tile as _spawn2, game_numba.py:172-212).  Such tables are produced by ml2048_upload / by
 *    the previous ml2048_rollout_chain (table_offset < nbytes) and by ml2048_devrng*_schedule (expanded_tables).
 *  - the same launch copies the NEXT rollout's packed schedule: next_nbytes (rounded up to 16) from
 *    next_src_host (pinned / registered host memory, 16-byte aligned - a schedule-stream slot: header, then from
 *    byte next_table_offset on plain 16-byte permutation rows) to next_dst (device, 16-byte aligned, not read by this
 *    launch), expanding those rows to the 48-byte form: next_dst receives the header unchanged and
 *    3 x (next_nbytes - next_table_offset) bytes of tables behind it.  Stream order alone then guarantees the data is there
 *    for the next launch: a chain of fused rollouts needs no copy stream and no event.  next_* = NULL/0/0:
 *    no upload.  (The reference keeps its tables in host memory, game_numba.py:589-591: no counterpart.)
 */
int ml2048_rollout_chain(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                         const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                         const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed,
                         uint64_t act_counter0, uint64_t *reset_flags, int32_t *last_reset,
                         const void *next_src_host, void *next_dst, int64_t next_nbytes,
                         int64_t next_table_offset);

/*
 * Host-side plumbing of the pipelined rollout.  Events are hipEvent_t handles (timing disabled) created by the library
 * so that a binding can record / wait / poll them without its framework's event objects:
 *   query: 1 = everything recorded into the event has finished (or nothing ever was), 0 = pending, < 0 error.
 * ml2048_rollout_chain_ids = ml2048_rollout_chain on `stream`, then rollout_event (may be NULL) recorded behind its
 * kernel, then - when scratch != NULL - ml2048_rollout_finish_ids on ids_stream (== stream: inline; another stream:
 * it first waits for rollout_event, so the scan runs beside whatever `stream` does next) and ids_done_event (may be
 * NULL) recorded behind the scan.  One call instead of five: the host side of a 70 us kernel has to stay well below it.
 * rollout_event is not recorded as a marker packet but attached to the kernel's dispatch as its completion event
 * (hipExtLaunchKernel's stop event): between two dependent rollout kernels a marker costs ~2 us.  The same is available
 * to the other rollout entry points through ml2048_rollout_attach_event.
 * (No counterpart in the reference, whose loop is synchronous Python, runner.py:74-117.)
 */
int ml2048_rollout_attach_event(void *event); /* one-shot, per thread: the next ml2048_rollout* launch completes `event` */
int ml2048_rollout_event_attached(void);      /* 1 = the last rollout launch of this thread carried the attached event */

/* The line table of the plain fused rollouts.  A line of four cells with exponents <= 15 is a 16-bit index and what
 * _push_row (game_numba.py:48-90) makes of it a 16-bit value; the rollout kernels without optional outputs keep the
 * 65 536 entries in LDS and move a board with four look-ups instead of ~85 vector instructions.  The table is constant
 * data in a caller-owned device buffer of ML2048_ROW_LUT_BYTES (16-byte aligned), filled once by ml2048_row_lut_fill and
 * handed to the next rollout launch of the calling thread by ml2048_rollout_attach_row_lut (one-shot; without it the
 * launch computes the move arithmetically - same results).  A wave whose boards hold a tile above 2^11, or a rollout of
 * more than 8 191 steps, takes the arithmetic move by itself. */
#define ML2048_ROW_LUT_BYTES 131072
int ml2048_row_lut_fill(void *stream, void *lut);
int ml2048_rollout_attach_row_lut(const void *lut);
int ml2048_event_create(void **event);
void ml2048_event_destroy(void *event);
int ml2048_event_record(void *event, void *stream);
int ml2048_event_query(void *event);
int ml2048_event_synchronize(void *event);
int ml2048_stream_wait_event(void *stream, void *event);
int ml2048_rollout_chain_ids(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                             const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                             const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed,
                             uint64_t act_counter0, uint64_t *reset_flags, int32_t *last_reset,
                             const void *next_src_host, void *next_dst, int64_t next_nbytes, int64_t next_table_offset,
                             void *rollout_event, void *ids_stream, int32_t *game_count, int32_t *step_resets,
                             int32_t *scratch, void *ids_done_event);

/*
 * Upload of a pre-drawn schedule: a copy KERNEL that reads device-accessible (pinned / registered)
 * host memory directly, so that the upload is an ordinary launch on `stream` and involves neither the
 * copy engine nor a host wait.  dst: device, src_host: pinned host memory, both 16-byte aligned; nbytes
 * is rounded up to a multiple of 16.  The bytes from table_offset on are permutation rows and are expanded to the
 * 48-byte form described above (dst then needs table_offset + 3 x (nbytes - table_offset) bytes); table_offset = nbytes
 * is a plain copy.
 */
int ml2048_upload(void *stream, void *dst, const void *src_host, int64_t nbytes, int64_t table_offset);
int ml2048_rollout(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                   const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                   const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed,
                   uint64_t act_counter0, uint8_t *actions_out, uint64_t *digests_out,
                   uint64_t *reset_flags, int32_t *last_reset);
/*
 * ml2048_rollout that also materialises every step's transition - the fused form of N x
 * ml2048_record_transition (Trainer.on_stepped, run_train3.py:125-155): each non-NULL pointer is a
 * [N][M] buffer (contiguous in the game dimension) with the REPLAY_SPEC dtypes
 * (src/ml2048/replay.py:10-20).  54 bytes are written per board-step when all nine are given, which
 * makes this variant HBM-write-bound rather than VALU-bound.
 */
typedef struct ml2048_traj {
    int8_t *state;               /* [N][M][16] board before the move (after the deal)  */
    uint8_t *valid_actions;      /* [N][M][4]  its mask                                */
    int8_t *action;              /* [N][M]     sampled action                          */
    float *action_log_prob;      /* [N][M]     -log(#valid) (or -log 4 unmasked)       */
    float *reward;               /* [N][M]     record field after the step             */
    int8_t *next_state;          /* [N][M][16]                                         */
    uint8_t *next_valid_actions; /* [N][M][4]                                          */
    int32_t *step;               /* [N][M]                                             */
    uint8_t *terminated;         /* [N][M]     bool                                    */
} ml2048_traj;
int ml2048_rollout_traj(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                        const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                        const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed,
                        uint64_t act_counter0, uint64_t *reset_flags, int32_t *last_reset,
                        const ml2048_traj *traj);

/* ml2048_rollout_traj on EXPANDED tables (perm_epochs as ml2048_rollout_chain reads it: ML2048_EXPANDED_ROW_BYTES per row,
 * produced by ml2048_upload with table_offset < nbytes - from host or device memory): the dot-product spawn instead of
 * sixteen probes, the re-deal and the spawned tile as LDS look-ups; same results. */
int ml2048_rollout_traj_expanded(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                                 const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                                 const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed, uint64_t act_counter0,
                                 uint64_t *reset_flags, int32_t *last_reset, const ml2048_traj *traj);

/* The same fused rollout in rocRAND spawn mode: no host schedule at all.  Equals N calls of
 * ml2048_prepare_fast(counter = spawn_counter0+t) / ml2048_sample_actions / ml2048_step_fast(
 * counter = spawn_counter0+t); bit-exact against the oracle's orc_*_fast, not against numba. */
int ml2048_rollout_fast(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                        double two_prob, uint64_t spawn_seed, uint64_t spawn_counter0, int reward_kind,
                        int masked, uint64_t act_seed, uint64_t act_counter0, uint8_t *actions_out,
                        uint64_t *digests_out, uint64_t *reset_flags, int32_t *last_reset);
/* ids for the boards dealt during a rollout + *game_count advanced by the total; step_resets:
 * device int32[N] out or NULL (boards dealt per step = len(prepare()[0]) of each step);
 * scratch: device int32[ml2048_rollout_scratch_ints(M, N)]. */
int ml2048_rollout_finish_ids(void *stream, int64_t M, const ml2048_state *st, int32_t N,
                              const uint64_t *reset_flags, const int32_t *last_reset,
                              int32_t *game_count, int32_t *step_resets, int32_t *scratch);

/* The id scan of a rollout issued later than its launch: ml2048_stream_wait_event(ids_stream, wait_event) (skipped when
 * wait_event is NULL) + ml2048_rollout_finish_ids on ids_stream + done_event (may be NULL) recorded behind it, as one
 * call.  VecGame.rollout(finish_ids="deferred") issues rollout k's scan on a side stream right AFTER the launch of
 * rollout k+1 (so nothing is queued between two rollout kernels, and the scan runs beside kernel k+1), and the last
 * rollout's scan on the launch stream itself when something finally needs the ids (VecGame.join_ids): no cross-stream
 * wait sits between the last kernel and the end of a run. */
int ml2048_rollout_ids_after(void *ids_stream, void *wait_event, int64_t M, const ml2048_state *st, int32_t N,
                             const uint64_t *reset_flags, const int32_t *last_reset, int32_t *game_count,
                             int32_t *step_resets, int32_t *scratch, void *done_event);

/*
 * Front end of the reference's CNN encoder (CNNEncoder.forward, src/ml2048/policy/_network.py:84-119),
 * leaky_relu of the three depthwise convolutions over the one-hot board - whole board
 * (:30-35, :98-100), rows (:42-47, :108-110), columns (:54-59, :114-116) - computed as weight gathers
 * straight from the 16-byte board; the one-hot is never materialised.  multiplier = 16, 16 classes:
 *   board      device uint8[N][16] exponents (a cell >= 16 has no class and contributes nothing; the
 *              reference's F.one_hot raises for it), 16-byte aligned
 *   w_full     device float[256][16]  = _depthwise_full.weight (256,1,16);   b_full float[256]
 *   w_rows     device float[256][4]   = _depthwise_hori.weight (256,1,1,4);  b_rows float[256]
 *   w_cols     device float[256][4]   = _depthwise_vert.weight (256,1,4,1);  b_cols float[256]
 *   out_full   device float[N][256]   = leaky_relu(_depthwise_full(x))  flattened     (16-byte aligned)
 *   out_rows   device float[N][256][4] = leaky_relu(_depthwise_hori(board)) (N,256,4,1)
 *   out_cols   device float[N][256][4] = leaky_relu(_depthwise_vert(board)) (N,256,1,4)
 * Sums run over cells in ascending order; equal to torch's convolutions to rounding, not bit for bit.
 */
int ml2048_encoder_front(void *stream, int64_t N, const uint8_t *board, const float *w_full, const float *b_full,
                         const float *w_rows, const float *b_rows, const float *w_cols, const float *b_cols,
                         float negative_slope, float *out_full, float *out_rows, float *out_cols);

/*
 * ---- one env step of a policy-driven loop as ONE launch -----------------------------------------------------
 *
 * The reference's driver alternates  prepare() -> observations() -> policy -> step()
 * (VecRunner.step_once, src/ml2048/runner.py:74-109).  Between step(t) and prepare(t+1) there is only host code
 * whose PCG64 draws do not depend on game state (step draws integers(0,1024), game_numba.py:670; prepare
 * draws random(), possibly the tables, and integers(0,1024), :622-626), so ml2048_step_prepare runs
 *     step(t)   [game_numba.py:660-738]   then   prepare(t+1)   [:619-658]
 * in one kernel, and can also emit what the next policy call consumes.  After it returns (stream order) the env is
 * in the state the reference has after prepare(t+1): finished games re-dealt with ids in ascending slot order,
 * *n_reset / reset_indices as ml2048_prepare yields them, prev_board / prev_valid as step(t) left them.
 *
 * ml2048_step_out: what step(t) RETURNED - the reference's VecStepResult views (:685-698) as they read between
 * step() and the next prepare(), which clobbers them for finished games.  Dense [M] arrays with the reference's
 * dtypes; every pointer may be NULL (skipped).  Pointing them at the [game] slices of the trainer's replay buffers
 * replaces Trainer.on_stepped's copies (run_train3.py:125-155; uint8 0/1 is what torch.bool stores, uint8 boards
 * are REPLAY_SPEC's int8 bytes).  `id` is the record's id before any re-deal (replay.py:147-151 reads it per slot).
 */
typedef struct ml2048_step_out {
    uint8_t *state;              /* [M][16]  board after move + spawn                      result["state"] */
    uint8_t *valid_actions;      /* [M][4]                                                 result["valid_actions"] */
    uint8_t *merged;             /* [M][16]  stale on an invalid move (H3)                 result["merged"] */
    int32_t *step;               /* [M]      stale on an invalid move                      result["step"] */
    float *reward;               /* [M]      stale on an invalid move                      result["reward"] */
    float *score;                /* [M]                                                    result["score"] */
    uint8_t *terminated;         /* [M]                                                    result["terminated"] */
    uint8_t *invalid;            /* [M]                                                    result["invalid"] */
    int32_t *id;                 /* [M]      _data["id"] as of step(t)                                  */
    uint8_t *prev_state;         /* [M][16]  copy of st.prev_board                         result["prev_state"] */
    uint8_t *prev_valid_actions; /* [M][4]   copy of st.prev_valid                         result["prev_valid_actions"] */
} ml2048_step_out;

/*
 *   actions / action_dtype / reward_kind / two_prob / idx_base        as ml2048_step
 *   randperm_step, two_mask_step, row_base_step                       step(t)'s permutation table, 2/4 mask and
 *                                                                     rand_step + rand_offset (:670,:681)
 *   randperm_prepare, two_mask_prepare, row_base_prepare              prepare(t+1)'s (:622-626, :651): another buffer than
 *                                                                     the step's when that prepare regenerated the tables
 *   two_mask_*         bit c <=> float64(randfloat[c]) < two_prob for the 16 cells - all the reference ever reads of
 *                      randfloat (it indexes by the CELL, :207; H1, H2); ml2048_hostrng_schedule's two_masks[] entries
 *   randperm_step == NULL selects the rocRAND spawn mode: spawn_seed with spawn_counter_step / spawn_counter_prepare
 *                                                                     as ml2048_step_fast / ml2048_prepare_fast take them
 *   game_count, n_reset, reset_indices (may be NULL)                  as ml2048_prepare
 *   scratch            ml2048_step_prepare_scratch_bytes() bytes of device memory, 8-byte aligned, ZEROED ONCE by the
 *                      caller and then owned by this env (epoch-tagged status words + the epoch itself, advanced on
 *                      the device, so no host-side launch counter is baked into the arguments; a hipGraph of several steps would still
 *                      need the per-step row bases in device memory - they are plain arguments here)
 *   onehot             optional float32[M][16][4][4] observation of the post-prepare boards (_network.py:91-95,105)
 *   board_i64          optional int64[M][16] of the same boards (_network.py:87-88)
 *   next_actions       optional uint8[M]: RandomPolicy's draw for step t+1 on the post-prepare masks
 *   next_logp          optional float32[M] log-probabilities of those draws (needs next_actions)
 *   act_seed, act_counter, act_masked                                 as ml2048_sample_actions
 *   hist_counts        optional int32[20]: += max-tile histogram of the games this step finished (runner.py:120-136)
 * Plain struct, passed by pointer and copied at the call: ctypes callers keep one instance and update a few fields.
 */
typedef struct ml2048_step_prepare_args {
    int64_t M, idx_base;
    ml2048_state st;
    const void *actions;
    int32_t action_dtype, reward_kind;
    double two_prob;
    const uint8_t *randperm_step;
    int64_t row_base_step;
    const uint8_t *randperm_prepare;
    int64_t row_base_prepare;
    uint32_t two_mask_step, two_mask_prepare;
    uint64_t spawn_seed, spawn_counter_step, spawn_counter_prepare;
    int32_t *game_count;
    int32_t *n_reset;
    int64_t *reset_indices;
    void *scratch;
    ml2048_step_out out;
    float *onehot;
    int64_t *board_i64;
    uint8_t *next_actions;
    float *next_logp;
    uint64_t act_seed, act_counter;
    int32_t act_masked, reserved;
    int32_t *hist_counts;
} ml2048_step_prepare_args;
int64_t ml2048_step_prepare_scratch_bytes(void);
int ml2048_step_prepare(void *stream, const ml2048_step_prepare_args *args);

/*
 * Episode statistics of the FIRST `rounds` games - the evaluation loop of eval_perf.py:64-102 with
 * ReplayRecorder (src/ml2048/replay.py:110-232) as its collector, on device.  Call once after every
 * ml2048_step (per-step API).  Per slot, age[] counts the steps its current game has lived (what the
 * recorder's buffer.steps counts, replay.py:167: every step, valid move or not); a new id in the slot
 * restarts the count; in the step that terminates a game with id_base <= id < id_base + rounds the kernel adds
 * (1, age, final score) to entry [max tile exponent] of count / step_sum / score_sum (eval_perf.py:88-99;
 * scores are integer-valued, so the integer sum equals the reference's float sum).
 *   age, seen_id    device int32[M], 16-byte aligned (as st->id; st->terminated 4-byte aligned); initialise age
 *                   to 0 and seen_id to INT32_MIN (or any id never used)
 *   count, step_sum, score_sum   device uint64[18], zero-initialised by the caller; index = max exponent
 *   id_base         the env's game counter when the evaluation started (VecGame._game_count survives reset(),
 *                   game_numba.py:582 vs :606, so the "first N games" of an evaluation are ids id_base..id_base+N-1;
 *                   0 for a freshly constructed env, which is the only case eval_perf.py has)
 * The evaluation is complete when sum(count) == rounds ("the first N games, not the first N terminated
 * games", README.md:112-118).
 */
int ml2048_episode_stats(void *stream, int64_t M, const ml2048_state *st, int32_t *age, int32_t *seen_id,
                         int32_t id_base, int32_t rounds, uint64_t *count, uint64_t *step_sum, uint64_t *score_sum);

/*
 * ---- rollout consumers (SURVEY 8f "next" rows) ------------------------------------------------
 *
 * One transition slice straight into the trainer's `[use, step, game]` replay buffers with the
 * REPLAY_SPEC dtypes (src/ml2048/replay.py:10-20), replacing the nine per-step copies of
 * Trainer.on_stepped (run_train3.py:125-155).  Every output pointer addresses the contiguous
 * [game] slice of its buffer for the current (use, step); any of them may be NULL (skipped).
 *   state <- prev_board (int8[M][16])      valid_actions <- prev_valid (bool[M][4])
 *   next_state <- board                    next_valid_actions <- valid
 *   reward (f32), terminated (bool = uint8 0/1), step (int32) <- the record fields
 *   action (int8[M]) <- actions (uint8[M]);  action_log_prob (f32[M]) <- logp
 */
int ml2048_record_transition(void *stream, int64_t M, const ml2048_state *st, const uint8_t *actions,
                             const float *logp, int8_t *state, uint8_t *valid_actions, int8_t *action,
                             float *action_log_prob, float *reward, int8_t *next_state,
                             uint8_t *next_valid_actions, int32_t *step, uint8_t *terminated);

/*
 * Generalized advantage estimation over the step dimension (compute_gae, src/ml2048/gae.py:50,
 * :58-68).  Tensors are [U][S][G] contiguous float32 (terminated: uint8/bool):
 *     delta = gamma * v1 * (!terminated) + reward - v0                                    (:50)
 *     tmp = 0;  for s = S-1..0:  tmp = delta[:, s] + (tmp * (gamma*lambda)) * (!terminated[:, s]);
 *                                adv[:, s] = tmp                                        (:65-68)
 * with the reference's float32 operation order (no fused multiply-add; gamma and the python-float
 * product gamma*lambda each rounded once to float32, as torch does with python scalars), so the
 * result equals the torch expression bit for bit.  delta_out may be NULL.
 */
int ml2048_gae(void *stream, int64_t U, int64_t S, int64_t G, const float *v0, const float *v1,
               const float *reward, const uint8_t *terminated, double gamma, double lambda_, float *adv,
               float *delta_out);

#ifdef __cplusplus
}
#endif
#endif
