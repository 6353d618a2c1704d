// step_prepare.hip - one env step of a policy-driven loop as ONE launch (ml2048_step_prepare, include/ml2048_hip.h).
//
// The reference's driver (VecRunner.step_once, src/ml2048/runner.py:74-109) alternates
//     prepare() -> observations() -> policy -> step()
// and the per-step API of this library mirrors it with one launch each (k_prep_*, k_sample, k_step, k_onehot:
// 3-4 launches and as many Python round trips per env step, ~34 us for ~15 us of GPU work at 65 536 boards).
// Nothing separates step(t) from prepare(t+1) but host code that does not depend on game state (the PCG64 draws are
// state-independent), so this kernel runs  step(t) ; prepare(t+1)  back to back and hands the caller
// exactly what the next policy call needs: the re-dealt live state, optionally its one-hot / int64 observation
// (src/ml2048/policy/_network.py:91-95,105) and, for the uniform RandomPolicy (policy/random.py:17-27), the next
// actions themselves.  What step() returned - the reference's VecStepResult views, which the next prepare()
// clobbers for finished games - is written to caller buffers (ml2048_step_out), e.g. straight into the
// trainer's replay slices (run_train3.py:125-155).
//
// Mapping: lane per board, 256-thread workgroups, at most 1024 of them; wave g owns the contiguous boards
// [g*64*R, (g+1)*64*R) and walks them 64 at a time (R = 1 up to 262 144 boards: the whole step lives in registers),
// so "ascending slot order" (game ids, the returned index list: game_numba.py:629,641-644) is wave-major, then
// iteration-major, then lane.  The ordered numbering needs an exclusive prefix of the per-workgroup re-deal counts.
// Every workgroup publishes its count in an epoch-tagged status word right after the step phase, does everything that
// does not need the prefix (the re-deal itself: its table row depends on the slot, not on the rank; observations;
// sampling), and only then sums the words of the workgroups below it (<= 1023 words, 4 per thread, one L2 round
// trip: no serial look-back chain).  With <= 1024 workgroups of 256 threads all of them are resident at once and
// workgroups are dispatched in index order, so waiting for lower indices cannot deadlock.
// (Measured and dropped: the observation of the boards that are not re-dealt written already in the step phase, so
// that its stores would drain under the barrier and the count exchange: C3 44.5 -> 54.3 us per env step - the barrier's
// s_waitcnt then waits for 64 KiB of stores per wave BEFORE the counts are published, and on gfx9 the probes' loads
// retire behind those stores either way.)
// (Measured and dropped: up to 65 536 workgroups - one pass for any M up to 16.7 M boards - with a two-level exchange,
// own group of 64 + totals of the lower groups: the words read grow with workgroups^2 / 128 and every one is a
// device-scope access; 4 M boards 180 vs 129 us per launch, 16.7 M 753 vs 490-690 us, and the second barrier costs
// C3 47.9 vs 44.9 us.)
// The epoch lives in device memory next to the status words (the last workgroup advances it once all others have
// published, i.e. have read it): no host-side launch counter is baked into the arguments.
//
// HBM traffic per board: the per-step kernel's 99 B (R board 16 + valid 4 + step 4 + score 4 + action 1; W prev
// 16 + 4; on a valid move W board 16 + merged 16 + valid 4 + step 4 + score 4 + reward 4 + flags 2) + 1 B
// (terminated read) + what the caller asks for: one-hot 1 024 B, int64 128 B, next action 1 + log-prob 4 B,
// ml2048_step_out up to 74 B; re-deals (< 1 % of boards per step) rewrite their 58-byte record.  No MFMA: byte work.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ml2048_hip.h"
#include "board_ops.hpp"
#include "device_helpers.hpp"

using namespace ml2048;
using namespace ml2048::dev;

// error text + state checks live in ml2048_hip.hip (one thread-local message buffer per library)
__attribute__((visibility("hidden"))) int ml2048_internal_fail(int code, const char *fmt, ...);
__attribute__((visibility("hidden"))) int ml2048_internal_check_state(int64_t M, const ml2048_state *st);
__attribute__((visibility("hidden"))) uint64_t ml2048_internal_two_threshold(double two_prob);
__attribute__((visibility("hidden"))) bool ml2048_internal_take_timer(hipEvent_t *start, hipEvent_t *stop);

namespace {

constexpr int kBlk = 256;         // threads per workgroup (4 waves)
constexpr int kMaxBlocks = 1024;  // status words; all workgroups of a launch are resident at once

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct Geometry {
    int R;     // 64-board iterations per wave
    int nblk;  // workgroups
};
Geometry geometry(int64_t M) {
    const int64_t waves = cdiv(M, 64);
    int64_t max_blocks = kMaxBlocks;
    if (const char *e = getenv("ML2048_SP_MAX_BLOCKS")) { // tests: exercise the multi-pass form at small M
        const long v = atol(e);
        if (v >= 1 && v < kMaxBlocks) max_blocks = v;
    }
    int64_t R = cdiv(cdiv(waves, kBlk / 64), max_blocks);
    if (R < 1) R = 1;
    return {(int)R, (int)cdiv(waves, (kBlk / 64) * R)};
}

// (16,4,4) float one-hot of the wave's boards, wave-per-board on the store side as k_onehot does: lane = class*4 +
// row writes the 4 columns, one coalesced 1 KiB store per board; optionally the (16,) int64 row of every board
__device__ __forceinline__ void emit_obs(const Board &b, int64_t base, int cnt, int lane, float *__restrict__ onehot,
                                         int64_t *__restrict__ board_i64) {
    if (onehot) {
        const uint32_t kkkk = ((uint32_t)lane >> 2) * 0x01010101u;
        const int r = lane & 3;
        for (int j = 0; j < cnt; ++j) {
            const uint32_t r0 = __builtin_amdgcn_readlane(b.r[0], j);
            const uint32_t r1 = __builtin_amdgcn_readlane(b.r[1], j);
            const uint32_t r2 = __builtin_amdgcn_readlane(b.r[2], j);
            const uint32_t r3 = __builtin_amdgcn_readlane(b.r[3], j);
            const uint32_t row = r == 0 ? r0 : r == 1 ? r1 : r == 2 ? r2 : r3;
            const uint32_t eq = z_bytes(row ^ kkkk); // 0x80 where cell == class
            float4 o;
            o.x = (eq & 0x00000080u) ? 1.0f : 0.0f;
            o.y = (eq & 0x00008000u) ? 1.0f : 0.0f;
            o.z = (eq & 0x00800000u) ? 1.0f : 0.0f;
            o.w = (eq & 0x80000000u) ? 1.0f : 0.0f;
#if defined(ML2048_OBS_NT) && ML2048_OBS_NT // A/B only: non-temporal observation stores LOSE here (C3: 56 vs 44 us per
            // env step) - the 256 MiB observation is rewritten every step and lives in the Infinity Cache
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f q = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(onehot) + (base + j) * 64 + lane);
#else
            reinterpret_cast<float4 *>(onehot)[(base + j) * 64 + lane] = o;
#endif
        }
    }
    if (board_i64) {
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
            const int e = t * 64 + lane; // element within the wave's 64 boards
            const int src = e >> 4, cell = e & 15;
            uint32_t w = __shfl(b.r[0], src, 64);
            const uint32_t w1 = __shfl(b.r[1], src, 64);
            const uint32_t w2 = __shfl(b.r[2], src, 64);
            const uint32_t w3 = __shfl(b.r[3], src, 64);
            const int q = cell >> 2;
            w = q == 0 ? w : q == 1 ? w1 : q == 2 ? w2 : w3;
            if (src < cnt) board_i64[base * 16 + e] = (int64_t)((w >> (8 * (cell & 3))) & 0xFFu);
        }
    }
}

template <bool FAST, bool SINGLE>
__global__ __launch_bounds__(kBlk) void k_step_prepare(const ml2048_step_prepare_args a, int R, uint64_t thr) {
    __shared__ int wave_cnt[kBlk / 64];
    __shared__ int red[kBlk / 64];
    __shared__ int hist[20];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = (int)blockIdx.x;
    const int64_t M = a.M;
    const ml2048_state st = a.st;
    const int64_t wbase = ((int64_t)b * (kBlk / 64) + wave) * 64 * (int64_t)R;
    unsigned long long *const status = reinterpret_cast<unsigned long long *>(a.scratch);
    uint32_t *const epoch_p = reinterpret_cast<uint32_t *>(status + kMaxBlocks);
    // read by every workgroup BEFORE it publishes its count; advanced / updated by the last one after it has seen
    // every other workgroup's count, i.e. after all of them have read these
    const uint32_t epoch = *epoch_p + 1u;
    const int32_t gc0 = *a.game_count;
    const uint32_t two_s = a.two_mask_step, two_p = a.two_mask_prepare; // H1: randfloat is indexed by the CELL
    if (a.hist_counts) { // wave-uniform
        if (threadIdx.x < 20) hist[threadIdx.x] = 0;
        __syncthreads();
    }

    // ---- phase 1: step(t), game_numba.py:660-738 ---------------------------------------------------------------
    StepRegs s;            // SINGLE: the board's post-step record stays in registers for phase 2
    uint32_t term = 0;     // SINGLE: terminated after the step (of a live lane)
    bool live = false;
    int my_cnt = 0;        // wave-uniform: boards of this wave to re-deal
    // one iteration's inputs; the multi-pass form requests iteration r + 1 before it works on r (a wave that walks its
    // boards 64 at a time has one request wave in flight otherwise: 16 M boards 491 us per launch vs 332 + 115 us for
    // the separate step and prepare kernels)
    struct Inputs {
        Board board;
        uint32_t valid, t_before, action, perm[4], stale_merged[4];
        int32_t step, id_before;
        float score, stale_reward;
    };
    auto load_inputs = [&](int r, Inputs &x) __attribute__((always_inline)) {
        const int64_t i0 = wbase + (int64_t)r * 64 + lane;
        const int64_t i = i0 < M ? i0 : M - 1; // tail lanes shadow the last board: loads only, no stores
        x.board = load_board(st.board, i);
        x.valid = reinterpret_cast<const uint32_t *>(st.valid)[i];
        x.step = st.step[i];
        x.score = st.score[i];
        x.t_before = st.terminated[i];
        x.action = load_action(a.actions, a.action_dtype, i);
        x.perm[0] = x.perm[1] = x.perm[2] = x.perm[3] = 0;
        if (!FAST) load_perm_row(a.randperm_step, a.row_base_step + a.idx_base + i, x.perm);
        // what an invalid move leaves untouched (H3) is still part of the step's outcome
        x.stale_merged[0] = x.stale_merged[1] = x.stale_merged[2] = x.stale_merged[3] = 0;
        x.stale_reward = 0.0f;
        x.id_before = 0;
        if (a.out.merged) {
            const uint4 m = reinterpret_cast<const uint4 *>(st.merged)[i];
            x.stale_merged[0] = m.x; x.stale_merged[1] = m.y; x.stale_merged[2] = m.z; x.stale_merged[3] = m.w;
        }
        if (a.out.reward) x.stale_reward = st.reward[i];
        if (a.out.id) x.id_before = st.id[i];
    };
    Inputs in;
    if (wbase < M) load_inputs(0, in);
    for (int r = 0; r < R; ++r) {
        const int64_t i0 = wbase + (int64_t)r * 64 + lane;
        if (wbase + (int64_t)r * 64 >= M) break; // wave-uniform
        live = i0 < M;
        const int64_t i = live ? i0 : M - 1;
        Inputs nxt = in;
        if (!SINGLE && r + 1 < R && wbase + (int64_t)(r + 1) * 64 < M) load_inputs(r + 1, nxt);
        s.board = in.board;
        s.valid = in.valid;
        s.step = in.step;
        s.score = in.score;
        const uint32_t t_before = in.t_before, action = in.action;
        const uint32_t perm[4] = {in.perm[0], in.perm[1], in.perm[2], in.perm[3]};
        const uint32_t stale_merged[4] = {in.stale_merged[0], in.stale_merged[1], in.stale_merged[2], in.stale_merged[3]};
        const float stale_reward = in.stale_reward;
        const int32_t id_before = in.id_before;
        in = nxt;

        uint32_t invalid = 1;
        term = t_before;
        if (live) {
            // VecGame.step's copies (:672-673): always, valid move or not
            store16(st.prev_board, i, s.board.r);
            reinterpret_cast<uint32_t *>(st.prev_valid)[i] = s.valid;
            if (a.out.prev_state) store16(a.out.prev_state, i, s.board.r);
            if (a.out.prev_valid_actions) reinterpret_cast<uint32_t *>(a.out.prev_valid_actions)[i] = s.valid;
        }
        s.reward = stale_reward;
        s.merged[0] = stale_merged[0]; s.merged[1] = stale_merged[1]; s.merged[2] = stale_merged[2]; s.merged[3] = stale_merged[3];
        s.terminated = t_before;
        if (live && step_pre(s, action, a.reward_kind)) {
            if (FAST) {
                spawn_word(s.board, rocrand_seq_word(a.spawn_seed, a.spawn_counter_step, (uint64_t)(a.idx_base + i)), (uint32_t)thr);
            } else {
                spawn_table(s.board, perm, two_s, 1);
            }
            step_post(s);
            invalid = 0;
            term = s.terminated;
            store16(st.board, i, s.board.r);
            store16(st.merged, i, s.merged);
            reinterpret_cast<uint32_t *>(st.valid)[i] = s.valid;
            st.step[i] = s.step;
            st.score[i] = s.score;
            st.reward[i] = s.reward;
            st.terminated[i] = (uint8_t)s.terminated;
            st.invalid[i] = 0;
        } else if (live) {
            st.invalid[i] = 1; // :738
        }
        if (live) { // the reference's VecStepResult (:685-698) as of before the next prepare()
            if (a.out.state) store16(a.out.state, i, s.board.r);
            if (a.out.valid_actions) reinterpret_cast<uint32_t *>(a.out.valid_actions)[i] = s.valid;
            if (a.out.merged) store16(a.out.merged, i, s.merged);
            if (a.out.step) a.out.step[i] = s.step;
            if (a.out.reward) a.out.reward[i] = s.reward;
            if (a.out.score) a.out.score[i] = s.score;
            if (a.out.terminated) a.out.terminated[i] = (uint8_t)term;
            if (a.out.invalid) a.out.invalid[i] = (uint8_t)invalid;
            if (a.out.id) a.out.id[i] = id_before;
        }
        const bool redeal = live && term != 0;
        my_cnt += __popcll(__ballot(redeal));
        if (a.hist_counts && redeal) { // RunnerStats._update_count, runner.py:120-136
            const uint32_t mx = max_cell(s.board);
            if (mx < 20) atomicAdd(&hist[mx], 1);
        }
    }
    if (lane == 0) wave_cnt[wave] = my_cnt;
    __syncthreads(); // (also: this wave's stores above are complete before phase 2 re-reads them)
    int tot = 0, below = 0; // re-deals of this workgroup / of its lower waves
#pragma unroll
    for (int w = 0; w < kBlk / 64; ++w) {
        tot += wave_cnt[w];
        if (w < wave) below += wave_cnt[w];
    }
    if (threadIdx.x == 0)
        __hip_atomic_store(status + b, ((unsigned long long)epoch << 32) | (uint32_t)tot, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (a.hist_counts && threadIdx.x < 20 && hist[threadIdx.x]) atomicAdd(&a.hist_counts[threadIdx.x], hist[threadIdx.x]);

    // ---- phase 2a: prepare(t+1) without the numbering (game_numba.py:634-656), observations, next actions --------
    const bool want_obs = a.onehot || a.board_i64;
    const bool want_act = a.next_actions != nullptr;
    for (int r = 0; r < R; ++r) {
        const int64_t base = wbase + (int64_t)r * 64;
        if (base >= M) break;
        const int64_t i0 = base + lane;
        if (!SINGLE) {
            live = i0 < M;
            term = live ? st.terminated[i0] : 0u;
        }
        const int64_t i = live ? i0 : M - 1;
        if (live && term != 0) {
            clear_regs(s); // entry.fill(0), :639
            if (FAST) {
                uint32_t w0, w1;
                rocrand_deal_words(a.spawn_seed, a.spawn_counter_prepare, (uint64_t)(a.idx_base + i), w0, w1);
                spawn_word(s.board, w0, (uint32_t)thr);
                spawn_word(s.board, w1, (uint32_t)thr);
            } else {
                const uint32_t head[4] = {load_perm_head(a.randperm_prepare, a.row_base_prepare + a.idx_base + i), 0, 0, 0};
                deal_table(s.board, head, two_p); // :648-655 on the zeroed board
            }
            s.valid = valid_actions(s.board); // :656
            store16(st.board, i, s.board.r);
            store16(st.merged, i, s.merged);
            reinterpret_cast<uint32_t *>(st.valid)[i] = s.valid;
            st.step[i] = 0;
            st.score[i] = 0.0f;
            st.reward[i] = 0.0f;
            st.invalid[i] = 0;
            if (SINGLE) st.terminated[i] = 0; // otherwise phase 2c clears it after it has numbered the game
        } else if (!SINGLE && (want_obs || want_act)) {
            s.board = load_board(st.board, i);
            s.valid = reinterpret_cast<const uint32_t *>(st.valid)[i];
        }
        if (want_obs) {
            const int cnt = (int)((M - base) < 64 ? (M - base) : 64);
            emit_obs(s.board, base, cnt, lane, a.onehot, a.board_i64);
        }
        if (want_act) { // RandomPolicy.sample_actions on the post-prepare mask (policy/random.py:17-27)
            const uint32_t v = a.act_masked ? s.valid : 0x01010101u;
            const uint32_t word = rocrand_sampler_word(a.act_seed, a.act_counter, (uint64_t)(a.idx_base + i), a.act_masked);
            uint32_t cnt;
            const uint32_t act = pick_action(v, word, a.act_masked, cnt);
            if (live) {
                a.next_actions[i] = (uint8_t)act;
                if (a.next_logp) a.next_logp[i] = neg_log_count(cnt);
            }
        }
    }

    // ---- phase 2b: exclusive prefix of the workgroup counts (every lower workgroup has published, or will) --------
    // Relaxed device-scope accesses on both sides: the word itself (tag + count) is all that is consumed, so nothing
    // has to be released or acquired.  (First build: release store + acquire loads = an L2 write-back per workgroup
    // and an L1 invalidate per probe, and the <= 4 probes of a thread one after the other: 36 us per launch at
    // 262 144 boards against 8 + 14 us for the separate step and prepare kernels.)  All probes of a thread are issued
    // before the first is looked at; a word that is not there yet is polled.
    constexpr int kProbes = kMaxBlocks / kBlk;
    unsigned long long v[kProbes];
#pragma unroll
    for (int q = 0; q < kProbes; ++q) {
        const int j = (int)threadIdx.x + q * kBlk;
        v[q] = j < b ? __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                     : ((unsigned long long)epoch << 32);
    }
    int part = 0;
#pragma unroll
    for (int q = 0; q < kProbes; ++q) {
        const int j = (int)threadIdx.x + q * kBlk;
        while ((uint32_t)(v[q] >> 32) != epoch) { // only ever true for j < b
            __builtin_amdgcn_s_sleep(1);
            v[q] = __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        part += (int)(uint32_t)v[q];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    int excl = 0;
#pragma unroll
    for (int w = 0; w < kBlk / 64; ++w) excl += red[w];
    if (b == (int)gridDim.x - 1 && threadIdx.x == 0) { // the last workgroup has seen everybody's count
        *a.n_reset = excl + tot;
        *a.game_count = gc0 + excl + tot;
        *epoch_p = epoch;
    }

    // ---- phase 2c: ids in ascending slot order (:641-644) and the returned index list (:629,658) ------------------
    if (tot == 0) return; // workgroup-uniform
    int run = excl + below;
    for (int r = 0; r < R; ++r) {
        const int64_t base = wbase + (int64_t)r * 64;
        if (base >= M) break;
        const int64_t i0 = base + lane;
        if (!SINGLE) {
            live = i0 < M;
            term = live ? st.terminated[i0] : 0u;
        }
        const bool redeal = live && term != 0;
        const unsigned long long bal = __ballot(redeal);
        if (redeal) {
            const int rank = run + __popcll(bal & ((1ull << lane) - 1ull));
            st.id[i0] = gc0 + rank;
            if (a.reset_indices) a.reset_indices[rank] = i0;
            if (!SINGLE) st.terminated[i0] = 0;
        }
        run += __popcll(bal);
    }
}

} // namespace

extern "C" {

int64_t ml2048_step_prepare_scratch_bytes(void) { return (int64_t)(kMaxBlocks + 8) * 8; }

int ml2048_step_prepare(void *stream, const ml2048_step_prepare_args *a) {
    if (!a) return ml2048_internal_fail(ML2048_E_INVALID, "args is NULL");
    if (int rc = ml2048_internal_check_state(a->M, &a->st)) return rc;
    const bool fast = a->randperm_step == nullptr;
    if (!a->actions) return ml2048_internal_fail(ML2048_E_INVALID, "actions is NULL");
    if (a->action_dtype < 0 || a->action_dtype > 2) return ml2048_internal_fail(ML2048_E_INVALID, "action_dtype=%d", a->action_dtype);
    if (a->reward_kind < 0 || a->reward_kind > 3) return ml2048_internal_fail(ML2048_E_INVALID, "reward_kind=%d", a->reward_kind);
    if (!fast) {
        if (!a->randperm_prepare) return ml2048_internal_fail(ML2048_E_INVALID, "randperm_prepare is NULL");
        if ((a->two_mask_step | a->two_mask_prepare) >> 16)
            return ml2048_internal_fail(ML2048_E_INVALID, "two_mask_* have 16 bits (one per cell)");
        if (((uintptr_t)a->randperm_step | (uintptr_t)a->randperm_prepare) % 16)
            return ml2048_internal_fail(ML2048_E_ALIGN, "randperm_* must be 16-byte aligned");
        if (a->row_base_step < 0 || a->row_base_prepare < 0)
            return ml2048_internal_fail(ML2048_E_INVALID, "row_base_* must be >= 0");
    }
    if (a->idx_base < 0) return ml2048_internal_fail(ML2048_E_INVALID, "idx_base must be >= 0");
    if (!a->game_count || !a->n_reset || !a->scratch)
        return ml2048_internal_fail(ML2048_E_INVALID, "game_count, n_reset and scratch must be non-NULL");
    if ((uintptr_t)a->scratch % 8) return ml2048_internal_fail(ML2048_E_ALIGN, "scratch must be 8-byte aligned");
    if (((uintptr_t)a->out.state | (uintptr_t)a->out.merged | (uintptr_t)a->out.prev_state | (uintptr_t)a->onehot) % 16)
        return ml2048_internal_fail(ML2048_E_ALIGN, "out.state/merged/prev_state and onehot must be 16-byte aligned");
    if (((uintptr_t)a->out.valid_actions | (uintptr_t)a->out.prev_valid_actions | (uintptr_t)a->out.step |
         (uintptr_t)a->out.reward | (uintptr_t)a->out.score | (uintptr_t)a->out.id | (uintptr_t)a->next_logp |
         (uintptr_t)a->hist_counts) % 4)
        return ml2048_internal_fail(ML2048_E_ALIGN, "4-byte outputs must be 4-byte aligned");
    if ((uintptr_t)a->board_i64 % 8) return ml2048_internal_fail(ML2048_E_ALIGN, "board_i64 must be 8-byte aligned");
    if (a->next_logp && !a->next_actions) return ml2048_internal_fail(ML2048_E_INVALID, "next_logp without next_actions");
    const Geometry g = geometry(a->M);
    const uint64_t thr = ml2048_internal_two_threshold(a->two_prob);
    const dim3 grid((unsigned)g.nblk), block(kBlk);
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // ml2048_timer_arm: events attached to this very dispatch
    (void)ml2048_internal_take_timer(&ev0, &ev1);
#define ML_SP(F, S) hipExtLaunchKernelGGL((k_step_prepare<F, S>), grid, block, 0, s, ev0, ev1, 0, *a, g.R, thr)
    if (fast) {
        if (g.R == 1) ML_SP(true, true);
        else ML_SP(true, false);
    } else {
        if (g.R == 1) ML_SP(false, true);
        else ML_SP(false, false);
    }
#undef ML_SP
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ml2048_internal_fail((int)e, "ml2048_step_prepare: %s", hipGetErrorString(e));
    return 0;
}

} // extern "C"
