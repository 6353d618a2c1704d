// ml2048_hip.hip - gfx950 (MI355X) kernels + the C ABI of include/ml2048_hip.h.
//
// Mapping: one board per lane (64 boards per wavefront).  A board is 16 bytes, so a wave's
// load/store of any 16-byte field is one fully coalesced 1 KiB global_load/store_dwordx4 and the
// 4-byte fields are one 256-byte dword access; all game logic is register-resident byte-parallel
// integer work (board_ops.hpp).  The work is HBM-bound integer/byte work: no MFMA anywhere.
// The one-hot encoder alone uses the wave-per-board mapping: the fp32 one-hot of one board is
// exactly 64 lanes x 16 B, one dwordx4 store per lane per board.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <rocrand/rocrand_kernel.h>

#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/ml2048_hip.h"
#include "board_ops.hpp"
#include "device_helpers.hpp"

using namespace ml2048;
using namespace ml2048::dev;

// The plain (no optional outputs) rollout instantiations are compiled in three further passes over this same file
// (build.py: -DML2048_ROLLOUT_PLAIN_TU=<launcher name>; a scheduling strategy is a per-file flag, hence the passes; the
// first pass picks the launcher and instantiates only the EXTRAS kernels itself):
//   _ilp   small M (one wave per SIMD): the lone-wave forms (ML2048_LONE_WAVE) + -mllvm -amdgpu-sched-strategy=max-ilp;
//   _lone  the same forms with the default strategy;
//   _dflt  large M: the general forms, default strategy.
// Same-box A/B of max-ilp: round 1's kernel at C2 105.7 -> 101.0 us (fewer s_nop between v_cmp and v_cndmask,
// independent chains interleaved); 1 M boards 852 -> 880 us (with several waves per SIMD the hardware interleaves
// already and the longer live ranges cost); 9-20 % slower on the trajectory-writing instantiations.  With round 2's
// shorter lazy loop (dot-product spawn, position-mask compaction) the default strategy is the faster one for UNMASKED
// table-mode rollouts (70.6 vs 72.1 us at C2, 106.0 vs 106.9 at 131 072 boards; max-memory-clause / iterative-maxocc the
// same, iterative-minreg 73.2), while the masked loop (102.2 vs 103.8 us) and the rocRAND mode (97.4 vs 98.0) still
// prefer max-ilp - both loops live in one kernel function, so the choice is made per launch: _lone for unmasked table
// mode, _ilp otherwise (tools/debug/ab_multi.sh).
struct ml2048_plain_launch {
    dim3 grid, block;
    hipStream_t stream;
    hipEvent_t ev_start, ev_stop;
    ml2048_state st;
    int64_t M, idx_base;
    int32_t N;
    const uint8_t *perm_epochs;
    const uint32_t *two_masks;
    const int32_t *schedule;
    int32_t n_epochs;
    int masked;
    uint64_t act_seed, act_counter0;
    uint64_t *reset_flags;
    int64_t flag_words;
    int32_t *last_reset;
    uint64_t spawn_seed, spawn_counter0, two_threshold;
    const uint4 *up_src;
    uint4 *up_dst;
    int64_t up_n16;
    unsigned up_blocks; // 0 = no chained upload, 1 = by an extra wave per workgroup, 2 = by each workgroup's first wave
    int64_t up_tab16;
    const uint16_t *row_lut; // the 65 536 line look-ups of ml2048_row_lut_fill, or NULL: the arithmetic move
    int board_waves;         // board-carrying waves per workgroup (the plain kernels' workgroup = one CU's share of the boards)
};
__attribute__((visibility("hidden"))) int ml2048_internal_launch_rollout_plain_ilp(int reward_kind, bool fast, bool inv_tables,
                                                                                   const ml2048_plain_launch &a);
__attribute__((visibility("hidden"))) int ml2048_internal_launch_rollout_plain_lone(int reward_kind, bool fast, bool inv_tables,
                                                                                    const ml2048_plain_launch &a);
__attribute__((visibility("hidden"))) int ml2048_internal_launch_rollout_plain_dflt(int reward_kind, bool fast, bool inv_tables,
                                                                                    const ml2048_plain_launch &a);

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

constexpr int kBlock = 256;      // threads per workgroup for the per-board kernels (4 waves)
// The plain rollout kernels hold the 128 KiB line table in LDS, i.e. ONE workgroup per compute unit: the workgroup is
// sized at launch to the CU's whole share of the boards, up to 16 waves (4 per SIMD).
#ifdef ML2048_LONE_WAVE
// the small-M passes serve at most 8 board waves per workgroup (+ the upload wave; launch_rollout): the smaller bound lets
// the compiler use 168 VGPRs instead of 128 (the rank / maxcell reward kinds spilled 8-14 VGPRs under the 1 024 bound)
constexpr int kRollMaxBlock = 576;
#else
constexpr int kRollMaxBlock = 1024;
#endif
constexpr int kRollMaxWaves = 16;
// rollout steps whose re-deal flags are parked between flushes: in LDS (128 steps), or - ML2048_LONE_WAVE, the
// small-M pass of the plain kernels - in a register pair, one step per lane (64 steps)
constexpr int kFlagChunk = 64; // (128 in LDS for the large-M pass until round 3: the LDS now holds the move's line table)
#ifndef ML2048_STEP_UNROLL
#define ML2048_STEP_UNROLL 8
#endif
constexpr int kStepUnroll = ML2048_STEP_UNROLL; // steps per iteration of the plain rollout loop (A/B: -DML2048_STEP_UNROLL=n)
constexpr int kPrepBlock = 1024; // boards per workgroup in the prepare scan

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

int check_state(int64_t M, const ml2048_state *st) {
    if (M <= 0) return fail(ML2048_E_INVALID, "M=%lld must be positive", (long long)M);
    if (M > (int64_t)1 << 31) return fail(ML2048_E_INVALID, "M=%lld too large", (long long)M);
    if (!st) return fail(ML2048_E_INVALID, "state is NULL");
    const void *p[] = {st->board, st->merged, st->valid, st->step, st->score, st->reward,
                       st->terminated, st->invalid, st->id, st->prev_board, st->prev_valid};
    const char *n[] = {"board", "merged", "valid", "step", "score", "reward",
                       "terminated", "invalid", "id", "prev_board", "prev_valid"};
    const unsigned al[] = {16, 16, 4, 4, 4, 4, 1, 1, 4, 16, 4};
    for (int i = 0; i < 11; ++i) {
        if (!p[i]) return fail(ML2048_E_INVALID, "state.%s is NULL", n[i]);
        if ((uintptr_t)p[i] % al[i]) return fail(ML2048_E_ALIGN, "state.%s must be %u-byte aligned", n[i], al[i]);
    }
    return 0;
}

// ---- reset -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_reset(ml2048_state st, int64_t M) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    const uint32_t z[4] = {0, 0, 0, 0};
    store16(st.board, i, z);
    store16(st.merged, i, z);
    store16(st.prev_board, i, z);
    reinterpret_cast<uint32_t *>(st.valid)[i] = 0;
    reinterpret_cast<uint32_t *>(st.prev_valid)[i] = 0;
    st.step[i] = 0;
    st.score[i] = 0.0f;
    st.reward[i] = 0.0f;
    st.id[i] = 0;
    st.terminated[i] = 1; // game_numba.py:614 - the first prepare() deals every game
    st.invalid[i] = 0;
}

// ---- step --------------------------------------------------------------------------------------
// Algorithmic traffic per board: read board 16 + valid 4 + step 4 + score 4 + action 1..8;
// write prev_board 16 + prev_valid 4 always; on a valid move board 16 + merged 16 + valid 4 +
// step 4 + score 4 + reward 4 + terminated 1 + invalid 1 (=99 B with a 1-byte action).
template <int KIND, bool FAST>
__global__ __launch_bounds__(kBlock) void k_step(ml2048_state st, int64_t M, int64_t idx_base,
                                                 const void *__restrict__ actions, int action_dtype,
                                                 double two_prob, int64_t row_base,
                                                 const uint8_t *__restrict__ randperm,
                                                 const float *__restrict__ randfloat, uint64_t seed,
                                                 uint64_t counter, uint64_t thr) {
    const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i0 < M;
    const int64_t i = live ? i0 : M - 1; // tail lanes shadow the last board: loads only, no stores

    // issue every load up front; the 2/4 mask (H1: randfloat is indexed by the CELL, so only
    // randfloat[0..15] matter) is formed by a wave ballot once the loads are in flight
    float rf = 0.0f;
    if (!FAST) rf = randfloat[threadIdx.x & 15];
    StepRegs s;
    s.board = load_board(st.board, i);
    s.valid = reinterpret_cast<const uint32_t *>(st.valid)[i];
    s.step = st.step[i];
    s.score = st.score[i];
    const uint32_t action = load_action(actions, action_dtype, i);
    uint32_t perm[4] = {0, 0, 0, 0};
    if (!FAST) load_perm_row(randperm, row_base + idx_base + i, perm);
    uint32_t two_mask = 0;
    if (!FAST) two_mask = (uint32_t)(__ballot((double)rf < two_prob) & 0xFFFFull);
    if (!live) return;

    // VecGame.step's copies (game_numba.py:672-673): always, valid move or not
    store16(st.prev_board, i, s.board.r);
    reinterpret_cast<uint32_t *>(st.prev_valid)[i] = s.valid;

    if (!step_pre(s, action, KIND)) {
        st.invalid[i] = 1; // :738
        return;
    }
    if (FAST) {
        spawn_word(s.board, rocrand_seq_word(seed, counter, (uint64_t)(idx_base + i)), (uint32_t)thr);
    } else {
        spawn_table(s.board, perm, two_mask, 1);
    }
    step_post(s);

    store16(st.board, i, s.board.r);
    store16(st.merged, i, s.merged);
    reinterpret_cast<uint32_t *>(st.valid)[i] = s.valid;
    st.step[i] = s.step;
    st.score[i] = s.score;
    st.reward[i] = s.reward;
    st.terminated[i] = (uint8_t)s.terminated;
    st.invalid[i] = 0;
}

// ---- step, wave-per-board variant ----------------------------------------------------------------------
// The mapping BASELINE.json's north_star sketches: one wavefront per board, the board's 16 bytes
// staged in LDS, 4 lanes each owning one line of the chosen direction, 16 lanes probing the
// permutation row for the spawn, 8 lanes evaluating the 8 line-movability tests.  Same semantics and
// bit-identical results as k_step; kept as a measured alternative: ~120 wave
// instructions per board against ~10 per board for the lane-per-board kernel.
__device__ __forceinline__ void line_movable_word(uint32_t x, bool &fwd, bool &bwd) {
    const uint32_t n = x >> 8;                       // byte j = cell j+1
    const uint32_t nzx = nz_bytes(x), zx = nzx ^ 0x80808080u;
    const uint32_t eq = z_bytes(x ^ n);
    fwd = ((nzx >> 8) & (zx | eq)) != 0;                                   // a cell can move toward 0
    bwd = (nzx & (((zx >> 8) | 0x80000000u) | eq) & 0x00808080u) != 0;     // ... toward 3
}

__global__ __launch_bounds__(kBlock) void k_step_wave(ml2048_state st, int64_t M, int64_t idx_base,
                                                      const void *__restrict__ actions, int action_dtype,
                                                      int reward_kind, double two_prob, int64_t row_base,
                                                      const uint8_t *__restrict__ randperm,
                                                      const float *__restrict__ randfloat) {
    __shared__ uint32_t lds[kBlock / 64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *B = reinterpret_cast<uint8_t *>(lds[wave]);
    const uint32_t two_mask = two_mask_from_table(randfloat, two_prob);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t b = (int64_t)blockIdx.x * (kBlock / 64) + wave; b < M; b += n_waves) {
        // ---- stage the board in LDS: lanes 0..3 carry one row word each (one 16-byte request)
        uint32_t roww = 0;
        if (lane < 4) {
            roww = reinterpret_cast<const uint32_t *>(st.board)[b * 4 + lane];
            lds[wave][lane] = roww;
            reinterpret_cast<uint32_t *>(st.prev_board)[b * 4 + lane] = roww; // game_numba.py:672
        }
        const uint32_t valid = reinterpret_cast<const uint32_t *>(st.valid)[b]; // wave-uniform loads
        const uint32_t action = load_action(actions, action_dtype, b);
        if (lane == 0) reinterpret_cast<uint32_t *>(st.prev_valid)[b] = valid;  // :673
        const bool ok = action < 4u && ((valid >> (8u * (action & 3u))) & 0xFFu) != 0u;
        if (!ok) {
            if (lane == 0) st.invalid[b] = 1; // :738
            continue;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); // lgkmcnt(0): the row words are in LDS (same wave: no barrier)
        Board prev;
        prev.r[0] = lds[wave][0]; prev.r[1] = lds[wave][1]; prev.r[2] = lds[wave][2]; prev.r[3] = lds[wave][3];

        // ---- move: lane j < 4 owns line j of the chosen direction (offset, stride) :93-122
        uint32_t m = 0;
        if (lane < 4) {
            int off, stride;
            switch (action) {
            case 0: off = 4 * lane; stride = 1; break;
            case 1: off = 4 * lane + 3; stride = -1; break;
            case 2: off = lane; stride = 4; break;
            default: off = 12 + lane; stride = -4; break;
            }
            uint32_t x = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) x |= (uint32_t)B[off + r * stride] << (8 * r);
            x = push_line(x, m);
#pragma unroll
            for (int r = 0; r < 4; ++r) B[off + r * stride] = (uint8_t)(x >> (8 * r));
        }
        __builtin_amdgcn_wave_barrier();
        // the <= 8 merge exponents, gathered from the four line owners
        uint32_t mexp_first = 0, mexp_second = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t mj = __builtin_amdgcn_readlane(m, j);
            mexp_first |= (mj & 0xFFu) << (8 * j);
            mexp_second |= (mj >> 8) << (8 * j);
        }
        const MergeStats ms = merge_stats(mexp_first, mexp_second);
        __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier();
        Board cur;
        cur.r[0] = lds[wave][0]; cur.r[1] = lds[wave][1]; cur.r[2] = lds[wave][2]; cur.r[3] = lds[wave][3];
        const float reward = (float)reward_value(reward_kind, cur, prev, ms);

        // ---- spawn: lanes 0..15 probe the permutation row in order (:198-210)
        const uint4 pv = reinterpret_cast<const uint4 *>(randperm)[(row_base + idx_base + b) & (ML2048_RAND_SIZE - 1)];
        const int q = (lane >> 2) & 3;
        const uint32_t pw = q == 0 ? pv.x : q == 1 ? pv.y : q == 2 ? pv.z : pv.w;
        const uint32_t c = ubfe(pw, 8u * (lane & 3), 4);
        const bool empty_here = lane < 16 && B[c] == 0;
        const unsigned long long hits = __ballot(empty_here);
        if (hits) {
            const int first = __builtin_ctzll(hits);
            const uint32_t cell = __builtin_amdgcn_readlane(c, first);
            if (lane == 0) B[cell] = (uint8_t)(2u - ubfe(two_mask, cell, 1));
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier();

        // ---- mask: lanes 0..3 rows, lanes 4..7 columns (:259-289)
        bool fwd = false, bwd = false;
        if (lane < 8) {
            const int off = lane < 4 ? 4 * lane : lane - 4, stride = lane < 4 ? 1 : 4;
            uint32_t x = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) x |= (uint32_t)B[off + r * stride] << (8 * r);
            line_movable_word(x, fwd, bwd);
        }
        const unsigned long long fb = __ballot(fwd), bb = __ballot(bwd);
        const uint32_t nv = ((fb & 0x0Full) ? 1u : 0u) | ((bb & 0x0Full) ? 0x100u : 0u) |
                            ((fb & 0xF0ull) ? 0x10000u : 0u) | ((bb & 0xF0ull) ? 0x1000000u : 0u);

        // ---- write back
        if (lane < 4) {
            reinterpret_cast<uint32_t *>(st.board)[b * 4 + lane] = lds[wave][lane];
            const uint32_t mw = lane == 0 ? ms.merged[0] : lane == 1 ? ms.merged[1] : lane == 2 ? ms.merged[2] : ms.merged[3];
            reinterpret_cast<uint32_t *>(st.merged)[b * 4 + lane] = mw;
        }
        if (lane == 0) {
            reinterpret_cast<uint32_t *>(st.valid)[b] = nv;
            st.step[b] = st.step[b] + 1;
            st.score[b] = ml2048::add_score(st.score[b], ms.sum_value);
            st.reward[b] = reward;
            st.terminated[b] = nv == 0u;
            st.invalid[b] = 0;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); // LDS reads done before the next board overwrites the slot
    }
}

// ---- prepare: count -> scan -> deal --------------------------------------------------------------
// scratch layout: [0, nblk) per-block counts -> exclusive prefix; [nblk] total; [nblk+1] snapshot
// of *game_count taken before any block of the deal kernel can advance it.
__global__ __launch_bounds__(kPrepBlock) void k_prep_count(const uint8_t *__restrict__ terminated,
                                                           int64_t M, int32_t *__restrict__ scratch) {
    __shared__ int wave_cnt[kPrepBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kPrepBlock + threadIdx.x;
    const bool t = i < M && terminated[i] != 0;
    const unsigned long long b = __ballot(t);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < kPrepBlock / 64; ++w) tot += wave_cnt[w];
        scratch[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(1024) void k_prep_scan(int32_t *__restrict__ scratch, int nblk,
                                                    const int32_t *__restrict__ game_count) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int j = base + threadIdx.x;
        const int v = j < nblk ? scratch[j] : 0;
        int incl = v; // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int n = __shfl_up(incl, off, 64);
            if (lane >= off) incl += n;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wave_base = 0;
        for (int w = 0; w < wave; ++w) wave_base += wave_tot[w];
        const int carry = carry_s;
        if (j < nblk) scratch[j] = carry + wave_base + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scratch[nblk] = carry_s;
        scratch[nblk + 1] = *game_count;
    }
}

template <bool FAST>
__global__ __launch_bounds__(kPrepBlock) void k_prep_deal(ml2048_state st, int64_t M, int64_t idx_base,
                                                          const uint8_t *__restrict__ randperm,
                                                          const float *__restrict__ randfloat,
                                                          double two_prob, int64_t row_base,
                                                          uint64_t seed, uint64_t counter, uint64_t thr,
                                                          const int32_t *__restrict__ scratch, int nblk,
                                                          int32_t *__restrict__ game_count,
                                                          int32_t *__restrict__ n_reset,
                                                          int64_t *__restrict__ reset_indices) {
    __shared__ int wave_cnt[kPrepBlock / 64];
    uint32_t two_mask = 0;
    if (!FAST) two_mask = two_mask_from_table(randfloat, two_prob);
    const int total = scratch[nblk];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *n_reset = total;
        *game_count = scratch[nblk + 1] + total; // only this thread writes it; readers use the snapshot
    }
    if (total == 0) return;
    const int64_t i = (int64_t)blockIdx.x * kPrepBlock + threadIdx.x;
    const bool t = i < M && st.terminated[i] != 0;
    const unsigned long long b = __ballot(t);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    if (!t) return;
    int rank = scratch[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += wave_cnt[w];

    StepRegs s;
    clear_regs(s); // entry.fill(0), game_numba.py:639
    if (FAST) {
        uint32_t w0, w1;
        rocrand_deal_words(seed, counter, (uint64_t)(idx_base + i), w0, w1);
        spawn_word(s.board, w0, (uint32_t)thr);
        spawn_word(s.board, w1, (uint32_t)thr);
    } else {
        uint32_t perm[4];
        load_perm_row(randperm, row_base + idx_base + i, perm);
        deal_table(s.board, perm, two_mask); // :648-655 on the zeroed board
    }
    s.valid = valid_actions(s.board); // :656

    store16(st.board, i, s.board.r);
    store16(st.merged, i, s.merged);
    reinterpret_cast<uint32_t *>(st.valid)[i] = s.valid;
    st.step[i] = 0;
    st.score[i] = 0.0f;
    st.reward[i] = 0.0f;
    st.terminated[i] = 0;
    st.invalid[i] = 0;
    st.id[i] = scratch[nblk + 1] + rank; // :641-644, ascending slot order
    if (reset_indices) reset_indices[rank] = i;
}

// The same prepare in ONE launch for any M up to kPrepFusedBlocks workgroups: a single-pass scan with
// decoupled look-back (Merrill & Garland).  Workgroup b counts its terminated boards, publishes the count
// in status[b] (AGGREGATE), walks back over its predecessors' status words adding aggregates until it meets
// an inclusive PREFIX, publishes its own inclusive prefix and deals.  Status words carry the launch's epoch
// (a process-wide counter), so they never need resetting between calls: a word from an older launch reads as
// "not published yet".  Workgroups are dispatched in index order and a workgroup only ever waits for lower
// indices, so the walk cannot deadlock.  *game_count is read by every workgroup BEFORE it publishes and
// written by the last workgroup AFTER its walk has seen all of them.  Three launches (count, scan, deal)
// cost ~22 us of launch gaps per env step at M = 65 536; this one ~7 us.  Used up to 64 workgroups (one
// look-back window, M <= 65 536): with more, all workgroups start at once, every one of them walks back
// window by window to the few prefixes that exist yet (O(blocks^2 / 64) probes; measured 297 us vs 122 us per
// env step at 4 M boards), and the three bandwidth-bound launches win again.
// The status accesses are RELAXED device-scope atomics: the word itself (tag, kind, count) is all that a reader
// consumes, and a release store / acquire load here means an L2 write-back / an L1 invalidate per access
// (step_prepare.hip measured that at 36 vs 15 us per launch).
constexpr int kPrepFusedBlocks = 64;
constexpr unsigned long long kStAggregate = 1ull << 32, kStPrefix = 2ull << 32;

template <bool FAST>
__global__ __launch_bounds__(kPrepBlock) void k_prep_fused(ml2048_state st, int64_t M, int64_t idx_base,
                                                           const uint8_t *__restrict__ randperm,
                                                           const float *__restrict__ randfloat, double two_prob,
                                                           int64_t row_base, uint64_t seed, uint64_t counter,
                                                           uint64_t thr, unsigned long long *__restrict__ status,
                                                           uint32_t epoch, int32_t *__restrict__ game_count,
                                                           int32_t *__restrict__ n_reset,
                                                           int64_t *__restrict__ reset_indices) {
    __shared__ int wave_cnt[kPrepBlock / 64];
    __shared__ int s_excl, s_gc0;
    uint32_t two_mask = 0; // a wave-wide ballot: every lane takes part, before anybody leaves
    if (!FAST) two_mask = two_mask_from_table(randfloat, two_prob);
    const int64_t i = (int64_t)blockIdx.x * kPrepBlock + threadIdx.x;
    const bool t = i < M && st.terminated[i] != 0;
    const unsigned long long bal = __ballot(t);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    if (wave == 0) { // the look-back is done by one wavefront: 64 predecessors per probe
        int tot = 0;
        for (int w = 0; w < kPrepBlock / 64; ++w) tot += wave_cnt[w];
        const int gc0 = *game_count; // before this workgroup publishes anything:
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // (the load has returned before the relaxed store below issues)
        const unsigned long long tag = (unsigned long long)epoch << 34;
        const int b = (int)blockIdx.x;
        int excl = 0;
        if (b == 0) {
            if (lane == 0) __hip_atomic_store(status, tag | kStPrefix | (uint32_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(status + b, tag | kStAggregate | (uint32_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int hi = b - 1; hi >= 0;) { // window = predecessors hi, hi-1, ..., hi-63 (lane 0 = nearest)
                const int j = hi - lane;
                unsigned long long v = 0;
                bool ready = true;
                if (j >= 0) {
                    v = __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ready = (v >> 34) == epoch && (v & (3ull << 32)) != 0;
                }
                const unsigned long long prefix_lanes = __ballot(j >= 0 && ready && (v & kStPrefix));
                // usable lanes: the contiguous run of published words from the nearest predecessor on
                const unsigned long long not_ready = __ballot(!ready);
                const int run = not_ready ? __builtin_ctzll(not_ready) : 64; // lanes [0, run) are published
                const int first_prefix = prefix_lanes ? __builtin_ctzll(prefix_lanes) : 64;
                if (first_prefix < run) { // an inclusive prefix within reach: sum up to and including it, done
                    int val = lane <= first_prefix ? (int)(uint32_t)v : 0;
                    for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
                    excl += val;
                    break;
                }
                if (run == 0) { // the nearest predecessor has not published yet
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                int val = lane < run && j >= 0 ? (int)(uint32_t)v : 0; // aggregates only: take them and move on
                for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
                excl += val;
                hi -= run;
            }
            if (lane == 0) __hip_atomic_store(status + b, tag | kStPrefix | (uint32_t)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl = excl;
            s_gc0 = gc0;
            if (b == (int)gridDim.x - 1) {
                *n_reset = excl + tot;
                *game_count = gc0 + excl + tot;
            }
        }
    }
    __syncthreads();
    if (!t) return;
    int rank = s_excl + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += wave_cnt[w];

    StepRegs s;
    clear_regs(s); // entry.fill(0), game_numba.py:639
    if (FAST) {
        uint32_t w0, w1;
        rocrand_deal_words(seed, counter, (uint64_t)(idx_base + i), w0, w1);
        spawn_word(s.board, w0, (uint32_t)thr);
        spawn_word(s.board, w1, (uint32_t)thr);
    } else {
        uint32_t perm[4];
        load_perm_row(randperm, row_base + idx_base + i, perm);
        deal_table(s.board, perm, two_mask); // :648-655 on the zeroed board
    }
    s.valid = valid_actions(s.board); // :656
    store16(st.board, i, s.board.r);
    store16(st.merged, i, s.merged);
    reinterpret_cast<uint32_t *>(st.valid)[i] = s.valid;
    st.step[i] = 0;
    st.score[i] = 0.0f;
    st.reward[i] = 0.0f;
    st.terminated[i] = 0;
    st.invalid[i] = 0;
    st.id[i] = s_gc0 + rank; // :641-644, ascending slot order
    if (reset_indices) reset_indices[rank] = i;
}

// Small M (<= 16 boards per thread of one 1024-thread workgroup): the whole prepare in ONE launch.
// Thread t owns the contiguous boards [t*B, (t+1)*B), so ranks are still in ascending slot order.
template <bool FAST>
__global__ __launch_bounds__(1024) void k_prep_small(ml2048_state st, int64_t M, int B, int64_t idx_base,
                                                     const uint8_t *__restrict__ randperm,
                                                     const float *__restrict__ randfloat, double two_prob,
                                                     int64_t row_base, uint64_t seed, uint64_t counter, uint64_t thr,
                                                     int32_t *__restrict__ game_count, int32_t *__restrict__ n_reset,
                                                     int64_t *__restrict__ reset_indices) {
    __shared__ int wave_tot[16];
    uint32_t two_mask = 0;
    if (!FAST) two_mask = two_mask_from_table(randfloat, two_prob);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t first = (int64_t)threadIdx.x * B;
    uint32_t mine = 0; // bit j <=> board first+j is terminated
    for (int j = 0; j < B; ++j) {
        const int64_t i = first + j;
        if (i < M && st.terminated[i] != 0) mine |= 1u << j;
    }
    const int cnt = __popc(mine);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = incl - cnt, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) base += wave_tot[w];
        total += wave_tot[w];
    }
    const int32_t gc0 = *game_count; // every thread reads it before the barrier below ...
    __syncthreads();
    if (threadIdx.x == 0) {          // ... and only then is it advanced
        *n_reset = total;
        *game_count = gc0 + total;
    }
    int rank = base;
    for (int j = 0; j < B; ++j) {
        if (!((mine >> j) & 1u)) continue;
        const int64_t i = first + j;
        StepRegs s;
        clear_regs(s);
        if (FAST) {
            uint32_t w0, w1;
            rocrand_deal_words(seed, counter, (uint64_t)(idx_base + i), w0, w1);
            spawn_word(s.board, w0, (uint32_t)thr);
            spawn_word(s.board, w1, (uint32_t)thr);
        } else {
            uint32_t perm[4];
            load_perm_row(randperm, row_base + idx_base + i, perm);
            deal_table(s.board, perm, two_mask);
        }
        s.valid = valid_actions(s.board);
        store16(st.board, i, s.board.r);
        store16(st.merged, i, s.merged);
        reinterpret_cast<uint32_t *>(st.valid)[i] = s.valid;
        st.step[i] = 0;
        st.score[i] = 0.0f;
        st.reward[i] = 0.0f;
        st.terminated[i] = 0;
        st.invalid[i] = 0;
        st.id[i] = gc0 + rank;
        if (reset_indices) reset_indices[rank] = i;
        ++rank;
    }
}

// ---- sampler -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sample(const uint8_t *__restrict__ valid, int64_t M,
                                                   int64_t idx_base, uint64_t seed, uint64_t counter,
                                                   int masked, uint8_t *__restrict__ actions,
                                                   float *__restrict__ logp) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    const uint32_t v = masked ? reinterpret_cast<const uint32_t *>(valid)[i] : 0x01010101u;
    const uint32_t word = rocrand_sampler_word(seed, counter, (uint64_t)(idx_base + i), masked);
    uint32_t cnt;
    const uint32_t a = pick_action(v, word, masked, cnt);
    actions[i] = (uint8_t)a;
    if (logp) logp[i] = neg_log_count(cnt);
}

__global__ __launch_bounds__(kBlock) void k_debug_philox(int64_t n, int64_t idx_base, uint64_t seed,
                                                         uint64_t counter, uint32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t w[4];
    rocrand_words(seed, counter, (uint64_t)(idx_base + i), w);
    reinterpret_cast<uint4 *>(out)[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- one-hot observation encode --------------------------------------------------------------------
// Wave-per-board on the store side: each wave loads 64 boards (one dwordx4 per lane), then for
// each of them every lane emits the 4 consecutive floats [4*lane, 4*lane+3] of that board's
// (16,4,4) one-hot = class k = lane>>2, row r = lane&3, columns 0..3: one coalesced 1 KiB
// global_store_dwordx4 per board.  1 KiB written per 16 B read.
__global__ __launch_bounds__(kBlock) void k_onehot(const uint8_t *__restrict__ board, int64_t M,
                                                   float *__restrict__ onehot,
                                                   int64_t *__restrict__ board_i64,
                                                   int32_t *__restrict__ bad_count) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    const uint32_t k = (uint32_t)lane >> 2; // class
    const int r = lane & 3;                 // board row
    const uint32_t kkkk = k * 0x01010101u;
    int bad = 0;
    for (int64_t base = wave_global * 64; base < M; base += n_waves * 64) {
        const int64_t mine = base + lane;
        Board b;
        b.r[0] = b.r[1] = b.r[2] = b.r[3] = 0;
        if (mine < M) b = load_board(board, mine);
        if (bad_count) {
#pragma unroll
            for (int q = 0; q < 4; ++q) // bytes >= 16  <=>  (x & 0xF0) != 0 for exponents < 0x80
                bad += __popc(nz_bytes(b.r[q] & 0xF0F0F0F0u));
        }
        const int cnt = (int)((M - base) < 64 ? (M - base) : 64);
        for (int j = 0; j < cnt; ++j) {
            // broadcast board j of this wave (wave-uniform j -> v_readlane_b32)
            const uint32_t r0 = __builtin_amdgcn_readlane(b.r[0], j);
            const uint32_t r1 = __builtin_amdgcn_readlane(b.r[1], j);
            const uint32_t r2 = __builtin_amdgcn_readlane(b.r[2], j);
            const uint32_t r3 = __builtin_amdgcn_readlane(b.r[3], j);
            const uint32_t row = r == 0 ? r0 : r == 1 ? r1 : r == 2 ? r2 : r3;
            const uint32_t eq = z_bytes(row ^ kkkk); // 0x80 where cell == k
            float4 o;
            o.x = (eq & 0x00000080u) ? 1.0f : 0.0f;
            o.y = (eq & 0x00008000u) ? 1.0f : 0.0f;
            o.z = (eq & 0x00800000u) ? 1.0f : 0.0f;
            o.w = (eq & 0x80000000u) ? 1.0f : 0.0f;
            reinterpret_cast<float4 *>(onehot)[(base + j) * 64 + lane] = o;
        }
        if (board_i64) {
            // 64 boards x 16 cells = 1024 int64, written 64 per instruction (512 B, coalesced)
#pragma unroll 4
            for (int t = 0; t < 16; ++t) {
                const int e = t * 64 + lane;   // element within the wave's 64 boards
                const int src = e >> 4, cell = e & 15;
                uint32_t w = __shfl(b.r[0], src, 64);
                const uint32_t w1 = __shfl(b.r[1], src, 64);
                const uint32_t w2 = __shfl(b.r[2], src, 64);
                const uint32_t w3 = __shfl(b.r[3], src, 64);
                const int q = cell >> 2;
                w = q == 0 ? w : q == 1 ? w1 : q == 2 ? w2 : w3;
                const int64_t v = (w >> (8 * (cell & 3))) & 0xFFu;
                if (base + src < M) board_i64[base * 16 + e] = v;
            }
        }
    }
    if (bad_count) {
        const uint64_t tot = wave_sum_u64((uint64_t)bad);
        if (lane == 0 && tot) atomicAdd(bad_count, (int)tot);
    }
}

// ---- max-tile histogram ----------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_hist(const uint8_t *__restrict__ board,
                                                 const uint8_t *__restrict__ terminated, int64_t M,
                                                 int32_t *__restrict__ counts) {
    __shared__ int h[20];
    if (threadIdx.x < 20) h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += (int64_t)gridDim.x * kBlock) {
        if (terminated && terminated[i] == 0) continue;
        const Board b = load_board(board, i);
        const uint32_t mx = max_cell(b);
        if (mx < 20) atomicAdd(&h[mx], 1);
    }
    __syncthreads();
    if (threadIdx.x < 20 && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}

// ---- AoS <-> SoA -----------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pack(ml2048_state st, int64_t M, uint4 *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    uint4 q0, q3;
    q0.x = (uint32_t)st.id[i];
    q0.y = (uint32_t)st.step[i];
    q0.z = __float_as_uint(st.score[i]);
    q0.w = __float_as_uint(st.reward[i]);
    q3.x = reinterpret_cast<const uint32_t *>(st.valid)[i];
    q3.y = (uint32_t)st.terminated[i] | ((uint32_t)st.invalid[i] << 8);
    q3.z = 0;
    q3.w = 0;
    rec[i * 4 + 0] = q0;
    rec[i * 4 + 1] = reinterpret_cast<const uint4 *>(st.board)[i];
    rec[i * 4 + 2] = reinterpret_cast<const uint4 *>(st.merged)[i];
    rec[i * 4 + 3] = q3;
}

__global__ __launch_bounds__(kBlock) void k_unpack(ml2048_state st, int64_t M, const uint4 *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    const uint4 q0 = rec[i * 4 + 0], q3 = rec[i * 4 + 3];
    st.id[i] = (int32_t)q0.x;
    st.step[i] = (int32_t)q0.y;
    st.score[i] = __uint_as_float(q0.z);
    st.reward[i] = __uint_as_float(q0.w);
    reinterpret_cast<uint4 *>(st.board)[i] = rec[i * 4 + 1];
    reinterpret_cast<uint4 *>(st.merged)[i] = rec[i * 4 + 2];
    reinterpret_cast<uint32_t *>(st.valid)[i] = q3.x;
    st.terminated[i] = (uint8_t)(q3.y & 0xFFu);
    st.invalid[i] = (uint8_t)((q3.y >> 8) & 0xFFu);
}

// ---- digest ----------------------------------------------------------------------------------------
__device__ __forceinline__ void block_add_u64(uint64_t v, uint64_t *out) {
    __shared__ uint64_t part[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum_u64(v);
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += part[w];
        atomicAdd(reinterpret_cast<unsigned long long *>(out), (unsigned long long)t);
    }
}

__global__ __launch_bounds__(kBlock) void k_digest(ml2048_state st, int64_t M, int64_t idx_base, int with_id,
                                                   uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t h = 0;
    if (i < M) {
        const Board b = load_board(st.board, i);
        const uint4 m = reinterpret_cast<const uint4 *>(st.merged)[i];
        const uint32_t mm[4] = {m.x, m.y, m.z, m.w};
        h = board_digest((uint64_t)(idx_base + i), b, mm, reinterpret_cast<const uint32_t *>(st.valid)[i],
                         st.terminated[i], st.invalid[i], st.step[i], st.id[i],
                         __float_as_uint(st.score[i]), __float_as_uint(st.reward[i]), with_id);
    }
    block_add_u64(h, out);
}

// ---- fused rollout -----------------------------------------------------------------------------------
// One lane keeps one board (and its whole record) in registers for all N steps; per step:
// prepare (re-deal if terminated) -> sample action (rocRAND Philox) -> move/merge/reward/spawn/mask.
// Global traffic per step is the 16-byte permutation row(s) (L1/L2-resident 16 KiB table), the
// optional 1-byte action, and 8 bytes of re-deal flags per wave; state is read once and written
// once per launch.
// EXTRAS = false compiles the optional outputs (actions_out, digests_out, trajectory) out entirely:
// with one wave per SIMD every taken branch costs an instruction-buffer refill, and a dozen
// "pointer is NULL, skip" branches per step were ~10 % of the kernel (SQ_WAIT_ANY, profiles/).
template <int KIND, bool FAST, bool EXTRAS, bool INV>
__global__ __launch_bounds__(EXTRAS ? kBlock : kRollMaxBlock) void k_rollout(ml2048_state st, int64_t M, int64_t idx_base, int N,
                                                    const uint8_t *__restrict__ perm_epochs,
                                                    const uint32_t *__restrict__ two_masks,
                                                    const int32_t *__restrict__ schedule, int n_epochs,
                                                    int masked, uint64_t act_seed, uint64_t act_counter0,
                                                    uint8_t *__restrict__ actions_out,
                                                    uint64_t *__restrict__ digests_out,
                                                    uint64_t *__restrict__ reset_flags, int64_t flag_words,
                                                    int32_t *__restrict__ last_reset, uint64_t spawn_seed,
                                                    uint64_t spawn_counter0, uint64_t thr, ml2048_traj traj,
                                                    const uint4 *__restrict__ up_src, uint4 *__restrict__ up_dst,
                                                    int64_t up_n16, unsigned up_blocks, int64_t up_tab16,
                                                    const uint16_t *__restrict__ row_lut, int board_waves) {
    // The re-deal flags (one 64-bit ballot per step) of up to kFlagChunk steps are parked so that the step loop holds
    // no global store (its acknowledgement would share the vmcnt queue with the prefetched loads): in LDS, or - the
    // pass for small M, where a lone wave per SIMD pays for every slot - in a VGPR pair, step q of the chunk in lane q,
    // written with v_writelane.  The LDS write shares lgkmcnt with the scalar loads of the schedule and its lane-0
    // store needs an exec-mask round trip: 3.7 us of the 95.7 us kernel at C2; the register form costs 1.6 us less
    // there and 1 % more at 1 M boards (two more VALU per step where other waves hid the LDS write anyway).
    // The schedule kernel of the NEXT rollout (csrc/devrng.hip) may share a CU with one of these
    // workgroups; the rollout is the critical path, so its waves win the issue arbitration.
    __builtin_amdgcn_s_setprio(1);
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = (int)(threadIdx.x >> 6);
    // Workgroup shape.  EXTRAS: 256 threads = 4 board waves, as every per-board kernel.  Plain: `board_waves` board waves
    // (+ one upload wave, up_blocks == 1), chosen at launch so that ONE workgroup carries a compute unit's whole share
    // of the boards - the line table below takes 128 of the CU's 160 KiB of LDS, so workgroups do not share a CU.
    const int bw = EXTRAS ? kBlock / 64 : board_waves;
    const int64_t wave_global = (int64_t)blockIdx.x * bw + wave_in_wg;
    const int64_t i = wave_global * 64 + lane;
#ifndef ML2048_LONE_WAVE
    __shared__ unsigned long long flag_park[EXTRAS ? kBlock / 64 : kRollMaxWaves][kFlagChunk];
#endif
    // The line table (board_ops.hpp move_board_lut): 65 536 x uint16 in LDS, copied from the caller's device buffer by
    // the whole workgroup (16-byte units, eight requests in flight per thread: ~2 us).  With several waves per SIMD the
    // ~85 VALU of the arithmetic move are what bounds the kernel and a look-up's round trip hides behind the other waves
    // (262 144 boards: 189 -> 158 us); a wave alone on its SIMD waits the ~65 cycles out every step and still gains
    // (C2: 68.9 -> 66.9 us, 65.1 with the table's entries spread over the LDS banks: LUT_SWZ, board_ops.hpp row_lut_slot -
    // the small-M passes store the table swizzled, the large-M pass plain).  (Round 3's first measurement of the
    // small-M form read 78 us: its loader staged the table through scratch memory.)
    constexpr bool HAS_LUT = !EXTRAS;
#ifdef ML2048_LONE_WAVE
    constexpr bool LUT_SWZ = true;
#else
    constexpr bool LUT_SWZ = false;
#endif
    __shared__ uint16_t lut_s[HAS_LUT ? kRowLutEntries : 8];
    // The re-deal by LDS table too (every plain pass, table mode): a freshly dealt board is a function of the first two
    // cells of its permutation row and of their two bits of the epoch's 2/4 mask - 1 024 boards of 16 bytes, filled by the
    // workgroup itself; one ds_read_b128 and 8 VALU instead of the ~31 of deal_table_nibbles, in a block that runs
    // whenever one of the wave's 64 games has ended (about every third step).  262 144 boards: 2.09e11 -> 2.16e11
    // board-steps/s end to end, masked 1.33e11 -> 1.39e11; 131 072 boards (small-M pass): 104.5 -> 101.7 us; C2 (one wave
    // per SIMD: the look-up's round trip is exposed) 68.9 us either way.  ML2048_DEAL_LDS=0: A/B.
#if !defined(ML2048_DEAL_LDS) || ML2048_DEAL_LDS
    constexpr bool DEAL_LDS = EXTRAS ? INV : (FAST || INV); // (rocRAND mode: board_ops.hpp deal_words_index)
#else
    constexpr bool DEAL_LDS = false;
#endif
    __shared__ uint4 deal_s[DEAL_LDS ? 1024 : 1];
    // ... and the spawned tile (device_helpers.hpp spawn_table_dot_tiles), where other waves share the SIMD (large-M pass)
#if !defined(ML2048_TILE_LDS) || ML2048_TILE_LDS
    // (a lone wave loses: C2 68.3 vs 66.9 us; the trajectory-writing instantiations on expanded tables take it too)
    constexpr bool TILE_LDS = !FAST && INV && !LUT_SWZ;
#else
    constexpr bool TILE_LDS = false;
#endif
    __shared__ uint4 tile_s[TILE_LDS ? 32 : 1];
    // ... and the orientation of the table move as byte selectors looked up by the action (board_ops.hpp move_selectors)
    // (where waves share a SIMD: 262 144 boards 147.3 -> 142.0 us; a lone wave gains nothing - C2 65.2-66.3 vs 65.6-65.9 us,
    // rocRAND mode 80.0 vs 79.4 - so the small-M passes keep the decoded form.  -DML2048_MOVE_SEL=0 / =1: off / on
    // in every pass, A/B)
#if !defined(ML2048_MOVE_SEL)
    constexpr bool MOVE_SEL = HAS_LUT && !LUT_SWZ;
#elif ML2048_MOVE_SEL
    constexpr bool MOVE_SEL = HAS_LUT;
#else
    constexpr bool MOVE_SEL = false;
#endif
    __shared__ __attribute__((aligned(32))) uint32_t msel_s[MOVE_SEL ? 32 : 8];
    if constexpr (MOVE_SEL) {
        if (threadIdx.x < 32) {
            uint32_t sel6[6];
            move_selectors(threadIdx.x >> 3, sel6);
            const uint32_t k = threadIdx.x & 7u;
            msel_s[threadIdx.x] = k == 0 ? sel6[0] : k == 1 ? sel6[1] : k == 2 ? sel6[2] : k == 3 ? sel6[3] : k == 4 ? sel6[4] : k == 5 ? sel6[5] : 0u;
        }
    }
    if constexpr (TILE_LDS) {
        if (threadIdx.x < 32) {
            const uint32_t c = threadIdx.x & 15u, v = 2u - (threadIdx.x >> 4);
            const uint32_t w = v << (8u * (c & 3u));
            tile_s[threadIdx.x] = make_uint4((c >> 2) == 0u ? w : 0u, (c >> 2) == 1u ? w : 0u, (c >> 2) == 2u ? w : 0u,
                                             (c >> 2) == 3u ? w : 0u);
        }
    }
    if constexpr (DEAL_LDS) {
        for (int e = (int)threadIdx.x; e < 1024; e += (int)blockDim.x) {
            // index = c1 | c2 << 4 | (mask bit of cell c1) << 8 | (mask bit of cell c2) << 9
            const uint32_t c1 = (uint32_t)e & 15u, c2 = ((uint32_t)e >> 4) & 15u;
            const uint32_t m = ((((uint32_t)e >> 8) & 1u) << c1) | ((((uint32_t)e >> 9) & 1u) << c2);
            Board d;
            deal_table_nibbles(d, (uint32_t)e & 0xFFu, m);
            deal_s[e] = make_uint4(d.r[0], d.r[1], d.r[2], d.r[3]);
        }
    }
    if constexpr (!EXTRAS) {
        if (HAS_LUT && row_lut) {
            const uint4 *g = reinterpret_cast<const uint4 *>(row_lut);
            uint4 *l = reinterpret_cast<uint4 *>(lut_s);
            constexpr int kUnits = kRowLutEntries * 2 / 16, kInFlight = 8;
            const int nthr = (int)blockDim.x;
            int k0 = (int)threadIdx.x;
            // full rounds of eight unconditional requests per thread (a guarded form made the compiler stage the eight
            // values through scratch memory: 32 MB of extra writes per launch at 262 144 boards), then the rest one by one
            // (LUT_SWZ: unit u = entries 8u .. 8u+7 of the plain table; row_lut_slot moves it to unit u ^ ((u >> 6) & 31)
            // and swaps its halves when bit 5 of u is set - the slot function XORs bits 2..7 of an index with bits 8..13)
            auto put = [&](int u, uint4 v) __attribute__((always_inline)) {
                if (LUT_SWZ) {
                    if (u & 32) v = make_uint4(v.z, v.w, v.x, v.y);
                    u ^= (u >> 6) & 31;
                }
                l[u] = v;
            };
            // (sixteen requests in flight for the small workgroups - two L2 round trips instead of three full rounds and
            // two single requests - measured slower: C2 66.3 vs 65.5 us)
            for (; k0 + (kInFlight - 1) * nthr < kUnits; k0 += kInFlight * nthr) {
                const uint4 v0 = g[k0], v1 = g[k0 + nthr], v2 = g[k0 + 2 * nthr], v3 = g[k0 + 3 * nthr];
                const uint4 v4 = g[k0 + 4 * nthr], v5 = g[k0 + 5 * nthr], v6 = g[k0 + 6 * nthr], v7 = g[k0 + 7 * nthr];
                put(k0, v0); put(k0 + nthr, v1); put(k0 + 2 * nthr, v2); put(k0 + 3 * nthr, v3);
                put(k0 + 4 * nthr, v4); put(k0 + 5 * nthr, v5); put(k0 + 6 * nthr, v6); put(k0 + 7 * nthr, v7);
            }
            for (; k0 < kUnits; k0 += nthr) put(k0, g[k0]);
        }
    }
    if (DEAL_LDS || TILE_LDS || MOVE_SEL || (HAS_LUT && row_lut)) __syncthreads();
    if constexpr (!EXTRAS) {
        // Chained upload: this launch also copies the NEXT rollout's packed schedule from device-accessible
        // host memory into the other device buffer - ordered before the next launch by the stream itself,
        // so the upload needs neither a copy stream nor an event (a cross-stream wait in front of every
        // rollout kernel cost ~19 us of idle GPU, tools/debug/marker_cost.py).  The copy is done by one extra
        // wave per workgroup (up_blocks == 1: it shares a SIMD with a board wave for the few microseconds of a PCIe
        // read and is gone) or, where the workgroup already has its 16 waves, by each workgroup's first wave before
        // it turns to its boards (up_blocks == 2; large M: the other waves of its SIMD fill the slots meanwhile).
        // Either way no board wave sits behind a PCIe read in the in-order vmcnt queue during its step loop.
        if (up_blocks != 0 && wave_in_wg == (up_blocks == 1 ? bw : 0)) {
            for (int64_t j = (int64_t)blockIdx.x * 64 + lane; j < up_n16; j += (int64_t)gridDim.x * 64)
                upload_unit(up_dst, up_src, j, up_tab16);
        }
        if (wave_in_wg >= bw || wave_global * 64 >= M) return; // the upload wave; a wave past the last board
        // flag words past the last board's wave (flag_words is rounded up to four waves): nobody else writes them
        if ((wave_global + 1) * 64 >= M)
            for (int64_t w = wave_global + 1; w < flag_words; ++w)
                for (int t = lane; t < N; t += 64) reset_flags[(int64_t)t * flag_words + w] = 0ull;
    }
#ifdef ML2048_LONE_WAVE
    uint32_t park_lo = 0, park_hi = 0;
    const unsigned long long live_lanes = __builtin_amdgcn_ballot_w64(i < M);
#endif
    const bool live = i < M;
    const int64_t ii = live ? i : M - 1; // tail lanes shadow the last board (loads only, no stores)

#if defined(ML2048_LONE_WAVE) && (!defined(ML2048_SCHED_AHEAD) || ML2048_SCHED_AHEAD)
    // The lane-held schedule of a 64-step chunk (see run_steps): its entries are requested a chunk ahead - the first
    // chunk's here, in front of the record's loads, the next chunk's at the start of the chunk before it - and only the
    // dependent gather of the entries' two-masks is left for the chunk's own start (one exposed round trip per chunk
    // instead of two; ML2048_SCHED_AHEAD=0: A/B).
    constexpr bool SCHED_AHEAD = !FAST && !EXTRAS;
    uint32_t nx_e = 0, nx_rp = 0, nx_rs = 0; // lane q: entry (chunk start + q + 1), epoch clamped
    auto fetch_chunk_entries = [&](int t0c) __attribute__((always_inline)) {
        const int j = t0c + lane + 1 < N ? t0c + lane + 1 : N - 1;
        const uint32_t e = (uint32_t)schedule[3 * j + 0];
        const uint32_t cap = (uint32_t)(n_epochs - 1);
        nx_e = e < cap ? e : cap;
        nx_rp = (uint32_t)schedule[3 * j + 1];
        nx_rs = (uint32_t)schedule[3 * j + 2];
    };
    if (SCHED_AHEAD) fetch_chunk_entries(0);
#elif defined(ML2048_LONE_WAVE)
    constexpr bool SCHED_AHEAD = false;
#endif
    StepRegs s;
    s.board = load_board(st.board, ii);
    {
        const uint4 m = reinterpret_cast<const uint4 *>(st.merged)[ii];
        s.merged[0] = m.x; s.merged[1] = m.y; s.merged[2] = m.z; s.merged[3] = m.w;
    }
    s.valid = reinterpret_cast<const uint32_t *>(st.valid)[ii];
    s.step = st.step[ii];
    s.score = st.score[ii];
    s.reward = st.reward[ii];
    s.terminated = st.terminated[ii];
    uint32_t invalid = st.invalid[ii];
    Board prev_board = load_board(st.prev_board, ii);
    uint32_t prev_valid = reinterpret_cast<const uint32_t *>(st.prev_valid)[ii];
    int32_t last = -1;
    // The merged[16] counters are only observable at the end (and in the optional digests), so
    // the loop carries the <= 8 merge exponents of the last valid move and expands them lazily.
    uint32_t mexp_first = 0, mexp_second = 0;
    bool merged_stale = false; // true once s.merged no longer reflects (mexp_first, mexp_second)
    // The sampler's stream (board_ops.hpp sampler_word): a masked draw consumes one word W(t), a plain uniform
    // draw two bits of W(t >> 4).  aw[0] is always the current word; one Philox block serves 4 masked or 64 plain
    // steps.  `seq_shift` = log2(draws per word).
    uint32_t aw[4];
    uint32_t act_bits = 0; // lazy loop: what is left of the current word after the draws made so far
    {
        const uint32_t seq_shift = masked ? 0u : 4u;
        const uint64_t j0 = act_counter0 >> seq_shift;
        rocrand_state_philox4x32_10 rs;
        rocrand_init(act_seed, (uint64_t)(idx_base + ii), j0 & ~3ull, &rs);
        const uint4 v = rocrand4(&rs);
        aw[0] = v.x; aw[1] = v.y; aw[2] = v.z; aw[3] = v.w;
        for (uint32_t q = 0; q < (uint32_t)(j0 & 3ull); ++q) { // wave-uniform, outside the loop
            aw[0] = aw[1]; aw[1] = aw[2]; aw[2] = aw[3];
        }
    }

    // spawn="rocrand": the step's spawn consumes word (spawn_counter0 + t) of the board's step stream (board_ops.hpp
    // spawn_word), so one Philox block serves four steps; sw[0] is always the current word
    uint32_t sw[4] = {0, 0, 0, 0};
    if (FAST) {
        rocrand_state_philox4x32_10 rs;
        rocrand_init(spawn_seed, (uint64_t)(idx_base + ii), spawn_counter0 & ~3ull, &rs);
        const uint4 v = rocrand4(&rs);
        sw[0] = v.x; sw[1] = v.y; sw[2] = v.z; sw[3] = v.w;
        for (uint32_t q = 0; q < (uint32_t)(spawn_counter0 & 3ull); ++q) { // wave-uniform, outside the loop
            sw[0] = sw[1]; sw[1] = sw[2]; sw[2] = sw[3];
        }
    }
    // Everything that does not depend on game state is fetched ahead of its use - at M = 65 536 there is
    // one wave per SIMD and nobody else to hide a load behind (profiles/r01_i_sq_counters.json): the
    // schedule entry (scalar loads) two steps ahead, the two-mask and both permutation rows one step
    // ahead.  The re-deal only reads the first two cells of its row, so that load is a single dword
    // (a dwordx4 with three dead result registers makes the compiler wait for the load mid-step before
    // it can reuse them).  Indices are clamped instead of branched around at the end of the schedule.
    // With the optional outputs (EXTRAS) the loop also issues stores, which share the in-order vmcnt
    // queue with the loads: a row fetched a step ahead would make the wait at the top of the next
    // step cover this step's stores as well (fused-traj at 1 M boards: 1.89 -> 2.42 ms).  There the rows
    // are fetched in the step that uses them and only the scalar schedule entry runs ahead.
    constexpr int VPF = EXTRAS ? 0 : 1;   // how many steps ahead the vector loads run
    int e1 = 0;
    int64_t rp1 = 0, rs1 = 0;             // schedule entry of step t + VPF
    uint32_t n_two_mask = 0, n_perm_p = 0; // VPF: data of step t, in flight since the previous iteration
    uint32_t n_perm_s[4] = {0, 0, 0, 0}, n_wlo_s[4] = {0, 0, 0, 0}, n_whi_s[4] = {0, 0, 0, 0};
    constexpr int64_t kEpochBytes = (int64_t)ML2048_RAND_SIZE * (INV ? 16 * kExpandedRowUnits : 16);
    // an epoch index is clamped to the tables that exist: a corrupted schedule then yields wrong
    // boards, never an out-of-bounds read (rows are already taken modulo the table size)
    const uint32_t e_last = (uint32_t)(n_epochs - 1);
    if (!FAST) {
        const uint32_t e0r = (uint32_t)schedule[0];
        e1 = (int)(e0r < e_last ? e0r : e_last);
        rp1 = schedule[1];
        rs1 = schedule[2];
        if (VPF) {
            const uint8_t *tab0 = perm_epochs + (int64_t)e1 * kEpochBytes;
            n_two_mask = two_masks[e1];
            n_perm_p = load_table_head<INV>(tab0, rp1 + idx_base + ii);
            load_table_row<INV>(tab0, rs1 + idx_base + ii, n_perm_s, n_wlo_s, n_whi_s);
            const int j1 = N > 1 ? 1 : 0;
            const uint32_t e1r = (uint32_t)schedule[3 * j1 + 0];
            e1 = (int)(e1r < e_last ? e1r : e_last);
            rp1 = schedule[3 * j1 + 1];
            rs1 = schedule[3 * j1 + 2];
        }
    }
    // Uniform (unmasked) sampling never reads the valid-action mask between steps, so when every board of
    // this wave starts with a mask that is consistent with its board, the loop runs in its LAZY form:
    // validity = "the move changes the board", terminated = "full and no equal neighbours"
    // (board_ops.hpp step_pre_lazy / no_move_left), and valid / prev_valid are derived once, after the last
    // step.  ~70 fewer VALU operations per board-step; identical results (tests/test_host_ops.py,
    // tests/test_gpu_parity.py).  The optional outputs need the mask every step: always eager.
    bool lazy = !EXTRAS && !masked && __all(valid_consistent(s.board, s.valid) ? 1 : 0);
    // The lazy loop carries no reward arithmetic at all: the score after the last step follows from the board and the
    // step counter (board_ops.hpp score_potential; `score` accumulates reward_fn_normal whatever reward_fn is, :729-731),
    // and the record's reward / merged[] are the LAST valid move's (H3), recomputed once from that move's board and
    // action (`pv`, `pv_action`), which is all the loop keeps.  ~17 VALU per board-step for reward_fn_normal, the whole
    // reward function for the others.  A wave with a board or a score outside the guard's range (never in play from a
    // deal) takes the eager loop.
    // The same holds with a maintained mask (masked sampling without optional outputs): `defer`.
    const bool defer = !EXTRAS && __all(phi_guard(s.board, s.score, N) ? 1 : 0);
    lazy = lazy && defer;
    int32_t phi_acc = 0; // score - potential, as of the game's start inside this rollout
    if (defer) phi_acc = (int32_t)s.score - score_potential(s.board, s.step);
    Board pv = s.board;      // the board the last valid move started from ...
    uint32_t pv_action = 4u; // ... and its action (4 = none since the deal / since the rollout began)
    // The loops without reward arithmetic keep nothing of a move but the board, so the move can be four look-ups in the
    // LDS line table instead of ~85 VALU (board_ops.hpp move_board_lut) - for waves whose boards stay inside the
    // table's domain for the whole rollout (lut_guard: tiles <= 2^11 now, at most 8 191 steps).
    bool use_lut = false;
    if constexpr (HAS_LUT) use_lut = defer && row_lut != nullptr && N <= kRowLutMaxSteps && __all(lut_guard(s.board) ? 1 : 0);
    auto run_steps = [&](auto lazy_tag, auto defer_tag, auto lut_tag) {
        constexpr bool LAZY = decltype(lazy_tag)::value;
        constexpr bool DEFER = decltype(defer_tag)::value; // no reward arithmetic in the loop (always with LAZY)
        constexpr bool LUT = decltype(lut_tag)::value;     // the move by table look-up (only with DEFER)
        // LAZY (always two bits per draw, sixteen draws per word): inside a chunk the groups of kStepUnroll steps start on
        // a multiple of kStepUnroll of the sampler's counter (`group_lead` single steps come first; the chunks themselves
        // start on multiples of kFlagChunk) - a new word can then only fall due at the first step of a group, and the
        // other steps carry no test for it (a wave-uniform branch per step that was taken fifteen times out of sixteen:
        // with one wave per SIMD every taken branch is an instruction-buffer refill nobody hides).
        constexpr bool WORD_AT_GROUP_START = LAZY && kStepUnroll > 1 && 16 % kStepUnroll == 0 && kFlagChunk % kStepUnroll == 0;
        const int group_lead = WORD_AT_GROUP_START ? (int)((0 - act_counter0) & (uint64_t)(kStepUnroll - 1)) : 0;
        for (int t0 = 0; t0 < N; t0 += kFlagChunk) {
            const int t1 = t0 + kFlagChunk < N ? t0 + kFlagChunk : N;
#ifdef ML2048_LONE_WAVE
            // The schedule of this chunk, one entry per lane (lane q: the entry step t0 + q fetches ahead, i.e.
            // step t0 + q + 1's): epoch as a byte offset into the tables, the two row bases, the epoch's two-mask.
            // The step loop broadcasts them with v_readlane - four VALU slots instead of ~16 scalar address
            // instructions, three scalar loads and their lgkmcnt waits per step, which a lone wave pays in full.
            uint32_t sch_eo = 0, sch_rp = 0, sch_rs = 0, sch_tm = 0;
            const uint32_t row_lane = (uint32_t)(idx_base + ii); // only its low 10 bits matter
            if (!FAST && VPF) {
#if !defined(ML2048_SCHED_AHEAD) || ML2048_SCHED_AHEAD
                if (SCHED_AHEAD) {
                    sch_eo = nx_e * (uint32_t)kEpochBytes;
                    sch_rp = nx_rp;
                    sch_rs = nx_rs;
                    sch_tm = two_masks[nx_e];
                    if (t0 + kFlagChunk < N) fetch_chunk_entries(t0 + kFlagChunk); // wave-uniform; used a chunk later
                } else
#endif
                {
                    const int j = t0 + lane + 1 < N ? t0 + lane + 1 : N - 1;
                    const uint32_t e = (uint32_t)schedule[3 * j + 0];
                    const uint32_t ec = e < e_last ? e : e_last;
                    sch_eo = ec * (uint32_t)kEpochBytes;
                    sch_rp = (uint32_t)schedule[3 * j + 1];
                    sch_rs = (uint32_t)schedule[3 * j + 2];
                    sch_tm = two_masks[ec];
                }
            }
#endif
            auto one_step = [&](const int t, auto word_check_tag) __attribute__((always_inline)) {
                constexpr bool WORD_CHECK = decltype(word_check_tag)::value; // may this step be the first of a sampler word
                // table mode: the pre-drawn host schedule; fast mode: rocRAND Philox words keyed
                // (spawn_seed; board; 2*(counter0+t) for the deal, +1 for the step) as the per-step API does
                uint32_t two_mask = n_two_mask;
                uint32_t perm_p = n_perm_p;
                uint32_t perm_s[4] = {n_perm_s[0], n_perm_s[1], n_perm_s[2], n_perm_s[3]};
                uint32_t wlo_s[4] = {n_wlo_s[0], n_wlo_s[1], n_wlo_s[2], n_wlo_s[3]};
                uint32_t whi_s[4] = {n_whi_s[0], n_whi_s[1], n_whi_s[2], n_whi_s[3]};
#ifdef ML2048_LONE_WAVE
                if (!FAST && VPF) {
                    // requests for step t+1 from the chunk's lane-held schedule (lane q = entry t0 + q + 1)
                    const int q = t - t0;
                    const uint32_t s_eo = (uint32_t)__builtin_amdgcn_readlane((int)sch_eo, q);
                    const uint32_t s_rp = (uint32_t)__builtin_amdgcn_readlane((int)sch_rp, q);
                    const uint32_t s_rs = (uint32_t)__builtin_amdgcn_readlane((int)sch_rs, q);
                    n_two_mask = (uint32_t)__builtin_amdgcn_readlane((int)sch_tm, q);
                    constexpr uint32_t kRowBytes = INV ? 16u * kExpandedRowUnits : 16u;
                    const uint32_t off_p = ((s_rp + row_lane) & (uint32_t)(ML2048_RAND_SIZE - 1)) * kRowBytes + s_eo;
                    const uint32_t off_s = ((s_rs + row_lane) & (uint32_t)(ML2048_RAND_SIZE - 1)) * kRowBytes + s_eo;
                    n_perm_p = *reinterpret_cast<const uint32_t *>(perm_epochs + off_p);
                    const uint4 *row = reinterpret_cast<const uint4 *>(perm_epochs + off_s);
                    if (INV) {
                        const uint2 p = *reinterpret_cast<const uint2 *>(row);
                        const uint4 v = row[1], w = row[2];
                        n_perm_s[0] = p.x; n_perm_s[1] = p.y;
                        n_wlo_s[0] = v.x; n_wlo_s[1] = v.y; n_wlo_s[2] = v.z; n_wlo_s[3] = v.w;
                        n_whi_s[0] = w.x; n_whi_s[1] = w.y; n_whi_s[2] = w.z; n_whi_s[3] = w.w;
                    } else {
                        const uint4 v = row[0];
                        n_perm_s[0] = v.x; n_perm_s[1] = v.y; n_perm_s[2] = v.z; n_perm_s[3] = v.w;
                    }
                } else
#endif
                if (!FAST) {
                    const uint8_t *tab1 = perm_epochs + (int64_t)e1 * kEpochBytes;
                    if (VPF) { // requests for step t+1, consumed one iteration later
                        n_two_mask = two_masks[e1];
                        n_perm_p = load_table_head<INV>(tab1, rp1 + idx_base + ii);
                        load_table_row<INV>(tab1, rs1 + idx_base + ii, n_perm_s, n_wlo_s, n_whi_s);
                    } else {
                        two_mask = two_masks[e1];
                        perm_p = load_table_head<INV>(tab1, rp1 + idx_base + ii);
                        load_table_row<INV>(tab1, rs1 + idx_base + ii, perm_s, wlo_s, whi_s);
                    }
                    const int j2 = t + 1 + VPF < N ? t + 1 + VPF : N - 1;
                    const uint32_t e2r = (uint32_t)schedule[3 * j2 + 0];
                    e1 = (int)(e2r < e_last ? e2r : e_last);
                    rp1 = schedule[3 * j2 + 1];
                    rs1 = schedule[3 * j2 + 2];
                }

                if (FAST && t > 0) { // the next word of the step stream (wave-uniform)
                    const uint64_t js = spawn_counter0 + (uint64_t)t;
                    if ((js & 3ull) == 0) {
                        rocrand_state_philox4x32_10 rs;
                        rocrand_init(spawn_seed, (uint64_t)(idx_base + ii), js, &rs);
                        const uint4 v = rocrand4(&rs);
                        sw[0] = v.x; sw[1] = v.y; sw[2] = v.z; sw[3] = v.w;
                    } else {
                        sw[0] = sw[1]; sw[1] = sw[2]; sw[2] = sw[3];
                    }
                }
                // ---- prepare (game_numba.py:619-658)
                const bool deal = s.terminated != 0;
                // (the builtin on the bool itself: __ballot(int) first turns the predicate into 0/1 with a select and
                // compares that again - two VALU slots per step for a lone wave)
#ifdef ML2048_LONE_WAVE
                const unsigned long long dealt = __builtin_amdgcn_ballot_w64(deal); // the tail lanes are masked at the flush
#else
                const unsigned long long dealt = __builtin_amdgcn_ballot_w64(deal && live);
#endif
#ifdef ML2048_LONE_WAVE
                write_lane64(park_lo, park_hi, dealt, t - t0);
#else
                if (lane == 0) flag_park[wave_in_wg][t - t0] = dealt;
#endif
                if (deal) {
                    clear_regs(s);
                    invalid = 0;
                    mexp_first = mexp_second = 0;
                    merged_stale = true;
                    if (FAST) {
                        uint32_t w0, w1;
                        rocrand_deal_words(spawn_seed, spawn_counter0 + (uint64_t)t, (uint64_t)(idx_base + ii), w0, w1);
                        if (DEAL_LDS) {
                            const uint4 d = deal_s[deal_words_index(w0, w1, (uint32_t)thr)];
                            s.board.r[0] = d.x; s.board.r[1] = d.y; s.board.r[2] = d.z; s.board.r[3] = d.w;
                        } else {
                            spawn_word(s.board, w0, (uint32_t)thr);
                            spawn_word(s.board, w1, (uint32_t)thr);
                        }
                    } else {
                        if (DEAL_LDS) {
                            const uint32_t c1 = perm_p & 15u, c2 = ubfe(perm_p, 4, 4);
                            const uint4 d = deal_s[(perm_p & 0xFFu) | (ubfe(two_mask, c1, 1) << 8) | (ubfe(two_mask, c2, 1) << 9)];
                            s.board.r[0] = d.x; s.board.r[1] = d.y; s.board.r[2] = d.z; s.board.r[3] = d.w;
                        } else if (INV) {
                            deal_table_nibbles(s.board, perm_p, two_mask);
                        } else {
                            const uint32_t head[4] = {perm_p, 0, 0, 0};
                            deal_table(s.board, head, two_mask);
                        }
                    }
                    if (!LAZY) s.valid = valid_actions(s.board);
                    if (DEFER) { // score 0 on the dealt board at step 0: its potential is 0
                        phi_acc = 0;
                        pv_action = 4u;
                    }
                    last = t;
                }
                // ---- sample (policy/random.py:17-27)
                uint32_t word;
                if constexpr (!WORD_CHECK) {
                    // inside a group of the lazy loop: the draws of this group are the next bits of `act_bits`, which
                    // the group's first step left behind - no counter arithmetic at all (three scalar slots per step)
                    word = act_bits;
                    act_bits >>= 2;
                } else {
                const uint64_t tt = act_counter0 + (uint64_t)t;
                const bool plain = LAZY || !masked; // two bits per draw; LAZY instantiations are never masked
                const uint32_t sub = plain ? (uint32_t)(tt & 15ull) : 0u;
                if (t > 0 && sub == 0) { // wave-uniform: the next word of the stream
                    const uint64_t j = plain ? (tt >> 4) : tt;
                    if ((j & 3ull) == 0) {
                        rocrand_state_philox4x32_10 rs;
                        rocrand_init(act_seed, (uint64_t)(idx_base + ii), j, &rs);
                        const uint4 v = rocrand4(&rs);
                        aw[0] = v.x; aw[1] = v.y; aw[2] = v.z; aw[3] = v.w;
                    } else {
                        aw[0] = aw[1]; aw[1] = aw[2]; aw[2] = aw[3];
                    }
                }
                word = aw[0] >> (2u * sub);
                act_bits = word >> 2;
                }
                uint32_t cnt = 4;
                const uint32_t action = LAZY ? (word & 3u) : pick_action(s.valid, word, masked, cnt);
                if (EXTRAS && actions_out && live) actions_out[(int64_t)t * M + i] = (uint8_t)action;
                // ---- step (game_numba.py:660-738)
                prev_board = s.board;
                if (!LAZY) prev_valid = s.valid;
                if (LAZY    ? (LUT ? (MOVE_SEL ? step_pre_lazy_pv_lut<LUT_SWZ>(s, action, pv, pv_action, lut_s, msel_s)
                                                : step_pre_lazy_pv_lut<LUT_SWZ>(s, action, pv, pv_action, lut_s))
                                   : step_pre_lazy_pv(s, action, pv, pv_action))
                    : DEFER ? (LUT ? (MOVE_SEL ? step_pre_lite_pv_lut<LUT_SWZ>(s, action, pv, pv_action, lut_s, msel_s)
                                                : step_pre_lite_pv_lut<LUT_SWZ>(s, action, pv, pv_action, lut_s))
                                   : step_pre_lite_pv(s, action, pv, pv_action))
                            : step_pre_lite(s, action, KIND, mexp_first, mexp_second)) {
                    merged_stale = true;
                    uint32_t empties = 0;
                    if (FAST) {
                        empties = spawn_word(s.board, sw[0], (uint32_t)thr);
                    } else {
                        if (TILE_LDS) empties = spawn_table_dot_tiles<LAZY>(s.board, perm_s[0], perm_s[1], wlo_s, whi_s, two_mask, tile_s);
                        else if (INV) empties = spawn_table_dot<LAZY>(s.board, perm_s[0], perm_s[1], wlo_s, whi_s, two_mask);
                        else spawn_table(s.board, perm_s, two_mask, 1);
                    }
                    if (LAZY && (FAST || INV)) step_post_lazy(s, empties);
                    else if (LAZY) step_post_lazy(s);
                    else step_post(s);
                    invalid = 0;
                } else {
                    invalid = 1;
                }
                // ---- optional trajectory: this step's transition into [N][M] buffers with the REPLAY_SPEC
                //      dtypes (replay.py:10-20), i.e. what N x ml2048_record_transition would write
                if (EXTRAS && live) {
                    const int64_t o = (int64_t)t * M + i;
                    // written once, read by another kernel much later: non-temporal stores (ML2048_TRAJ_NT=0: A/B) -
                    // same box, 1 M boards: 1.43-1.56 ms vs 1.55-1.94 ms per launch; 262 144: 428 vs 456 us; C2: 167 vs 175 us.
                    // Measured and dropped on the same boxes (profiles/r02_f_traj_ab.log): the epoch's permutation table
                    // staged in LDS so that the loop holds no vector load (whose vmcnt wait also covers the previous
                    // step's stores on gfx9): +10 % at C2 (13 exposed refills), +2 % at 1 M; the seven narrow fields
                    // staged in LDS and written as one full-width store per field per workgroup (15 store instructions
                    // per workgroup-step instead of 36, one barrier per step): +30 % at C2, +11 % at 262 144, +-0 at 1 M.
#if !defined(ML2048_TRAJ_NT) || ML2048_TRAJ_NT
#define ML_TSTORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define ML_TSTORE(ptr, val) (*(ptr) = (val))
#endif
                    if (traj.state) {
                        uint32_t *d = reinterpret_cast<uint32_t *>(traj.state) + 4 * o;
                        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                        const v4u q = {prev_board.r[0], prev_board.r[1], prev_board.r[2], prev_board.r[3]};
                        ML_TSTORE(reinterpret_cast<v4u *>(d), q);
                    }
                    if (traj.next_state) {
                        uint32_t *d = reinterpret_cast<uint32_t *>(traj.next_state) + 4 * o;
                        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                        const v4u q = {s.board.r[0], s.board.r[1], s.board.r[2], s.board.r[3]};
                        ML_TSTORE(reinterpret_cast<v4u *>(d), q);
                    }
                    if (traj.valid_actions) ML_TSTORE(reinterpret_cast<uint32_t *>(traj.valid_actions) + o, prev_valid);
                    if (traj.next_valid_actions) ML_TSTORE(reinterpret_cast<uint32_t *>(traj.next_valid_actions) + o, s.valid);
                    if (traj.action) ML_TSTORE(traj.action + o, (int8_t)action);
                    if (traj.action_log_prob) ML_TSTORE(traj.action_log_prob + o, neg_log_count(cnt));
                    if (traj.reward) ML_TSTORE(traj.reward + o, s.reward);
                    if (traj.step) ML_TSTORE(traj.step + o, s.step);
                    if (traj.terminated) ML_TSTORE(traj.terminated + o, (uint8_t)(s.terminated != 0));
                }
#undef ML_TSTORE
                if (EXTRAS && digests_out) {
                    if (merged_stale) {
                        merge_buckets(mexp_first, mexp_second, s.merged);
                        merged_stale = false;
                    }
                    uint64_t h = 0;
                    if (live)
                        h = board_digest((uint64_t)(idx_base + i), s.board, s.merged, s.valid, s.terminated, invalid,
                                         s.step, 0, __float_as_uint(s.score), __float_as_uint(s.reward), 0);
                    h = wave_sum_u64(h);
                    if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(digests_out + t), (unsigned long long)h);
                }
            };
            // kStepUnroll steps per loop iteration: the registers that carry the prefetched table rows and the board
            // from one step to the next are renamed instead of copied at every step boundary (the compiler does not
            // unroll this loop by itself: it contains convergent operations and has a run-time trip count)
            constexpr int U = EXTRAS ? 1 : kStepUnroll;
            int t = t0;
            if constexpr (WORD_AT_GROUP_START) {
                const int lead_end = t0 + group_lead < t1 ? t0 + group_lead : t1;
                for (; t < lead_end; ++t) one_step(t, std::true_type{});
            }
            if constexpr (U > 1) {
                for (; t + U <= t1; t += U) {
                    if constexpr (WORD_AT_GROUP_START) {
                        one_step(t, std::true_type{});
#pragma unroll
                        for (int u = 1; u < U; ++u) one_step(t + u, std::false_type{});
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) one_step(t + u, std::true_type{});
                    }
                }
            }
            for (; t < t1; ++t) one_step(t, std::true_type{});
#ifdef ML2048_LONE_WAVE
            // flush this chunk's re-deal flags: lane q holds step t0 + q
            if (lane < t1 - t0) // (lanes past M shadow the last board, re-deals included: not boards, not flags)
                reset_flags[(int64_t)(t0 + lane) * flag_words + wave_global] =
                    ((uint64_t)park_lo | ((uint64_t)park_hi << 32)) & live_lanes;
#else
            // flush this chunk's re-deal flags (wave-private LDS row: a wave-scope fence is enough)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int q = lane; q < t1 - t0; q += 64)
                reset_flags[(int64_t)(t0 + q) * flag_words + wave_global] = flag_park[wave_in_wg][q];
#endif
        }
    };
    if constexpr (EXTRAS) {
        run_steps(std::false_type{}, std::false_type{}, std::false_type{});
    } else {
        if (!defer) {
            run_steps(std::false_type{}, std::false_type{}, std::false_type{});
        } else {
            if (lazy) {
                if (HAS_LUT && use_lut) run_steps(std::true_type{}, std::true_type{}, std::bool_constant<HAS_LUT>{});
                else run_steps(std::true_type{}, std::true_type{}, std::false_type{});
                s.valid = valid_actions(s.board);
                prev_valid = valid_actions(prev_board);
            } else {
                if (HAS_LUT && use_lut) run_steps(std::false_type{}, std::true_type{}, std::bool_constant<HAS_LUT>{});
                else run_steps(std::false_type{}, std::true_type{}, std::false_type{});
            }
            s.score = (float)(phi_acc + score_potential(s.board, s.step));
            if (merged_stale) { // a deal or a valid move happened: the record shows the last valid move since the deal
                s.reward = 0.0f;
                s.merged[0] = s.merged[1] = s.merged[2] = s.merged[3] = 0;
                if (pv_action < 4u) {
                    const MoveResult mv = move_board(pv, pv_action);
                    const MergeStats ms = merge_stats(mv.mexp_first, mv.mexp_second);
                    s.merged[0] = ms.merged[0]; s.merged[1] = ms.merged[1]; s.merged[2] = ms.merged[2]; s.merged[3] = ms.merged[3];
                    s.reward = (float)reward_value(KIND, mv.board, pv, ms);
                }
                merged_stale = false;
            }
        }
    }
    if (merged_stale) merge_buckets(mexp_first, mexp_second, s.merged);

    if (!live) return;
    store16(st.board, i, s.board.r);
    store16(st.merged, i, s.merged);
    store16(st.prev_board, i, prev_board.r);
    reinterpret_cast<uint32_t *>(st.valid)[i] = s.valid;
    reinterpret_cast<uint32_t *>(st.prev_valid)[i] = prev_valid;
    st.step[i] = s.step;
    st.score[i] = s.score;
    st.reward[i] = s.reward;
    st.terminated[i] = (uint8_t)s.terminated;
    st.invalid[i] = (uint8_t)invalid;
    last_reset[i] = last;
}

// ids after a rollout.  scratch: [0, N*W) exclusive prefix of popcounts inside each step's flag
// row; [N*W, N*W+N] exclusive prefix of per-step totals (last entry = grand total).
__global__ __launch_bounds__(256) void k_ids_rowscan(const uint64_t *__restrict__ flags, int64_t W, int N,
                                                     int32_t *__restrict__ scratch,
                                                     const int32_t *__restrict__ game_count, int32_t snap_slot,
                                                     int32_t *__restrict__ step_resets) {
    __shared__ int wave_tot[4];
    __shared__ int carry_s;
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < W; base += 256) {
        const int64_t j = base + threadIdx.x;
        const int v = j < W ? __popcll(flags[(int64_t)t * W + j]) : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int n = __shfl_up(incl, off, 64);
            if (lane >= off) incl += n;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wave_base = 0;
        for (int w = 0; w < wave; ++w) wave_base += wave_tot[w];
        const int carry = carry_s;
        if (j < W) scratch[(int64_t)t * W + j] = carry + wave_base + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + wave_base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scratch[(int64_t)N * W + t] = carry_s; // per-step total (scanned by the consumer)
        if (step_resets) step_resets[t] = carry_s;
        // the game counter as of before this rollout: k_ids_assign reads it from here while one of
        // its threads writes the new value
        if (t == 0) scratch[snap_slot] = *game_count;
    }
}

__global__ __launch_bounds__(64) void k_ids_stepscan(int32_t *__restrict__ tot, int N,
                                                     int32_t *__restrict__ step_resets) {
    // single wave, serial over chunks of 64 steps; N is small (hundreds..thousands)
    const int lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < N; base += 64) {
        const int j = base + lane;
        const int v = j < N ? tot[j] : 0;
        if (step_resets && j < N) step_resets[j] = v;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int n = __shfl_up(incl, off, 64);
            if (lane >= off) incl += n;
        }
        if (j < N) tot[j] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) tot[N] = carry;
}

// SCAN = true: the per-step totals are still unscanned; every workgroup scans them into LDS itself
// (N ints - cheaper than a separate single-wave kernel and its launch gap).  SCAN = false: scratch
// already holds the exclusive prefix (k_ids_stepscan; used when N is too large for LDS).
template <bool SCAN>
__global__ __launch_bounds__(kBlock) void k_ids_assign(ml2048_state st, int64_t M, int N, int64_t W,
                                                       const uint64_t *__restrict__ flags,
                                                       const int32_t *__restrict__ last_reset,
                                                       const int32_t *__restrict__ scratch,
                                                       int32_t *__restrict__ game_count, int32_t gc_snapshot_slot) {
    extern __shared__ int32_t step_lds[]; // SCAN: [N + 1] exclusive prefix of the per-step totals
    __shared__ int32_t chunk_excl[kBlock + 1];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int32_t *tot = scratch + (int64_t)N * W;
    const int32_t *step_base = tot;
    if (SCAN) {
        const int tid = threadIdx.x;
        const int per = (N + kBlock - 1) / kBlock;
        const int lo = tid * per < N ? tid * per : N, hi = lo + per < N ? lo + per : N;
        int sum = 0;
        for (int j = lo; j < hi; ++j) sum += tot[j];
        chunk_excl[tid + 1] = sum;
        if (tid == 0) chunk_excl[0] = 0;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) { // inclusive Hillis-Steele over the 256 chunk sums
            const int v = (tid + 1 > off) ? chunk_excl[tid + 1 - off] : 0;
            __syncthreads();
            chunk_excl[tid + 1] += v;
            __syncthreads();
        }
        int run = chunk_excl[tid];
        for (int j = lo; j < hi; ++j) {
            step_lds[j] = run;
            run += tot[j];
        }
        if (tid == kBlock - 1) step_lds[N] = chunk_excl[kBlock];
        __syncthreads();
        step_base = step_lds;
    }
    const int32_t gc0 = scratch[gc_snapshot_slot];
    if (i == 0) *game_count = gc0 + step_base[N];
    if (i >= M) return;
    const int t = last_reset[i];
    if (t < 0) return;
    const int64_t w = i >> 6;
    const unsigned long long f = flags[(int64_t)t * W + w];
    const int rank = scratch[(int64_t)t * W + w] + __popcll(f & ((1ull << (i & 63)) - 1ull));
    st.id[i] = gc0 + step_base[t] + rank;
}

// ---- episode statistics of the first `rounds` games (eval_perf.py:64-102 + replay.py:110-232) ------------
// The reference evaluates a policy by recording every game from its deal to its termination
// (ReplayRecorder: buffer.steps += 1 per on_stepped call, replay.py:167; the last appended score is the
// final score, :186-192) and, for the games whose id < rounds, adding (1, steps, score) to the entry of their
// max tile (eval_perf.py:88-99) - "the first N games, not the first N terminated games" (README.md:112-118).
// Here that is one pass after every step: `age` counts the steps a slot's current game has lived (a new id in
// the slot restarts it), and a game is folded into the table in the step that terminates it.
constexpr int kStatsBlock = 1024;
__global__ __launch_bounds__(kStatsBlock) void k_episode_stats(ml2048_state st, int64_t M, int32_t *__restrict__ age,
                                                          int32_t *__restrict__ seen_id, int32_t id_base, int32_t rounds,
                                                          unsigned long long *__restrict__ count,
                                                          unsigned long long *__restrict__ step_sum,
                                                          unsigned long long *__restrict__ score_sum) {
    // Terminations are rare per board but land on a handful of table entries: they are accumulated per
    // workgroup in LDS over a grid-stride loop and flushed with at most 54 global atomics per workgroup
    // (one global atomic per terminated board serialised on ~10 addresses: 186 us at 4 M boards; 2048 small
    // workgroups still 52 us; 512 workgroups of 1024 threads: see profiles/).
    __shared__ unsigned long long acc[3][18];
    if (threadIdx.x < 54) acc[threadIdx.x / 18][threadIdx.x % 18] = 0ull;
    __syncthreads();
    // four boards per lane: 16-byte accesses to id / age / seen_id, 4 bytes of flags
    auto one = [&](int64_t i, int32_t id, int32_t a, int32_t &seen, uint32_t term) -> int32_t {
        if (id != seen) { // a new game was dealt into this slot
            a = 0;
            seen = id;
        }
        if (a >= 0) { // a < 0: already folded in, waiting for its re-deal
            a += 1;
            if (term) {
                if ((uint32_t)(id - id_base) < (uint32_t)rounds) { // id_base <= id < id_base + rounds
                    const Board b = load_board(st.board, i);
                    const uint32_t mc = max_cell(b);
                    atomicAdd(&acc[0][mc], 1ull);
                    atomicAdd(&acc[1][mc], (unsigned long long)a);
                    // scores are integer-valued float32 (sums of tile values), so the integer sum is exact
                    atomicAdd(&acc[2][mc], (unsigned long long)(long long)st.score[i]);
                }
                a = -1;
            }
        }
        return a;
    };
    const int64_t quads = M / 4;
    for (int64_t q = (int64_t)blockIdx.x * kStatsBlock + threadIdx.x; q < quads; q += (int64_t)gridDim.x * kStatsBlock) {
        const int4 id4 = reinterpret_cast<const int4 *>(st.id)[q];
        int4 a4 = reinterpret_cast<const int4 *>(age)[q];
        int4 s4 = reinterpret_cast<const int4 *>(seen_id)[q];
        const uint32_t t4 = reinterpret_cast<const uint32_t *>(st.terminated)[q];
        const int4 s4_in = s4;
        a4.x = one(4 * q + 0, id4.x, a4.x, s4.x, t4 & 0xFFu);
        a4.y = one(4 * q + 1, id4.y, a4.y, s4.y, (t4 >> 8) & 0xFFu);
        a4.z = one(4 * q + 2, id4.z, a4.z, s4.z, (t4 >> 16) & 0xFFu);
        a4.w = one(4 * q + 3, id4.w, a4.w, s4.w, t4 >> 24);
        reinterpret_cast<int4 *>(age)[q] = a4;
        if (s4.x != s4_in.x || s4.y != s4_in.y || s4.z != s4_in.z || s4.w != s4_in.w) reinterpret_cast<int4 *>(seen_id)[q] = s4;
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(M - 4 * quads)) { // the last M % 4 boards
        const int64_t i = 4 * quads + threadIdx.x;
        int32_t seen = seen_id[i];
        age[i] = one(i, st.id[i], age[i], seen, st.terminated[i]);
        seen_id[i] = seen;
    }
    __syncthreads();
    if (threadIdx.x < 54) {
        const unsigned long long v = acc[threadIdx.x / 18][threadIdx.x % 18];
        if (v) {
            unsigned long long *dst = threadIdx.x < 18 ? count : (threadIdx.x < 36 ? step_sum : score_sum);
            atomicAdd(dst + threadIdx.x % 18, v);
        }
    }
}

// ---- transition record (run_train3.py:125-155) -------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_record(ml2048_state st, int64_t M, const uint8_t *__restrict__ actions,
                                                   const float *__restrict__ logp, int8_t *__restrict__ o_state,
                                                   uint8_t *__restrict__ o_valid, int8_t *__restrict__ o_action,
                                                   float *__restrict__ o_logp, float *__restrict__ o_reward,
                                                   int8_t *__restrict__ o_next_state, uint8_t *__restrict__ o_next_valid,
                                                   int32_t *__restrict__ o_step, uint8_t *__restrict__ o_term) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    if (o_state) reinterpret_cast<uint4 *>(o_state)[i] = reinterpret_cast<const uint4 *>(st.prev_board)[i];
    if (o_next_state) reinterpret_cast<uint4 *>(o_next_state)[i] = reinterpret_cast<const uint4 *>(st.board)[i];
    if (o_valid) reinterpret_cast<uint32_t *>(o_valid)[i] = reinterpret_cast<const uint32_t *>(st.prev_valid)[i];
    if (o_next_valid) reinterpret_cast<uint32_t *>(o_next_valid)[i] = reinterpret_cast<const uint32_t *>(st.valid)[i];
    if (o_reward) o_reward[i] = st.reward[i];
    if (o_step) o_step[i] = st.step[i];
    if (o_term) o_term[i] = st.terminated[i] != 0;
    if (o_action && actions) o_action[i] = (int8_t)actions[i];
    if (o_logp && logp) o_logp[i] = logp[i];
}

// ---- GAE scan (gae.py:50, :58-68) ---------------------------------------------------------------------
// One lane per (use, game) pair walks the step dimension backwards; lanes of a wave are consecutive
// games, so every access is a coalesced 256-byte row of the [U][S][G] tensors.
__global__ __launch_bounds__(kBlock) void k_gae(int64_t U, int64_t S, int64_t G, const float *__restrict__ v0,
                                                const float *__restrict__ v1, const float *__restrict__ reward,
                                                const uint8_t *__restrict__ terminated, float gamma, float coef,
                                                float *__restrict__ adv, float *__restrict__ delta_out) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= U * G) return;
    const int64_t u = j / G, g = j - u * G;
    float tmp = 0.0f;
    for (int64_t s = S - 1; s >= 0; --s) {
        const int64_t k = (u * S + s) * G + g;
        const float mask = terminated[k] ? 0.0f : 1.0f;
        // delta = gamma * v1 * mask + reward - v0, evaluated left to right in float32 (:50)
        const float d = __fsub_rn(__fadd_rn(__fmul_rn(__fmul_rn(gamma, v1[k]), mask), reward[k]), v0[k]);
        // tmp = tmp * coef; tmp = delta + tmp * mask (:66-67)
        tmp = __fadd_rn(d, __fmul_rn(__fmul_rn(tmp, coef), mask));
        adv[k] = tmp;
        if (delta_out) delta_out[k] = d;
    }
}

// spawn="rocrand": a tile is a 2 when the low half of its word is below ceil(two_prob * 2^16) (board_ops.hpp spawn_word)
uint64_t two_threshold(double two_prob) {
    double t = two_prob * 65536.0;
    if (!(t > 0.0)) return 0;
    if (t >= 65536.0) return 65536ull;
    uint64_t c = (uint64_t)t;
    if ((double)c < t) c += 1;
    return c;
}

template <bool FAST>
int launch_step(hipStream_t s, int64_t M, int64_t idx_base, const ml2048_state *st, const void *actions,
                int action_dtype, int reward_kind, double two_prob, int64_t row_base,
                const uint8_t *randperm, const float *randfloat, uint64_t seed, uint64_t counter) {
    const dim3 grid((unsigned)cdiv(M, kBlock)), block(kBlock);
    const uint64_t thr = two_threshold(two_prob);
#define ML_LAUNCH(K)                                                                                   \
    hipLaunchKernelGGL((k_step<K, FAST>), grid, block, 0, s, *st, M, idx_base, actions, action_dtype, \
                       two_prob, row_base, randperm, randfloat, seed, counter, thr)
    switch (reward_kind) {
    case ML2048_REWARD_NORMAL: ML_LAUNCH(REWARD_NORMAL); break;
    case ML2048_REWARD_IMPROVED: ML_LAUNCH(REWARD_IMPROVED); break;
    case ML2048_REWARD_RANK: ML_LAUNCH(REWARD_RANK); break;
    case ML2048_REWARD_MAXCELL: ML_LAUNCH(REWARD_MAXCELL); break;
    default: return fail(ML2048_E_INVALID, "reward_kind=%d", reward_kind);
    }
#undef ML_LAUNCH
    return check_launch("ml2048_step");
}

template <bool FAST>
int launch_prepare(hipStream_t s, int64_t M, int64_t idx_base, const ml2048_state *st,
                   const uint8_t *randperm, const float *randfloat, double two_prob, int64_t row_base,
                   uint64_t seed, uint64_t counter, int32_t *game_count, int32_t *n_reset,
                   int64_t *reset_indices, int32_t *scratch) {
    if (!game_count || !n_reset || !scratch)
        return fail(ML2048_E_INVALID, "game_count, n_reset and scratch must be non-NULL");
    if (M <= 16 * 1024) { // one workgroup does count + scan + deal
        hipLaunchKernelGGL((k_prep_small<FAST>), dim3(1), dim3(1024), 0, s, *st, M, (int)cdiv(M, 1024), idx_base,
                           randperm, randfloat, two_prob, row_base, seed, counter, two_threshold(two_prob),
                           game_count, n_reset, reset_indices);
        return check_launch("ml2048_prepare");
    }
    const int nblk = (int)cdiv(M, kPrepBlock);
    if (nblk <= kPrepFusedBlocks) { // single pass with decoupled look-back; scratch = 64-bit status words
        if ((uintptr_t)scratch % 8) return fail(ML2048_E_ALIGN, "scratch must be 8-byte aligned");
        static std::atomic<uint32_t> g_epoch{1};
        uint32_t epoch = g_epoch.fetch_add(1, std::memory_order_relaxed) & 0x3FFFFFFFu;
        if (epoch == 0) epoch = g_epoch.fetch_add(1, std::memory_order_relaxed) & 0x3FFFFFFFu; // 0 = "never written"
        hipLaunchKernelGGL((k_prep_fused<FAST>), dim3(nblk), dim3(kPrepBlock), 0, s, *st, M, idx_base, randperm, randfloat,
                           two_prob, row_base, seed, counter, two_threshold(two_prob),
                           reinterpret_cast<unsigned long long *>(scratch), epoch, game_count, n_reset, reset_indices);
        return check_launch("ml2048_prepare");
    }
    hipLaunchKernelGGL(k_prep_count, dim3(nblk), dim3(kPrepBlock), 0, s, st->terminated, M, scratch);
    hipLaunchKernelGGL(k_prep_scan, dim3(1), dim3(1024), 0, s, scratch, nblk, game_count);
    hipLaunchKernelGGL((k_prep_deal<FAST>), dim3(nblk), dim3(kPrepBlock), 0, s, *st, M, idx_base, randperm,
                       randfloat, two_prob, row_base, seed, counter, two_threshold(two_prob), scratch, nblk,
                       game_count, n_reset, reset_indices);
    return check_launch("ml2048_prepare");
}

} // namespace

#ifdef ML2048_ROLLOUT_PLAIN_TU
int ML2048_ROLLOUT_PLAIN_TU(int reward_kind, bool fast, bool inv_tables, const ml2048_plain_launch &a) {
    const ml2048_traj traj = {};
#define ML_PLAIN(K, F, I)                                                                                              \
    hipExtLaunchKernelGGL((k_rollout<K, F, false, I>), a.grid, a.block, 0, a.stream, a.ev_start, a.ev_stop, 0, a.st, a.M,  \
                          a.idx_base, a.N, a.perm_epochs, a.two_masks, a.schedule, a.n_epochs, a.masked, a.act_seed,   \
                          a.act_counter0, (uint8_t *)nullptr, (uint64_t *)nullptr, a.reset_flags, a.flag_words,        \
                          a.last_reset, a.spawn_seed, a.spawn_counter0, a.two_threshold, traj, a.up_src, a.up_dst,     \
                          a.up_n16, a.up_blocks, a.up_tab16, a.row_lut, a.board_waves)
#define ML_PLAIN_KIND(K)                 \
    if (fast) {                          \
        ML_PLAIN(K, true, false);        \
    } else if (inv_tables) {             \
        ML_PLAIN(K, false, true);        \
    } else {                             \
        ML_PLAIN(K, false, false);       \
    }
    switch (reward_kind) {
    case ML2048_REWARD_NORMAL: ML_PLAIN_KIND(REWARD_NORMAL); break;
    case ML2048_REWARD_IMPROVED: ML_PLAIN_KIND(REWARD_IMPROVED); break;
    case ML2048_REWARD_RANK: ML_PLAIN_KIND(REWARD_RANK); break;
    case ML2048_REWARD_MAXCELL: ML_PLAIN_KIND(REWARD_MAXCELL); break;
    default: return ML2048_E_INVALID;
    }
#undef ML_PLAIN_KIND
#undef ML_PLAIN
    return 0;
}
#else
// ================================================================================================
// shared with the other kernel files of the library (step_prepare.hip): one error buffer, one set of state checks
__attribute__((visibility("hidden"))) int ml2048_internal_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
__attribute__((visibility("hidden"))) int ml2048_internal_check_state(int64_t M, const ml2048_state *st) { return check_state(M, st); }
__attribute__((visibility("hidden"))) uint64_t ml2048_internal_two_threshold(double two_prob) { return two_threshold(two_prob); }

extern "C" {

int ml2048_abi_version(void) { return ML2048_ABI_VERSION; }

const char *ml2048_last_error(void) { return g_err; }

int64_t ml2048_scratch_ints(int64_t M) { return M > 0 ? 2 * cdiv(M, kPrepBlock) + 2 : 2; }

int ml2048_reset(void *stream, int64_t M, const ml2048_state *st) {
    if (int rc = check_state(M, st)) return rc;
    hipLaunchKernelGGL(k_reset, dim3((unsigned)cdiv(M, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, *st, M);
    return check_launch("ml2048_reset");
}

int ml2048_prepare(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st,
                   const uint8_t *randperm, const float *randfloat, double two_prob, int64_t row_base,
                   int32_t *game_count, int32_t *n_reset, int64_t *reset_indices, int32_t *scratch) {
    if (int rc = check_state(M, st)) return rc;
    if (!randperm || !randfloat) return fail(ML2048_E_INVALID, "randperm/randfloat is NULL");
    if ((uintptr_t)randperm % 16) return fail(ML2048_E_ALIGN, "randperm must be 16-byte aligned");
    if (row_base < 0 || idx_base < 0) return fail(ML2048_E_INVALID, "row_base/idx_base must be >= 0");
    return launch_prepare<false>((hipStream_t)stream, M, idx_base, st, randperm, randfloat, two_prob, row_base, 0, 0,
                                 game_count, n_reset, reset_indices, scratch);
}

int ml2048_prepare_fast(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, double two_prob,
                        uint64_t seed, uint64_t counter, int32_t *game_count, int32_t *n_reset,
                        int64_t *reset_indices, int32_t *scratch) {
    if (int rc = check_state(M, st)) return rc;
    if (idx_base < 0) return fail(ML2048_E_INVALID, "idx_base must be >= 0");
    return launch_prepare<true>((hipStream_t)stream, M, idx_base, st, nullptr, nullptr, two_prob, 0, seed, counter,
                                game_count, n_reset, reset_indices, scratch);
}

int ml2048_step(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, const void *actions,
                int action_dtype, int reward_kind, double two_prob, int64_t row_base,
                const uint8_t *randperm, const float *randfloat) {
    if (int rc = check_state(M, st)) return rc;
    if (!actions || !randperm || !randfloat) return fail(ML2048_E_INVALID, "actions/randperm/randfloat is NULL");
    if ((uintptr_t)randperm % 16) return fail(ML2048_E_ALIGN, "randperm must be 16-byte aligned");
    if (action_dtype < 0 || action_dtype > 2) return fail(ML2048_E_INVALID, "action_dtype=%d", action_dtype);
    if (row_base < 0 || idx_base < 0) return fail(ML2048_E_INVALID, "row_base/idx_base must be >= 0");
    return launch_step<false>((hipStream_t)stream, M, idx_base, st, actions, action_dtype, reward_kind, two_prob,
                              row_base, randperm, randfloat, 0, 0);
}

int ml2048_step_wave(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, const void *actions,
                     int action_dtype, int reward_kind, double two_prob, int64_t row_base,
                     const uint8_t *randperm, const float *randfloat) {
    if (int rc = check_state(M, st)) return rc;
    if (!actions || !randperm || !randfloat) return fail(ML2048_E_INVALID, "actions/randperm/randfloat is NULL");
    if ((uintptr_t)randperm % 16) return fail(ML2048_E_ALIGN, "randperm must be 16-byte aligned");
    if (action_dtype < 0 || action_dtype > 2) return fail(ML2048_E_INVALID, "action_dtype=%d", action_dtype);
    if (reward_kind < 0 || reward_kind > 3) return fail(ML2048_E_INVALID, "reward_kind=%d", reward_kind);
    if (row_base < 0 || idx_base < 0) return fail(ML2048_E_INVALID, "row_base/idx_base must be >= 0");
    int64_t blocks = cdiv(M, kBlock / 64);
    if (blocks > 256 * 8) blocks = 256 * 8; // 8 workgroups per CU, grid-stride over the boards
    hipLaunchKernelGGL(k_step_wave, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, *st, M, idx_base,
                       actions, action_dtype, reward_kind, two_prob, row_base, randperm, randfloat);
    return check_launch("ml2048_step_wave");
}

int ml2048_step_fast(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, const void *actions,
                     int action_dtype, int reward_kind, double two_prob, uint64_t seed, uint64_t counter) {
    if (int rc = check_state(M, st)) return rc;
    if (!actions) return fail(ML2048_E_INVALID, "actions is NULL");
    if (action_dtype < 0 || action_dtype > 2) return fail(ML2048_E_INVALID, "action_dtype=%d", action_dtype);
    if (idx_base < 0) return fail(ML2048_E_INVALID, "idx_base must be >= 0");
    return launch_step<true>((hipStream_t)stream, M, idx_base, st, actions, action_dtype, reward_kind, two_prob, 0,
                             nullptr, nullptr, seed, counter);
}

int ml2048_sample_actions(void *stream, int64_t M, int64_t idx_base, const uint8_t *valid, uint64_t seed,
                          uint64_t counter, int masked, uint8_t *actions, float *logp) {
    if (M <= 0) return fail(ML2048_E_INVALID, "M=%lld must be positive", (long long)M);
    if (!actions || (masked && !valid)) return fail(ML2048_E_INVALID, "actions/valid is NULL");
    if (masked && (uintptr_t)valid % 4) return fail(ML2048_E_ALIGN, "valid must be 4-byte aligned");
    hipLaunchKernelGGL(k_sample, dim3((unsigned)cdiv(M, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, valid, M,
                       idx_base, seed, counter, masked, actions, logp);
    return check_launch("ml2048_sample_actions");
}

int ml2048_encode_onehot(void *stream, int64_t M, const uint8_t *board, float *onehot, int64_t *board_i64,
                         int32_t *bad_count) {
    if (M <= 0) return fail(ML2048_E_INVALID, "M=%lld must be positive", (long long)M);
    if (!board || !onehot) return fail(ML2048_E_INVALID, "board/onehot is NULL");
    if ((uintptr_t)board % 16 || (uintptr_t)onehot % 16) return fail(ML2048_E_ALIGN, "board/onehot must be 16-byte aligned");
    // 64 boards per wave iteration; cap the grid and stride (each wave keeps >= 1 chunk)
    int64_t waves = cdiv(M, 64);
    int64_t blocks = cdiv(waves, kBlock / 64);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_onehot, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, board, M, onehot,
                       board_i64, bad_count);
    return check_launch("ml2048_encode_onehot");
}

int ml2048_hist_maxcell(void *stream, int64_t M, const uint8_t *board, const uint8_t *terminated, int32_t *counts) {
    if (M <= 0) return fail(ML2048_E_INVALID, "M=%lld must be positive", (long long)M);
    if (!board || !counts) return fail(ML2048_E_INVALID, "board/counts is NULL");
    if ((uintptr_t)board % 16) return fail(ML2048_E_ALIGN, "board must be 16-byte aligned");
    int64_t blocks = cdiv(M, kBlock);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_hist, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, board, terminated, M, counts);
    return check_launch("ml2048_hist_maxcell");
}

int ml2048_pack_records(void *stream, int64_t M, const ml2048_state *st, void *records) {
    if (int rc = check_state(M, st)) return rc;
    if (!records || (uintptr_t)records % 16) return fail(ML2048_E_ALIGN, "records must be non-NULL, 16-byte aligned");
    hipLaunchKernelGGL(k_pack, dim3((unsigned)cdiv(M, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, *st, M, (uint4 *)records);
    return check_launch("ml2048_pack_records");
}

int ml2048_unpack_records(void *stream, int64_t M, const ml2048_state *st, const void *records) {
    if (int rc = check_state(M, st)) return rc;
    if (!records || (uintptr_t)records % 16) return fail(ML2048_E_ALIGN, "records must be non-NULL, 16-byte aligned");
    hipLaunchKernelGGL(k_unpack, dim3((unsigned)cdiv(M, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, *st, M, (const uint4 *)records);
    return check_launch("ml2048_unpack_records");
}

int ml2048_digest(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int with_id, uint64_t *out) {
    if (int rc = check_state(M, st)) return rc;
    if (!out || (uintptr_t)out % 8) return fail(ML2048_E_ALIGN, "out must be non-NULL, 8-byte aligned");
    hipLaunchKernelGGL(k_digest, dim3((unsigned)cdiv(M, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, *st, M, idx_base, with_id, out);
    return check_launch("ml2048_digest");
}

int ml2048_debug_philox(void *stream, int64_t n, int64_t idx_base, uint64_t seed, uint64_t counter, uint32_t *out) {
    if (n <= 0 || !out || (uintptr_t)out % 16) return fail(ML2048_E_INVALID, "n/out");
    hipLaunchKernelGGL(k_debug_philox, dim3((unsigned)cdiv(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n, idx_base, seed, counter, out);
    return check_launch("ml2048_debug_philox");
}

int64_t ml2048_rollout_flag_words(int64_t M) { return M > 0 ? cdiv(M, kBlock) * (kBlock / 64) : 0; }

int64_t ml2048_rollout_scratch_ints(int64_t M, int32_t N) {
    if (M <= 0 || N <= 0) return 0;
    return (int64_t)N * ml2048_rollout_flag_words(M) + N + 2;
}

// schedule upload: 16 B per lane straight from device-accessible host memory
__global__ __launch_bounds__(kBlock) void k_upload(uint4 *__restrict__ dst, const uint4 *__restrict__ src, int64_t n16,
                                                   int64_t tab16) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (int64_t)gridDim.x * kBlock)
        upload_unit(dst, src, i, tab16);
}

int ml2048_upload(void *stream, void *dst, const void *src_host, int64_t nbytes, int64_t table_offset) {
    if (!dst || !src_host || nbytes <= 0) return fail(ML2048_E_INVALID, "dst/src_host/nbytes");
    if (table_offset < 0 || table_offset > nbytes || (table_offset < nbytes && (table_offset % 16 || (nbytes - table_offset) % 16)))
        return fail(ML2048_E_INVALID, "table_offset=%lld (nbytes=%lld)", (long long)table_offset, (long long)nbytes);
    if (((uintptr_t)dst | (uintptr_t)src_host) % 16) return fail(ML2048_E_ALIGN, "dst and src_host must be 16-byte aligned");
    const int64_t n16 = (nbytes + 15) / 16;
    int64_t blocks = cdiv(n16, kBlock);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_upload, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, (uint4 *)dst,
                       (const uint4 *)src_host, n16, table_offset >= nbytes ? n16 : table_offset / 16);
    return check_launch("ml2048_upload");
}

// ---- kernel timer: a HIP event pair attached to the rollout DISPATCH itself (hipExtLaunchKernel),
// so that the measured interval is the kernel and nothing else (events recorded around a launch also
// time two marker packets, ~4 % at C2) ---------------------------------------------------------------
struct ml2048_timer_impl {
    hipEvent_t start, stop;
    bool used;
};
static thread_local ml2048_timer_impl *g_armed_timer = nullptr;
// ml2048_rollout_chain_ids: the rollout's marker event rides on the kernel's own dispatch (its completion signal)
// instead of a marker packet behind it - one-shot, consumed by the next launch_rollout of this thread
static thread_local hipEvent_t g_attach_stop_event = nullptr;
static thread_local const uint16_t *g_attach_row_lut = nullptr; // one-shot: the next rollout launch's line table
static thread_local bool g_stop_event_attached = false;
// Every public rollout entry point holds one of these: whatever the call does - launch, or fail an argument check before
// it gets that far - the thread's one-shots (armed timer, attached marker event, attached line table) are spent when it
// returns, so none of them can ride on an unrelated later launch.
struct rollout_one_shots {
    ~rollout_one_shots() {
        g_armed_timer = nullptr;
        g_attach_stop_event = nullptr;
        g_attach_row_lut = nullptr;
    }
};

int ml2048_timer_create(void **timer) {
    if (!timer) return fail(ML2048_E_INVALID, "timer is NULL");
    ml2048_timer_impl *t = new ml2048_timer_impl{nullptr, nullptr, false};
    hipError_t e = hipEventCreate(&t->start);
    if (e == hipSuccess) e = hipEventCreate(&t->stop);
    if (e != hipSuccess) {
        delete t;
        fail(ML2048_E_INVALID, "hipEventCreate: %s", hipGetErrorString(e));
        return (int)e; // > 0: a hipError_t, as everywhere in this ABI
    }
    *timer = t;
    return 0;
}

int ml2048_timer_arm(void *timer) {
    if (!timer) return fail(ML2048_E_INVALID, "timer is NULL");
    g_armed_timer = (ml2048_timer_impl *)timer;
    return 0;
}

int ml2048_timer_elapsed_ms(void *timer, float *ms) {
    ml2048_timer_impl *t = (ml2048_timer_impl *)timer;
    if (!t || !ms) return fail(ML2048_E_INVALID, "timer/ms is NULL");
    if (!t->used) return fail(ML2048_E_INVALID, "timer was never attached to a launch");
    hipError_t e = hipEventSynchronize(t->stop);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, t->start, t->stop);
    if (e != hipSuccess) {
        fail(ML2048_E_INVALID, "hipEventElapsedTime: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

void ml2048_timer_destroy(void *timer) {
    ml2048_timer_impl *t = (ml2048_timer_impl *)timer;
    if (!t) return;
    if (g_armed_timer == t) g_armed_timer = nullptr;
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
}

} // extern "C"
// the armed one-shot timer, for launches made by the other kernel files (step_prepare.hip)
__attribute__((visibility("hidden"))) bool ml2048_internal_take_timer(hipEvent_t *start, hipEvent_t *stop) {
    ml2048_timer_impl *t = g_armed_timer;
    g_armed_timer = nullptr;
    if (!t) return false;
    t->used = true;
    *start = t->start;
    *stop = t->stop;
    return true;
}
extern "C" {

// largest M that runs the small-M builds of the plain rollout kernels (see ml2048_plain_launch); ML2048_PLAIN_ILP_MAX
// overrides it for A/B runs
static int64_t plain_ilp_max_boards() {
    static const int64_t v = [] {
        const char *e = getenv("ML2048_PLAIN_ILP_MAX");
        // one wave per SIMD: as soon as waves share a SIMD the large-M pass with its LDS tables (line table, re-deal,
        // spawned tile) wins - 98 304 boards 96.6 vs 102.5 us, 131 072: 97.1 vs 103.3, 196 608: 138 vs 151 - while a
        // lone wave pays every look-up's round trip (65 536: 78 vs 68 us; tools/debug/ilpmax_lut.sh, lut_ab.sh)
        return e ? (int64_t)atoll(e) : (int64_t)65536;
    }();
    return v;
}

// compute units of the current device (the plain rollout sizes its workgroups to a CU's share of the boards)
static int cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

static int launch_rollout(hipStream_t s, bool fast, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                          const uint8_t *perm_epochs, const uint32_t *two_masks, const int32_t *schedule, int32_t n_epochs,
                          int reward_kind, int masked, uint64_t act_seed, uint64_t act_counter0, uint8_t *actions_out,
                          uint64_t *digests_out, uint64_t *reset_flags, int32_t *last_reset, uint64_t spawn_seed,
                          uint64_t spawn_counter0, double two_prob, const ml2048_traj *traj_or_null,
                          const void *up_src = nullptr, void *up_dst = nullptr, int64_t up_nbytes = 0,
                          int64_t up_table_offset = -1, bool inv_tables = false, const uint16_t *row_lut = nullptr) {
    ml2048_traj traj = {};
    // the per-thread one-shots (armed timer, attached marker event, attached line table) belong to THIS call whether it
    // launches or fails its checks: none of them may stay armed for an unrelated later launch
    ml2048_timer_impl *const timer = g_armed_timer;
    g_armed_timer = nullptr;
    const hipEvent_t attached_event = g_attach_stop_event;
    g_attach_stop_event = nullptr;
    g_stop_event_attached = false;
    if (!row_lut) row_lut = g_attach_row_lut;
    g_attach_row_lut = nullptr;
    if (up_table_offset < 0) up_table_offset = up_nbytes; // no table units: a plain copy
    if (up_table_offset > up_nbytes || (up_table_offset < up_nbytes && (up_table_offset % 16 || (up_nbytes - up_table_offset) % 16)))
        return fail(ML2048_E_INVALID, "next_table_offset=%lld", (long long)up_table_offset);
    if (up_src || up_dst || up_nbytes) {
        if (!up_src || !up_dst || up_nbytes <= 0) return fail(ML2048_E_INVALID, "next_src_host/next_dst/next_nbytes");
        if (((uintptr_t)up_src | (uintptr_t)up_dst) % 16) return fail(ML2048_E_ALIGN, "next_src_host and next_dst must be 16-byte aligned");
    }
    const int64_t up_n16 = (up_nbytes + 15) / 16;
    const int64_t up_tab16 = up_table_offset >= up_nbytes ? up_n16 : up_table_offset / 16;
    if ((uintptr_t)row_lut % 16) return fail(ML2048_E_ALIGN, "row_lut must be 16-byte aligned");
    if (traj_or_null) {
        traj = *traj_or_null;
        if (((uintptr_t)traj.state | (uintptr_t)traj.next_state) % 16)
            return fail(ML2048_E_ALIGN, "traj.state/next_state must be 16-byte aligned");
        if (((uintptr_t)traj.valid_actions | (uintptr_t)traj.next_valid_actions | (uintptr_t)traj.reward |
             (uintptr_t)traj.step | (uintptr_t)traj.action_log_prob) % 4)
            return fail(ML2048_E_ALIGN, "4-byte trajectory outputs must be 4-byte aligned");
    }
    const dim3 grid((unsigned)cdiv(M, kBlock)), block(kBlock); // the EXTRAS instantiations: 4 board waves per workgroup
    const int64_t W = ml2048_rollout_flag_words(M);
    const uint64_t thr = two_threshold(two_prob);
    const bool extras = actions_out || digests_out || traj_or_null;
    // an armed timer brackets THIS dispatch with its events
    hipEvent_t ev_start = timer ? timer->start : nullptr, ev_stop = timer ? timer->stop : nullptr;
    if (timer) timer->used = true;
    if (!timer && attached_event) { // (a timed dispatch keeps its own pair: the caller records the marker then)
        ev_stop = attached_event;
        g_stop_event_attached = true;
    }
    if (inv_tables && fast) return fail(ML2048_E_UNSUPPORTED, "expanded tables: table-mode rollouts only");
    if (!fast && n_epochs > 65536) // (table rows are addressed by 32-bit byte offsets from the tables' base: 65 536 x 48 KiB < 4 GiB)
        return fail(ML2048_E_INVALID, "n_epochs=%d: at most 65536 table epochs per launch", n_epochs);
    if (extras && up_n16 > 0) return fail(ML2048_E_UNSUPPORTED, "chained upload: plain rollouts only");
    if (!extras) { // the plain instantiations live in their own compilation passes (see ml2048_plain_launch)
        // one workgroup = one compute unit's share of the boards (the LDS line table leaves room for one workgroup per
        // CU): up to 16 board waves, plus the upload wave where there is room for it
        const int64_t waves = cdiv(M, 64);
        const int bw = (int)std::min<int64_t>(kRollMaxWaves, std::max<int64_t>(1, cdiv(waves, cu_count())));
        const unsigned up_mode = up_n16 > 0 ? (bw < kRollMaxWaves ? 1u : 2u) : 0u;
        ml2048_plain_launch a;
        a.grid = dim3((unsigned)cdiv(waves, bw)), a.block = dim3(64u * (unsigned)(bw + (up_mode == 1u ? 1 : 0)));
        a.stream = s, a.ev_start = ev_start, a.ev_stop = ev_stop;
        a.board_waves = bw;
        a.st = *st, a.M = M, a.idx_base = idx_base, a.N = N;
        a.perm_epochs = perm_epochs, a.two_masks = two_masks, a.schedule = schedule, a.n_epochs = n_epochs;
        a.masked = masked, a.act_seed = act_seed, a.act_counter0 = act_counter0;
        a.reset_flags = reset_flags, a.flag_words = W, a.last_reset = last_reset;
        a.spawn_seed = spawn_seed, a.spawn_counter0 = spawn_counter0, a.two_threshold = thr;
        a.up_src = (const uint4 *)up_src, a.up_dst = (uint4 *)up_dst, a.up_n16 = up_n16, a.up_blocks = up_mode, a.up_tab16 = up_tab16;
        const bool small = M <= plain_ilp_max_boards() && bw <= 8; // (the small-M passes' launch bound: 8 + 1 waves)
        // copying the line table into LDS costs ~2.5 us per launch; a step gains ~0.05 us from it in the small-M passes
        // and ~0.3 us in the large-M pass: short rollouts move arithmetically
        a.row_lut = N >= (small ? 64 : 16) ? row_lut : nullptr;
        if (small ? (!fast && !masked ? ml2048_internal_launch_rollout_plain_lone(reward_kind, fast, inv_tables, a)
                                      : ml2048_internal_launch_rollout_plain_ilp(reward_kind, fast, inv_tables, a))
                  : ml2048_internal_launch_rollout_plain_dflt(reward_kind, fast, inv_tables, a))
            return fail(ML2048_E_INVALID, "reward_kind=%d", reward_kind);
        return check_launch("ml2048_rollout");
    }
#define ML_LAUNCH2(K, F) ML_LAUNCH3(K, F, false)
#define ML_LAUNCH3(K, F, I)                                                                                  \
    hipExtLaunchKernelGGL((k_rollout<K, F, true, I>), grid, block, 0, s, ev_start, ev_stop, 0, *st, M, idx_base, N, perm_epochs, two_masks, \
                       schedule, n_epochs, masked, act_seed, act_counter0, actions_out, digests_out,         \
                       reset_flags, W,                                                                       \
                       last_reset, spawn_seed, spawn_counter0, thr, traj, (const uint4 *)nullptr, (uint4 *)nullptr, (int64_t)0, 0u, (int64_t)0, \
                       (const uint16_t *)nullptr, kBlock / 64)
#define ML_KIND(F)                                                                   \
    switch (reward_kind) {                                                           \
    case ML2048_REWARD_NORMAL: ML_LAUNCH2(REWARD_NORMAL, F); break;                  \
    case ML2048_REWARD_IMPROVED: ML_LAUNCH2(REWARD_IMPROVED, F); break;              \
    case ML2048_REWARD_RANK: ML_LAUNCH2(REWARD_RANK, F); break;                      \
    case ML2048_REWARD_MAXCELL: ML_LAUNCH2(REWARD_MAXCELL, F); break;                \
    default: return fail(ML2048_E_INVALID, "reward_kind=%d", reward_kind);           \
    }
    if (fast) {
        ML_KIND(true)
    } else if (inv_tables) { // optional outputs on expanded tables: dot-product spawn, re-deal and spawned tile by LDS look-ups
        switch (reward_kind) {
        case ML2048_REWARD_NORMAL: ML_LAUNCH3(REWARD_NORMAL, false, true); break;
        case ML2048_REWARD_IMPROVED: ML_LAUNCH3(REWARD_IMPROVED, false, true); break;
        case ML2048_REWARD_RANK: ML_LAUNCH3(REWARD_RANK, false, true); break;
        case ML2048_REWARD_MAXCELL: ML_LAUNCH3(REWARD_MAXCELL, false, true); break;
        default: return fail(ML2048_E_INVALID, "reward_kind=%d", reward_kind);
        }
    } else {
        ML_KIND(false)
    }
#undef ML_KIND
#undef ML_LAUNCH3
#undef ML_LAUNCH2
    return check_launch("ml2048_rollout");
}

int ml2048_rollout(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                   const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                   const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed, uint64_t act_counter0,
                   uint8_t *actions_out, uint64_t *digests_out, uint64_t *reset_flags, int32_t *last_reset) {
    const rollout_one_shots spent_on_return;
    if (int rc = check_state(M, st)) return rc;
    if (N <= 0 || n_epochs <= 0) return fail(ML2048_E_INVALID, "N=%d n_epochs=%d must be positive", N, n_epochs);
    if (!perm_epochs || !two_masks || !schedule || !reset_flags || !last_reset)
        return fail(ML2048_E_INVALID, "perm_epochs/two_masks/schedule/reset_flags/last_reset is NULL");
    if ((uintptr_t)perm_epochs % 16) return fail(ML2048_E_ALIGN, "perm_epochs must be 16-byte aligned");
    if (idx_base < 0) return fail(ML2048_E_INVALID, "idx_base must be >= 0");
    return launch_rollout((hipStream_t)stream, false, M, idx_base, st, N, perm_epochs, two_masks, schedule, n_epochs, reward_kind,
                          masked, act_seed, act_counter0, actions_out, digests_out, reset_flags, last_reset, 0, 0, 0.0,
                          nullptr);
}

int ml2048_rollout_chain(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                         const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                         const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed, uint64_t act_counter0,
                         uint64_t *reset_flags, int32_t *last_reset, const void *next_src_host, void *next_dst,
                         int64_t next_nbytes, int64_t next_table_offset) {
    const rollout_one_shots spent_on_return;
    if (int rc = check_state(M, st)) return rc;
    if (N <= 0 || n_epochs <= 0) return fail(ML2048_E_INVALID, "N=%d n_epochs=%d must be positive", N, n_epochs);
    if (!perm_epochs || !two_masks || !schedule || !reset_flags || !last_reset)
        return fail(ML2048_E_INVALID, "perm_epochs/two_masks/schedule/reset_flags/last_reset is NULL");
    if ((uintptr_t)perm_epochs % 16) return fail(ML2048_E_ALIGN, "perm_epochs must be 16-byte aligned");
    if (idx_base < 0) return fail(ML2048_E_INVALID, "idx_base must be >= 0");
    return launch_rollout((hipStream_t)stream, false, M, idx_base, st, N, perm_epochs, two_masks, schedule, n_epochs, reward_kind,
                          masked, act_seed, act_counter0, nullptr, nullptr, reset_flags, last_reset, 0, 0, 0.0, nullptr,
                          next_src_host, next_dst, next_nbytes, next_table_offset, /*inv_tables=*/true);
}

int ml2048_rollout_traj(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                        const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                        const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed, uint64_t act_counter0,
                        uint64_t *reset_flags, int32_t *last_reset, const ml2048_traj *traj) {
    const rollout_one_shots spent_on_return;
    if (int rc = check_state(M, st)) return rc;
    if (N <= 0 || n_epochs <= 0) return fail(ML2048_E_INVALID, "N=%d n_epochs=%d must be positive", N, n_epochs);
    if (!perm_epochs || !two_masks || !schedule || !reset_flags || !last_reset || !traj)
        return fail(ML2048_E_INVALID, "perm_epochs/two_masks/schedule/reset_flags/last_reset/traj is NULL");
    if ((uintptr_t)perm_epochs % 16) return fail(ML2048_E_ALIGN, "perm_epochs must be 16-byte aligned");
    if (idx_base < 0) return fail(ML2048_E_INVALID, "idx_base must be >= 0");
    return launch_rollout((hipStream_t)stream, false, M, idx_base, st, N, perm_epochs, two_masks, schedule, n_epochs, reward_kind,
                          masked, act_seed, act_counter0, nullptr, nullptr, reset_flags, last_reset, 0, 0, 0.0, traj);
}

int ml2048_rollout_traj_expanded(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                                 const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                                 const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed, uint64_t act_counter0,
                                 uint64_t *reset_flags, int32_t *last_reset, const ml2048_traj *traj) {
    const rollout_one_shots spent_on_return;
    if (int rc = check_state(M, st)) return rc;
    if (N <= 0 || n_epochs <= 0) return fail(ML2048_E_INVALID, "N=%d n_epochs=%d must be positive", N, n_epochs);
    if (!perm_epochs || !two_masks || !schedule || !reset_flags || !last_reset || !traj)
        return fail(ML2048_E_INVALID, "perm_epochs/two_masks/schedule/reset_flags/last_reset/traj is NULL");
    if ((uintptr_t)perm_epochs % 16) return fail(ML2048_E_ALIGN, "perm_epochs must be 16-byte aligned");
    if (idx_base < 0) return fail(ML2048_E_INVALID, "idx_base must be >= 0");
    return launch_rollout((hipStream_t)stream, false, M, idx_base, st, N, perm_epochs, two_masks, schedule, n_epochs, reward_kind,
                          masked, act_seed, act_counter0, nullptr, nullptr, reset_flags, last_reset, 0, 0, 0.0, traj,
                          nullptr, nullptr, 0, -1, /*inv_tables=*/true);
}

int ml2048_rollout_fast(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                        double two_prob, uint64_t spawn_seed, uint64_t spawn_counter0, int reward_kind, int masked,
                        uint64_t act_seed, uint64_t act_counter0, uint8_t *actions_out, uint64_t *digests_out,
                        uint64_t *reset_flags, int32_t *last_reset) {
    const rollout_one_shots spent_on_return;
    if (int rc = check_state(M, st)) return rc;
    if (N <= 0) return fail(ML2048_E_INVALID, "N=%d must be positive", N);
    if (!reset_flags || !last_reset) return fail(ML2048_E_INVALID, "reset_flags/last_reset is NULL");
    if (idx_base < 0) return fail(ML2048_E_INVALID, "idx_base must be >= 0");
    return launch_rollout((hipStream_t)stream, true, M, idx_base, st, N, nullptr, nullptr, nullptr, 1, reward_kind, masked,
                          act_seed, act_counter0, actions_out, digests_out, reset_flags, last_reset, spawn_seed,
                          spawn_counter0, two_prob, nullptr);
}

int ml2048_rollout_finish_ids(void *stream, int64_t M, const ml2048_state *st, int32_t N,
                              const uint64_t *reset_flags, const int32_t *last_reset, int32_t *game_count,
                              int32_t *step_resets, int32_t *scratch) {
    if (int rc = check_state(M, st)) return rc;
    if (N <= 0 || !reset_flags || !last_reset || !game_count || !scratch)
        return fail(ML2048_E_INVALID, "N/reset_flags/last_reset/game_count/scratch");
    hipStream_t s = (hipStream_t)stream;
    const int64_t W = ml2048_rollout_flag_words(M);
    const int64_t snap = (int64_t)N * W + N + 1;
    if (snap > 0x7FFFFFFF) return fail(ML2048_E_UNSUPPORTED, "N*M too large for one finish_ids call");
    hipLaunchKernelGGL(k_ids_rowscan, dim3((unsigned)N), dim3(256), 0, s, reset_flags, W, N, scratch, game_count,
                       (int32_t)snap, step_resets);
    const dim3 grid((unsigned)cdiv(M, kBlock)), block(kBlock);
    if (N <= 8192) { // two launches: every workgroup of the assignment scans the N step totals itself
        hipLaunchKernelGGL((k_ids_assign<true>), grid, block, (size_t)(N + 1) * sizeof(int32_t), s, *st, M, N, W,
                           reset_flags, last_reset, scratch, game_count, (int32_t)snap);
    } else {
        hipLaunchKernelGGL(k_ids_stepscan, dim3(1), dim3(64), 0, s, scratch + (int64_t)N * W, N, (int32_t *)nullptr);
        hipLaunchKernelGGL((k_ids_assign<false>), grid, block, 0, s, *st, M, N, W, reset_flags, last_reset, scratch,
                           game_count, (int32_t)snap);
    }
    return check_launch("ml2048_rollout_finish_ids");
}

// ---- events of the pipelined rollout (host-side plumbing: every call below is one HIP runtime call) ----------------
int ml2048_event_create(void **event) {
    if (!event) return fail(ML2048_E_INVALID, "event is NULL");
    hipEvent_t e;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) return fail((int)rc, "hipEventCreateWithFlags: %s", hipGetErrorString(rc));
    *event = (void *)e;
    return 0;
}

void ml2048_event_destroy(void *event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}

int ml2048_event_record(void *event, void *stream) {
    if (!event) return fail(ML2048_E_INVALID, "event is NULL");
    const hipError_t rc = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    return rc == hipSuccess ? 0 : fail((int)rc, "hipEventRecord: %s", hipGetErrorString(rc));
}

// 1 = everything recorded into the event has finished (or nothing ever was), 0 = still pending
int ml2048_event_query(void *event) {
    if (!event) return fail(ML2048_E_INVALID, "event is NULL");
    const hipError_t rc = hipEventQuery((hipEvent_t)event);
    if (rc == hipSuccess) return 1;
    if (rc == hipErrorNotReady) {
        (void)hipGetLastError(); // "not ready" is an answer, not an error to leave behind
        return 0;
    }
    return fail(ML2048_E_INVALID, "hipEventQuery: %s", hipGetErrorString(rc));
}

int ml2048_event_synchronize(void *event) {
    if (!event) return fail(ML2048_E_INVALID, "event is NULL");
    const hipError_t rc = hipEventSynchronize((hipEvent_t)event);
    return rc == hipSuccess ? 0 : fail((int)rc, "hipEventSynchronize: %s", hipGetErrorString(rc));
}

int ml2048_stream_wait_event(void *stream, void *event) {
    if (!event) return fail(ML2048_E_INVALID, "event is NULL");
    const hipError_t rc = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
    return rc == hipSuccess ? 0 : fail((int)rc, "hipStreamWaitEvent: %s", hipGetErrorString(rc));
}

// ml2048_rollout_chain + the marker behind its kernel + ml2048_rollout_finish_ids (on the same or on a side stream) in
// ONE call: what VecGame.rollout() issues per plain rollout, without a Python round trip per piece
// One-shot, per thread: the next ml2048_rollout* launch carries `event` as its dispatch's completion event (no marker
// packet behind the kernel).  ml2048_rollout_event_attached() tells whether that launch did (a timed dispatch keeps its
// timer's pair; then the caller records the event itself).
int ml2048_rollout_attach_event(void *event) {
    if (!event) return fail(ML2048_E_INVALID, "event is NULL");
    g_attach_stop_event = (hipEvent_t)event;
    return 0;
}

int ml2048_rollout_event_attached(void) { return g_stop_event_attached ? 1 : 0; }

// ---- the line table of the plain rollout kernels (board_ops.hpp row_lut_entry / move_board_lut) ----------------------
__global__ __launch_bounds__(kBlock) void k_fill_row_lut(uint16_t *__restrict__ lut) {
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx < (uint32_t)kRowLutEntries) lut[idx] = row_lut_entry(idx);
}

int ml2048_row_lut_fill(void *stream, void *lut) {
    if (!lut || (uintptr_t)lut % 16) return fail(ML2048_E_ALIGN, "lut must be a 16-byte aligned device buffer of ML2048_ROW_LUT_BYTES");
    hipLaunchKernelGGL(k_fill_row_lut, dim3(kRowLutEntries / kBlock), dim3(kBlock), 0, (hipStream_t)stream, (uint16_t *)lut);
    return check_launch("ml2048_row_lut_fill");
}

int ml2048_rollout_attach_row_lut(const void *lut) {
    g_attach_row_lut = (const uint16_t *)lut;
    return 0;
}

int ml2048_rollout_chain_ids(void *stream, int64_t M, int64_t idx_base, const ml2048_state *st, int32_t N,
                             const uint8_t *perm_epochs, const uint32_t *two_masks, int32_t n_epochs,
                             const int32_t *schedule, int reward_kind, int masked, uint64_t act_seed, uint64_t act_counter0,
                             uint64_t *reset_flags, int32_t *last_reset, const void *next_src_host, void *next_dst,
                             int64_t next_nbytes, int64_t next_table_offset, void *rollout_event, void *ids_stream,
                             int32_t *game_count, int32_t *step_resets, int32_t *scratch, void *ids_done_event) {
    const rollout_one_shots spent_on_return;
    const bool ids = scratch != nullptr;
    if (ids && ids_stream != stream && !rollout_event)
        return fail(ML2048_E_INVALID, "an id scan on another stream needs rollout_event to wait for");
    if (ids && !game_count) return fail(ML2048_E_INVALID, "game_count is NULL");
    // ML2048_ATTACH_EVENT=0 records the marker behind the kernel instead (A/B: tools/debug/attach_ab.sh - the marker
    // packet costs ~2 us between two dependent rollout kernels, ~4 us per rollout in a 20-launch window)
    static const bool attach = [] { const char *e = getenv("ML2048_ATTACH_EVENT"); return !e || atoi(e) != 0; }();
    if (attach && rollout_event) g_attach_stop_event = (hipEvent_t)rollout_event;
    const int rc0 = ml2048_rollout_chain(stream, M, idx_base, st, N, perm_epochs, two_masks, n_epochs, schedule, reward_kind, masked,
                                         act_seed, act_counter0, reset_flags, last_reset, next_src_host, next_dst, next_nbytes,
                                         next_table_offset);
    g_attach_stop_event = nullptr;
    if (rc0) return rc0;
    if (rollout_event && !g_stop_event_attached)
        if (int rc = ml2048_event_record(rollout_event, stream)) return rc;
    if (!ids) return 0;
    if (ids_stream != stream)
        if (int rc = ml2048_stream_wait_event(ids_stream, rollout_event)) return rc;
    if (int rc = ml2048_rollout_finish_ids(ids_stream, M, st, N, reset_flags, last_reset, game_count, step_resets, scratch)) return rc;
    if (ids_done_event) return ml2048_event_record(ids_done_event, ids_stream);
    return 0;
}

// ml2048_stream_wait_event(ids_stream, wait_event) + ml2048_rollout_finish_ids on ids_stream + done_event recorded behind
// it, as ONE call: the id scan of the PREVIOUS rollout, issued by VecGame.rollout() right after the next rollout's launch
int ml2048_rollout_ids_after(void *ids_stream, void *wait_event, int64_t M, const ml2048_state *st, int32_t N,
                             const uint64_t *reset_flags, const int32_t *last_reset, int32_t *game_count,
                             int32_t *step_resets, int32_t *scratch, void *done_event) {
    if (wait_event)
        if (int rc = ml2048_stream_wait_event(ids_stream, wait_event)) return rc;
    if (int rc = ml2048_rollout_finish_ids(ids_stream, M, st, N, reset_flags, last_reset, game_count, step_resets, scratch)) return rc;
    if (done_event) return ml2048_event_record(done_event, ids_stream);
    return 0;
}

int ml2048_episode_stats(void *stream, int64_t M, const ml2048_state *st, int32_t *age, int32_t *seen_id,
                         int32_t id_base, int32_t rounds, uint64_t *count, uint64_t *step_sum, uint64_t *score_sum) {
    if (int rc = check_state(M, st)) return rc;
    if (!age || !seen_id || !count || !step_sum || !score_sum) return fail(ML2048_E_INVALID, "age/seen_id/count/step_sum/score_sum is NULL");
    if (rounds < 0) return fail(ML2048_E_INVALID, "rounds=%d", rounds);
    if (((uintptr_t)age | (uintptr_t)seen_id | (uintptr_t)st->id) % 16 || (uintptr_t)st->terminated % 4)
        return fail(ML2048_E_ALIGN, "age, seen_id and st->id must be 16-byte aligned, st->terminated 4-byte aligned");
    int64_t blocks = cdiv(cdiv(M, 4), kStatsBlock); // grid-stride over 4-board units: big workgroups keep the flushes rare
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_episode_stats, dim3((unsigned)blocks), dim3(kStatsBlock), 0, (hipStream_t)stream, *st, M, age,
                       seen_id, id_base, rounds, (unsigned long long *)count, (unsigned long long *)step_sum,
                       (unsigned long long *)score_sum);
    return check_launch("ml2048_episode_stats");
}

int ml2048_record_transition(void *stream, int64_t M, const ml2048_state *st, const uint8_t *actions,
                             const float *logp, int8_t *state, uint8_t *valid_actions, int8_t *action,
                             float *action_log_prob, float *reward, int8_t *next_state, uint8_t *next_valid_actions,
                             int32_t *step, uint8_t *terminated) {
    if (int rc = check_state(M, st)) return rc;
    if (((uintptr_t)state | (uintptr_t)next_state) % 16)
        return fail(ML2048_E_ALIGN, "state/next_state slices must be 16-byte aligned");
    if (((uintptr_t)valid_actions | (uintptr_t)next_valid_actions | (uintptr_t)reward | (uintptr_t)step |
         (uintptr_t)action_log_prob) % 4)
        return fail(ML2048_E_ALIGN, "4-byte outputs must be 4-byte aligned");
    hipLaunchKernelGGL(k_record, dim3((unsigned)cdiv(M, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, *st, M, actions,
                       logp, state, valid_actions, action, action_log_prob, reward, next_state, next_valid_actions, step,
                       terminated);
    return check_launch("ml2048_record_transition");
}

int ml2048_gae(void *stream, int64_t U, int64_t S, int64_t G, const float *v0, const float *v1, const float *reward,
               const uint8_t *terminated, double gamma, double lambda_, float *adv, float *delta_out) {
    if (U <= 0 || S <= 0 || G <= 0) return fail(ML2048_E_INVALID, "U=%lld S=%lld G=%lld must be positive",
                                                  (long long)U, (long long)S, (long long)G);
    if (!v0 || !v1 || !reward || !terminated || !adv) return fail(ML2048_E_INVALID, "v0/v1/reward/terminated/adv is NULL");
    const float coef = (float)(gamma * lambda_); // `coef = gamma * lambda_` is a python float (gae.py:56)
    hipLaunchKernelGGL(k_gae, dim3((unsigned)cdiv(U * G, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, U, S, G, v0, v1,
                       reward, terminated, (float)gamma, coef, adv, delta_out);
    return check_launch("ml2048_gae");
}

} // extern "C"
#endif // ML2048_ROLLOUT_PLAIN_TU
