"""GPU parity of the PLAIN fused-rollout instantiations - the kernels bench.py times - against the CPU oracle.

`VecGame.rollout()` without `record_actions / digests / trajectory` launches `k_rollout<KIND, FAST, EXTRAS=false, INV>`
(csrc/ml2048_hip.hip): lazy valid-mask loop, eight steps per loop iteration, inverse-permutation spawn on the
expanded tables of the chained upload (`ml2048_rollout_chain`), and one of two separately compiled passes chosen by
the board count (M <= 196 608: the lone-wave forms, default scheduling - max-ilp for masked sampling and the rocRAND mode;
larger M: the general forms).  None of
that code runs in the EXTRAS=true instantiation the digest-checking tests of test_gpu_parity.py use.

The actions of such a rollout are a pure function of (act_seed, act_counter + t, board index) - and of the valid
mask when masked - so the oracle replays them without any recording: game_numba.py:619-738 step for step, with
`oracle.sample_actions` as the action source.  Every comparison is bit-exact (integer / byte / index work).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ACT_SEED = 7


@pytest.fixture(scope="module")
def amd():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU box"
    import ml2048_amd

    ml2048_amd._lib.load()  # must be the in-tree HIP library; raises otherwise
    return ml2048_amd


def _assert_same(env, ref, what):
    a, b = env._data, ref._data
    for f in ("id", "step", "score", "reward", "board", "merged", "valid_actions", "terminated", "invalid"):
        assert np.array_equal(a[f], b[f]), f"{what}: field {f} differs"
    assert a.tobytes() == b.tobytes(), what
    assert np.array_equal(env._prev_state, ref._prev_state), f"{what}: prev_state"
    assert np.array_equal(env._prev_valid_actions, ref._prev_valid_actions), f"{what}: prev_valid_actions"


def _run(amd, oracle, M, lengths, reward, schedule, counter0, *, masked=False, seed=20241008, spawn="table",
         ids="inline"):
    """Back-to-back plain rollouts of the given lengths (the second and later launches of a host-async env consume
    a schedule uploaded by the previous launch), each replayed by the oracle."""
    kw = {}
    if spawn == "rocrand":
        env = amd.VecGame(M, reward, spawn="rocrand", spawn_seed=0xC0FFEE)
        ref = oracle.OracleVecGame(M, reward, fast_seed=0xC0FFEE)
    else:
        env = amd.VecGame(M, reward, schedule=schedule)
        ref = oracle.OracleVecGame(M, reward)
    env.reset(seed)
    ref.reset(seed)
    counter = counter0
    outs = []
    for N in lengths:  # all launches first: nothing between two rollout kernels but what the product itself puts there
        outs.append(env.rollout(N, masked=masked, act_seed=ACT_SEED, act_counter=counter,
                                finish_ids="deferred" if ids == "deferred" else True, **kw))
        counter += N
    counter = counter0
    expect_resets = []
    for N in lengths:
        resets = np.zeros(N, np.int32)
        for t in range(N):
            (ridx,) = ref.prepare()
            resets[t] = ridx.size
            a, _ = oracle.sample_actions(ref._data["valid_actions"], ACT_SEED, counter + t, masked=masked)
            ref.step(a)
        counter += N
        expect_resets.append(resets)
    env.join_ids()
    for k, (out, exp) in enumerate(zip(outs, expect_resets)):
        assert np.array_equal(out["step_resets"].cpu().numpy(), exp), f"step_resets of rollout {k}"
    _assert_same(env, ref, f"M={M} lengths={lengths} {reward} {schedule}")
    assert env._game_count == ref._game_count
    assert env._rand_step == ref._rand_step  # closes the schedule stream: generator == what was executed
    if spawn == "table":
        assert np.array_equal(env._randperm, ref._randperm) and np.array_equal(env._randfloat, ref._randfloat)
    # and the per-step API continues from exactly that state
    (i1,) = env.prepare()
    (i2,) = ref.prepare()
    assert np.array_equal(i1, i2)
    a = np.random.default_rng(M).integers(0, 4, size=M).astype(np.uint8)
    env.step(a)
    ref.step(a)
    _assert_same(env, ref, "per-step call after the rollouts")
    return env, ref


# ---- the default bench path: schedule="host-async" -> ml2048_rollout_chain (expanded tables, INV spawn) -----------
def test_plain_c2_twice_back_to_back(amd, oracle):
    """BASELINE config 2 exactly as bench.py runs it (M=65 536, N=128, unmasked, host-async, deferred id scan):
    the small-M pass, second launch on a chained upload."""
    env, ref = _run(amd, oracle, 65536, [128, 128], "normal", "host-async", 0, ids="deferred")
    assert ref._game_count > 65536  # games ended and were re-dealt inside the window


@pytest.mark.parametrize("M,lengths,reward,counter0", [
    (65536, [127, 65, 200], "improved", 13),   # remainder loop of the 8-step unroll; 64-step flag chunk edges
    (65537, [128, 64], "rank", 0),             # smallest M of the large-M pass (LDS tables): 5 board waves per workgroup
    (131073, [128, 65], "maxcell", 77),        # 9 board waves per workgroup, ragged tail wave
    (196609, [65, 128], "improved", 40),       # 13 board waves per workgroup, the last workgroup partly past the boards
    (262144, [128, 128], "normal", 3),         # C4's per-GPU shard / C3's size
    (1000, [200, 65, 1], "improved", 31),      # ragged small M
    (63, [130], "rank", 5),
])
def test_plain_chain_matrix(amd, oracle, M, lengths, reward, counter0):
    _run(amd, oracle, M, lengths, reward, "host-async", counter0, ids="deferred" if M % 2 == 0 else "inline")


@pytest.mark.parametrize("M,lengths,reward,counter0", [
    (1, [64, 63, 1], "normal", 0),             # one board; the small-M pass takes the line table from 64 steps on
    (64, [64], "improved", 15),                # exactly one full wave
    (65, [63, 64, 65], "rank", 2),             # a full wave and a one-lane wave, around the table's step threshold
    (70000, [16, 15, 17], "normal", 63),       # the large-M pass takes the table from 16 steps on
])
def test_plain_chain_at_the_table_thresholds(amd, oracle, M, lengths, reward, counter0):
    """Launches just below and at the step counts from which a pass is given the LDS line table (launch_rollout: N >= 64 in
    the small-M passes, N >= 16 in the large-M pass), at the smallest board counts."""
    _run(amd, oracle, M, lengths, reward, "host-async", counter0, ids="deferred")


def test_plain_2m_boards_large_pass(amd, oracle):
    """BASELINE config 4's total size as one plain rollout (large-M pass, 32 768 wavefronts)."""
    _run(amd, oracle, 2097152, [128], "normal", "host-async", 1)


def test_plain_chain_with_the_copying_schedule_hand_over(amd, oracle, monkeypatch):
    """ML2048_STREAM_PINNED=0: the schedule stream owns its slots and every schedule is copied into a pinned staging ring
    (the fallback when the slots would not fit the pinned budget: very long rollouts) - same launches, same results."""
    monkeypatch.setenv("ML2048_STREAM_PINNED", "0")
    _run(amd, oracle, 4099, [128, 128, 65, 65], "improved", "host-async", 9, ids="deferred")


def test_plain_chain_without_the_id_scan_then_with(amd, oracle):
    """finish_ids=False leaves ids and the game counter alone (the caller numbers the games later or not at all); the
    boards, flags and last-deal steps are those of the numbered run."""
    M, N = 8192, 128
    a = amd.VecGame(M, "normal", schedule="host-async")
    b = amd.VecGame(M, "normal", schedule="host-async")
    a.reset(5)
    b.reset(5)
    oa = a.rollout(N, masked=False, act_seed=ACT_SEED, act_counter=0, finish_ids=False)
    ob = b.rollout(N, masked=False, act_seed=ACT_SEED, act_counter=0, finish_ids=True)
    assert oa["step_resets"] is None and ob["step_resets"].shape == (N,)
    assert set(oa) == set(ob) == {"actions", "digests", "step_resets", "reset_flags", "last_reset"}
    assert np.array_equal(oa["reset_flags"].cpu().numpy(), ob["reset_flags"].cpu().numpy())
    assert np.array_equal(oa["last_reset"].cpu().numpy(), ob["last_reset"].cpu().numpy())
    assert np.array_equal(a.board.cpu().numpy(), b.board.cpu().numpy())
    assert int(ob["step_resets"].sum().item()) == b._game_count   # every deal of the window, the first M included
    assert a._game_count == 0                                     # nothing was numbered


# ---- the same kernels on the plain 16-byte table layout (schedule="host": ml2048_rollout, probing spawn) -----------
@pytest.mark.parametrize("M,lengths,reward,counter0", [
    (65536, [128, 65], "maxcell", 9),
    (262144, [200, 127], "improved", 21),
])
def test_plain_sync_host_schedule(amd, oracle, M, lengths, reward, counter0):
    _run(amd, oracle, M, lengths, reward, "host", counter0)


# ---- schedule drawn on the GPU, expanded tables straight from csrc/devrng2.hip -------------------------------------
@pytest.mark.parametrize("M,lengths,reward,counter0", [
    (65536, [128, 65, 128], "normal", 9),
    (262144, [128], "rank", 18),
])
def test_plain_device_schedule(amd, oracle, M, lengths, reward, counter0):
    _run(amd, oracle, M, lengths, reward, "device", counter0)


# ---- masked-uniform sampling without optional outputs: the eager-mask loop of the plain instantiations -------------
@pytest.mark.parametrize("M,lengths,reward", [
    (65536, [128, 127], "normal"),
    (262144, [65], "maxcell"),
])
def test_plain_masked(amd, oracle, M, lengths, reward):
    _run(amd, oracle, M, lengths, reward, "host-async", 4, masked=True)


# ---- rocRAND spawn mode, plain instantiations (ml2048_rollout_fast without optional outputs) -----------------------
@pytest.mark.parametrize("M,lengths,masked", [
    (65536, [128, 64], False),
    (262144, [65], True),
])
def test_plain_rocrand_spawn(amd, oracle, M, lengths, masked):
    _run(amd, oracle, M, lengths, "improved", None, 11, masked=masked, spawn="rocrand")


# ---- the large-M pass moves boards by LDS table look-up (board_ops.hpp move_board_lut) -----------------------------
def test_plain_large_pass_without_the_line_table(amd, oracle, monkeypatch):
    """ML2048_ROW_LUT=0: no table is handed to the launch and the large-M pass moves boards arithmetically (what it also
    does by itself for waves outside the table's guard) - same results."""
    monkeypatch.setenv("ML2048_ROW_LUT", "0")
    _run(amd, oracle, 262144, [128, 65], "normal", "host-async", 7, ids="deferred")
    _run(amd, oracle, 196608, [65], "rank", "host-async", 2, masked=True)


def test_plain_large_pass_on_boards_outside_the_table_guard(amd, oracle):
    """Tiles above 2^11 (the table's guard: lut_guard) loaded into some boards of a large-M env: those waves take the
    arithmetic move, the others the look-ups; tiles 2^12 ... 2^15 merge and move exactly as the oracle's."""
    M, N = 139264, 96                              # 2 176 waves: 9 per workgroup, large-M pass
    env = amd.VecGame(M, "improved", schedule="host-async")
    ref = oracle.OracleVecGame(M, "improved")
    env.reset(23)
    ref.reset(23)
    env.prepare()
    ref.prepare()
    d = ref._data
    rng = np.random.default_rng(2)
    for idx in rng.choice(M, size=600, replace=False):   # ~600 of the 2 176 waves get one board with big tiles
        cells = rng.choice(16, size=4, replace=False)
        d["board"][idx, cells] = rng.integers(12, 16, size=4)
        if rng.random() < 0.5:                            # two equal big tiles side by side: they merge on the first move
            d["board"][idx, 0] = d["board"][idx, 1] = rng.integers(12, 15)
        d["valid_actions"][idx] = oracle.compute_valid_actions(d["board"][idx])[0]
    env.load_records(d)
    env.rollout(N, masked=False, act_seed=ACT_SEED, act_counter=3)
    for t in range(N):
        ref.prepare()
        a, _ = oracle.sample_actions(ref._data["valid_actions"], ACT_SEED, 3 + t, masked=False)
        ref.step(a)
    _assert_same(env, ref, "boards outside the table guard")
    assert env._game_count == ref._game_count
    assert int(ref._data["board"].max()) >= 13           # the big tiles were really in play


# ---- the lazy loop's score comes from the board (board_ops.hpp score_potential); states outside its guard -----------
def test_plain_rollout_on_states_outside_the_score_guard(amd, oracle):
    """Fractional / huge / negative scores and tiles above 2^15 loaded into the env: those waves take the eager loop
    (per-step float32 accumulation exactly as the reference); the others the lazy one.  All equal the oracle."""
    M, N = 4096, 150
    env = amd.VecGame(M, "normal", schedule="host-async")
    ref = oracle.OracleVecGame(M, "normal")
    env.reset(11)
    ref.reset(11)
    arng = np.random.default_rng(1)
    for _ in range(30):   # some play first, in lock-step: boards, scores, steps of games in progress
        env.prepare()
        ref.prepare()
        a = arng.integers(0, 4, size=M)
        env.step(a)
        ref.step(a)
    env.prepare()
    ref.prepare()
    d = ref._data
    d["score"][0:64] += 0.5                    # one whole wave fractional
    d["score"][64] = 4194304.0                 # one lane of the next wave beyond the guard's range
    d["score"][200] = 16777215.0               # float32 adds stop being exact during the rollout
    d["score"][300] = -8.0
    d["board"][400, 5] = 16                    # a 65 536 tile
    d["valid_actions"][400] = oracle.compute_valid_actions(d["board"][400])[0]
    env.load_records(d)
    # the rollout's first prepare() is the oracle's next prepare(): nothing is terminated after the one above
    env.rollout(N, masked=False, act_seed=ACT_SEED, act_counter=5)
    for t in range(N):
        ref.prepare()
        a, _ = oracle.sample_actions(ref._data["valid_actions"], ACT_SEED, 5 + t, masked=False)
        ref.step(a)
    _assert_same(env, ref, "states outside the score guard")
    assert env._game_count == ref._game_count


def test_a_failed_rollout_call_spends_the_threads_one_shots(amd):
    """ml2048_timer_arm / ml2048_rollout_attach_event / ml2048_rollout_attach_row_lut are one-shots of the calling thread for
    "the next rollout launch".  A rollout call that fails an argument check is that next call: nothing may stay armed and ride
    on an unrelated later launch (include/ml2048_hip.h: the one-shot notes)."""
    import ctypes as C

    L = amd._lib.load()
    env = amd.VecGame(4096, amd.reward_fn_normal, device="cuda:0", schedule="host-async")
    env.reset(3)
    tm = C.c_void_p()
    amd._lib.check(L.ml2048_timer_create(C.byref(tm)), "timer_create")
    try:
        amd._lib.check(L.ml2048_timer_arm(tm), "timer_arm")
        # N = 0 fails the entry check before anything is launched
        rc = L.ml2048_rollout_chain(None, 4096, 0, env._st(), 0, None, None, 1, None, 0, 0, 0, 0, None, None, None, None, 0, 0)
        assert rc == -1 and b"N=0" in L.ml2048_last_error()
        env.rollout(16, masked=False, act_seed=1)  # an unrelated launch of the same thread
        env.join_ids()
        ms = C.c_float(-1.0)
        assert L.ml2048_timer_elapsed_ms(tm, C.byref(ms)) == -1 and b"never attached" in L.ml2048_last_error()
        # and armed again, the timer does bracket the next launch
        env.rollout(16, masked=False, act_seed=1, act_counter=16, kernel_timer=tm)
        env.join_ids()
        amd._lib.check(L.ml2048_timer_elapsed_ms(tm, C.byref(ms)), "timer_elapsed_ms")
        assert 0.0 < ms.value < 50.0
    finally:
        L.ml2048_timer_destroy(tm)
