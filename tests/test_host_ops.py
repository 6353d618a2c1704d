"""CPU check of the DEVICE arithmetic: ml2048_amd/csrc/board_ops.hpp compiled with g++ through
tests/csrc/host_ops_harness.cpp and compared with the golden vectors and the oracle.

This is a test harness for the bit-parallel board code that the HIP kernels inline; it is not a
product path (the product has no CPU implementation and fails loudly without the HIP library).
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "host_ops_harness.cpp")
LIB = os.path.join(HERE, "csrc", "libhost_ops_harness.so")
HDR = os.path.join(HERE, "..", "ml2048_amd", "csrc", "board_ops.hpp")


@pytest.fixture(scope="module")
def H():
    if not os.path.exists(LIB) or max(os.path.getmtime(SRC), os.path.getmtime(HDR)) > os.path.getmtime(LIB):
        subprocess.check_call(
            ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-o", LIB, SRC]
        )
    L = C.CDLL(LIB)
    vp, i64, i32, u64, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_double
    L.h_lines18.argtypes = [vp, vp]
    L.h_move.argtypes = [vp, i64, vp, vp, vp, vp]
    L.h_valid.argtypes = [vp, i64, vp, vp, vp]
    L.h_transpose.argtypes = [vp, vp]
    L.h_spawn_forms_check.argtypes = [vp, vp, vp, i64]
    L.h_spawn_forms_check.restype = i64
    L.h_compact_check.argtypes = [vp, i64]
    L.h_compact_check.restype = i64
    L.h_move_forms_check.argtypes = [vp, i64]
    L.h_move_forms_check.restype = i64
    L.h_move_lut_check.argtypes = [vp, i64, vp]
    L.h_move_lut_check.restype = i64
    L.h_deal_words_check.argtypes = [vp, i64, C.c_uint32]
    L.h_deal_words_check.restype = i64
    L.h_lazy_mask_check.argtypes = [vp, i64]
    L.h_lazy_mask_check.restype = i64
    L.h_step.argtypes = [vp, i64, i64, vp, vp, vp, i32, dbl, i64, vp, vp]
    L.h_prepare.argtypes = [vp, i64, i64, i64, vp, vp, dbl, vp]
    L.h_prepare.restype = i64
    L.h_step_fast.argtypes = [vp, i64, i64, vp, vp, vp, i32, dbl, u64, u64]
    L.h_prepare_fast.argtypes = [vp, i64, i64, dbl, u64, u64, vp]
    L.h_prepare_fast.restype = i64
    L.h_sample.argtypes = [vp, i64, i64, u64, u64, i32, vp, vp]
    L.h_digest.argtypes = [vp, i64, i64, i32]
    L.h_digest.restype = u64
    L.h_philox.argtypes = [vp, vp, vp]
    return L


def _p(a):
    return a.ctypes.data


def test_push_line_exhaustive(H, golden_dir):
    g = np.load(os.path.join(golden_dir, "lines18.npz"))
    n = 18**4
    result = np.zeros((n, 4), np.uint8)
    mexp = np.zeros((n, 2), np.uint8)
    H.h_lines18(_p(result), _p(mexp))
    assert np.array_equal(result, g["result"])
    # merge exponents -> 18-slot bucket array as the reference fills it
    buckets = np.zeros((n, 18), np.uint8)
    for j in range(2):
        m = mexp[:, j]
        rows = np.flatnonzero(m)
        np.add.at(buckets, (rows, m[rows]), 1)
    assert np.array_equal(buckets, g["buckets"])
    # forward-movable <=> push changes the line; tested through valid_actions below


def _random_boards(rng, n, hi=18):
    b = rng.integers(0, hi, size=(n, 16)).astype(np.uint8)
    # sprinkle many zeros and equal neighbours
    mask = rng.random((n, 16)) < 0.35
    b[mask] = 0
    dup = rng.random((n, 16)) < 0.3
    b[:, 1:][dup[:, 1:]] = b[:, :-1][dup[:, 1:]]
    return b


def test_transpose(H):
    a = np.arange(16, dtype=np.uint8)
    out = np.zeros(16, np.uint8)
    H.h_transpose(_p(a), _p(out))
    assert np.array_equal(out.reshape(4, 4), a.reshape(4, 4).T)


def test_move_and_stats_vs_oracle(H, oracle):
    rng = np.random.default_rng(5)
    n = 20000
    boards = _random_boards(rng, n)
    actions = rng.integers(0, 4, size=n).astype(np.uint8)
    out = np.zeros((n, 16), np.uint8)
    merged = np.zeros((n, 16), np.uint8)
    stats = np.zeros((n, 3), np.uint32)
    H.h_move(_p(boards), n, _p(actions), _p(out), _p(merged), _p(stats))
    for i in range(n):
        ob, om = oracle.step_board(boards[i], int(actions[i]))
        assert np.array_equal(ob, out[i]), i
        assert np.array_equal(om[:16], merged[i]), i  # buckets 16/17 dropped (H9)
        assert oracle.reward("normal", ob, boards[i], om[:16]) == float(stats[i, 0])
        assert oracle.reward("rank", ob, boards[i], om[:16]) == float(stats[i, 1])
        assert int(om[:16].sum()) == int(stats[i, 2])


def test_merged_value_sum_all_byte_values(H):
    """Every byte value 0..255 in every position, pairs, and random words: sum of 2^(num+1) over the bytes
    1..15 of a pair of merge-exponent words (0 = no merge; >= 16 dropped, H9)."""
    rng = np.random.default_rng(11)
    cells = np.concatenate([
        np.array([[v if j == k else 0 for j in range(8)] for k in range(8) for v in range(256)], np.uint8),
        np.array([[v, w, v, w, w, v, w, v] for v in range(24) for w in range(24)], np.uint8),
        rng.integers(0, 16, size=(50000, 8)).astype(np.uint8),
        rng.integers(0, 20, size=(50000, 8)).astype(np.uint8)])
    words = np.ascontiguousarray(cells).view(np.uint32).reshape(-1)
    out = np.zeros(len(cells), np.uint32)
    H.h_merged_value_sum.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    H.h_merged_value_sum.restype = None
    H.h_merged_value_sum(_p(words), len(cells), _p(out))
    c = cells.astype(np.int64)
    want = np.where((c >= 1) & (c <= 15), 2 << np.minimum(c, 20), 0).sum(axis=1)
    assert np.array_equal(out.astype(np.int64), want)


def test_score_is_a_function_of_board_and_step(H, oracle):
    """board_ops.hpp score_potential: in every game played from a deal, after every prepare() and every step(),
    score == sum over tiles of (e-3)*2^e + 8 + 4*step - what lets the fused rollout's lazy loop drop the reward
    arithmetic (reward_fn_normal accumulates into score whatever reward_fn is, game_numba.py:729-731).  Checked move
    by move on the reference-derived oracle, with invalid moves (H3) and both 2/4 probabilities; and the guard that
    keeps every intermediate sum an exact float32 integer."""
    H.h_score_potential.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int32] + [C.c_void_p] * 2
    H.h_score_potential.restype = None
    games = 0
    for two_prob, kind in ((0.8, "improved"), (0.3, "normal")):
        M = 2048
        env = oracle.OracleVecGame(M, kind, two_prob=two_prob)
        env.reset(5)
        rng = np.random.default_rng(0)
        pot = np.zeros(M, np.int32)
        guard = np.zeros(M, np.uint8)
        for t in range(500):
            for phase in range(2):
                if phase == 0:
                    env.prepare()
                else:
                    env.step(rng.integers(0, 4, size=M))
                d = env._data
                board = np.ascontiguousarray(d["board"])
                step = np.ascontiguousarray(d["step"])
                score = np.ascontiguousarray(d["score"])
                H.h_score_potential(_p(board), _p(step), _p(score), M, 128, _p(pot), _p(guard))
                assert np.array_equal(pot.astype(np.float32), score), (two_prob, t, phase)
                assert guard.all()
        games += env._game_count
    assert games > 20000
    # the guard: tiles above 2^15, negative / fractional / large scores are sent to the eager loop
    boards = np.zeros((5, 16), np.uint8)
    boards[0, 3] = 16
    scores = np.array([0.0, -1.0, 0.5, 2097152.0, 2097151.0], np.float32)
    steps = np.zeros(5, np.int32)
    pot = np.zeros(5, np.int32)
    guard = np.zeros(5, np.uint8)
    H.h_score_potential(_p(boards), _p(steps), _p(scores), 5, 128, _p(pot), _p(guard))
    assert guard.tolist() == [0, 0, 0, 0, 1]
    # ... and so is every rollout in which two 2^16 tiles could meet: the reference drops that merge's reward (H9: bucket
    # 16 is outside merged[16]) while the identity would count it.  A 2^17 tile needs a tile sum of 2^17: the guard asks
    # for tile_sum + 4 * steps < 2^17.
    boards = np.zeros((4, 16), np.uint8)
    boards[0, :4] = 15                              # tile sum 131 072 already
    boards[1, :3] = 15                              # 98 304: fine for 128 steps ...
    boards[2, :3] = 15                              # ... not for 8 192 (98 304 + 32 768)
    boards[3, :2] = 1                               # a fresh deal: up to 32 766 steps
    scores = np.zeros(4, np.float32)
    steps = np.zeros(4, np.int32)
    pot = np.zeros(4, np.int32)
    guard = np.zeros(4, np.uint8)
    H.h_score_potential(_p(boards), _p(steps), _p(scores), 4, 128, _p(pot), _p(guard))
    assert guard.tolist() == [0, 1, 1, 1]
    H.h_score_potential(_p(boards), _p(steps), _p(scores), 4, 8192, _p(pot), _p(guard))
    assert guard.tolist() == [0, 0, 0, 1]
    H.h_score_potential(_p(boards), _p(steps), _p(scores), 4, 32766, _p(pot), _p(guard))
    assert guard.tolist() == [0, 0, 0, 1]           # 4 + 4 * 32 766 = 131 068
    H.h_score_potential(_p(boards), _p(steps), _p(scores), 4, 32767, _p(pot), _p(guard))
    assert guard.tolist() == [0, 0, 0, 0]           # 4 + 4 * 32 767 = 2^17


def test_valid_empties_maxcell_vs_oracle(H, oracle):
    rng = np.random.default_rng(6)
    n = 20000
    boards = np.concatenate([_random_boards(rng, n - 3), np.zeros((1, 16), np.uint8),
                             np.full((1, 16), 17, np.uint8),
                             (np.arange(16, dtype=np.uint8) % 2 + 1).reshape(1, 16)])
    valid = np.zeros((n, 4), np.uint8)
    empties = np.zeros(n, np.uint16)
    maxcell = np.zeros(n, np.uint8)
    H.h_valid(_p(boards), n, _p(valid), _p(empties), _p(maxcell))
    for i in range(n):
        ov, _ = oracle.compute_valid_actions(boards[i])
        assert np.array_equal(ov, valid[i]), (i, boards[i])
    exp_empty = ((boards == 0) * (1 << np.arange(16))).sum(axis=1)
    assert np.array_equal(exp_empty.astype(np.uint16), empties)
    assert np.array_equal(boards.max(axis=1), maxcell)


def test_move_all_lines_at_once_equals_line_by_line(H):
    """move_board (four lines bytewise in four cell words) == four push_line calls, board and merge
    exponents, for every one of the 18^4 lines placed in each of the four line positions of a board (all
    four directions) and for random boards up to exponent 17."""
    import itertools

    lines = np.array(list(itertools.product(range(18), repeat=4)), dtype=np.uint8)   # 104976 x 4
    rng = np.random.default_rng(3)
    boards = rng.integers(0, 18, size=(lines.shape[0], 4, 4)).astype(np.uint8)
    boards[np.arange(lines.shape[0]), np.arange(lines.shape[0]) % 4, :] = lines       # as a row
    cols = boards.copy().transpose(0, 2, 1).copy()                                      # and as a column
    sparse = _random_boards(rng, 100000)
    dense = rng.integers(1, 4, size=(100000, 16)).astype(np.uint8)
    allb = np.ascontiguousarray(np.concatenate([boards.reshape(-1, 16), cols.reshape(-1, 16), sparse, dense]))
    assert H.h_move_forms_check(_p(allb), allb.shape[0]) == 0


def test_move_by_table_lookup_equals_the_arithmetic_move(H):
    """move_board_lut (four 16-bit line look-ups in the 65 536-entry table of _push_row results) == move_board for
    every one of the 16^4 lines the table covers, placed in each line position of a board (all four directions), and
    for random boards; boards outside the table's domain (a tile above 2^15, two 2^15 tiles merging) are skipped by
    the harness and must be refused by lut_guard."""
    import itertools

    lines = np.array(list(itertools.product(range(16), repeat=4)), dtype=np.uint8)   # 65536 x 4
    rng = np.random.default_rng(5)
    boards = rng.integers(0, 16, size=(lines.shape[0], 4, 4)).astype(np.uint8)
    boards[np.arange(lines.shape[0]), np.arange(lines.shape[0]) % 4, :] = lines       # as a row
    cols = boards.copy().transpose(0, 2, 1).copy()                                      # and as a column
    sparse = _random_boards(rng, 100000)                                                # exponents up to 17: some skipped
    play = rng.integers(0, 12, size=(200000, 16)).astype(np.uint8)                      # what lut_guard accepts
    play[rng.random(play.shape) < 0.4] = 0
    dense = rng.integers(1, 4, size=(100000, 16)).astype(np.uint8)
    allb = np.ascontiguousarray(np.concatenate([boards.reshape(-1, 16), cols.reshape(-1, 16), sparse, play, dense]))
    skipped = np.zeros(1, np.int64)
    assert H.h_move_lut_check(_p(allb), allb.shape[0], _p(skipped)) == 0
    assert 0 < skipped[0] < allb.shape[0]          # the 2^15 + 2^15 lines and the boards above 2^15 were there


def test_counter_based_redeal_as_a_table_index(H):
    """deal_words_index (rocRAND spawn mode's re-deal as an index into the LDS table of dealt boards) == spawn_word twice
    on an empty board, for random word pairs, the edge words, and thresholds 0, 0.8, 1."""
    rng = np.random.default_rng(8)
    words = rng.integers(0, 2**32, size=(400000, 2), dtype=np.uint64).astype(np.uint32)
    edges = np.array([[0, 0], [0xFFFFFFFF, 0xFFFFFFFF], [0xF0000000, 0xEFFF0000], [0x10000000, 0x11110000], [0, 0xFFFF0000],
                      [0xFFFF0000, 0], [0x7FFFCCCC, 0x8000CCCD]], np.uint32)
    allw = np.ascontiguousarray(np.concatenate([edges, words]))
    for thr in (0, 52429, 65536, 1, 65535):
        assert H.h_deal_words_check(_p(allw), allw.shape[0], thr) == 0


def test_compaction_by_position_masks_equals_the_bubble(H):
    """compact_cells (every cell OR-ed into the position it lands on) against six pull_into_empty steps: all 18^4
    lines in every byte lane, and random cell words with many zeros."""
    rng = np.random.default_rng(4)
    lines = np.array(list(np.ndindex(18, 18, 18, 18)), dtype=np.uint8)              # (104976, 4): cells a, b, c, d
    words = np.zeros((lines.shape[0], 4, 4), np.uint8)                              # [case][cell word][byte lane]
    for lane in range(4):
        words[:, :, lane] = np.roll(lines, lane * 1009, axis=0)                     # different lines side by side
    rnd = rng.integers(0, 18, size=(300000, 4, 4)).astype(np.uint8)
    rnd[rng.random(rnd.shape) < 0.5] = 0
    allw = np.ascontiguousarray(np.concatenate([words, rnd]).reshape(-1, 16)).view(np.uint32)
    assert H.h_compact_check(_p(allw), allw.shape[0]) == 0


def test_spawn_from_inverse_permutation_equals_probing_spawn(H):
    """spawn_table_inv (one SDWA shift per cell + OR tree on the row's inverse permutation) and spawn_table_dot (dot
    products with the expanded row's weights - what the plain fused rollout runs) place the same tile as spawn_table
    for random boards (sparse, dense, full, empty) and random permutations; the expanded row is checked field by field."""
    rng = np.random.default_rng(21)
    n = 200000
    boards = _random_boards(rng, n)
    boards[: n // 4] = rng.integers(1, 6, size=(n // 4, 16))            # full boards: nothing to place
    k = n // 4
    holes = rng.integers(0, 16, size=k)
    boards[k:2 * k] = rng.integers(1, 6, size=(k, 16))
    boards[np.arange(k, 2 * k), holes] = 0                               # exactly one empty cell
    boards[-1] = 0
    perms = np.argsort(rng.random((n, 16)), axis=1).astype(np.uint8)
    masks = rng.integers(0, 1 << 16, size=n).astype(np.uint32)
    assert H.h_spawn_forms_check(_p(np.ascontiguousarray(boards)), _p(np.ascontiguousarray(perms)), _p(masks), n) == 0


def test_lazy_mask_forms_agree_with_the_eager_mask(H):
    """What the fused rollout uses when nothing reads the mask between steps (uniform sampling):
    `terminated` from "full and no equal neighbours", validity from "the move changes the board"."""
    rng = np.random.default_rng(16)
    n = 200000
    dense = rng.integers(1, 5, size=(n // 2, 16)).astype(np.uint8)       # full boards, many dead ends
    dense[rng.random(dense.shape) < 0.03] = 0
    checker = (np.add.outer(np.arange(4), np.arange(4)) % 2 + 1).reshape(1, 16).astype(np.uint8)
    boards = np.concatenate([_random_boards(rng, n // 2 - 3), dense, checker, checker[:, ::-1] * 3,
                             np.full((1, 16), 17, np.uint8)])
    assert H.h_lazy_mask_check(_p(boards), boards.shape[0]) == 0


def _rec(oracle, M):
    return np.zeros(M, dtype=oracle.RECORD_DTYPE)


def _replay_host(H, oracle, g, M, seed, kind):
    rng = oracle.HostRng(seed)
    data = _rec(oracle, M)
    data["terminated"] = 1
    prev_s = np.zeros((M, 16), np.uint8)
    prev_v = np.zeros((M, 4), np.uint8)
    gc = C.c_int64(0)
    actions = g["actions"]
    for s in range(actions.shape[0]):
        _, base_p = rng.draw_prepare()
        n = H.h_prepare(_p(data), M, 0, base_p, _p(rng.randperm), _p(rng.randfloat), 0.8, C.addressof(gc))
        assert n == g["n_reset"][s]
        a = np.ascontiguousarray(actions[s], dtype=np.uint8)
        base_s = rng.draw_step()
        H.h_step(_p(data), M, 0, _p(a), _p(prev_s), _p(prev_v), oracle.REWARD_KINDS[kind], 0.8, base_s,
                 _p(rng.randperm), _p(rng.randfloat))
        hs = hashlib.sha256(data.tobytes())
        hs.update(prev_s.tobytes())
        hs.update(prev_v.tobytes())
        assert hs.hexdigest() == str(g["step_sha"][s]), f"step {s}"
    assert gc.value == int(g["game_count"])


def test_device_arith_rollout_c1(H, oracle, golden_dir):
    _replay_host(H, oracle, np.load(os.path.join(golden_dir, "rollout_c1.npz")), 1024, 12345, "normal")


def test_device_arith_rollout_long(H, oracle, golden_dir):
    _replay_host(H, oracle, np.load(os.path.join(golden_dir, "rollout_long.npz")), 256, 1, "normal")


@pytest.mark.parametrize("kind", ["normal", "improved", "rank", "maxcell"])
def test_device_arith_rollout_rewards(H, oracle, golden_dir, kind):
    g = np.load(os.path.join(golden_dir, "rollout_rewards.npz"))
    sub = {k[len(kind) + 1:]: g[k] for k in g.files if k.startswith(kind + "_")}
    _replay_host(H, oracle, sub, 128, 77, kind)


def test_high_exponent_merges_match_oracle(H, oracle):
    """H9: merges of 65536/131072 tiles (bucket index 16/17) - board, reward, merged, mask."""
    M = 64
    rng = np.random.default_rng(9)
    host = oracle.HostRng(3)
    data = _rec(oracle, M)
    boards = rng.integers(14, 18, size=(M, 16)).astype(np.uint8)
    boards[rng.random((M, 16)) < 0.2] = 0
    for i in range(M):
        data["board"][i] = boards[i]
        v, _ = oracle.compute_valid_actions(boards[i])
        data["valid_actions"][i] = v
    ref = data.copy()
    actions = rng.integers(0, 4, size=M).astype(np.uint8)
    p1, v1 = np.zeros((M, 16), np.uint8), np.zeros((M, 4), np.uint8)
    p2, v2 = np.zeros((M, 16), np.uint8), np.zeros((M, 4), np.uint8)
    for kind in range(4):
        a, b = data.copy(), ref.copy()
        H.h_step(_p(a), M, 0, _p(actions), _p(p1), _p(v1), kind, 0.8, 17, _p(host.randperm), _p(host.randfloat))
        oracle.lib().orc_step(_p(b), M, 0, _p(actions), 0, _p(p2), _p(v2), kind, 0.8, 17,
                              _p(host.randperm), _p(host.randfloat))
        assert a.tobytes() == b.tobytes(), kind


def test_fast_mode_matches_oracle(H, oracle):
    M, N = 512, 200
    env = oracle.OracleVecGame(M, "improved", fast_seed=0xC0FFEE)
    env.reset(0)
    data = _rec(oracle, M)
    data["terminated"] = 1
    prev_s = np.zeros((M, 16), np.uint8)
    prev_v = np.zeros((M, 4), np.uint8)
    gc = C.c_int64(0)
    for s in range(N):
        (idx,) = env.prepare()
        n = H.h_prepare_fast(_p(data), M, 0, 0.8, C.c_uint64(0xC0FFEE), C.c_uint64(s), C.addressof(gc))
        assert n == idx.size
        a_o, lp_o = oracle.sample_actions(env._data["valid_actions"], 77, s, masked=(s % 2 == 0))
        a_h = np.zeros(M, np.uint8)
        lp_h = np.zeros(M, np.float32)
        vh = np.ascontiguousarray(data["valid_actions"])
        H.h_sample(_p(vh), M, 0, C.c_uint64(77), C.c_uint64(s), int(s % 2 == 0),
                   _p(a_h), _p(lp_h))
        assert np.array_equal(a_o, a_h)
        assert lp_o.tobytes() == lp_h.tobytes()
        env.step(a_o)
        H.h_step_fast(_p(data), M, 0, _p(a_h), _p(prev_s), _p(prev_v), 1, 0.8, C.c_uint64(0xC0FFEE), C.c_uint64(s))
        assert data.tobytes() == env._data.tobytes(), s
        assert H.h_digest(_p(data), M, 0, 1) == env.digest()
        assert H.h_digest(_p(data), M, 0, 0) == env.digest(with_id=False)
    assert gc.value == env._game_count and gc.value > M  # some games ended and were re-dealt


def test_sampler_statistics(oracle):
    """uniform over valid actions (policy/random.py:17-27 semantics); logp = -log(#valid)."""
    M = 200000
    rng = np.random.default_rng(1)
    valid = (rng.random((M, 4)) < 0.6).astype(np.uint8)
    a, lp = oracle.sample_actions(valid, 5, 9, masked=True)
    cnt = valid.sum(axis=1)
    has = cnt > 0
    assert np.all(valid[np.arange(M)[has], a[has]] == 1)
    assert np.all(a[~has] == 0) and np.all(lp[~has] == 0)
    np.testing.assert_allclose(lp[has], -np.log(cnt[has]).astype(np.float32), rtol=1e-6)
    sel = np.flatnonzero((valid == [1, 1, 1, 1]).all(axis=1))
    freq = np.bincount(a[sel], minlength=4) / sel.size
    assert np.all(np.abs(freq - 0.25) < 0.01)


def test_philox_matches_oracle(H, oracle):
    rng = np.random.default_rng(2)
    for _ in range(64):
        ctr = rng.integers(0, 2**32, size=4, dtype=np.uint64).astype(np.uint32)
        key = rng.integers(0, 2**32, size=2, dtype=np.uint64).astype(np.uint32)
        out = np.zeros(4, np.uint32)
        H.h_philox(_p(ctr), _p(key), _p(out))
        assert np.array_equal(out, oracle.philox(ctr, key))
