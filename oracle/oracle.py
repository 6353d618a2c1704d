"""ctypes front end of the CPU oracle (oracle/ml2048_oracle.c).

TEST INFRASTRUCTURE ONLY - see the header of ml2048_oracle.h.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.
Parity status: PINNED (tests/test_oracle_golden.py).

`OracleVecGame` reproduces the host side of the reference's `VecGame`
(src/ml2048/game_numba.py:522-698): the same numpy PCG64 draw order, the same 64-byte record
array, the same method names, with the per-board work done by the C restatement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libml2048_oracle.so")

REWARD_KINDS = {"normal": 0, "improved": 1, "rank": 2, "maxcell": 3}
_ACT_CODE = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.int64): 2}

# game_numba.py:537-551
RECORD_DTYPE = np.dtype(
    [
        ("id", np.int32, ()),
        ("step", np.int32, ()),
        ("score", np.float32, ()),
        ("reward", np.float32, ()),
        ("board", np.uint8, (16,)),
        ("merged", np.uint8, (16,)),
        ("valid_actions", np.uint8, (4,)),
        ("terminated", np.uint8, ()),
        ("invalid", np.uint8, ()),
        ("_padding", np.uint8, 10),
    ],
    align=True,
)
assert RECORD_DTYPE.itemsize == 64


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ml2048_oracle.c")
    hdr = os.path.join(_HERE, "ml2048_oracle.h")
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr) if os.path.exists(p)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib(path: Optional[str] = None) -> C.CDLL:
    global _lib
    if path is not None:
        return _bind(C.CDLL(path))
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L: C.CDLL) -> C.CDLL:
    vp, i64, i32, u64, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_double
    L.orc_abi_version.restype = i32
    L.orc_num_threads.restype = i32
    L.orc_set_num_threads.argtypes = [i32]
    L.orc_push_row.argtypes = [vp, i32, i32, vp]
    L.orc_lines18_table.argtypes = [vp, vp, vp]
    L.orc_step_kernel.argtypes = [vp, vp, i64]
    L.orc_spawn2.argtypes = [vp, vp, i64, vp, dbl, i32]
    L.orc_spawn2.restype = i32
    L.orc_line_movable.argtypes = [i32, i32, i32, i32, vp, vp]
    L.orc_compute_valid_actions.argtypes = [vp, vp]
    L.orc_compute_valid_actions.restype = i32
    L.orc_reward.argtypes = [i32, vp, vp, vp]
    L.orc_reward.restype = dbl
    L.orc_reset_records.argtypes = [vp, i64, vp, vp]
    L.orc_prepare.argtypes = [vp, i64, i64, i64, vp, vp, dbl, vp, vp]
    L.orc_prepare.restype = i64
    L.orc_step.argtypes = [vp, i64, i64, vp, i32, vp, vp, i32, dbl, i64, vp, vp]
    L.orc_encode_onehot.argtypes = [vp, i64, vp]
    L.orc_encode_onehot.restype = i64
    L.orc_hist_maxcell.argtypes = [vp, i64, vp, i64, i64, vp]
    L.orc_philox4x32_10.argtypes = [vp, vp, vp]
    L.orc_sample_actions.argtypes = [vp, i64, i64, i64, u64, u64, i32, vp, vp]
    L.orc_gae.argtypes = [i64, i64, i64, vp, vp, vp, vp, dbl, dbl, vp, vp]
    L.orc_digest.argtypes = [vp, i64, i64, i32]
    L.orc_digest.restype = u64
    L.orc_step_fast.argtypes = [vp, i64, i64, vp, i32, vp, vp, i32, dbl, u64, u64]
    L.orc_prepare_fast.argtypes = [vp, i64, i64, dbl, u64, u64, vp, vp]
    L.orc_prepare_fast.restype = i64
    return L


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


# ---- thin function wrappers -------------------------------------------------------------------
def push_row(line, nbuckets: int = 18):
    b = np.array(line, dtype=np.uint8)
    buckets = np.zeros(nbuckets, np.uint8)
    lib().orc_push_row(_p(b), 0, 1, _p(buckets))
    return b, buckets


def lines18_table():
    n = 18**4
    result = np.zeros((n, 4), np.uint8)
    buckets = np.zeros((n, 18), np.uint8)
    movable = np.zeros((n, 2), np.uint8)
    lib().orc_lines18_table(_p(result), _p(buckets), _p(movable))
    return result, buckets, movable


def step_board(board, action: int):
    b = np.array(board, dtype=np.uint8)
    merged = np.zeros(18, np.uint8)
    lib().orc_step_kernel(_p(b), _p(merged), int(action))
    return b, merged


def line_movable(n1, n2, n3, n4):
    f, b = C.c_int(0), C.c_int(0)
    lib().orc_line_movable(int(n1), int(n2), int(n3), int(n4), C.addressof(f), C.addressof(b))
    return bool(f.value), bool(b.value)


def compute_valid_actions(board):
    b = np.ascontiguousarray(board, dtype=np.uint8)
    out = np.zeros(4, np.uint8)
    any_ = lib().orc_compute_valid_actions(_p(b), _p(out))
    return out, bool(any_)


def reward(kind, state, prev, merged) -> float:
    k = REWARD_KINDS[kind] if isinstance(kind, str) else int(kind)
    s = np.ascontiguousarray(state, dtype=np.uint8)
    p = np.ascontiguousarray(prev, dtype=np.uint8)
    m = np.ascontiguousarray(merged, dtype=np.uint8)
    return float(lib().orc_reward(k, _p(s), _p(p), _p(m)))


def philox(ctr, key):
    c = np.array(ctr, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def sample_actions(valid, seed, counter, masked=True, idx_base=0):
    valid = np.ascontiguousarray(valid, dtype=np.uint8)
    M = valid.shape[0]
    actions = np.zeros(M, np.uint8)
    logp = np.zeros(M, np.float32)
    lib().orc_sample_actions(
        _p(valid), 4, M, idx_base, C.c_uint64(seed), C.c_uint64(counter), int(masked), _p(actions), _p(logp)
    )
    return actions, logp


def encode_onehot(board):
    board = np.ascontiguousarray(board, dtype=np.uint8)
    M = board.shape[0]
    out = np.zeros((M, 16, 4, 4), np.float32)
    bad = lib().orc_encode_onehot(_p(board), M, _p(out))
    return out, int(bad)


def hist_maxcell(board, terminated=None):
    board = np.ascontiguousarray(board, dtype=np.uint8)
    counts = np.zeros(20, np.int32)
    t = None if terminated is None else np.ascontiguousarray(terminated, dtype=np.uint8)
    lib().orc_hist_maxcell(_p(board), 16, _p(t), 1, board.shape[0], _p(counts))
    return counts


def gae(v0, v1, reward, terminated, gamma, lambda_):
    """-> (adv, delta), float32 [U,S,G] (gae.py:50, :58-68)"""
    v0 = np.ascontiguousarray(v0, np.float32)
    v1 = np.ascontiguousarray(v1, np.float32)
    reward = np.ascontiguousarray(reward, np.float32)
    term = np.ascontiguousarray(terminated).astype(np.uint8)
    U, S, G = v0.shape
    adv = np.empty_like(v0)
    delta = np.empty_like(v0)
    lib().orc_gae(U, S, G, _p(v0), _p(v1), _p(reward), _p(term), float(gamma), float(lambda_), _p(adv), _p(delta))
    return adv, delta


# ---- host RNG schedule: game_numba.py:589-591, 606-611, 622-626, 670 -----------------------------
class HostRng:
    """The reference's PCG64 draw order, kept on the host exactly as VecGame does."""

    def __init__(self, seed=None):
        self.rand = np.random.default_rng(seed)
        self.rand_step = 0
        self.randperm = np.empty((1024, 16), np.uint8)
        self.randfloat = np.empty((1024,), np.float32)
        self.randperm[:, :] = np.arange(16).reshape((1, 16))
        self.regen()

    def regen(self):
        self.rand.permuted(self.randperm, axis=1, out=self.randperm)
        self.rand.random(dtype=np.float32, out=self.randfloat)

    def draw_prepare(self):
        """-> (regenerated?, row_base = rand_step + rand_offset)"""
        regen = False
        if self.rand.random() >= 0.9 or self.rand_step >= 1024:
            self.rand_step = 0
            self.regen()
            regen = True
        off = int(self.rand.integers(0, 1024))
        return regen, self.rand_step + off

    def draw_step(self):
        off = int(self.rand.integers(0, 1024))
        base = self.rand_step + off
        self.rand_step += 1
        return base


class OracleVecGame:
    """Reference-shaped VecGame (game_numba.py:522-698) on top of the C restatement."""

    def __init__(self, size, reward_fn="normal", *, two_prob=0.8, idx_base=0, fast_seed=None):
        if size <= 0:
            raise ValueError(f"size={size}")
        self._size = size
        self._two_prob = float(two_prob)
        self._reward_kind = REWARD_KINDS[reward_fn] if isinstance(reward_fn, str) else int(reward_fn)
        self._data = np.empty((size,), dtype=RECORD_DTYPE)
        self._prev_state = np.empty((size, 16), np.uint8)
        self._prev_valid_actions = np.empty((size, 4), np.uint8)
        self._game_count = 0
        self._idx_base = idx_base
        self._fast_seed = fast_seed  # not None -> fast (Philox) spawn mode
        self._fast_counter = 0
        self.reset()

    # game_numba.py:606-617
    def reset(self, seed=None):
        self._rng = HostRng(seed)
        self._fast_counter = 0
        lib().orc_reset_records(_p(self._data), self._size, _p(self._prev_state), _p(self._prev_valid_actions))

    @property
    def _rand_step(self):
        return self._rng.rand_step

    @property
    def _randperm(self):
        return self._rng.randperm

    @property
    def _randfloat(self):
        return self._rng.randfloat

    def observations(self):
        return self._data["board"], self._data["valid_actions"]

    # game_numba.py:619-658
    def prepare(self):
        indices = np.empty(self._size, np.int64)
        gc = C.c_int64(self._game_count)
        if self._fast_seed is None:
            _, row_base = self._rng.draw_prepare()
            n = lib().orc_prepare(
                _p(self._data), self._size, self._idx_base, row_base, _p(self._rng.randperm),
                _p(self._rng.randfloat), self._two_prob, C.addressof(gc), _p(indices),
            )
        else:
            n = lib().orc_prepare_fast(
                _p(self._data), self._size, self._idx_base, self._two_prob,
                C.c_uint64(self._fast_seed), C.c_uint64(self._fast_counter), C.addressof(gc), _p(indices),
            )
        self._game_count = gc.value
        return (indices[:n].copy(),)

    # game_numba.py:660-698
    def step(self, actions):
        actions = np.ascontiguousarray(actions)
        assert actions.shape == (self._size,), actions.shape
        if actions.dtype not in _ACT_CODE:
            actions = actions.astype(np.int64)
        if self._fast_seed is None:
            rand_seed = self._rng.draw_step()
            lib().orc_step(
                _p(self._data), self._size, self._idx_base, _p(actions), _ACT_CODE[actions.dtype],
                _p(self._prev_state), _p(self._prev_valid_actions), self._reward_kind, self._two_prob,
                rand_seed, _p(self._rng.randperm), _p(self._rng.randfloat),
            )
        else:
            lib().orc_step_fast(
                _p(self._data), self._size, self._idx_base, _p(actions), _ACT_CODE[actions.dtype],
                _p(self._prev_state), _p(self._prev_valid_actions), self._reward_kind, self._two_prob,
                C.c_uint64(self._fast_seed), C.c_uint64(self._fast_counter),
            )
            self._fast_counter += 1
        d = self._data
        return {
            "state": d["board"], "valid_actions": d["valid_actions"], "merged": d["merged"],
            "step": d["step"], "reward": d["reward"], "score": d["score"],
            "terminated": d["terminated"], "invalid": d["invalid"],
            "prev_state": self._prev_state, "prev_valid_actions": self._prev_valid_actions,
        }

    # game_numba.py:593-604
    def summary(self):
        counts = hist_maxcell(self._data["board"])
        total = int(counts.sum())
        entries = [(2 ** int(k), int(c), int(c) / total) for k, c in enumerate(counts) if c]
        entries.sort(key=lambda s: s[0], reverse=True)
        return entries

    def digest(self, with_id: bool = True) -> int:
        return int(lib().orc_digest(_p(self._data), self._size, self._idx_base, int(with_id)))


def eval_first_games(game: "OracleVecGame", next_actions, rounds: int, max_steps: int = 1_000_000):
    """TEST INFRASTRUCTURE: the reference's evaluation loop restated in plain Python - eval_perf.py:64-102
    driving ReplayRecorder's bookkeeping (src/ml2048/replay.py:133-200; thresholds = batch size as
    eval_perf.py:69 sets them): a game is recorded from the prepare() that deals it; buffer.steps += 1 per
    on_stepped (:167); on termination its (max tile, steps, final score) is queued (:178-200); the consumer adds
    games with id < rounds to stats[2**maxcell] and, on the first id >= rounds, stops new recordings (:84-90).
    `next_actions(game)` returns the action vector of the next step.  Returns ({tile: [count, step_sum,
    score_sum]}, steps played)."""
    recording = {}            # game id -> [slot, steps]   (dict order = recording order, as in the reference)
    recording_threshold = game._size
    stats = {}
    remaining, steps = rounds, 0
    while remaining > 0:
        if steps >= max_steps:
            raise RuntimeError("evaluation did not finish")
        (new_indices,) = game.prepare()
        free = min(new_indices.size, recording_threshold - len(recording))   # replay.py:140-143
        for i in range(max(0, free)):
            slot = int(new_indices[i])
            recording[int(game._data[slot]["id"])] = [slot, 0]
        result = game.step(next_actions(game))
        steps += 1
        ready = []
        for gid, rec in recording.items():                                     # replay.py:165-192
            rec[1] += 1
            if result["terminated"][rec[0]]:
                ready.append((gid, rec[1], int(result["state"][rec[0]].max()), float(result["score"][rec[0]])))
        for gid, _, _, _ in ready:
            recording.pop(gid)
        for gid, nsteps, maxcell, score in ready:                              # eval_perf.py:84-99
            if remaining <= 0:
                break
            if gid >= rounds:
                recording_threshold = 0
                continue
            remaining -= 1
            e = stats.setdefault(2 ** maxcell, [0, 0, 0.0])
            e[0] += 1
            e[1] += nsteps
            e[2] += score
    return stats, steps
