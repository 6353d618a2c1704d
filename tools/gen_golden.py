#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by executing the reference's own source.

Runs ONLY in the build container (needs /root/reference; that tree never travels to the GPU box).
`numba` is not installed here, so a tiny stand-in module (`njit` = identity decorator,
`prange` = range) is placed in sys.modules; the reference's `@njit` bodies in
`src/ml2048/game_numba.py` then execute unmodified as plain CPython over numpy arrays.

The single place where CPython+numpy 2 semantics differ from numba's typing is integer width in
the reward functions (`merged[7] * 256` on a uint8 scalar raises under NEP 50; numba computes in
int64).  The reference reward functions are therefore *called on int64 copies of their inputs*
(see `_as_i64`) - still the reference's source, evaluated at numba's integer width - and are
anchored by the known-answer cell of `playground.ipynb` (cell 11: 6144.0 / 42.0 / 2052.0).

Outputs (all small, committed):
  tests/golden/lines18.npz        exhaustive 18^4 `_push_row` + `_line_movable` table
  tests/golden/rng_schedule.npz   host PCG64 tables + draw schedule for several seeds
  tests/golden/rollout_c1.npz     config C1: M=1024, N=32, seed 12345, masked-uniform actions
  tests/golden/rollout_long.npz   M=256, N=1200, unmasked actions (termination/reset/ids/stale)
  tests/golden/rollout_rewards.npz  M=128, N=300 for each of the 4 reward functions
  tests/golden/reward_triples.npz random (state, prev, merged) -> 4 reward values
  tests/golden/eval_first_games.npz  eval_perf.py's loop (ReplayRecorder as collector) on M=64, rounds=150:
                                  the action sequence and the resulting per-max-tile table
  tests/golden/known_answer.json  the playground.ipynb cell-11 vector, re-derived here
  tests/golden/MANIFEST.json      sha256 of every file + numpy version used

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import hashlib
import itertools
import json
import os
import sys
import time
import types

import numpy as np

REFERENCE_SRC = "/root/reference/src"


def _install_numba_stub() -> None:
    stub = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    stub.njit = njit
    stub.prange = range
    stub.set_num_threads = lambda n: None
    sys.modules["numba"] = stub


def _load_reference():
    _install_numba_stub()
    sys.path.insert(0, REFERENCE_SRC)
    import ml2048.game_numba as gn  # noqa: E402

    return gn


def _as_i64(fn):
    """Call a reference reward function at numba's integer width (int64)."""

    def wrapped(state, prev_state, merged):
        return fn(
            np.asarray(state).astype(np.int64),
            np.asarray(prev_state).astype(np.int64),
            np.asarray(merged).astype(np.int64),
        )

    wrapped.__name__ = fn.__name__
    return wrapped


def sha(a: np.ndarray | bytes) -> str:
    if isinstance(a, np.ndarray):
        a = np.ascontiguousarray(a).tobytes()
    return hashlib.sha256(a).hexdigest()


# --------------------------------------------------------------------------------------
def gen_lines(gn, out):
    n = 18**4
    lines = np.zeros((n, 4), np.uint8)
    result = np.zeros((n, 4), np.uint8)
    buckets = np.zeros((n, 18), np.uint8)
    movable = np.zeros((n, 2), np.uint8)
    for i, line in enumerate(itertools.product(range(18), repeat=4)):
        lines[i] = line
        row = np.array(line, dtype=np.uint8)
        b = np.zeros(18, np.uint8)
        gn._push_row(row, 0, 1, b)
        result[i] = row
        buckets[i] = b
        f, bk = gn._line_movable(*[np.uint8(v) for v in line])
        movable[i] = (bool(f), bool(bk))
    np.savez_compressed(
        os.path.join(out, "lines18.npz"), result=result, buckets=buckets, movable=movable
    )
    return {
        "lines18.result.sha256": sha(result),
        "lines18.buckets.sha256": sha(buckets),
        "lines18.movable.sha256": sha(movable),
    }


def gen_rng(gn, out):
    """Tables after reset(seed) and the (u, regen, off_prepare, off_step) draw schedule."""
    payload = {}
    info = {}
    K = 400
    for seed in (0, 12345, 20241008):
        g = gn.VecGame(4)
        g.reset(seed)
        perm0 = g._randperm.copy()
        flt0 = g._randfloat.copy()
        sched = np.zeros((K, 4), np.int64)  # rand_step_at_prepare, row_base_prepare, row_base_step, regen
        table_sha = []
        nregen = 0
        perms = [perm0]
        flts = [flt0]
        # replay the draw order of prepare()/step() (game_numba.py:622-626, :670) through the
        # reference object itself, recording the private counters around each call.
        for k in range(K):
            before = g._randperm.copy()
            g.prepare()
            regen = int(not np.array_equal(before, g._randperm))
            # recover rand_offset for prepare: not stored by the reference; re-derive by
            # cloning the generator state is not possible after the fact, so instead we
            # snapshot the bit generator before the call in a second pass below.
            sched[k, 3] = regen
            if regen:
                nregen += 1
                perms.append(g._randperm.copy())
                flts.append(g._randfloat.copy())
            sched[k, 0] = g._rand_step
            g.step(np.zeros(4, np.int64))
        # second pass: re-derive the offsets with an independent Generator replaying the
        # documented draw order; cross-check it reproduces regen flags, tables and rand_step.
        rng = np.random.default_rng(seed)
        perm = np.tile(np.arange(16, dtype=np.uint8), (1024, 1))
        flt = np.empty(1024, np.float32)
        rng.permuted(perm, axis=1, out=perm)
        rng.random(dtype=np.float32, out=flt)
        assert np.array_equal(perm, perm0) and np.array_equal(flt, flt0)
        rand_step = 0
        ei = 0
        for k in range(K):
            u = rng.random()
            regen = int(u >= 0.9 or rand_step >= 1024)
            if regen:
                rand_step = 0
                rng.permuted(perm, axis=1, out=perm)
                rng.random(dtype=np.float32, out=flt)
                ei += 1
                assert np.array_equal(perm, perms[ei]) and np.array_equal(flt, flts[ei])
            assert regen == sched[k, 3], (seed, k)
            assert rand_step == sched[k, 0], (seed, k)
            off_p = int(rng.integers(0, 1024))
            off_s = int(rng.integers(0, 1024))
            sched[k, 1] = rand_step + off_p
            sched[k, 2] = rand_step + off_s
            rand_step += 1
        assert np.array_equal(perm, g._randperm) and np.array_equal(flt, g._randfloat)
        payload[f"s{seed}_perm0"] = perm0
        payload[f"s{seed}_flt0"] = flt0
        payload[f"s{seed}_sched"] = sched
        payload[f"s{seed}_perm_last"] = g._randperm.copy()
        payload[f"s{seed}_flt_last"] = g._randfloat.copy()
        payload[f"s{seed}_table_sha"] = np.array(
            [sha(p) + sha(f) for p, f in zip(perms, flts)]
        )
        info[f"rng.s{seed}.perm0.sha256"] = sha(perm0)
        info[f"rng.s{seed}.flt0.sha256"] = sha(flt0)
        info[f"rng.s{seed}.nregen"] = nregen
    np.savez_compressed(os.path.join(out, "rng_schedule.npz"), **payload)
    return info


def _masked_uniform_actions(rng, valid):
    """C1 action recipe: k=min(int(u*cnt),cnt-1); action = k-th set bit (L,R,U,D)."""
    m = valid.shape[0]
    u = rng.random(m)
    actions = np.zeros(m, np.int64)
    for i in range(m):
        bits = np.flatnonzero(valid[i])
        cnt = bits.size
        if cnt == 0:
            actions[i] = 0
            continue
        k = min(int(u[i] * cnt), cnt - 1)
        actions[i] = bits[k]
    return actions


def _run_rollout(gn, M, N, seed, reward_fn, action_mode, action_seed, keep_rewards=False):
    g = gn.VecGame(M, reward_fn)
    g.reset(seed)
    arng = np.random.default_rng(action_seed)
    step_sha = []
    actions_all = np.zeros((N, M), np.uint8)
    n_reset = np.zeros(N, np.int64)
    rewards = np.zeros((N, M), np.float32) if keep_rewards else None
    h_all = hashlib.sha256()
    for s in range(N):
        (idx,) = g.prepare()
        n_reset[s] = idx.size
        _, valid = g.observations()
        if action_mode == "masked":
            a = _masked_uniform_actions(arng, valid)
        else:
            a = arng.integers(0, 4, size=M)
        actions_all[s] = a
        res = g.step(a)
        b = g._data.tobytes()
        h_all.update(b)
        hs = hashlib.sha256(b)
        hs.update(res["prev_state"].tobytes())
        hs.update(res["prev_valid_actions"].tobytes())
        step_sha.append(hs.hexdigest())
        if keep_rewards:
            rewards[s] = res["reward"]
    out = dict(
        actions=actions_all,
        n_reset=n_reset,
        step_sha=np.array(step_sha),
        final_records=np.frombuffer(g._data.tobytes(), np.uint8).reshape(M, 64).copy(),
        final_prev_state=g._prev_state.copy(),
        final_prev_valid=g._prev_valid_actions.copy(),
        game_count=np.int64(g._game_count),
        rand_step=np.int64(g._rand_step),
        all_sha=np.array(h_all.hexdigest()),
    )
    if keep_rewards:
        out["rewards"] = rewards
    return out


def gen_rollouts(gn, out):
    info = {}
    t0 = time.time()
    c1 = _run_rollout(gn, 1024, 32, 12345, gn.reward_fn_normal, "masked", 999)
    np.savez_compressed(os.path.join(out, "rollout_c1.npz"), **c1)
    info["rollout_c1.all_sha256"] = str(c1["all_sha"])
    print(f"  c1 {time.time()-t0:.1f}s  sha={c1['all_sha']}", flush=True)

    t0 = time.time()
    lg = _run_rollout(gn, 256, 1200, 1, gn.reward_fn_normal, "unmasked", 4242)
    np.savez_compressed(os.path.join(out, "rollout_long.npz"), **lg)
    info["rollout_long.all_sha256"] = str(lg["all_sha"])
    info["rollout_long.game_count"] = int(lg["game_count"])
    print(f"  long {time.time()-t0:.1f}s games={int(lg['game_count'])}", flush=True)

    t0 = time.time()
    payload = {}
    for name in ("normal", "improved", "rank", "maxcell"):
        fn = getattr(gn, f"reward_fn_{name}")
        r = _run_rollout(gn, 128, 300, 77, fn, "unmasked", 5151, keep_rewards=True)
        for k, v in r.items():
            payload[f"{name}_{k}"] = v
        info[f"rollout_rewards.{name}.all_sha256"] = str(r["all_sha"])
    np.savez_compressed(os.path.join(out, "rollout_rewards.npz"), **payload)
    print(f"  rewards {time.time()-t0:.1f}s", flush=True)
    return info


def gen_eval(gn, out):
    """The reference's evaluation loop (eval_perf.py:64-102) with the reference's own ReplayRecorder
    (src/ml2048/replay.py:110-232) and VecGame, driven by pre-drawn actions instead of a policy network:
    masked-uniform for the first configuration (no invalid moves, as a trained policy), unmasked for the
    second (invalid moves make the recorder's `steps` differ from the record's `step` field)."""
    import torch  # the recorder takes the action vector as a tensor (replay.py:160)
    import ml2048.replay as rp

    payload, info = {}, {}
    for tag, M, rounds, seed, mode in (("masked", 64, 150, 31, "masked"), ("unmasked", 48, 120, 32, "unmasked")):
        game = gn.VecGame(M)
        game.reset(seed)
        recorder = rp.ReplayRecorder(M, M)
        arng = np.random.default_rng(900 + seed)
        stats = {}
        remaining, steps = rounds, 0
        actions_all = []
        while remaining > 0:
            (new_indices,) = game.prepare()                      # VecRunner.step_once, runner.py:80-109
            recorder.on_prepared(game, new_indices)
            _, valid = game.observations()
            a = _masked_uniform_actions(arng, valid) if mode == "masked" else arng.integers(0, 4, size=M)
            actions_all.append(np.asarray(a, np.uint8))
            result = game.step(a)
            recorder.on_stepped(game, result, torch.from_numpy(np.asarray(a)), None)
            steps += 1
            while recorder.ready_buffers and remaining > 0:      # eval_perf.py:84-99
                buffer = recorder.ready_buffers.popleft()
                if buffer.id >= rounds:
                    recorder.recording_threshold = 0
                    continue
                remaining -= 1
                e = stats.setdefault(2 ** buffer.maxcell, [0, 0, 0.0])
                e[0] += 1
                e[1] += buffer.steps
                e[2] += buffer.score
        keys = sorted(stats)
        payload[f"{tag}_M"] = np.int64(M)
        payload[f"{tag}_rounds"] = np.int64(rounds)
        payload[f"{tag}_seed"] = np.int64(seed)
        payload[f"{tag}_actions"] = np.stack(actions_all)
        payload[f"{tag}_tiles"] = np.array(keys, np.int64)
        payload[f"{tag}_count"] = np.array([stats[k][0] for k in keys], np.int64)
        payload[f"{tag}_step_sum"] = np.array([stats[k][1] for k in keys], np.int64)
        payload[f"{tag}_score_sum"] = np.array([stats[k][2] for k in keys], np.float64)
        info[f"eval_first_games.{tag}.steps"] = steps
        info[f"eval_first_games.{tag}.table"] = {str(k): stats[k] for k in keys}
        print(f"  eval {tag}: {steps} steps, table {info[f'eval_first_games.{tag}.table']}", flush=True)
    np.savez_compressed(os.path.join(out, "eval_first_games.npz"), **payload)
    return info


def gen_reward_triples(gn, out):
    rng = np.random.default_rng(31337)
    T = 4096
    state = rng.integers(0, 18, size=(T, 16)).astype(np.uint8)
    prev = rng.integers(0, 18, size=(T, 16)).astype(np.uint8)
    merged = rng.integers(0, 3, size=(T, 16)).astype(np.uint8)
    merged[:, 0] = 0  # "merged[0] should be always 0" (game_numba.py:419)
    # a few rows with the largest legal counts (8 merges in one bucket)
    merged[:16] = 0
    merged[np.arange(16), np.arange(16)] = 8
    merged[0, 0] = 0
    vals = np.zeros((T, 4), np.float64)
    fns = [gn.reward_fn_normal, gn.reward_fn_improved, gn.reward_fn_rank, gn.reward_fn_maxcell]
    for i in range(T):
        for j, fn in enumerate(fns):
            vals[i, j] = fn(state[i], prev[i], merged[i])
    np.savez_compressed(
        os.path.join(out, "reward_triples.npz"), state=state, prev=prev, merged=merged, values=vals
    )
    return {"reward_triples.values.sha256": sha(vals)}


def gen_known_answer(gn, raw, out):
    """playground.ipynb cell 11 (legacy np.random.seed(12322))."""
    np.random.seed(12322)
    state = np.random.randint(8, 11, (16,), dtype=np.int8)
    merged = np.zeros_like(state, dtype=np.int8)
    prev_state = state.copy()
    gn._step_left(state, merged)
    ka = {
        "source": "playground.ipynb cell 11 (outputs at JSON lines ~3900-3906)",
        "prev_state": prev_state.tolist(),
        "state": state.tolist(),
        "merged": merged.tolist(),
        "reward_fn_normal": float(gn.reward_fn_normal(state, prev_state, merged)),
        "reward_fn_rank": float(gn.reward_fn_rank(state, prev_state, merged)),
        "reward_fn_maxcell": float(gn.reward_fn_maxcell(state, prev_state, merged)),
        "reward_fn_improved": float(gn.reward_fn_improved(state, prev_state, merged)),
    }
    # the notebook's printed outputs - asserted here so a drift is caught at generation time
    assert ka["prev_state"] == [10, 10, 8, 10, 10, 9, 8, 10, 9, 10, 9, 9, 10, 10, 9, 9]
    assert ka["state"] == [11, 8, 10, 0, 10, 9, 8, 10, 9, 10, 10, 0, 11, 10, 0, 0]
    assert ka["merged"] == [0] * 9 + [2, 2] + [0] * 5
    assert ka["reward_fn_normal"] == 6144.0
    assert ka["reward_fn_rank"] == 42.0
    assert ka["reward_fn_maxcell"] == 2052.0
    with open(os.path.join(out, "known_answer.json"), "w") as f:
        json.dump(ka, f, indent=1)
    return {"known_answer": "ok"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default=None, help="regenerate one group only (e.g. eval); the others keep their MANIFEST entries")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)

    gn = _load_reference()
    raw = {n: getattr(gn, f"reward_fn_{n}") for n in ("normal", "improved", "rank", "maxcell")}
    # evaluate the reference reward functions at numba's integer width (module global lookup at
    # game_numba.py:729 and :453 picks these up)
    for n, fn in raw.items():
        setattr(gn, f"reward_fn_{n}", _as_i64(fn))

    info = {"numpy": np.__version__, "generated_by": "tools/gen_golden.py"}
    if args.only and os.path.exists(os.path.join(out, "MANIFEST.json")):
        with open(os.path.join(out, "MANIFEST.json")) as f:
            info.update({k: v for k, v in json.load(f).items() if k != "files"})
    for name, fn in (
        ("known_answer", lambda: gen_known_answer(gn, raw, out)),
        ("lines", lambda: gen_lines(gn, out)),
        ("rng", lambda: gen_rng(gn, out)),
        ("reward_triples", lambda: gen_reward_triples(gn, out)),
        ("rollouts", lambda: gen_rollouts(gn, out)),
        ("eval", lambda: gen_eval(gn, out)),
    ):
        if args.only and name != args.only:
            continue
        t0 = time.time()
        info.update(fn())
        print(f"{name}: {time.time()-t0:.1f}s", flush=True)

    files = {}
    for fn in sorted(os.listdir(out)):
        if fn == "MANIFEST.json":
            continue
        with open(os.path.join(out, fn), "rb") as f:
            files[fn] = hashlib.sha256(f.read()).hexdigest()
    info["files"] = files
    with open(os.path.join(out, "MANIFEST.json"), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in info.items() if k != "files"}, indent=1))


if __name__ == "__main__":
    main()
