"""Pin the CPU oracle (oracle/ml2048_oracle.c) against the reference's golden vectors.

Fixtures in tests/golden/ were produced by tools/gen_golden.py, which executes the reference's
own source (src/ml2048/game_numba.py); known_answer.json is the reference's only known-answer
vector (playground.ipynb cell 11).
"""
import hashlib
import itertools
import json
import os

import numpy as np
import pytest


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_manifest_files_intact(golden_dir):
    with open(os.path.join(golden_dir, "MANIFEST.json")) as f:
        man = json.load(f)
    for fn, h in man["files"].items():
        with open(os.path.join(golden_dir, fn), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == h, fn


def test_known_answer_playground_cell11(oracle, golden_dir):
    with open(os.path.join(golden_dir, "known_answer.json")) as f:
        ka = json.load(f)
    prev = np.array(ka["prev_state"], np.uint8)
    # literal values printed in the notebook
    assert prev.tolist() == [10, 10, 8, 10, 10, 9, 8, 10, 9, 10, 9, 9, 10, 10, 9, 9]
    state, merged = oracle.step_board(prev, 0)  # _step_left
    assert state.tolist() == [11, 8, 10, 0, 10, 9, 8, 10, 9, 10, 10, 0, 11, 10, 0, 0]
    assert merged[:16].tolist() == [0] * 9 + [2, 2] + [0] * 5
    assert oracle.reward("normal", state, prev, merged[:16]) == 6144.0
    assert oracle.reward("rank", state, prev, merged[:16]) == 42.0
    assert oracle.reward("maxcell", state, prev, merged[:16]) == 2052.0
    assert oracle.reward("improved", state, prev, merged[:16]) == ka["reward_fn_improved"]


def test_lines18_exhaustive(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "lines18.npz"))
    result, buckets, movable = oracle.lines18_table()
    # spot-check the single-line entry points against the batch one
    for i, line in enumerate(itertools.product(range(18), repeat=4)):
        if i % 37 == 0:
            r, b = oracle.push_row(line)
            assert np.array_equal(r, result[i]) and np.array_equal(b, buckets[i])
            assert oracle.line_movable(*line) == tuple(bool(v) for v in movable[i])
    assert np.array_equal(result, g["result"])
    assert np.array_equal(buckets, g["buckets"])
    assert np.array_equal(movable, g["movable"])
    # hashes recorded in SURVEY.md 8c (probed from the reference independently of this repo)
    assert _sha(result) == "a15964ffa44267c33e7717bcf29ceed7f60bd5e049458c470e990e2d64ee5f26"
    assert _sha(buckets) == "9bf9875f4a0c38ccc2ed3c5ea3b56db9f27da9e34e448182cf06ff35d2d533a4"


def test_reward_triples(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "reward_triples.npz"))
    kinds = ["normal", "improved", "rank", "maxcell"]
    state, prev, merged, values = g["state"], g["prev"], g["merged"], g["values"]
    for i in range(state.shape[0]):
        for j, k in enumerate(kinds):
            assert oracle.reward(k, state[i], prev[i], merged[i]) == values[i, j], (i, k)


@pytest.mark.parametrize("seed", [0, 12345, 20241008])
def test_host_rng_schedule(oracle, golden_dir, seed):
    g = np.load(os.path.join(golden_dir, "rng_schedule.npz"))
    rng = oracle.HostRng(seed)
    assert np.array_equal(rng.randperm, g[f"s{seed}_perm0"])
    assert np.array_equal(rng.randfloat, g[f"s{seed}_flt0"])
    sched = g[f"s{seed}_sched"]
    table_sha = list(g[f"s{seed}_table_sha"])
    ei = 0
    for k in range(sched.shape[0]):
        regen, base_p = rng.draw_prepare()
        assert int(regen) == sched[k, 3]
        if regen:
            ei += 1
            assert _sha(rng.randperm) + _sha(rng.randfloat) == table_sha[ei]
        assert rng.rand_step == sched[k, 0]
        assert base_p == sched[k, 1]
        assert rng.draw_step() == sched[k, 2]
    assert np.array_equal(rng.randperm, g[f"s{seed}_perm_last"])
    assert np.array_equal(rng.randfloat, g[f"s{seed}_flt_last"])


def test_rng_probe_values_seed12345(oracle):
    # SURVEY.md 8c-ii probe
    rng = oracle.HostRng(12345)
    assert rng.randperm[0].tolist() == [10, 11, 12, 6, 4, 1, 14, 7, 3, 8, 0, 2, 9, 5, 15, 13]
    np.testing.assert_allclose(rng.randfloat[:4], [0.15963054, 0.11662817, 0.12755048, 0.79059356], rtol=1e-7)


def _replay(oracle, g, M, seed, reward):
    env = oracle.OracleVecGame(M, reward)
    env.reset(seed)
    actions = g["actions"]
    h_all = hashlib.sha256()
    for s in range(actions.shape[0]):
        (idx,) = env.prepare()
        assert idx.size == g["n_reset"][s], s
        assert np.all(np.diff(idx) > 0)
        res = env.step(actions[s])
        b = env._data.tobytes()
        h_all.update(b)
        hs = hashlib.sha256(b)
        hs.update(res["prev_state"].tobytes())
        hs.update(res["prev_valid_actions"].tobytes())
        assert hs.hexdigest() == str(g["step_sha"][s]), f"step {s}"
    assert h_all.hexdigest() == str(g["all_sha"])
    assert env._game_count == int(g["game_count"])
    assert env._rand_step == int(g["rand_step"])
    assert np.array_equal(np.frombuffer(env._data.tobytes(), np.uint8).reshape(M, 64), g["final_records"])
    return env


def test_rollout_c1(oracle, golden_dir):
    """Config C1: M=1024, N=32, reset(12345), masked-uniform actions."""
    g = np.load(os.path.join(golden_dir, "rollout_c1.npz"))
    assert str(g["all_sha"]) == "0cfbbf64a3fe274f550f5b3103914c6d6ec06e57583f06535ecedae50348ff2a"
    env = _replay(oracle, g, 1024, 12345, "normal")
    assert float(env._data["score"].astype(np.float64).sum()) == 190020.0
    assert int(env._data["board"].max()) == 6
    assert env._rand_step == 11 and env._game_count == 1024


def test_rollout_long_termination_reset_ids(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "rollout_long.npz"))
    env = _replay(oracle, g, 256, 1, "normal")
    assert env._game_count == 2572  # many games ended and were re-dealt with ordered ids


@pytest.mark.parametrize("kind", ["normal", "improved", "rank", "maxcell"])
def test_rollout_each_reward_fn(oracle, golden_dir, kind):
    g = np.load(os.path.join(golden_dir, "rollout_rewards.npz"))
    sub = {k[len(kind) + 1 :]: g[k] for k in g.files if k.startswith(kind + "_")}
    env = oracle.OracleVecGame(128, kind)
    env.reset(77)
    for s in range(sub["actions"].shape[0]):
        env.prepare()
        res = env.step(sub["actions"][s])
        assert np.array_equal(res["reward"], sub["rewards"][s]), (kind, s)
        hs = hashlib.sha256(env._data.tobytes())
        hs.update(res["prev_state"].tobytes())
        hs.update(res["prev_valid_actions"].tobytes())
        assert hs.hexdigest() == str(sub["step_sha"][s])


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10."""
    z = oracle.philox([0, 0, 0, 0], [0, 0])
    assert [hex(v) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)
    assert [hex(v) for v in f] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    p = oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])
    assert [hex(v) for v in p] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
