"""Episode statistics of the first N games: the reference's evaluation loop
(eval_perf.py:64-102 with ReplayRecorder, src/ml2048/replay.py:110-232).

CPU: the oracle's restatement of that loop against tables produced by the reference's own recorder
(tests/golden/eval_first_games.npz, tools/gen_golden.py --only eval).  GPU: the device kernel
(ml2048_episode_stats through ml2048_amd.EpisodeStats) against the same fixtures and against the oracle
on larger random runs."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def amd():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ml2048_amd

    ml2048_amd._lib.load()  # must be the in-tree HIP library; raises otherwise
    return ml2048_amd


def _golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "eval_first_games.npz"))
    table = {int(t): [int(c), int(s), float(v)] for t, c, s, v in
             zip(g[f"{tag}_tiles"], g[f"{tag}_count"], g[f"{tag}_step_sum"], g[f"{tag}_score_sum"])}
    return int(g[f"{tag}_M"]), int(g[f"{tag}_rounds"]), int(g[f"{tag}_seed"]), g[f"{tag}_actions"], table


@pytest.mark.parametrize("tag", ["masked", "unmasked"])
def test_oracle_eval_loop_matches_the_reference_recorder(oracle, golden_dir, tag):
    M, rounds, seed, actions, want = _golden(golden_dir, tag)
    game = oracle.OracleVecGame(M, "normal")
    game.reset(seed)
    it = iter(actions)
    got, steps = oracle.eval_first_games(game, lambda g: next(it), rounds)
    assert steps == actions.shape[0]          # the reference stopped after exactly as many steps
    assert {k: [v[0], v[1], float(v[2])] for k, v in got.items()} == want
    assert sum(v[0] for v in got.values()) == rounds


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["masked", "unmasked"])
def test_device_episode_stats_match_the_reference_recorder(amd, golden_dir, tag):
    M, rounds, seed, actions, want = _golden(golden_dir, tag)
    env = amd.VecGame(M, "normal")
    env.reset(seed)
    stats = amd.EpisodeStats(env, rounds)
    for s in range(actions.shape[0]):
        assert not stats.done, s              # not finished before the reference was
        env.prepare()
        env.step(actions[s])
        stats.on_stepped()
    assert stats.completed() == rounds
    got = {k: [v["count"], v["step_sum"], v["score_sum"]] for k, v in stats.table().items()}
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("M,rounds,masked", [(512, 1500, True), (1000, 700, False), (4096, 4096, True)])
def test_device_episode_stats_vs_oracle(amd, oracle, M, rounds, masked):
    """Larger runs with the device sampler: same table, same number of steps until the first `rounds`
    games are all over; stepping on afterwards (games with larger ids keep terminating) changes nothing."""
    env = amd.VecGame(M, "improved")
    env.reset(77)
    pol = amd.RandomPolicy(seed=3, masked=masked)
    stats = amd.EpisodeStats(env, rounds)
    ref = oracle.OracleVecGame(M, "improved")
    ref.reset(77)
    played = []

    def next_actions(game):
        env.prepare_torch()
        a, _ = pol.sample_actions_u8(env.valid_actions)
        env.step_torch(a)
        stats.on_stepped()
        played.append(stats.completed())
        return a.cpu().numpy()

    want, steps = oracle.eval_first_games(ref, next_actions, rounds)
    assert played[-1] == rounds and (len(played) < 2 or played[-2] < rounds)   # done at the same step
    got = {k: [v["count"], v["step_sum"], v["score_sum"]] for k, v in stats.table().items()}
    assert got == {k: [v[0], v[1], float(v[2])] for k, v in want.items()}
    assert env.digest() == ref.digest()
    for _ in range(50):
        env.prepare_torch()
        a, _ = pol.sample_actions_u8(env.valid_actions)
        env.step_torch(a)
        stats.on_stepped()
    assert {k: [v["count"], v["step_sum"], v["score_sum"]] for k, v in stats.table().items()} == got


@pytest.mark.gpu
def test_evaluate_helper_runs_a_policy_to_completion(amd):
    env = amd.VecGame(256, "normal")
    env.reset(5)
    table, steps = amd.evaluate(env, amd.RandomPolicy(seed=1, masked=True), 300, check_every=8)
    assert sum(e["count"] for e in table.values()) == 300 and steps % 8 == 0
    assert abs(sum(e["share"] for e in table.values()) - 1.0) < 1e-12


@pytest.mark.gpu
def test_evaluate_on_a_reused_env_counts_from_its_game_counter(amd):
    """`_game_count` survives reset() (game_numba.py:582 vs :606): on an env that has already played, "the first
    N games" are the N ids dealt after the reset.  Same table as a fresh env with the same seed."""
    fresh = amd.VecGame(256, "normal")
    fresh.reset(5)
    want, steps_want = amd.evaluate(fresh, amd.RandomPolicy(seed=1, masked=True), 300, check_every=8)
    used = amd.VecGame(256, "normal")
    used.reset(9)
    used.rollout(200, masked=True)            # plays and re-deals: the game counter is far above zero
    assert used._game_count > 256
    used.reset(5)
    got, steps_got = amd.evaluate(used, amd.RandomPolicy(seed=1, masked=True), 300, check_every=8, max_steps=20000)
    assert got == want and steps_got == steps_want
