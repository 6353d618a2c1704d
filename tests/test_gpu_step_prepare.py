"""GPU parity of the one-launch policy step (`ml2048_step_prepare` = step(t) + prepare(t+1) in one kernel, with the
optional dense step outcome, one-hot / int64 observation, fused RandomPolicy draw and finished-games histogram)
against the CPU oracle driven exactly as the reference's loop is (runner.py:74-109: prepare -> policy -> step).
Integer / byte / index work: bit-exact everywhere."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("id", "step", "score", "reward", "board", "merged", "valid_actions", "terminated", "invalid")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


@pytest.fixture(scope="module")
def amd(torch_cuda):
    import ml2048_amd

    ml2048_amd._lib.load()
    return ml2048_amd


def _same_state(env, ref, what):
    a, b = env._data, ref._data
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), f"{what}: field {f} differs"
    assert a.tobytes() == b.tobytes(), what
    assert np.array_equal(env._prev_state, ref._prev_state), what
    assert np.array_equal(env._prev_valid_actions, ref._prev_valid_actions), what


def _pair(amd, oracle, M, reward, spawn, seed, two_prob=0.8):
    if spawn == "rocrand":
        env = amd.VecGame(M, reward, spawn="rocrand", spawn_seed=0xFACE, two_prob=two_prob)
        ref = oracle.OracleVecGame(M, reward, fast_seed=0xFACE, two_prob=two_prob)
    else:
        env = amd.VecGame(M, reward, two_prob=two_prob)
        ref = oracle.OracleVecGame(M, reward, two_prob=two_prob)
    env.reset(seed)
    ref.reset(seed)
    (i1,) = env.prepare()
    (i2,) = ref.prepare()
    assert np.array_equal(i1, i2)
    return env, ref


def _drive(amd, oracle, torch, M, N, reward, *, masked, spawn="table", seed=20241008, full_every=8, with_out=True,
           two_prob=0.8):
    """N fused env steps; the fused sampler provides the actions from step 1 on."""
    env, ref = _pair(amd, oracle, M, reward, spawn, seed, two_prob)
    pol = amd.RandomPolicy(seed=11, masked=masked)
    out = env.new_step_out() if with_out else None
    hist = torch.zeros(20, dtype=torch.int32, device="cuda")
    want_hist = np.zeros(20, np.int32)
    a, _ = pol.sample_actions_u8(env.valid_actions)   # actions of step 0 from the standalone sampler
    for t in range(N):
        ao, lpo = oracle.sample_actions(ref._data["valid_actions"], 11, t, masked=masked)
        assert np.array_equal(a.cpu().numpy(), ao), f"actions of step {t}"
        res = env.step_prepare_torch(a, out=out, sample=pol, hist=hist)
        r = ref.step(ao)
        if with_out and (t % full_every == 0 or t == N - 1):   # the step's outcome as of BEFORE the re-deal
            for name, key in (("state", "state"), ("valid_actions", "valid_actions"), ("merged", "merged"),
                              ("step", "step"), ("reward", "reward"), ("score", "score"),
                              ("terminated", "terminated"), ("invalid", "invalid"),
                              ("prev_state", "prev_state"), ("prev_valid_actions", "prev_valid_actions")):
                assert np.array_equal(out[name].cpu().numpy(), r[key]), f"step {t}: out[{name}]"
            assert np.array_equal(out["id"].cpu().numpy(), ref._data["id"]), f"step {t}: out[id]"
        want_hist += oracle.hist_maxcell(r["state"], r["terminated"])
        (ridx,) = ref.prepare()
        if t % full_every == 0 or t == N - 1:
            k = int(res["n_reset"].item())
            assert k == ridx.size, t
            assert np.array_equal(res["reset_indices"][:k].cpu().numpy(), ridx), t
            _same_state(env, ref, f"after fused step {t}")
            assert env._game_count == ref._game_count
        else:
            assert env.digest() == ref.digest(), f"digest differs after fused step {t}"
        a, lp = res["actions"], res["log_probs"]
        if t == N - 1:
            _, lpo2 = oracle.sample_actions(ref._data["valid_actions"], 11, t + 1, masked=masked)
            assert lp.cpu().numpy().tobytes() == lpo2.tobytes()
    assert np.array_equal(hist.cpu().numpy(), want_hist) and want_hist.sum() > 0 or N < 60
    assert pol.counter == N + 1
    assert env._rand_step == ref._rand_step
    return env, ref


@pytest.mark.parametrize("M,N,reward,masked", [
    (65536, 128, "normal", False),    # C2's shape: 256 workgroups, everything in registers; invalid moves (H3)
    (65536, 96, "improved", True),
    (1000, 200, "rank", True),        # ragged tail wave, long enough for many table regenerations
    (65, 150, "maxcell", False),
    (1, 120, "normal", True),
    (262144, 48, "normal", True),     # C3's size: 1024 workgroups
])
def test_step_prepare_vs_oracle_single_pass(amd, oracle, torch_cuda, M, N, reward, masked):
    _drive(amd, oracle, torch_cuda, M, N, reward, masked=masked)


@pytest.mark.parametrize("M,N,reward,masked", [
    (262145, 40, "improved", False),  # first size whose waves walk 2 x 64 boards
    (600000, 40, "normal", True),     # 3 iterations per wave, ragged
    (2097152, 24, "rank", False),     # C4's total size
])
def test_step_prepare_vs_oracle_large(amd, oracle, torch_cuda, M, N, reward, masked):
    _drive(amd, oracle, torch_cuda, M, N, reward, masked=masked, full_every=13)


@pytest.mark.parametrize("M,blocks,N,reward,masked", [
    (5000, 7, 60, "improved", True),      # 3 x 64 boards per wave, ragged
    (262144, 100, 30, "normal", False),   # 11 iterations per wave
    (70001, 64, 40, "maxcell", True),
])
def test_step_prepare_vs_oracle_multi_pass(amd, oracle, torch_cuda, monkeypatch, M, blocks, N, reward, masked):
    """The multi-pass form (more than 262 144 boards: a wave walks its boards 64 at a time, phase 2 re-reads what
    phase 1 wrote) with many iterations per wave, forced at test sizes through ML2048_SP_MAX_BLOCKS."""
    monkeypatch.setenv("ML2048_SP_MAX_BLOCKS", str(blocks))
    _drive(amd, oracle, torch_cuda, M, N, reward, masked=masked, full_every=7)
    # observations of the multi-pass form
    import torch.nn.functional as F
    torch = torch_cuda
    env = amd.VecGame(M)
    env.reset(3)
    env.prepare()
    pol = amd.RandomPolicy(seed=2, masked=True)
    obs = torch.empty((M, 16, 4, 4), dtype=torch.float32, device="cuda")
    a, _ = pol.sample_actions_u8(env.valid_actions)
    for t in range(70):
        a = env.step_prepare_torch(a, onehot=obs, sample=pol)["actions"]
    x = F.one_hot(env.board.cpu().long(), 16).float()
    assert torch.equal(obs.cpu(), torch.reshape(torch.permute(x, (0, 2, 1)), (-1, 16, 4, 4)))


@pytest.mark.parametrize("M", [4096, 300000])
def test_step_prepare_rocrand_spawn(amd, oracle, torch_cuda, M):
    _drive(amd, oracle, torch_cuda, M, 70, "improved", masked=True, spawn="rocrand", seed=5)


@pytest.mark.parametrize("spawn,two_prob", [("table", 0.37), ("table", 1.0), ("rocrand", 0.37), ("table", 0.0)])
def test_step_prepare_other_two_prob(amd, oracle, torch_cuda, spawn, two_prob):
    """H2: the 2/4 decision is `float64(randfloat[cell]) < two_prob`; the fused kernel takes it as a host-computed
    mask (table mode) or a 32-bit threshold (rocRAND mode)."""
    _drive(amd, oracle, torch_cuda, 3000, 90, "normal", masked=True, spawn=spawn, seed=8, two_prob=two_prob)


def test_step_prepare_without_optional_outputs_and_mixed_with_the_other_apis(amd, oracle, torch_cuda):
    """No out / obs / sampler / histogram; fused steps interleaved with per-step calls and a fused rollout: one draw
    stream, one state."""
    torch = torch_cuda
    M = 20000
    env, ref = _pair(amd, oracle, M, "normal", "table", 77)
    arng = np.random.default_rng(4)

    def fused(n):
        for _ in range(n):
            a = arng.integers(0, 4, size=M).astype(np.uint8)
            env.step_prepare_torch(torch.from_numpy(a).cuda())
            ref.step(a)
            ref.prepare()
        _same_state(env, ref, "fused")

    def per_step(n):
        for _ in range(n):
            a = arng.integers(0, 4, size=M)
            env.step(a)
            ref.step(a)
            (i1,) = env.prepare()
            (i2,) = ref.prepare()
            assert np.array_equal(i1, i2)
        _same_state(env, ref, "per-step")

    fused(40)
    per_step(25)
    fused(60)
    # a fused rollout starts with its own prepare (a no-op here: nothing is terminated), so the streams stay aligned
    out = env.rollout(50, masked=False, act_seed=3, act_counter=9, record_actions=True)
    acts = out["actions"].cpu().numpy()
    for t in range(50):
        ref.prepare()
        ref.step(acts[t])
    (i1,) = env.prepare()
    (i2,) = ref.prepare()
    assert np.array_equal(i1, i2)
    fused(30)
    assert env._game_count == ref._game_count and env._rand_step == ref._rand_step


@pytest.mark.parametrize("M", [1, 100, 4097, 70000, 262145])
def test_step_prepare_observations(amd, torch_cuda, M):
    """The fused one-hot / int64 observation == the reference's own expression on the post-prepare boards."""
    torch = torch_cuda
    import torch.nn.functional as F

    env = amd.VecGame(M)
    env.reset(3)
    env.prepare()
    pol = amd.RandomPolicy(seed=2, masked=True)
    obs = torch.empty((M, 16, 4, 4), dtype=torch.float32, device="cuda")
    long = torch.empty((M, 16), dtype=torch.int64, device="cuda")
    a, _ = pol.sample_actions_u8(env.valid_actions)
    for t in range(130):
        res = env.step_prepare_torch(a, onehot=obs, board_i64=long, sample=pol)
        a = res["actions"]
        if t % 43 == 0 or t == 129:
            board = env.board.cpu()
            x = F.one_hot(board.long(), 16).float()          # src/ml2048/policy/_network.py:91-95,105
            want = torch.reshape(torch.permute(x, (0, 2, 1)), (-1, 16, 4, 4))
            assert torch.equal(obs.cpu(), want), t
            assert torch.equal(long.cpu(), board.long()), t
            assert torch.equal(obs, env.onehot())
    assert env._game_count > M or M > 100000   # games were finished and re-dealt under the observation


def test_step_prepare_into_replay_slices(amd, oracle, torch_cuda):
    """ml2048_step_out pointed at [step] slices of REPLAY_SPEC-typed buffers (int8 / bool views): the direct-to-replay
    write of run_train3.py:125-155, equal to ml2048_record_transition after a separate step."""
    torch = torch_cuda
    from ml2048_amd import train_ops

    M, S = 3000, 40
    a_env, b_env = amd.VecGame(M, "improved"), amd.VecGame(M, "improved")
    for e in (a_env, b_env):
        e.reset(8)
        e.prepare()
    pa, pb = amd.RandomPolicy(seed=6, masked=True), amd.RandomPolicy(seed=6, masked=True)
    A = train_ops.new_buffers((1, S, M), "cuda", spec=train_ops.REPLAY_SPEC)
    B = train_ops.new_buffers((1, S, M), "cuda", spec=train_ops.REPLAY_SPEC)
    act, lp = pa.sample_actions_u8(a_env.valid_actions)
    for s in range(S):
        out = {"prev_state": A["state"][0, s], "prev_valid_actions": A["valid_actions"][0, s], "reward": A["reward"][0, s],
               "state": A["next_state"][0, s], "valid_actions": A["next_valid_actions"][0, s], "step": A["step"][0, s],
               "terminated": A["terminated"][0, s]}
        A["action"][0, s].copy_(act.view(torch.int8))
        A["action_log_prob"][0, s].copy_(lp)
        res = a_env.step_prepare_torch(act, out=out, sample=pa)
        act, lp = res["actions"], res["log_probs"]
        # the same step through the separate launches
        b_act, b_lp = pb.sample_actions_u8(b_env.valid_actions)
        b_env.step_torch(b_act)
        train_ops.record_transition(b_env, b_act, b_lp, B, 0, s)
        b_env.prepare_torch()
    for k in A:
        assert torch.equal(A[k], B[k]), k
    assert a_env._data.tobytes() == b_env._data.tobytes()


class _FirstValidPolicy:
    """A deterministic non-random policy with the reference's contract (state long (N,16), valid bool (N,4))."""

    def sample_actions(self, state, valid_actions, *, generator=None):
        import torch

        assert state.dtype == torch.long and valid_actions.dtype == torch.bool
        pref = (state.sum(dim=1) % 4).unsqueeze(1)                       # a state-dependent preference order
        order = (pref + torch.arange(4, device=state.device)) % 4
        ok = torch.gather(valid_actions, 1, order)
        first = torch.argmax(ok.to(torch.uint8), dim=1, keepdim=True)
        a = torch.gather(order, 1, first).squeeze(1)
        return a, torch.zeros(a.shape, dtype=torch.float32, device=a.device)


@pytest.mark.parametrize("policy_kind", ["random-masked", "random-unmasked", "torch"])
def test_fused_runner_equals_per_step_runner(amd, oracle, torch_cuda, policy_kind):
    """VecRunner(fused=True) - one launch per env step - emits the same events with the same contents as the
    reference-shaped loop (runner.py:74-109), and RunnerStats / EpisodeStats fed from them agree."""
    torch = torch_cuda
    M, N = 3000, 260
    envs = [amd.VecGame(M, "improved") for _ in range(2)]
    for e in envs:
        e.reset(21)
    runners = [amd.VecRunner(envs[0], M), amd.VecRunner(envs[1], M, fused=True)]

    def make_policy():
        if policy_kind == "torch":
            return _FirstValidPolicy()
        return amd.RandomPolicy(seed=4, masked=policy_kind == "random-masked")

    pols = [make_policy(), make_policy()]
    logs = [[], []]
    stats = [amd.RunnerStats(), amd.RunnerStats()]
    evals = [amd.EpisodeStats(envs[0], 2000), amd.EpisodeStats(envs[1], 2000)]
    for k in range(2):
        def on_prepared(env, idx, k=k):
            logs[k].append(("prepared", idx.copy()))

        def on_stepped(env, result, actions, lp, k=k):
            keys = ("state", "valid_actions", "merged", "step", "reward", "score", "terminated", "invalid",
                    "prev_state", "prev_valid_actions")
            logs[k].append(("stepped", {n: result[n].cpu().numpy().copy() for n in keys}, actions.cpu().numpy().copy(),
                            lp.cpu().numpy().copy()))

        runners[k].add_callback("prepared", on_prepared)
        runners[k].add_callback("stepped", on_stepped)
        runners[k].add_callback("stepped", stats[k].on_stepped)
        runners[k].add_callback("stepped", evals[k].on_stepped)
    for k in range(2):
        runners[k].step_many(pols[k], N)
    # the fused runner has already run (and announced) the prepare of step N; the per-step one does it next
    (idx,) = envs[0].prepare()
    logs[0].append(("prepared", idx))
    assert len(logs[0]) == len(logs[1]) == 2 * N + 1
    for i, (x, y) in enumerate(zip(logs[0], logs[1])):
        assert x[0] == y[0], i
        if x[0] == "prepared":
            assert np.array_equal(x[1], y[1]), i
        else:
            for n in x[1]:
                assert np.array_equal(x[1][n], y[1][n]), (i, n)
            assert np.array_equal(x[2].astype(np.int64), y[2].astype(np.int64)) and x[3].tobytes() == y[3].tobytes(), i
    assert envs[0]._data.tobytes() == envs[1]._data.tobytes()
    for s in stats:
        s.sync()
    assert np.array_equal(stats[0].counts, stats[1].counts) and stats[0].terminated_count == stats[1].terminated_count > 0
    assert evals[0].table() == evals[1].table() and evals[0].completed() > 0


def test_evaluate_fused_equals_evaluate(amd):
    a, b = amd.VecGame(512, "normal"), amd.VecGame(512, "normal")
    a.reset(5)
    b.reset(5)
    ta, sa = amd.evaluate(a, amd.RandomPolicy(seed=1, masked=True), 700, check_every=8)
    tb, sb = amd.evaluate(b, amd.RandomPolicy(seed=1, masked=True), 700, check_every=8, fused=True)
    assert ta == tb and sa == sb
