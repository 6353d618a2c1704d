"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI,
against (a) the golden fixtures produced from the reference's own source and (b) the CPU oracle on
identical seeded inputs, plus size-independent properties at BASELINE.json's full sizes.
Integer / byte / index work: every comparison is bit-exact.
"""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


@pytest.fixture(scope="module")
def amd(torch_cuda):
    import ml2048_amd

    ml2048_amd._lib.load()  # must be the in-tree HIP library; raises otherwise
    return ml2048_amd


def _records_sha(env):
    h = hashlib.sha256(env._data.tobytes())
    h.update(env._prev_state.tobytes())
    h.update(env._prev_valid_actions.tobytes())
    return h.hexdigest()


def _assert_same_state(env, orc_env, what=""):
    a = env._data
    b = orc_env._data
    for f in ("id", "step", "score", "reward", "board", "merged", "valid_actions", "terminated", "invalid"):
        assert np.array_equal(a[f], b[f]), f"{what}: field {f} differs"
    assert a.tobytes() == b.tobytes(), what
    assert np.array_equal(env._prev_state, orc_env._prev_state), what
    assert np.array_equal(env._prev_valid_actions, orc_env._prev_valid_actions), what


# ---- (a) the reference's golden vectors, replayed on the GPU ---------------------------------------
def _replay_golden(amd, g, M, seed, reward):
    env = amd.VecGame(M, reward)
    env.reset(seed)
    actions = g["actions"]
    for s in range(actions.shape[0]):
        (idx,) = env.prepare()
        assert idx.dtype == np.int64 and idx.size == g["n_reset"][s], s
        assert np.all(np.diff(idx) > 0)
        env.step(actions[s])
        assert _records_sha(env) == str(g["step_sha"][s]), f"step {s}"
    assert env._game_count == int(g["game_count"])
    assert env._rand_step == int(g["rand_step"])
    assert np.array_equal(env._data.view(np.uint8).reshape(M, 64), g["final_records"])
    return env


def test_golden_c1(amd, golden_dir):
    """Config C1 of BASELINE.json (M=1024, N=32, reset(12345), masked-uniform actions)."""
    g = np.load(os.path.join(golden_dir, "rollout_c1.npz"))
    env = _replay_golden(amd, g, 1024, 12345, amd.reward_fn_normal)
    assert float(env.score.double().sum().item()) == 190020.0
    assert env.summary()[0][0] == 2**6


def test_golden_long_run_termination_reset_ids(amd, golden_dir):
    g = np.load(os.path.join(golden_dir, "rollout_long.npz"))
    _replay_golden(amd, g, 256, 1, None)


@pytest.mark.parametrize("kind", ["normal", "improved", "rank", "maxcell"])
def test_golden_each_reward_fn(amd, golden_dir, kind):
    g = np.load(os.path.join(golden_dir, "rollout_rewards.npz"))
    sub = {k[len(kind) + 1:]: g[k] for k in g.files if k.startswith(kind + "_")}
    env = amd.VecGame(128, getattr(amd, f"reward_fn_{kind}"))
    env.reset(77)
    for s in range(sub["actions"].shape[0]):
        env.prepare()
        res = env.step(sub["actions"][s].astype(np.int64))
        assert np.array_equal(res["reward"], sub["rewards"][s]), (kind, s)
        assert res["reward"].dtype == np.float32 and res["step"].dtype == np.int32
        assert _records_sha(env) == str(sub["step_sha"][s])


# ---- (b) HIP vs oracle on identical seeds --------------------------------------------------------------
def test_rocrand_philox_equals_oracle_philox(amd, oracle, torch_cuda):
    """The rocRAND device generator the kernels use == Philox4x32-10 as the oracle states it."""
    torch = torch_cuda
    L = amd._lib.load()
    n = 1000
    for seed, counter, base in [(0, 0, 0), (7, 3, 0), (0xDEADBEEFCAFEF00D, 2**33 + 5, 2**32 - 10), (1, 2**40, 123)]:
        out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        amd._lib.check(L.ml2048_debug_philox(None, n, base, seed, counter, out.data_ptr()), "philox")
        got = out.cpu().numpy().view(np.uint32)
        for i in (0, 1, 2, 17, 511, 999):
            idx = base + i
            exp = oracle.philox([counter & 0xFFFFFFFF, counter >> 32, idx & 0xFFFFFFFF, idx >> 32],
                                [seed & 0xFFFFFFFF, seed >> 32])
            assert np.array_equal(got[i], exp), (seed, counter, idx)


@pytest.mark.parametrize("masked", [True, False])
def test_sampler_vs_oracle(amd, oracle, torch_cuda, masked):
    torch = torch_cuda
    M = 100003
    rng = np.random.default_rng(3)
    valid = (rng.random((M, 4)) < 0.6).astype(np.uint8)
    pol = amd.RandomPolicy(seed=99, masked=masked, idx_base=5)
    pol.counter = 41
    a, lp = pol.sample_actions_u8(torch.from_numpy(valid).cuda())
    ao, lpo = oracle.sample_actions(valid, 99, 41, masked=masked, idx_base=5)
    assert np.array_equal(a.cpu().numpy(), ao)
    assert lp.cpu().numpy().tobytes() == lpo.tobytes()
    # reference contract (policy/random.py:17-27): long actions, float log-probs = -log(#valid)
    a2, lp2 = pol.sample_actions(None, torch.from_numpy(valid).cuda().bool())
    assert a2.dtype == torch.long and lp2.dtype == torch.float32 and a2.shape == (M,)
    if masked:
        cnt = valid.sum(axis=1)
        has = cnt > 0
        ref_lp = torch.distributions.Categorical(probs=torch.from_numpy(valid[has]).float()).log_prob(a2.cpu()[has])
        assert torch.allclose(lp2.cpu()[has], ref_lp, rtol=1e-6, atol=2.5e-7)  # fp32; torch clamps p=1 to 1-eps, so log p = -1.2e-7 there


def _run_pair(amd, oracle, M, N, seed, reward, *, masked=False, check_every=1, full_every=16, spawn="table",
              idx_base=0):
    kw = dict(idx_base=idx_base)
    if spawn == "rocrand":
        env = amd.VecGame(M, reward, spawn="rocrand", spawn_seed=0xABCDEF, **kw)
        ref = oracle.OracleVecGame(M, reward if isinstance(reward, str) else reward.name, fast_seed=0xABCDEF, **kw)
    else:
        env = amd.VecGame(M, reward, **kw)
        ref = oracle.OracleVecGame(M, reward if isinstance(reward, str) else reward.name, **kw)
    env.reset(seed)
    ref.reset(seed)
    pol = amd.RandomPolicy(seed=7, masked=masked, idx_base=idx_base)
    for s in range(N):
        (idx,) = env.prepare()
        (ridx,) = ref.prepare()
        assert np.array_equal(idx, ridx), s
        _, valid = env.observations_torch()
        a, _ = pol.sample_actions_u8(valid)
        env.step_torch(a)
        ref.step(a.cpu().numpy())
        if s % check_every == 0:
            assert env.digest() == ref.digest(), f"digest differs at step {s}"
        if s % full_every == 0 or s == N - 1:
            _assert_same_state(env, ref, f"step {s}")
    assert env._game_count == ref._game_count
    return env, ref


def test_c2_stepwise_vs_oracle(amd, oracle):
    """Config C2: M=65536, N=128, reset(20241008), unmasked uniform actions (exercises invalid
    moves, H3), digest every step + full field-by-field compare every 16 steps."""
    env, ref = _run_pair(amd, oracle, 65536, 128, 20241008, "normal")
    assert ref._game_count > 65536  # games ended and were re-dealt inside the window


@pytest.mark.parametrize("M", [1, 63, 64, 65, 1000, 1025])
def test_ragged_sizes(amd, oracle, M):
    _run_pair(amd, oracle, M, 150, 5, "improved", masked=(M % 2 == 0), full_every=10)


@pytest.mark.parametrize("kind", ["rank", "maxcell"])
def test_other_rewards_vs_oracle(amd, oracle, kind):
    _run_pair(amd, oracle, 4096, 200, 11, kind, full_every=25)


def test_fast_rocrand_spawn_vs_oracle(amd, oracle):
    env, ref = _run_pair(amd, oracle, 8192, 200, 0, "normal", spawn="rocrand", full_every=20)
    # statistically the same game: ~80 % of spawned tiles are 2s (two_prob)


def test_idx_base_shards_equal_one_global_game(amd, oracle):
    """Global mode: two shards with idx_base 0 / M/2 reproduce one game of size M
    (everything but the ids, which need the cross-shard scan)."""
    M, N, seed = 4096, 60, 3
    whole = oracle.OracleVecGame(M, "normal")
    whole.reset(seed)
    shards = [amd.VecGame(M // 2, "normal", idx_base=k * (M // 2)) for k in range(2)]
    for sh in shards:
        sh.reset(seed)
    arng = np.random.default_rng(1)
    for s in range(N):
        whole.prepare()
        for sh in shards:
            sh.prepare_torch()
        a = arng.integers(0, 4, size=M).astype(np.uint8)
        whole.step(a)
        for k, sh in enumerate(shards):
            sh.step(a[k * (M // 2):(k + 1) * (M // 2)])
        total = sum(sh.digest(with_id=False) for sh in shards) & 0xFFFFFFFFFFFFFFFF
        assert total == whole.digest(with_id=False), s
    got = np.concatenate([sh._data for sh in shards])
    for f in ("step", "score", "reward", "board", "merged", "valid_actions", "terminated", "invalid"):
        assert np.array_equal(got[f], whole._data[f]), f


def test_global_mode_shards_equal_one_reference_game_including_ids(amd, oracle):
    """SURVEY 8e global mode: three shards (here on one GPU) == one game of the total size, ids and
    the returned reset indices included."""
    from ml2048_amd.shard import GlobalVecGame

    M, N, seed = 3001, 400, 12
    g = GlobalVecGame(M, "improved", devices=["cuda:0"] * 3)
    ref = oracle.OracleVecGame(M, "improved")
    g.reset(seed)
    ref.reset(seed)
    arng = np.random.default_rng(2)
    for s in range(N):
        (i1,) = g.prepare()
        (i2,) = ref.prepare()
        assert np.array_equal(i1, i2), s
        a = arng.integers(0, 4, size=M)
        r1 = g.step(a)
        r2 = ref.step(a)
        if s % 50 == 0 or s == N - 1:
            assert g._data.tobytes() == ref._data.tobytes(), s
            assert np.array_equal(r1["prev_state"], r2["prev_state"])
    assert g._game_count == ref._game_count > M
    assert g.summary() == ref.summary()


def test_high_exponent_boards_h9_h10(amd, oracle, torch_cuda):
    """Tiles 2^14..2^17: merges whose bucket index is 16/17 (H9) and un-encodable one-hots (H10)."""
    M = 512
    rng = np.random.default_rng(9)
    ref = oracle.OracleVecGame(M, "maxcell")
    ref.reset(4)
    boards = rng.integers(13, 18, size=(M, 16)).astype(np.uint8)
    boards[rng.random((M, 16)) < 0.25] = 0
    for i in range(M):
        ref._data["board"][i] = boards[i]
        ref._data["valid_actions"][i] = oracle.compute_valid_actions(boards[i])[0]
        ref._data["terminated"][i] = 0
    env = amd.VecGame(M, amd.reward_fn_maxcell)
    env.reset(4)
    env.load_records(ref._data)
    a = rng.integers(0, 4, size=M)
    for _ in range(3):
        env.step(a)
        ref.step(a)
        _assert_same_state(env, ref)
    with pytest.raises(RuntimeError):
        env.onehot(check=True)


# ---- the one-wavefront-per-board step kernel (north_star's mapping) gives the same bits ------------------
def test_wave_per_board_step_golden_c1(amd, golden_dir):
    g = np.load(os.path.join(golden_dir, "rollout_c1.npz"))
    env = amd.VecGame(1024, amd.reward_fn_normal, step_mapping="wave")
    env.reset(12345)
    for s in range(g["actions"].shape[0]):
        env.prepare()
        env.step(g["actions"][s])
        assert _records_sha(env) == str(g["step_sha"][s]), f"step {s}"


@pytest.mark.parametrize("M,kind", [(65536, "normal"), (1000, "improved"), (77, "maxcell"), (4096, "rank")])
def test_wave_per_board_step_vs_oracle(amd, oracle, M, kind):
    env = amd.VecGame(M, kind, step_mapping="wave")
    ref = oracle.OracleVecGame(M, kind)
    env.reset(20241008)
    ref.reset(20241008)
    pol = amd.RandomPolicy(seed=11, masked=False)
    for s in range(100):
        env.prepare_torch()
        ref.prepare()
        a, _ = pol.sample_actions_u8(env.valid_actions)
        env.step_torch(a)
        ref.step(a.cpu().numpy())
        assert env.digest() == ref.digest(), s
    _assert_same_state(env, ref)


# ---- fused rollout ----------------------------------------------------------------------------------
def _check_rollout(amd, oracle, M, N, seed, reward, masked):
    env = amd.VecGame(M, reward)
    env.reset(seed)
    out = env.rollout(N, masked=masked, act_seed=7, act_counter=100, record_actions=True, digests=True)
    actions = out["actions"].cpu().numpy()
    dig = out["digests"].cpu().numpy().view(np.uint64)
    resets = out["step_resets"].cpu().numpy()
    ref = oracle.OracleVecGame(M, reward)
    ref.reset(seed)
    for s in range(N):
        (ridx,) = ref.prepare()
        assert resets[s] == ridx.size, s
        if masked:  # the recorded actions must be what the sampler yields on the oracle's masks
            ao, _ = oracle.sample_actions(ref._data["valid_actions"], 7, 100 + s, masked=True)
            assert np.array_equal(ao, actions[s]), s
        ref.step(actions[s])
        assert int(dig[s]) == ref.digest(with_id=False), f"rollout digest differs at step {s}"
    _assert_same_state(env, ref, "after rollout")  # includes the ids assigned after the fact
    assert env._game_count == ref._game_count
    assert env._rand_step == ref._rand_step
    # the per-step API continues seamlessly after a fused rollout
    (i1,) = env.prepare()
    (i2,) = ref.prepare()
    assert np.array_equal(i1, i2)
    a = np.random.default_rng(0).integers(0, 4, size=M)
    env.step(a)
    ref.step(a)
    _assert_same_state(env, ref, "step after rollout")


def test_rollout_c2_fused_vs_oracle(amd, oracle):
    """Config C2 as ONE launch: per-step digests, sampled actions, re-deal counts, final records
    and ordered game ids all equal the oracle's."""
    _check_rollout(amd, oracle, 65536, 128, 20241008, "normal", masked=False)


def test_rollout_c4_total_size_on_one_gpu_vs_oracle(amd, oracle):
    """BASELINE config 4's TOTAL size (M = 2 097 152 boards = 8 x 262 144) as one fused rollout on one
    GPU, against the oracle: per-step digests of every board, re-deal counts, final records and ids.
    (The 8-GPU run shards this into 8 independent replicas; this is the largest single-shard case.)"""
    M, N = 2097152, 48
    env = amd.VecGame(M, "normal")
    env.reset(20241008)
    out = env.rollout(N, masked=False, act_seed=7, act_counter=0, record_actions=True, digests=True)
    actions = out["actions"].cpu().numpy()
    dig = out["digests"].cpu().numpy().view(np.uint64)
    resets = out["step_resets"].cpu().numpy()
    ref = oracle.OracleVecGame(M, "normal")
    ref.reset(20241008)
    for s in range(N):
        (ridx,) = ref.prepare()
        assert resets[s] == ridx.size, s
        ref.step(actions[s])
        assert int(dig[s]) == ref.digest(with_id=False), f"digest differs at step {s}"
    assert env.digest() == ref.digest()          # includes the ids assigned after the fact
    assert env._game_count == ref._game_count >= M
    assert env._data.tobytes() == ref._data.tobytes()


@pytest.mark.parametrize("M,N,masked,reward", [(1000, 300, True, "improved"), (65, 64, False, "rank"),
                                               (4096, 256, True, "maxcell"),
                                               (130, 8300, False, "normal")])  # N > 8192: the 3-kernel id scan
def test_rollout_small_cases(amd, oracle, M, N, masked, reward):
    _check_rollout(amd, oracle, M, N, 12345, reward, masked)


def test_rollout_fast_mode_vs_oracle(amd, oracle):
    """Fused rollout in rocRAND spawn mode == the oracle's counter-based spawn, step for step."""
    M, N = 20000, 160
    env = amd.VecGame(M, "improved", spawn="rocrand", spawn_seed=0x5EED)
    ref = oracle.OracleVecGame(M, "improved", fast_seed=0x5EED)
    env.reset(0)
    ref.reset(0)
    for part in range(2):  # two consecutive rollouts: the spawn counter carries over
        out = env.rollout(N // 2, masked=True, act_seed=9, act_counter=part * (N // 2), record_actions=True,
                          digests=True)
        acts = out["actions"].cpu().numpy()
        dig = out["digests"].cpu().numpy().view(np.uint64)
        for s in range(N // 2):
            (ridx,) = ref.prepare()
            assert int(out["step_resets"][s].item()) == ridx.size
            ref.step(acts[s])
            assert int(dig[s]) == ref.digest(with_id=False), (part, s)
    _assert_same_state(env, ref, "fast rollout")
    assert env._game_count == ref._game_count > M
    # and the per-step fast API continues from the same counters
    env.prepare()
    ref.prepare()
    a = np.random.default_rng(1).integers(0, 4, size=M)
    env.step(a)
    ref.step(a)
    _assert_same_state(env, ref, "fast step after fast rollout")


# ---- the PCG64 draw schedule computed on the GPU (csrc/devrng.hip) --------------------------------------
# seed 519423726 draws 26 table regenerations in its first 128 steps: more than one window of generator outputs
# covers, so the data-parallel kernels hand the rest of the schedule to the single-workgroup continuation
@pytest.mark.parametrize("seed", [0, 12345, 20241008, 987654321, 519423726])
def test_device_schedule_equals_host_schedule(amd, torch_cuda, seed):
    """Schedule, epoch tables, 2/4 masks, final generator state, tables and rand_step: the device
    kernel against the native host implementation (itself checked against numpy)."""
    from ml2048_amd.hostrng import HostRng

    env = amd.VecGame(64, schedule="device")
    env.reset(seed)
    ref = HostRng(seed)
    for n in (128, 1, 7, 300, 128, 1100):
        perm, sched, mask, cap, _ = env._device_schedule(n)
        torch_cuda.cuda.synchronize()
        ne = int(env._sched_pending_slot["ne"].item())
        s2, e2, m2 = ref.draw_schedule(n, 0.8)
        assert ne == e2.shape[0], (seed, n)
        assert np.array_equal(sched.cpu().numpy(), s2), (seed, n)
        assert np.array_equal(perm[:ne].cpu().numpy(), e2), (seed, n)
        assert np.array_equal(mask[:ne].cpu().numpy().view(np.uint32), m2), (seed, n)
    assert env._rand_step == ref.rand_step  # pulls the device generator back to the host
    assert np.array_equal(env._randperm, ref.randperm) and np.array_equal(env._randfloat, ref.randfloat)
    assert env._rng.rand.bit_generator.state == ref.rand.bit_generator.state
    assert env._rng.rand.random() == ref.rand.random()


def test_device_schedule_rollouts_equal_host_schedule_rollouts(amd):
    """Fused rollouts with the schedule drawn on the GPU leave exactly the state that host-drawn
    ones do, and the per-step API continues from it."""
    M = 8192
    a = amd.VecGame(M, "improved", schedule="device")
    b = amd.VecGame(M, "improved", schedule="host")
    a.reset(31)
    b.reset(31)
    for k, n in enumerate((128, 64, 128, 200)):
        a.rollout(n, masked=True, act_seed=2, act_counter=1000 * k)
        b.rollout(n, masked=True, act_seed=2, act_counter=1000 * k)
    assert a._data.tobytes() == b._data.tobytes()
    assert a._game_count == b._game_count and a._rand_step == b._rand_step
    act = np.random.default_rng(5).integers(0, 4, size=M)
    for env in (a, b):
        env.prepare()
        env.step(act)
    assert a._data.tobytes() == b._data.tobytes()
    a.rollout(50)  # and back to the device schedule after host-side draws
    b.rollout(50)
    assert a.digest() == b.digest()


def test_runner_step_many_fuses_random_rollouts(amd):
    """VecRunner.step_many(RandomPolicy, n) without listeners is one fused launch; with a listener it is the
    reference's loop.  Both leave the same records, ids, game counter, generator state and policy counter."""
    M = 3000
    a, b = amd.VecGame(M, "rank"), amd.VecGame(M, "rank")
    a.reset(11)
    b.reset(11)
    pa, pb = amd.RandomPolicy(seed=9, masked=False), amd.RandomPolicy(seed=9, masked=False)
    ra, rb = amd.VecRunner(a, M), amd.VecRunner(b, M)
    seen = []
    rb.add_callback(amd.VecRunner.EVENT_STEPPED, lambda env, result, actions, lp: seen.append(1))
    for n in (40, 1, 128):
        ra.step_many(pa, n)   # fused
        rb.step_many(pb, n)   # per step (a listener is registered)
        assert a._data.tobytes() == b._data.tobytes(), n
        assert a._game_count == b._game_count and a._rand_step == b._rand_step
        assert pa.counter == pb.counter
    assert len(seen) == 169
    ra.step_once(pa)
    rb.step_once(pb)
    assert a.digest() == b.digest()


@pytest.mark.parametrize("mode", ["host-async", "host"])
def test_deferred_and_inline_id_scans_mixed(amd, mode):
    """Deferred and inline id scans in any order number the games as all-inline scans do (the pipelined plain rollout
    issues a deferred scan lazily: an inline one behind it must not overtake it)."""
    M = 8192
    a = amd.VecGame(M, "normal", schedule=mode)
    b = amd.VecGame(M, "normal", schedule=mode)
    a.reset(5)
    b.reset(5)
    c = 0
    for fin in ["deferred", True, "deferred", "deferred", True, True, "deferred", False, "deferred", True]:
        a.rollout(100, masked=False, act_seed=3, act_counter=c, finish_ids=fin)
        b.rollout(100, masked=False, act_seed=3, act_counter=c, finish_ids=True if fin else False)
        c += 100
    assert np.array_equal(a.game_id.cpu().numpy(), b.game_id.cpu().numpy())
    assert a._game_count == b._game_count and a._data.tobytes() == b._data.tobytes()


def test_deferred_id_scan_equals_inline(amd):
    """rollout(finish_ids="deferred") (id scan on a side stream, overlapped with the next rollout) leaves
    the same ids, game counter, per-step re-deal counts and records as the inline scan - also when the
    per-step API, a differently sized rollout or an inline rollout follows."""
    M = 8192
    a = amd.VecGame(M, "normal", schedule="host-async")
    b = amd.VecGame(M, "normal", schedule="host")
    a.reset(404)
    b.reset(404)
    act = np.random.default_rng(9).integers(0, 4, size=M)
    for k, n in enumerate((128, 128, 32, 200, 128)):
        oa = a.rollout(n, masked=True, act_seed=3, act_counter=500 * k, finish_ids="deferred")
        ob = b.rollout(n, masked=True, act_seed=3, act_counter=500 * k)
        if k == 1:  # two deferred scans back to back, then the per-step API (which numbers new games)
            for env in (a, b):
                env.prepare()
                env.step(act)
        if k == 3:
            a.join_ids()
            assert np.array_equal(oa["step_resets"].cpu().numpy(), ob["step_resets"].cpu().numpy())
            oa = a.rollout(64, masked=False, act_seed=4)  # an inline scan right after a deferred one
            ob = b.rollout(64, masked=False, act_seed=4)
            assert np.array_equal(oa["step_resets"].cpu().numpy(), ob["step_resets"].cpu().numpy())
    assert np.array_equal(a.game_id.cpu().numpy(), b.game_id.cpu().numpy())
    assert a._game_count == b._game_count
    assert a._data.tobytes() == b._data.tobytes()
    assert a.digest() == b.digest()


def test_async_host_schedule_equals_sync(amd):
    """schedule='host-async' (native producer thread drawing ahead) == schedule='host', including
    switching to the per-step API in between (the drawn-ahead schedules are then discarded)."""
    M = 4096
    a = amd.VecGame(M, "rank", schedule="host-async")
    b = amd.VecGame(M, "rank", schedule="host")
    a.reset(77)
    b.reset(77)
    act = np.random.default_rng(5).integers(0, 4, size=M)
    for rnd in range(3):
        for k in range(4):
            a.rollout(96, masked=False, act_seed=1, act_counter=rnd * 1000 + k * 96)
            b.rollout(96, masked=False, act_seed=1, act_counter=rnd * 1000 + k * 96)
        assert a.digest() == b.digest()
        assert a._rand_step == b._rand_step  # closes the stream, re-synchronises the host generator
        for env in (a, b):
            env.prepare()
            env.step(act)
        assert a._data.tobytes() == b._data.tobytes()
        a.rollout(17)  # a different length re-opens the stream
        b.rollout(17)
    assert a._data.tobytes() == b._data.tobytes() and a._game_count == b._game_count
    a.reset(3)
    b.reset(3)
    a.rollout(40)
    b.rollout(40)
    assert a.digest() == b.digest()


def test_rollout_equals_stepwise_gpu(amd):
    """Same seeds -> the fused launch and N per-step launches leave identical device state."""
    M, N = 32768, 96
    a = amd.VecGame(M)
    b = amd.VecGame(M)
    a.reset(42)
    b.reset(42)
    a.rollout(N, masked=True, act_seed=5, act_counter=0)
    pol = amd.RandomPolicy(seed=5, masked=True)
    for _ in range(N):
        b.prepare_torch()
        act, _ = pol.sample_actions_u8(b.valid_actions)
        b.step_torch(act)
    assert a._data.tobytes() == b._data.tobytes()
    assert a.digest() == b.digest()


# ---- observation encode ---------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 64, 100, 4097])
def test_onehot_vs_reference_expression(amd, oracle, torch_cuda, M):
    torch = torch_cuda
    import torch.nn.functional as F

    env = amd.VecGame(M)
    env.reset(1)
    env.rollout(40, masked=True)
    out, long = env.onehot(with_long=True)
    board = env.board.cpu()
    # the reference's own expression, src/ml2048/policy/_network.py:91-95,105
    x = F.one_hot(board.long(), 16).float()
    x = torch.permute(x, (0, 2, 1))
    ref = torch.reshape(x, (-1, 16, 4, 4))
    assert torch.equal(out.cpu(), ref)
    assert long.dtype == torch.int64 and torch.equal(long.cpu(), board.long())
    o2, bad = oracle.encode_onehot(board.numpy())
    assert bad == 0 and np.array_equal(out.cpu().numpy(), o2)
    env.onehot(check=True)  # no un-encodable tile in random play


def test_histogram_and_summary(amd, oracle):
    env = amd.VecGame(5000)
    env.reset(8)
    env.rollout(150, masked=False)
    board = env.board.cpu().numpy()
    term = env.terminated.cpu().numpy()
    assert np.array_equal(env.histogram().cpu().numpy(), oracle.hist_maxcell(board))
    assert np.array_equal(env.histogram(terminated_only=True).cpu().numpy(), oracle.hist_maxcell(board, term))
    mx, cnt = np.unique(board.max(axis=1), return_counts=True)
    exp = sorted(((2 ** int(m), int(c), int(c) / 5000) for m, c in zip(mx, cnt)), key=lambda s: -s[0])
    assert env.summary() == exp  # game_numba.py:593-604


# ---- properties at full size (no oracle needed) ------------------------------------------------------------
def test_c3_properties_at_full_size(amd, torch_cuda):
    """Config C3 size (M=262144, masked sampling + one-hot): invariants of the game rules."""
    torch = torch_cuda
    M = 262144
    env = amd.VecGame(M, amd.reward_fn_normal)
    env.reset(20241008)
    pol = amd.RandomPolicy(seed=1, masked=True)
    score_acc = torch.zeros(M, dtype=torch.float64, device="cuda")
    obs = torch.empty((M, 16, 4, 4), dtype=torch.float32, device="cuda")
    for s in range(48):
        idx, n = env.prepare_torch()
        k = int(n.item())
        score_acc[idx[:k]] = 0
        a, logp = pol.sample_actions_u8(env.valid_actions)
        before = env.board.clone()
        res = env.step_torch(a)
        # masked sampling never picks an invalid move unless the game is over (none is, post-prepare)
        assert int(res["invalid"].sum().item()) == 0
        assert torch.equal(res["prev_state"], before)
        # every board changed, and gained exactly (1 - merges) tiles
        nz_before = (before != 0).sum(dim=1)
        nz_after = (res["state"] != 0).sum(dim=1)
        merges = res["merged"].sum(dim=1)
        assert torch.equal(nz_after, nz_before - merges + 1)
        assert bool((res["state"] != before).any(dim=1).all())
        # reward_fn_normal == sum merged[k] * 2^(k+1); score is its running sum since the deal
        w = (2 ** (torch.arange(16, device="cuda") + 1)).double()
        r = (res["merged"].double() * w).sum(dim=1)
        assert torch.equal(res["reward"].double(), r)
        score_acc += r
        assert torch.equal(res["score"].double(), score_acc)
        # terminated <=> no valid action left
        assert torch.equal(res["terminated"] != 0, res["valid_actions"].sum(dim=1) == 0)
        env.onehot(out=obs)
        assert float(obs.sum().item()) == 16.0 * M  # exactly one class per cell
        assert torch.equal(obs.argmax(dim=1).reshape(M, 16).to(torch.uint8), env.board)
    assert torch.all(env.step_count <= 48)


def test_valid_mask_iff_move_changes_board(amd, torch_cuda):
    """valid_actions[d] <=> moving in direction d changes the board."""
    torch = torch_cuda
    M = 65536
    env = amd.VecGame(M)
    env.reset(3)
    env.rollout(60, masked=True)
    env.prepare_torch()
    mask = env.valid_actions.clone()
    snap = env._data.copy()
    for d in range(4):
        env.load_records(snap)
        before = env.board.clone()
        res = env.step_torch(torch.full((M,), d, dtype=torch.uint8, device="cuda"))
        changed = (res["state"] != before).any(dim=1)
        assert torch.equal(changed, mask[:, d] != 0)
        assert torch.equal(res["invalid"] != 0, mask[:, d] == 0)


# ---- API behaviour ----------------------------------------------------------------------------------------
def test_api_errors_and_dtypes(amd, torch_cuda):
    torch = torch_cuda
    with pytest.raises(ValueError):
        amd.VecGame(-1)
    with pytest.raises(TypeError):
        amd.VecGame(8, lambda s, p, m: 0.0)
    env = amd.VecGame(8)
    env.reset(0)
    env.prepare()
    with pytest.raises(AssertionError):
        env.step(np.zeros(7, np.int64))  # game_numba.py:668
    for dt in (np.uint8, np.int32, np.int64, np.int16):
        res = env.step(np.zeros(8, dt))
    assert res["state"].dtype == np.uint8 and res["state"].shape == (8, 16)
    assert res["valid_actions"].shape == (8, 4) and res["merged"].shape == (8, 16)
    assert res["terminated"].dtype == np.uint8 and res["invalid"].dtype == np.uint8
    assert res["score"].dtype == np.float32 and res["prev_state"].shape == (8, 16)
    # out-of-range actions are invalid moves, not undefined behaviour (H11)
    res = env.step(np.array([4, -1, 255, 0, 1, 2, 3, 100], np.int64))
    assert res["invalid"][[0, 1, 2, 7]].tolist() == [1, 1, 1, 1]
    b, v = env.observations()
    assert b.shape == (8, 16) and v.shape == (8, 4) and b.dtype == np.uint8
    assert env._data.dtype.itemsize == 64 and env._data["id"].tolist() == list(range(8))


def test_game_count_survives_reset_h7(amd):
    env = amd.VecGame(16)
    env.reset(1)
    env.prepare()
    assert env._game_count == 16
    env.reset(1)  # game_numba.py:582 vs :606 - reset() does not clear _game_count
    (idx,) = env.prepare()
    assert idx.tolist() == list(range(16)) and env._game_count == 32
    assert env._data["id"].tolist() == list(range(16, 32))


def test_runner_events_and_stats(amd, oracle, torch_cuda):
    M = 2048
    env = amd.VecGame(M)
    env.reset(2)
    runner = amd.VecRunner(env, 0)
    stats = amd.RunnerStats()
    seen = {"prepared": 0, "stepped": 0, "new": 0}
    expect = np.zeros(20, np.int32)

    def on_prepared(e, idx):
        seen["prepared"] += 1
        seen["new"] += idx.size

    def on_stepped(e, result, actions, logp):
        seen["stepped"] += 1
        assert set(result) == {"state", "valid_actions", "merged", "step", "reward", "score", "terminated",
                               "invalid", "prev_state", "prev_valid_actions"}
        expect[:] += oracle.hist_maxcell(result["state"].cpu().numpy(), result["terminated"].cpu().numpy())

    runner.add_callback("prepared", on_prepared)
    runner.add_callback("stepped", on_stepped)
    runner.add_callback("stepped", stats.on_stepped)
    runner.step_many(amd.RandomPolicy(seed=3, masked=False), 200)
    assert seen["prepared"] == 200 and seen["stepped"] == 200
    assert seen["new"] == env._game_count
    stats.sync()
    assert np.array_equal(stats.counts, expect) and stats.terminated_count == expect.sum() > 0
    assert amd.RunnerStats.combine([stats, stats]).terminated_count == 2 * stats.terminated_count
