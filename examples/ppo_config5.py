#!/usr/bin/env python3
"""BASELINE.json config 5: a full PPO loop (the counterpart of the reference's run_train3.py)
driving the HIP VecGame with a PyTorch-ROCm CNN actor-critic: M=2048 games, N=64 steps per epoch,
use=2 (the last two rollouts are trained on), reward_fn_improved.

Everything between the environment and the network stays on the GPU: observations are the HIP
one-hot (M,16,4,4) tensor, actions go back as a device tensor, transitions are written into the
[use, step, game] buffers by one HIP kernel (ml2048_record_transition) and the advantage scan is one
HIP kernel (ml2048_gae).  The dense network math (conv / linear / Adam) is stock PyTorch-ROCm, as
the task's north star prescribes ("the existing CNN").

The network and the losses restate the reference's architecture and formulas
(src/ml2048/policy/_network.py, src/ml2048/policy/actor_critic.py:113-226, :325-443) in this
file's own words; hyper-parameters are run_train3.py:62-78.

    python examples/ppo_config5.py --epochs 30
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch
from torch import nn
from torch.nn import functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ml2048_amd as amd  # noqa: E402
from ml2048_amd import train_ops  # noqa: E402

CLASSES, CELLS, ACTIONS = 16, 16, 4


class Encoder(nn.Module):
    """Three depthwise views of the one-hot board (whole board, rows, columns), each followed by a
    pointwise mix, concatenated to 12 positions and folded by one width-12 convolution
    (_network.py:12-133).  Input: the (N,16,4,4) float one-hot the HIP kernel emits."""

    def __init__(self, features: int = 1024, mult: int = 16):
        super().__init__()
        ch = features // 16
        self.ch, self.features = ch, features
        self.dw_full = nn.Conv1d(CLASSES, CLASSES * mult, CELLS, groups=CLASSES)
        self.pw_full = nn.Conv1d(CLASSES * mult, ch * 4, 1)
        self.dw_rows = nn.Conv2d(CLASSES, CLASSES * mult, (1, 4), groups=CLASSES)
        self.pw_rows = nn.Conv2d(CLASSES * mult, ch, 1)
        self.dw_cols = nn.Conv2d(CLASSES, CLASSES * mult, (4, 1), groups=CLASSES)
        self.pw_cols = nn.Conv2d(CLASSES * mult, ch, 1)
        self.fold = nn.Conv1d(ch, features, 12)
        for m in (self.dw_full, self.dw_rows, self.dw_cols, self.fold):
            nn.init.zeros_(m.bias)

    def forward(self, onehot: torch.Tensor) -> torch.Tensor:
        # Every convolution here spans its whole input extent (or is 1x1), so each is a batched
        # matrix product; evaluating them as einsum / linear on the modules' own parameters keeps the
        # reference's parameterisation while running on hipBLASLt GEMMs (MIOpen's grouped conv1d
        # fails to launch for N >= 16384 on this stack).
        n, m = onehot.shape[0], self.dw_full.out_channels // CLASSES
        flat = onehot.reshape(n, CLASSES, CELLS)
        a = torch.einsum("ncl,cml->ncm", flat, self.dw_full.weight.view(CLASSES, m, CELLS)).reshape(n, -1)
        a = F.leaky_relu(a + self.dw_full.bias)
        a = F.leaky_relu(F.linear(a, self.pw_full.weight.flatten(1), self.pw_full.bias)).reshape(n, self.ch, 4)
        b = torch.einsum("ncrw,cmw->ncmr", onehot, self.dw_rows.weight.view(CLASSES, m, 4)).reshape(n, -1, 4)
        b = F.leaky_relu(b + self.dw_rows.bias[:, None])
        b = F.leaky_relu(torch.einsum("nkr,ok->nor", b, self.pw_rows.weight.flatten(1)) + self.pw_rows.bias[:, None])
        c = torch.einsum("ncrw,cmr->ncmw", onehot, self.dw_cols.weight.view(CLASSES, m, 4)).reshape(n, -1, 4)
        c = F.leaky_relu(c + self.dw_cols.bias[:, None])
        c = F.leaky_relu(torch.einsum("nkw,ok->now", c, self.pw_cols.weight.flatten(1)) + self.pw_cols.bias[:, None])
        x = torch.cat((a, b, c), dim=2).flatten(1)
        return F.leaky_relu(F.linear(x, self.fold.weight.flatten(1), self.fold.bias))

    @torch.no_grad()
    def forward_boards(self, board: torch.Tensor) -> torch.Tensor:
        """Inference straight from (N,16) boards: the three depthwise convolutions + leaky_relu come from ONE
        HIP gather kernel (ml2048_encoder_front; the one-hot is never built), the dense part
        stays on hipBLASLt.  Same function as forward(onehot) up to float rounding."""
        n = board.shape[0]
        a, b, c = train_ops.encoder_front(board, self.dw_full, self.dw_rows, self.dw_cols)
        a = F.leaky_relu(F.linear(a, self.pw_full.weight.flatten(1), self.pw_full.bias)).reshape(n, self.ch, 4)
        b = F.leaky_relu(torch.einsum("nkr,ok->nor", b, self.pw_rows.weight.flatten(1)) + self.pw_rows.bias[:, None])
        c = F.leaky_relu(torch.einsum("nkw,ok->now", c, self.pw_cols.weight.flatten(1)) + self.pw_cols.bias[:, None])
        x = torch.cat((a, b, c), dim=2).flatten(1)
        return F.leaky_relu(F.linear(x, self.fold.weight.flatten(1), self.fold.bias))


def _mlp(i, h1, h2, o, out_gain):
    layers = [nn.Linear(i, h1), nn.ReLU(), nn.Linear(h1, h2), nn.ReLU(), nn.Linear(h2, o)]
    for m, g in ((layers[0], math.sqrt(2)), (layers[2], math.sqrt(2)), (layers[4], out_gain)):
        nn.init.orthogonal_(m.weight, g)
        nn.init.zeros_(m.bias)
    return nn.Sequential(*layers)


def onehot_of(state_i8: torch.Tensor) -> torch.Tensor:
    """(N,16) int8 exponents -> (N,16,4,4) float one-hot (training batches come from the buffers)."""
    return F.one_hot(state_i8.long(), CLASSES).float().permute(0, 2, 1).reshape(-1, CLASSES, 4, 4)


class ActorCritic(nn.Module, amd.Policy):
    """Shared encoder + actor / critic heads (actor_critic.py:240-323)."""

    def __init__(self, features: int = 1024):
        super().__init__()
        self.encoder = Encoder(features)
        self.actor = _mlp(features, 256, 64, ACTIONS, 0.01)
        self.critic = _mlp(features, 256, 64, 1, 1.0)

    def logits(self, onehot):
        z = self.actor(self.encoder(onehot))
        return z - z.detach().max(dim=-1, keepdim=True).values

    def value(self, onehot):
        return self.critic(self.encoder(onehot)).squeeze(-1).float()

    @torch.no_grad()
    def value_of_states(self, state_i8, chunk=65536, fused=True):
        """Critic over a long list of (N,16) int8 states, in chunks."""
        if fused:
            out = [self.critic(self.encoder.forward_boards(state_i8[i:i + chunk])).squeeze(-1).float()
                   for i in range(0, state_i8.shape[0], chunk)]
        else:
            out = [self.value(onehot_of(state_i8[i:i + 16384])) for i in range(0, state_i8.shape[0], 16384)]
        return torch.cat(out)

    @torch.no_grad()
    def act(self, onehot, valid_bool, board=None):
        """Sample actions; with `board` ((N,16) uint8) the encoder runs on the fused front end."""
        if board is not None:
            z = self.actor(self.encoder.forward_boards(board))
            z = z - z.max(dim=-1, keepdim=True).values
        else:
            z = self.logits(onehot)
        z = torch.where(valid_bool, z, torch.finfo(z.dtype).min)
        dist = torch.distributions.Categorical(logits=z)
        a = torch.multinomial(dist.probs, 1, True).squeeze(-1)
        return a, dist.log_prob(a)


def masked_entropy(logits, mask):
    lo = torch.finfo(logits.dtype).min
    dist = torch.distributions.Categorical(logits=torch.where(mask, logits, lo))
    plogp = dist.probs * torch.clamp(dist.logits, min=lo)
    return -torch.where(mask, plogp, 0).sum(-1)


def ppo_losses(net, batch, p):
    """PPO-clip + step-weighted entropy bonus + one-step TD critic loss
    (actor_critic.py:113-226)."""
    oh, mask = onehot_of(batch["state"]), batch["valid_actions"]
    logits = net.logits(oh)
    dist = torch.distributions.Categorical(logits=logits + torch.where(mask, 0, -10.0e5))
    logp = dist.log_prob(batch["action"].long())
    adv = batch["adv"]
    adv = adv / (torch.sqrt(torch.sum(adv * adv) / adv.numel()) * 3)
    adv = torch.tanh(adv) * torch.sqrt(adv.abs() + 0.6917418778812134)
    ratio = torch.exp(logp - batch["action_log_prob"])
    clipped = torch.clip(ratio, 1 - p["ppo_epsilon"], 1 + p["ppo_epsilon"])
    policy = torch.minimum(ratio * adv, clipped * adv)
    step = batch["step"].float()
    std, mean = torch.std_mean(step)
    weight = (torch.tanh((step - mean) / std * 2 - 1) + 1) * 0.4 + 0.2
    entropy = p["entropy_coef"] * weight * masked_entropy(logits, mask)
    v0 = net.value(oh)
    with torch.no_grad():
        v1 = net.value(onehot_of(batch["next_state"]))
    q0 = p["gamma"] * torch.where(batch["terminated"], 0, v1) + batch["reward"]
    critic = p["critic_coef"] * F.mse_loss(q0, v0)
    return -policy.sum(), -entropy.sum(), critic


def learn(net, data, p, seed=None):
    """One pass over the flattened [use*step*game] data in random minibatches with a fresh Adam
    (actor_critic.py:325-443; the reference re-creates the optimizer on every call, :361)."""
    lr = p["actor_lr"]
    opt = torch.optim.Adam([{"params": net.encoder.parameters(), "lr": lr},
                            {"params": net.actor.parameters(), "lr": lr},
                            {"params": net.critic.parameters(), "lr": lr}], lr=lr)
    total = data["state"].shape[0]
    g = None
    if seed is not None:
        g = torch.Generator(data["state"].device)
        g.manual_seed(seed)
    order = torch.randperm(total, device=data["state"].device, generator=g)
    bs = p["actor_batch_size"]
    out = []
    for start in range(0, total, bs):
        size = min(bs, total - start)
        if size < bs and size < bs // 2:  # replay.py:42-47
            break
        idx = order[start:start + size]
        batch = {k: v[idx] for k, v in data.items()}
        pl, el, cl = ppo_losses(net, batch, p)
        opt.zero_grad()
        (pl + el + cl).backward()
        opt.step()
        out.append(torch.stack((pl.detach(), el.detach(), cl.detach())))
    return torch.stack(out).mean(0).tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--games", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--use", type=int, default=2)
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--json", default=None)
    ap.add_argument("--fused-step", type=int, default=1,
                    help="one env launch per step (ml2048_step_prepare: step + transition record into the replay "
                         "slices + re-deal + terminated-games histogram) instead of prepare / step / record / histogram")
    ap.add_argument("--fused-front", type=int, default=1,
                    help="no-grad encoder passes (rollout policy, GAE critic passes) through ml2048_encoder_front")
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    dev = torch.device("cuda:0")
    lr_factor = 1 / 2**4
    p = {"gamma": 0.997, "lambda": 0.9, "ppo_epsilon": 0.1, "actor_lr": 4.0e-4 * lr_factor,
         "critic_lr": 1.0e-3 * lr_factor, "actor_batch_size": 1024, "entropy_coef": 0.00025,
         "critic_coef": 1e-5 / 2**10}  # run_train3.py:62-78
    M, S, U = args.games, args.steps, args.use
    env = amd.VecGame(M, amd.reward_fn_improved, device=dev)
    env.reset(args.seed)
    net = ActorCritic().to(dev)
    net_eval = ActorCritic().to(dev)
    buffers = train_ops.new_buffers((U, S, M), dev)
    stats = amd.RunnerStats()
    # ml2048_step_out pointed at the [game] slices of the replay buffers, one dict per (use, step)
    step_outs = [[{"prev_state": buffers["state"][u, s], "prev_valid_actions": buffers["valid_actions"][u, s],
                   "reward": buffers["reward"][u, s], "state": buffers["next_state"][u, s],
                   "valid_actions": buffers["next_valid_actions"][u, s], "step": buffers["step"][u, s],
                   "terminated": buffers["terminated"][u, s]} for s in range(S)] for u in range(U)]
    obs = torch.empty((M, 16, 4, 4), dtype=torch.float32, device=dev)
    log = []
    env_time = net_time = train_time = 0.0
    import gc
    gc.collect()
    gc.freeze()  # the set-up above lives as long as the run: keep it out of the cyclic collector's sight (README.md)
    for epoch in range(args.epochs):
        net_eval.load_state_dict(net.state_dict())
        stats.reset()
        ui = epoch % U
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_env = 0.0
        if args.fused_step:
            # VecRunner.step_once (runner.py:74-109) with ONE env launch per step: the policy acts on the live
            # (post-prepare) state; ml2048_step_prepare steps, writes the transition straight into this step's replay
            # slices (Trainer.on_stepped's nine copies, run_train3.py:125-155), adds the finished games to the
            # histogram and re-deals them - the state the next policy call observes
            te = time.perf_counter()
            if not env._prepared:
                env.prepare_torch()
                if not args.fused_front:
                    env.onehot(out=obs)
            t_env += time.perf_counter() - te
            for si in range(S):
                a, lp = net_eval.act(obs, env.valid_actions.bool(), board=env.board if args.fused_front else None)
                te = time.perf_counter()
                a8 = a.to(torch.uint8)
                buffers["action"][ui, si].copy_(a8.view(torch.int8))
                buffers["action_log_prob"][ui, si].copy_(lp)
                env.step_prepare_torch(a8, out=step_outs[ui][si], hist=stats.device_counts(env),
                                       onehot=None if args.fused_front else obs)
                t_env += time.perf_counter() - te
        else:
            for si in range(S):  # the reference-shaped loop: prepare / step / record / histogram launches
                te = time.perf_counter()
                env.prepare_torch()
                if not args.fused_front:
                    env.onehot(out=obs)
                t_env += time.perf_counter() - te
                a, lp = net_eval.act(obs, env.valid_actions.bool(), board=env.board if args.fused_front else None)
                te = time.perf_counter()
                res = env.step_torch(a.to(torch.uint8))
                train_ops.record_transition(env, a, lp, buffers, ui, si)
                stats.on_stepped(env, res, a, lp)
                t_env += time.perf_counter() - te
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_use = min(epoch + 1, U)
        view = {k: v[:n_use] for k, v in buffers.items()}
        # gae.py:39-48: two critic passes, then the HIP scan
        v0 = net_eval.value_of_states(view["state"].reshape(-1, 16), fused=bool(args.fused_front)).reshape(n_use, S, M)
        v1 = net_eval.value_of_states(view["next_state"].reshape(-1, 16), fused=bool(args.fused_front)).reshape(n_use, S, M)
        train_ops.gae(v0, v1, view["reward"], view["terminated"], gamma=p["gamma"], lambda_=p["lambda"],
                      adv=view["adv"] if view["adv"].is_contiguous() else None)
        data = {k: v.flatten(0, 2) for k, v in view.items()}
        q = dict(p)
        q["actor_lr"] = p["actor_lr"] * 32 / math.sqrt(1024 + epoch)  # run_train3.py:163-164
        losses = learn(net, data, q, seed=epoch)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        env_time += t_env
        net_time += (t1 - t0) - t_env
        train_time += t2 - t1
        top = env.summary()[:3]
        entry = {"epoch": epoch, "rollout_s": t1 - t0, "train_s": t2 - t1, "env_host_s": t_env,
                 "env_steps_per_s_in_loop": M * S / (t1 - t0), "mean_score": float(env.score.mean().item()),
                 "top_tiles": [(int(a), int(b)) for a, b, _ in top], "terminated": stats.summary()[:3],
                 "losses": losses}
        log.append(entry)
        print(json.dumps(entry), flush=True)
    summary = {"config": f"C5: games={M} steps={S} use={U} reward=improved", "epochs": args.epochs,
               "fused_step": bool(args.fused_step), "fused_front": bool(args.fused_front),
               "env_steps_per_s_in_loop": M * S * args.epochs / (env_time + net_time),
               "env_host_time_share_of_rollout": env_time / (env_time + net_time),
               "epoch_s": (env_time + net_time + train_time) / args.epochs,
               "mean_score_first": log[0]["mean_score"], "mean_score_last": log[-1]["mean_score"]}
    print(json.dumps(summary), flush=True)
    if args.json:
        json.dump({"summary": summary, "log": log}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
