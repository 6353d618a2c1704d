"""us per env step of a RandomPolicy-driven per-step loop: the 3-launch per-step API vs the one-launch
ml2048_step_prepare (fused sampler; optionally the fused one-hot observation).
    python tools/bench_step_prepare.py [M] [steps] [obs 0/1] [masked 0/1]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import ml2048_amd as amd

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
OBS = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
MASKED = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False


def timed(fn, n):
    fn(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, t_host / n * 1e6


obs = torch.empty((M, 16, 4, 4), dtype=torch.float32, device="cuda") if OBS else None

env = amd.VecGame(M)
env.reset(1)
pol = amd.RandomPolicy(seed=7, masked=MASKED)


def per_step(n):
    for _ in range(n):
        env.prepare_torch()
        a, _ = pol.sample_actions_u8(env.valid_actions)
        env.step_torch(a)
        if obs is not None:
            env.onehot(out=obs)


env2 = amd.VecGame(M)
env2.reset(1)
env2.prepare_torch()
pol2 = amd.RandomPolicy(seed=7, masked=MASKED)
state = {"a": pol2.sample_actions_u8(env2.valid_actions)[0]}


def fused(n):
    a = state["a"]
    for _ in range(n):
        a = env2.step_prepare_torch(a, sample=pol2, onehot=obs)["actions"]
    state["a"] = a


us1, h1 = timed(per_step, N)
us2, h2 = timed(fused, N)
per = 1128 if OBS else 99
print(f"M={M} obs={int(OBS)} masked={int(MASKED)}: per-step API {us1:.1f} us/env step (host {h1:.1f}); "
      f"step_prepare {us2:.1f} us/env step (host {h2:.1f}) = {M / us2 * 1e-3:.2f}e9 board-steps/s, "
      f"{per * M / us2 * 1e-6:.2f} TB/s of the {per} B/board-step")
