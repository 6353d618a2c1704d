"""Host-side cost breakdown of the per-step API (run on the GPU box)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ml2048_amd as amd

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = amd.VecGame(M)
env.reset(1)
pol = amd.RandomPolicy(seed=7, masked=False)


def loop(n):
    for _ in range(n):
        env.prepare_torch()
        a, _ = pol.sample_actions_u8(env.valid_actions)
        env.step_torch(a)


loop(50)
torch.cuda.synchronize()
t = time.perf_counter()
loop(500)
torch.cuda.synchronize()
print("per env step: %.1f us" % ((time.perf_counter() - t) / 500 * 1e6))
pr = cProfile.Profile()
pr.enable()
loop(500)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
