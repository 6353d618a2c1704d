"""Host-side cost breakdown of the per-step API (prepare_torch / sample / step_torch) at M = 65536."""
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
pol = amd.RandomPolicy(seed=1, masked=False)


def steps(n):
    for _ in range(n):
        env.prepare_torch()
        a, _ = pol.sample_actions_u8(env.valid_actions)
        env.step_torch(a)


steps(200)
torch.cuda.synchronize()
t = time.perf_counter()
steps(1000)
t_host = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print("host enqueue %.1f us/env step, end-to-end %.1f us/env step" % (t_host / 1000 * 1e6, t_all / 1000 * 1e6))
pr = cProfile.Profile()
pr.enable()
steps(1000)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
