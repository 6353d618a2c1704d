"""cProfile of the host side of a step_prepare loop (M=65536, fused sampler)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import ml2048_amd as amd

M = 65536
env = amd.VecGame(M)
env.reset(1)
env.prepare_torch()
pol = amd.RandomPolicy(seed=7, masked=False)
a = pol.sample_actions_u8(env.valid_actions)[0]
for _ in range(300):
    a = env.step_prepare_torch(a, sample=pol)["actions"]
torch.cuda.synchronize()


def loop(n):
    global a
    for _ in range(n):
        a = env.step_prepare_torch(a, sample=pol)["actions"]


pr = cProfile.Profile()
pr.enable()
loop(5000)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
