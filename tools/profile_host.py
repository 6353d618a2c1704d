"""Host-side cost breakdown of VecGame.rollout (run on the GPU box)."""
import cProfile
import pstats
import sys
import time

import torch

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ml2048_amd as amd

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = amd.VecGame(M)
env.reset(1)
for _ in range(5):
    env.rollout(128, prefetch_next=True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50):
    env.rollout(128, prefetch_next=True)
torch.cuda.synchronize()
print("prefetch: %.3f ms/rollout" % ((time.perf_counter() - t) / 50 * 1e3))
t = time.perf_counter()
for _ in range(50):
    env.rollout(128, prefetch_next=False)
torch.cuda.synchronize()
print("no prefetch: %.3f ms/rollout" % ((time.perf_counter() - t) / 50 * 1e3))
r = env._rng
t = time.perf_counter()
for _ in range(50):
    r.draw_schedule(128, 0.8)
print("draw_schedule alone: %.3f ms" % ((time.perf_counter() - t) / 50 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    env.rollout(128, prefetch_next=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
