"""Host-side cost breakdown of VecGame.rollout at C2 (run on the GPU box).

    python tools/profile_host.py [M] [schedule-mode]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ml2048_amd as amd

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mode = sys.argv[2] if len(sys.argv) > 2 else "host-async"
env = amd.VecGame(M, schedule=mode)
env.reset(1)
for _ in range(20):
    env.rollout(128, masked=False)
torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter()
    for _ in range(100):
        env.rollout(128, masked=False)
    t_host = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print("%s: host enqueue %.3f ms/rollout, end-to-end %.3f ms/rollout" % (mode, t_host / 100 * 1e3, t_all / 100 * 1e3), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    env.rollout(128, masked=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
