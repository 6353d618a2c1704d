"""cProfile of the host side of the default bench loop (C2 fused rollouts, host-async schedule, deferred id scan)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("ML2048_HOST_THREADS", "8")
import torch

import ml2048_amd as amd

M, N = 65536, 128
env = amd.VecGame(M, schedule="host-async")
env.reset(1)
c = 0
for _ in range(100):
    env.rollout(N, masked=False, act_seed=7, act_counter=c, finish_ids="deferred")
    c += N
torch.cuda.synchronize()
# host time per call when the GPU is NOT the bottleneck: sync every call
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.rollout(N, masked=False, act_seed=7, act_counter=c, finish_ids="deferred")
    ts.append(time.perf_counter() - t0)
    c += N
ts.sort()
print("host time of one rollout() call, GPU idle: median %.1f us, p10 %.1f, p90 %.1f" % (ts[100] * 1e6, ts[20] * 1e6, ts[180] * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    env.rollout(N, masked=False, act_seed=7, act_counter=c, finish_ids="deferred")
    c += N
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
