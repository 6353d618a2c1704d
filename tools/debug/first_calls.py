"""The first rollout() calls of a fresh process (the driver's 5 warm-up + 20 timed launches): where does each call's host
time go?  Prints every call: interval, native launch call, torch.empty, producer wait, GPU wait."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import ml2048_amd as amd
from ml2048_amd import game as G

os.environ.setdefault("ML2048_HOST_THREADS", "8")
M, N = 65536, 128
env = amd.VecGame(M, schedule="host-async")
env.reset(20241008)
L = env._L
acc = {"take": 0.0, "sync": 0.0, "launch": 0.0, "alloc": 0.0}
real_take, real_sync, real_launch, real_empty = L.ml2048_sched_stream_take_slot, G._Event.synchronize, L.ml2048_rollout_chain_ids, torch.empty


def wrap(key, fn):
    def f(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[key] += time.perf_counter() - t
        return r
    return f


L.ml2048_sched_stream_take_slot = wrap("take", real_take)
L.ml2048_rollout_chain_ids = wrap("launch", real_launch)
torch.empty = wrap("alloc", real_empty)
G._Event.synchronize = lambda self: wrap("sync", real_sync)(self)
gc.collect()
gc.freeze()
counter = 0
rows = []
t_prev = time.perf_counter()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
    if k == 5:
        torch.cuda.synchronize()
        t_prev = time.perf_counter()
    for key in acc:
        acc[key] = 0.0
    env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids="deferred")
    counter += N
    t = time.perf_counter()
    rows.append((k, (t - t_prev) * 1e6, acc["launch"] * 1e6, acc["alloc"] * 1e6, acc["take"] * 1e6, acc["sync"] * 1e6))
    t_prev = t
torch.cuda.synchronize()
tail = (time.perf_counter() - t_prev) * 1e6
print("call | interval us | native launch | torch.empty | producer wait | GPU wait")
for r in rows:
    print("%4d | %9.1f | %9.1f | %9.1f | %9.1f | %9.1f" % r)
print("final synchronize %.1f us; timed region (calls 5..) %.1f us" % (tail, sum(r[1] for r in rows[5:]) + tail))
