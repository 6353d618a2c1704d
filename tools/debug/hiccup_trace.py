"""Where do the occasional slow stretches of the pipelined C2 rollout come from?  Times, per rollout() call, the wait for
the producer (ml2048_sched_stream_take_slot) and the wait for the GPU to hand a slot back (event synchronize), and
lists the slowest calls.    python tools/debug/hiccup_trace.py [rollouts]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import ml2048_amd as amd
from ml2048_amd import game as G

os.environ.setdefault("ML2048_HOST_THREADS", "8")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
M, N = 65536, 128
env = amd.VecGame(M, schedule="host-async")
env.reset(20241008)
L = env._L
acc = {"take": 0.0, "sync": 0.0, "launch": 0.0, "alloc": 0.0}
real_launch = L.ml2048_rollout_chain_ids
real_empty = torch.empty


def launch(*a):
    t = time.perf_counter()
    r = real_launch(*a)
    acc["launch"] += time.perf_counter() - t
    return r


def empty(*a, **k):
    t = time.perf_counter()
    r = real_empty(*a, **k)
    acc["alloc"] += time.perf_counter() - t
    return r


L.ml2048_rollout_chain_ids = launch
torch.empty = empty
real_take = L.ml2048_sched_stream_take_slot
real_sync = G._Event.synchronize


def take(*a):
    t = time.perf_counter()
    r = real_take(*a)
    acc["take"] += time.perf_counter() - t
    return r


def sync(self):
    t = time.perf_counter()
    real_sync(self)
    acc["sync"] += time.perf_counter() - t


L.ml2048_sched_stream_take_slot = take
G._Event.synchronize = sync
counter = 0
for _ in range(50):
    env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids="deferred")
    counter += N
torch.cuda.synchronize()
if os.environ.get("HICCUP_GC", "1") == "0":
    import gc
    gc.disable()
    print("cyclic GC disabled")
elif os.environ.get("HICCUP_GC", "1") == "freeze":
    import gc
    gc.collect()
    gc.freeze()
    print("gc.collect() + gc.freeze() before the loop")
rows = []
t_prev = time.perf_counter()
t_begin = t_prev
for k in range(K):
    acc["take"] = acc["sync"] = acc["launch"] = acc["alloc"] = 0.0
    env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids="deferred")
    counter += N
    t = time.perf_counter()
    rows.append((t - t_prev, acc["take"], acc["sync"], k, acc["launch"], acc["alloc"]))
    t_prev = t
torch.cuda.synchronize()
total = time.perf_counter() - t_begin
print(f"{K} rollouts in {total * 1e3:.1f} ms = {total / K * 1e6:.1f} us each")
d = sorted(rows, reverse=True)[:12]
print("slowest calls: interval us | waiting for the producer | waiting for the GPU | rest (Python + launch) | index")
for dt, tk, sy, k, la, al in d:
    print(f"  {dt * 1e6:9.1f} | {tk * 1e6:9.1f} | {sy * 1e6:9.1f} | {(dt - tk - sy) * 1e6:9.1f} | {k}   (native launch call {la * 1e6:.1f}, torch.empty {al * 1e6:.1f})")
import statistics
print("median interval %.1f us; sum over calls: producer wait %.1f ms, GPU wait %.1f ms, rest %.1f ms" % (
    statistics.median(r[0] for r in rows) * 1e6, sum(r[1] for r in rows) * 1e3, sum(r[2] for r in rows) * 1e3,
    sum(r[0] - r[1] - r[2] for r in rows) * 1e3))
big = [r for r in rows if r[0] > 300e-6]
print(f"{len(big)} calls longer than 300 us, {sum(r[0] for r in big) * 1e3:.1f} ms in all")
