"""How late does torch.cuda.synchronize() return after the GPU has finished a 20-rollout window?  The same window waited
for (A) by synchronize alone, (B) by polling an event recorded behind the window first.   python tools/debug/sync_wakeup.py"""
import gc
import os
import statistics as st
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import ml2048_amd as amd

os.environ.setdefault("ML2048_HOST_THREADS", "9")
M, N, K = 65536, 128, 20
env = amd.VecGame(M, amd.reward_fn_normal, device="cuda:0", schedule="host-async")
env.reset(1)
cnt = [0]


def window(k):
    for _ in range(k):
        env.rollout(N, masked=False, act_seed=7, act_counter=cnt[0], finish_ids="deferred")
        cnt[0] += N
    env.join_ids()


gc.collect(); gc.freeze()
t = time.perf_counter()
while time.perf_counter() - t < 0.05:
    window(10)
torch.cuda.synchronize()
res = {"sync": [], "poll": []}
for rep in range(24):
    mode = "sync" if rep % 2 == 0 else "poll"
    window(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    window(K)
    if mode == "poll":
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
    torch.cuda.synchronize()
    res[mode].append((time.perf_counter() - t0) * 1e6)
for m, v in res.items():
    print(m, "window us: median %.1f min %.1f max %.1f  -> per rollout %.2f" % (st.median(v), min(v), max(v), st.median(v) / K))
