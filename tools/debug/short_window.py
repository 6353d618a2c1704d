"""Host-side timeline of bench.py's short timed window (K=20 rollouts at C2) after a long warm-up: when does each
rollout() call return, how long do the calls take, what does the window cost beyond K kernels."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import gc
import torch
import ml2048_amd as amd

os.environ.setdefault("ML2048_HOST_THREADS", "9")
M, N, K = 65536, 128, 20
W = int(os.environ.get("SW_WARM", "260"))
PAUSE = float(os.environ.get("SW_PAUSE_MS", "0"))
env = amd.VecGame(M, schedule="host-async")
env.reset(20241008)
counter = 0
gc.collect(); gc.freeze()
for rep in range(4):
    for _ in range(W):
        env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids="deferred"); counter += N
    env.join_ids()
    torch.cuda.synchronize()
    if PAUSE: time.sleep(PAUSE * 1e-3)
    t0 = time.perf_counter()
    rets = []
    for k in range(K):
        env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids="deferred"); counter += N
        rets.append((time.perf_counter() - t0) * 1e6)
    env.join_ids()
    tj = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e6
    calls = [rets[0]] + [rets[i] - rets[i - 1] for i in range(1, K)]
    print(f"rep {rep}: total {total:.0f} us = {total / K:.1f} us per rollout; join issued at {tj:.0f} us")
    print("   call durations (us):", " ".join(f"{c:.0f}" for c in calls))
    print("   returns at     (us):", " ".join(f"{r:.0f}" for r in rets))
