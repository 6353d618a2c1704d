"""Host-side enqueue time of rollout() per schedule mode vs the GPU time (is the host ahead of the GPU?)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd

for mode in ("device", "host-async"):
    env = amd.VecGame(65536, "normal", schedule=mode)
    env.reset(1)
    for _ in range(30):
        env.rollout(128, masked=False, act_seed=1, finish_ids="deferred")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        env.rollout(128, masked=False, act_seed=1, finish_ids="deferred")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: host enqueue {1e6*(t1-t0)/200:.1f} us/rollout, total {1e6*(t2-t0)/200:.1f} us/rollout")
    env.close() if hasattr(env, "close") else None
