"""Pure host cost of VecGame.rollout(): tiny kernels (M=4096, N=16) so that the GPU is never the limit."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd

for mode in ("host-async", "device"):
    env = amd.VecGame(4096, "normal", schedule=mode)
    env.reset(1)
    for _ in range(50):
        env.rollout(16, masked=False, act_seed=1, finish_ids="deferred")
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(500):
            env.rollout(16, masked=False, act_seed=1, finish_ids="deferred")
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{mode}: host {1e6*(t1-t0)/500:.1f} us/rollout, end to end {1e6*(t2-t0)/500:.1f} us/rollout", flush=True)
