import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd
env = amd.VecGame(65536, schedule="host-async")
env.reset(1)
torch.cuda.synchronize()
ts = []
for i in range(80):
    t = time.perf_counter()
    env.rollout(128, masked=False, finish_ids="deferred")
    ts.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize()
print("host ms per rollout call:", " ".join("%.2f" % x for x in ts))
