import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd
env = amd.VecGame(65536, schedule=sys.argv[1] if len(sys.argv) > 1 else "host-async")
env.reset(1)
torch.cuda.synchronize()
out = []
for blk in range(20):
    t = time.perf_counter()
    for _ in range(20):
        env.rollout(128, masked=False)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t) / 20 * 1e3)
print(" ".join("%.3f" % x for x in out))
