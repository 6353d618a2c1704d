import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd
from ml2048_amd import game
env = amd.VecGame(65536, schedule="host-async")
env.reset(1)
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, g)
for n in ("_next_schedule_async", "_stage_slot", "_stage_upload"):
    wrap(env, n)
torch.cuda.synchronize()
rows = []
for i in range(400):
    acc.clear()
    t = time.perf_counter()
    env.rollout(128, masked=False)
    dt = time.perf_counter() - t
    rows.append((i, dt * 1e3, dict(acc)))
torch.cuda.synchronize()
for i, dt, a in rows:
    if dt > 0.6 or i < 6:
        print(i, "%.3f ms" % dt, {k: "%.3f" % (v * 1e3) for k, v in a.items()})
