import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd
env = amd.VecGame(65536, schedule="host-async")
env.reset(1)
parts = []
def upload(slot, nbytes):
    dev = env.device
    if env._copy_stream is None:
        env._copy_stream = torch.cuda.Stream(device=dev)
    t0 = time.perf_counter()
    cur = torch.cuda.current_stream(dev)
    with torch.cuda.stream(env._copy_stream):
        t1 = time.perf_counter()
        env._copy_stream.wait_event(slot[3])
        t2 = time.perf_counter()
        slot[1][:nbytes].copy_(slot[0][:nbytes], non_blocking=True)
        t3 = time.perf_counter()
        slot[2].record(env._copy_stream)
        t4 = time.perf_counter()
    t5 = time.perf_counter()
    cur.wait_event(slot[2])
    t6 = time.perf_counter()
    parts.append([x * 1e3 for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)])
    return slot[1]
env._stage_upload = upload
torch.cuda.synchronize()
for i in range(400):
    env.rollout(128, masked=False)
torch.cuda.synchronize()
for i, p in enumerate(parts):
    if sum(p) > 0.5 and i > 3:
        print(i, ["%.3f" % x for x in p])
import statistics as st
print("median parts", ["%.4f" % st.median(p[k] for p in parts) for k in range(6)])
