"""Timeline of the round driver's short run (W=5 warm-up rollouts, barrier, K=20 timed rollouts at C2): host time of
every rollout() call and the GPU's gaps between the kernels, to see what the 2 ms timed region spends outside kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import ml2048_amd as amd

os.environ.setdefault("ML2048_HOST_THREADS", "8")
M, N, K, W = 65536, 128, 20, 5
env = amd.VecGame(M, schedule="host-async")
env.reset(20241008)
counter = 0
for rep in range(3):
    for _ in range(W):
        env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids="deferred")
        counter += N
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    ref = torch.cuda.Event(enable_timing=True)
    host = []
    t0 = time.perf_counter()
    ref.record()
    for k in range(K):
        env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids="deferred", kernel_events=evs[k])
        counter += N
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    starts = [ref.elapsed_time(a) for a, _ in evs]
    ends = [ref.elapsed_time(b) for _, b in evs]
    print(f"rep {rep}: total {total:.3f} ms for {K} rollouts = {total / K * 1e3:.1f} us each")
    print("  host return (ms):", " ".join(f"{h:.2f}" for h in host))
    print("  kernel start    :", " ".join(f"{s:.2f}" for s in starts))
    print("  kernel dur (us) :", " ".join(f"{(e - s) * 1e3:.0f}" for s, e in zip(starts, ends)))
    print("  gap before (us) :", " ".join(f"{(starts[i] - (ends[i - 1] if i else 0)) * 1e3:.0f}" for i in range(K)))
    print(f"  last kernel end {ends[-1]:.3f} ms; tail to sync {total - ends[-1]:.3f} ms")
