"""k_episode_stats alone: time per launch and achieved HBM rate (17 B/board algorithmic: id 4 + age 4 R/W 8 +
seen_id 4 + terminated 1; board/score only for the <1 % of boards that terminate).

    python tools/bench_eval_stats.py [M]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ml2048_amd as amd

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
env = amd.VecGame(M)
env.reset(1)
pol = amd.RandomPolicy(seed=1, masked=False)
stats = amd.EpisodeStats(env, M)
for _ in range(40):  # some games in progress, some terminating
    env.prepare_torch()
    a, _ = pol.sample_actions_u8(env.valid_actions)
    env.step_torch(a)
    stats.on_stepped()
evs = []
for _ in range(30):
    env.prepare_torch()
    a, _ = pol.sample_actions_u8(env.valid_actions)
    env.step_torch(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    stats.on_stepped()
    e1.record()
    evs.append((e0, e1))
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
# NOTE: events recorded around one small launch also time two marker packets (~8 us); run this script under
# `rocprofv3 --kernel-trace --stats` for the kernel's own duration (profiles/r01_k_episode_stats_4m_kernel_stats.csv)
print("M=%d: k_episode_stats %.2f us/launch incl. event markers, %.2f TB/s of the 17 B/board (frac %.2f of 8 TB/s)" %
      (M, ms * 1e3, 17 * M / (ms * 1e-3) / 1e12, 17 * M / (ms * 1e-3) / 8e12))
