"""Lane-per-board vs one-wavefront-per-board step kernel under the same conditions: same seeds,
same actions, alternating launches in one process, with a fused rollout on a third environment
between them so the GPU stays at its loaded clocks.  Run under rocprofv3 --kernel-trace --stats and
read k_step<...> vs k_step_wave.

    python tools/compare_step_mappings.py <boards> <steps>
"""
import sys

import torch

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ml2048_amd as amd

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lane = amd.VecGame(M, step_mapping="lane")
wave = amd.VecGame(M, step_mapping="wave")
heat = amd.VecGame(262144)
for e in (lane, wave, heat):
    e.reset(7)
pol = amd.RandomPolicy(seed=3, masked=False)
heat.rollout(128)
for t in range(T):
    lane.prepare_torch()
    wave.prepare_torch()
    a, _ = pol.sample_actions_u8(lane.valid_actions)
    heat.rollout(16, finish_ids=False)
    lane.step_torch(a)
    heat.rollout(16, finish_ids=False)
    wave.step_torch(a)
torch.cuda.synchronize()
assert lane.digest() == wave.digest(), "the two mappings must agree bit for bit"
print("ok", M, T, hex(lane.digest()))
