"""Encoder front end: the fused gather kernel against the torch path it replaces
(one-hot from ml2048_encode_onehot + three depthwise contractions + bias + leaky_relu, evaluated as in
examples/ppo_config5.py), same weights, same boards.

    python tools/bench_encoder_front.py [N]          (run under rocprofv3 --kernel-trace --stats for per-kernel times)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ml2048_amd as amd
from ml2048_amd import train_ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda:0")
env = amd.VecGame(N)
env.reset(1)
env.rollout(40, masked=True, act_seed=1)
board = env.board
dw_full = torch.nn.Conv1d(16, 256, 16, groups=16).to(dev)
dw_rows = torch.nn.Conv2d(16, 256, (1, 4), groups=16).to(dev)
dw_cols = torch.nn.Conv2d(16, 256, (4, 1), groups=16).to(dev)


@torch.no_grad()
def torch_path():
    onehot = env.onehot()                                   # (N,16,4,4) float, HIP kernel (1 KiB/board)
    flat = onehot.reshape(N, 16, 16)
    a = torch.einsum("ncl,cml->ncm", flat, dw_full.weight.view(16, 16, 16)).reshape(N, -1)
    a = F.leaky_relu(a + dw_full.bias)
    b = torch.einsum("ncrw,cmw->ncmr", onehot, dw_rows.weight.view(16, 16, 4)).reshape(N, -1, 4)
    b = F.leaky_relu(b + dw_rows.bias[:, None])
    c = torch.einsum("ncrw,cmr->ncmw", onehot, dw_cols.weight.view(16, 16, 4)).reshape(N, -1, 4)
    c = F.leaky_relu(c + dw_cols.bias[:, None])
    return a, b, c


out = (torch.empty((N, 256), device=dev), torch.empty((N, 256, 4), device=dev), torch.empty((N, 256, 4), device=dev))


@torch.no_grad()
def fused_path():
    return train_ops.encoder_front(board, dw_full, dw_rows, dw_cols, out=out)


ta, fa = torch_path(), fused_path()
for x, y in zip(ta, fa):
    assert torch.allclose(x, y, rtol=1e-5, atol=1e-6)
for name, fn in (("torch path", torch_path), ("fused kernel", fused_path)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 20
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("N=%d %-13s %.3f ms per forward; 9 216 B/board of activations = %.2f TB/s" % (N, name, dt * 1e3, 9216 * N / dt / 1e12))
