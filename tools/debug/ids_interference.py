"""Does the deferred id scan (two small kernels on a side stream, beside the next rollout's kernel) slow the rollout
kernel down?  Kernel time by the dispatch-attached timer with finish_ids = "deferred" / True / False."""
import ctypes as C
import gc
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import ml2048_amd as amd
from ml2048_amd import _lib

os.environ.setdefault("ML2048_HOST_THREADS", "9")
M, N, K = 65536, 128, 600
L = _lib.load()
for mode in ("deferred", False, True, "deferred", False):
    env = amd.VecGame(M, schedule="host-async")
    env.reset(20241008)
    counter = 0
    for _ in range(50):
        env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids=mode)
        counter += N
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()
    timers = []
    t0 = time.perf_counter()
    for k in range(K):
        tm = None
        if k % 19 == 9:
            tm = C.c_void_p()
            _lib.check(L.ml2048_timer_create(C.byref(tm)), "timer_create")
            timers.append(tm)
        env.rollout(N, masked=False, act_seed=7, act_counter=counter, finish_ids=mode, kernel_timer=tm)
        counter += N
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vals = []
    for tm in timers:
        ms = C.c_float(0.0)
        _lib.check(L.ml2048_timer_elapsed_ms(tm, C.byref(ms)), "timer_elapsed_ms")
        vals.append(ms.value)
        L.ml2048_timer_destroy(tm)
    print(f"finish_ids={mode!s:9}: {dt / K * 1e6:6.1f} us per rollout, kernel {sum(vals) / len(vals) * 1e3:6.2f} us")
    del env
