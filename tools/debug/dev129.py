import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd
from ml2048_amd.hostrng import HostRng
for n in (128, 129, 130, 257):
    for pre in (0, 1):
        a = amd.VecGame(64, "normal", schedule="device")
        a.reset(519423726)
        ref = HostRng(519423726)
        if pre:
            a.prepare(); a.step(np.zeros(64, np.uint8))
            ref.draw_prepare(); ref.draw_step()
        perm, sched, mask, cap, _ = a._device_schedule(n)
        torch.cuda.synchronize()
        words = a._devrng.cpu().numpy().view(np.uint64)
        err = int(words[5] >> 32)
        ne = int(a._sched_pending_slot["ne"].item())
        s2, e2, m2 = ref.draw_schedule(n, 0.8)
        ok = err == 0 and ne == e2.shape[0] and np.array_equal(sched.cpu().numpy(), s2)
        print("n", n, "pre", pre, "err", err, "ne", ne, "host ne", e2.shape[0], "sched ok", ok, flush=True)
