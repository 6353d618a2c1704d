import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd
from ml2048_amd.hostrng import HostRng
M = 4099
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
bg = amd.VecGame(65536, "normal", schedule="host-async")   # background load on the same GPU
bg.reset(1)
t0 = time.time()
rounds = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    a = amd.VecGame(M, "normal", schedule="device")
    b = amd.VecGame(M, "normal", schedule="host")
    seed = int(rng.integers(0, 2**31))
    a.reset(seed); b.reset(seed)
    ref = HostRng(seed)
    counter = 0
    hist = []
    for _ in range(int(rng.integers(5, 25))):
        n = int(rng.choice([1, 2, 7, 33, 64, 128, 200]))
        dig = rng.random() < 0.4
        fin = "deferred" if rng.random() < 0.5 else True
        masked = bool(rng.integers(0, 2))
        bg.rollout(128, masked=False, finish_ids="deferred")
        oa = a.rollout(n, masked=masked, act_seed=5, act_counter=counter, digests=dig, finish_ids=fin)
        slot = a._sched_pending_slot
        bg.rollout(128, masked=False, finish_ids="deferred")
        ob = b.rollout(n, masked=masked, act_seed=5, act_counter=counter, digests=dig, finish_ids=fin)
        s2, e2, m2 = ref.draw_schedule(n, 0.8)
        counter += n
        hist.append((n, dig, fin, masked))
        bad = False
        if dig:
            da, db = oa["digests"].cpu().numpy(), ob["digests"].cpu().numpy()
            if not np.array_equal(da, db):
                print("DIGEST MISMATCH first step", int(np.flatnonzero(da != db)[0]), "of", n, hist[-1]); bad = True
        if a.digest() != b.digest():
            print("STATE MISMATCH", hist[-1]); bad = True
        if bad:
            torch.cuda.synchronize()
            ne = int(slot["ne"].item())
            sched = slot["sched"].cpu().numpy()
            print("ne", ne, "host ne", e2.shape[0], "sched equal", np.array_equal(sched, s2), "masks equal",
                  np.array_equal(slot["mask"][:ne].cpu().numpy().view(np.uint32), m2))
            if not dig:
                pass
            else:
                perm = slot["perm"].view(-1)[:(n + 1) * 1024 * 16].view(n + 1, 1024, 16)[:ne].cpu().numpy()
                print("perm equal", np.array_equal(perm, e2))
                for e in range(ne):
                    rows = np.flatnonzero((perm[e] != e2[e]).any(axis=1))
                    print(" epoch", e, "differing rows", rows.size, rows[:12])
                    if rows.size:
                        r = rows[0]
                        print("   dev ", perm[e][r].tolist()); print("   host", e2[e][r].tolist())
            # recompute the swap stage on the host from the workspace's js words
            ws = a._devrng_ws.cpu().numpy()
            def al(x): return (x + 255) & ~255
            o_pow2 = 0; o_ctl = al(32 * 24); o_F = o_ctl + 256; o_L = o_F + al(((1 << 19) + 64) * 4)
            o_reg = o_L + 6 * al((1 << 19) * 4); o_js = o_reg + al(256 * 16)
            ctl = ws[o_ctl:o_ctl + 32].view(np.int32)
            print("ctl", ctl.tolist())
            n_reg = int(ctl[2])
            js = ws[o_js:o_js + n_reg * 1024 * 8].view(np.uint64).reshape(n_reg, 1024)
            if dig and n_reg == ne - 1:
                tab = perm[0].copy()
                for e in range(n_reg):
                    for r in range(1024):
                        w = int(js[e, r])
                        d = tab[r]
                        for i in range(15, 0, -1):
                            j = (w >> (4 * i)) & 15
                            d[i], d[j] = d[j], d[i]
                    bad_rows = np.flatnonzero((tab != perm[e + 1]).any(axis=1))
                    host_bad = np.flatnonzero((tab != e2[e + 1]).any(axis=1))
                    print(" regen", e, "host-applied js vs device epoch: rows differing", bad_rows.size, "| js-applied vs numpy:", host_bad.size)
            print("history", hist)
            sys.exit(1)
    rounds += 1
print("ok", rounds)
