import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd
M = 8192
seq = (128, 64, 128, 200)
def run(mode_a, mode_b, sync):
    a = amd.VecGame(M, "improved", schedule=mode_a)
    b = amd.VecGame(M, "improved", schedule=mode_b)
    a.reset(31); b.reset(31)
    da, db = [], []
    for k, n in enumerate(seq):
        a.rollout(n, masked=True, act_seed=2, act_counter=1000 * k)
        if sync: torch.cuda.synchronize()
        b.rollout(n, masked=True, act_seed=2, act_counter=1000 * k)
        if sync: torch.cuda.synchronize()
        if sync:
            da.append(a.digest()); db.append(b.digest())
    if not sync:
        da.append(a.digest()); db.append(b.digest())
    return da, db
ref = run("host", "host", True)[0]
print("ref", ["%016x" % (x & (2**64-1)) for x in ref])
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for ma, mb in (("device", "host"), ("host", "host"), ("device", "device"), ("host-async", "host")):
        da, db = run(ma, mb, False)
        ok_a, ok_b = da[-1] == ref[-1], db[-1] == ref[-1]
        if not (ok_a and ok_b):
            print("rep", rep, ma, mb, "a_ok", ok_a, "b_ok", ok_b, flush=True)
print("done")
