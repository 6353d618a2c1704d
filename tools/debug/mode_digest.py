import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ml2048_amd as amd
M = 8192
for mode in sys.argv[1:]:
    e = amd.VecGame(M, "improved", schedule=mode)
    e.reset(31)
    out = []
    for k, n in enumerate((128, 64, 128, 200)):
        e.rollout(n, masked=True, act_seed=2, act_counter=1000 * k)
        out.append("%016x" % (e.digest() & (2**64 - 1)))
    print(mode.ljust(12), " ".join(out), flush=True)
