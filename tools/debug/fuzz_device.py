import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ml2048_amd as amd
M = 1031
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
t0 = time.time()
rounds = 0
while time.time() - t0 < 40:
    a = amd.VecGame(M, "normal", schedule="device")
    b = amd.VecGame(M, "normal", schedule="host")
    seed = int(rng.integers(0, 2**31))
    a.reset(seed); b.reset(seed)
    hist = [("reset", seed)]
    counter = 0
    for _ in range(int(rng.integers(5, 25))):
        op = rng.integers(0, 10)
        if op < 7:
            n = int(rng.choice([1, 2, 7, 33, 64, 128, 129, 200, 256, 300]))
            masked = bool(rng.integers(0, 2))
            dig = rng.random() < 0.2
            hist.append(("rollout", n, masked, dig))
            a.rollout(n, masked=masked, act_seed=5, act_counter=counter, digests=dig)
            b.rollout(n, masked=masked, act_seed=5, act_counter=counter, digests=dig)
            counter += n
        elif op < 9:
            act = rng.integers(0, 4, size=M).astype(np.uint8)
            hist.append(("step",))
            for e in (a, b):
                e.prepare(); e.step(act)
        else:
            hist.append(("read",))
            assert (a._rand_step, a._game_count) == (b._rand_step, b._game_count), hist
        if a.digest() != b.digest():
            print("MISMATCH after", hist[-1], "history:", hist)
            sa, sb = a._rng.rand.bit_generator.state, b._rng.rand.bit_generator.state
            print("rand_step", a._rand_step, b._rand_step, "rng equal", sa == sb)
            sys.exit(1)
    rounds += 1
print("ok", rounds)
