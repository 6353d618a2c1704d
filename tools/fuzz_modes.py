"""Soak test: four VecGame instances (schedule = host / host-async / device, and a per-step-only
reference) driven through a random interleaving of fused rollouts of random lengths, per-step
prepare/step pairs, summary()/property reads and occasional reset()s.  All four must stay
bit-identical.  Run on the GPU box:  python tools/fuzz_modes.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ml2048_amd as amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
M = 4099
rng = np.random.default_rng(2048)
kinds = ["normal", "improved", "rank", "maxcell"]
t0 = time.time()
rounds = ops = 0
last_report = t0
while time.time() - t0 < budget:
    if time.time() - last_report > 60:  # a line a minute: a silent run looks hung to the job runner
        last_report = time.time()
        print(f"... {rounds} rounds, {ops} operations after {last_report - t0:.0f}s", flush=True)
    kind = kinds[rounds % 4]
    envs = {m: amd.VecGame(M, kind, schedule=m) for m in ("host", "host-async", "device")}
    step_env = amd.VecGame(M, kind)
    masked = bool(rounds % 3 != 1)  # unmasked rounds run the lazy-mask loop and the inverse-permutation spawn
    pol = amd.RandomPolicy(seed=5, masked=masked)
    last_n = 128
    seed = int(rng.integers(0, 2**31))
    for e in list(envs.values()) + [step_env]:
        e.reset(seed)
    counter = 0
    for _ in range(int(rng.integers(5, 25))):
        op = rng.integers(0, 12)
        if op >= 10:  # fused one-launch steps (step + following prepare) on the three envs, the per-step pair on the fourth
            k = int(rng.choice([1, 2, 5, 130]))
            for _ in range(k):
                a = rng.integers(0, 4, size=M).astype(np.uint8)
                at = torch.from_numpy(a).cuda()
                for e in envs.values():
                    if not e._prepared:
                        e.prepare_torch()
                    e.step_prepare_torch(at)
                if not step_env._prepared:
                    step_env.prepare_torch()
                step_env.step_torch(at)
                step_env.prepare_torch()
        elif op < 6:  # fused rollout: often the same length again (chained upload of the schedule taken ahead),
            # with the id scan inline or deferred, sometimes with per-step digests (drops the look-ahead)
            n = last_n if rng.random() < 0.5 else int(rng.choice([1, 2, 7, 33, 64, 128, 200]))
            last_n = n
            # one to three rollouts back to back, each with its own id-scan mode and nothing that joins in between (a
            # deferred scan is issued lazily: the next rollout's own scan - inline or deferred - must not overtake it)
            for _rep in range(int(rng.choice([1, 1, 2, 3]))):
                fin = "deferred" if rng.random() < 0.5 else True
                dig = rng.random() < 0.15
                outs = {}
                for m, e in envs.items():
                    outs[m] = e.rollout(n, masked=masked, act_seed=5, act_counter=counter, finish_ids=fin, digests=dig)
                if dig:
                    d0 = outs["host"]["digests"].cpu().numpy()
                    for m in outs:
                        assert np.array_equal(outs[m]["digests"].cpu().numpy(), d0), (rounds, ops, m, "digests")
                for _ in range(n):
                    step_env.prepare_torch()
                    pol.counter = counter
                    a, _ = pol.sample_actions_u8(step_env.valid_actions)
                    step_env.step_torch(a)
                    counter += 1
        elif op < 8:  # per-step pair on everyone with explicit actions
            a = rng.integers(0, 4, size=M).astype(np.uint8)
            for e in list(envs.values()) + [step_env]:
                e.prepare()
                e.step(a)
        elif op == 8:  # reads that force generator hand-backs
            vals = {m: (e._rand_step, e._game_count) for m, e in envs.items()}
            assert len(set(vals.values())) == 1, vals
            assert vals["host"] == (step_env._rand_step, step_env._game_count)
        else:
            seed = int(rng.integers(0, 2**31))
            counter = 0
            for e in list(envs.values()) + [step_env]:
                e.reset(seed)
        ops += 1
        ref = step_env.digest()
        for m, e in envs.items():
            assert e.digest() == ref, (rounds, ops, m)
    ref = step_env._data.tobytes()
    for m, e in envs.items():
        assert e._data.tobytes() == ref, (rounds, m)
    rounds += 1
print(f"fuzz ok: {rounds} rounds, {ops} operations in {time.time()-t0:.0f}s")
