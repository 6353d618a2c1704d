"""where does a large-M plain rollout on loaded boards with big tiles differ from the oracle"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import ml2048_amd as amd
from oracle import oracle

M, N = int(os.environ.get("DBG_M", "139264")), 96
env = amd.VecGame(M, "improved", schedule="host-async"); ref = oracle.OracleVecGame(M, "improved")
env.reset(23); ref.reset(23); env.prepare(); ref.prepare()
d = ref._data
rng = np.random.default_rng(2)
for idx in rng.choice(M, size=600, replace=False):
    cells = rng.choice(16, size=4, replace=False)
    d["board"][idx, cells] = rng.integers(12, 16, size=4)
    if rng.random() < 0.5:
        d["board"][idx, 0] = d["board"][idx, 1] = rng.integers(12, 15)
    d["valid_actions"][idx] = oracle.compute_valid_actions(d["board"][idx])[0]
start = d.copy()
env.load_records(d)
env.rollout(N, masked=False, act_seed=7, act_counter=3)
hist = []
for t in range(N):
    ref.prepare()
    a, _ = oracle.sample_actions(ref._data["valid_actions"], 7, 3 + t, masked=False)
    ref.step(a)
    hist.append(ref._data["board"].max(axis=1).copy())
a, b = env._data, ref._data
for f in ("id", "step", "score", "reward", "board", "merged", "valid_actions", "terminated", "invalid"):
    x, y = a[f], b[f]
    bad = np.nonzero((x != y).reshape(M, -1).any(axis=1))[0]
    print(f, "differs at", bad.size, "boards", bad[:10])
    for i in bad[:4]:
        print("  idx", i, "wave", i // 64, "gpu", x[i], "oracle", y[i], "start board", start["board"][i], "max tile over time", max(h[i] for h in hist))
