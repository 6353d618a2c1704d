import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import ml2048_amd as amd
M = 8192
for mode in ("host-async", "host"):
    a = amd.VecGame(M, "normal", schedule=mode); b = amd.VecGame(M, "normal", schedule=mode)
    a.reset(5); b.reset(5)
    c = 0
    for k, fin in enumerate(["deferred", True, "deferred", "deferred", True, True, "deferred"]):
        a.rollout(100, masked=False, act_seed=3, act_counter=c, finish_ids=fin)
        b.rollout(100, masked=False, act_seed=3, act_counter=c, finish_ids=True)
        c += 100
    ia, ib = a.game_id.cpu().numpy(), b.game_id.cpu().numpy()
    print(mode, "ids equal:", np.array_equal(ia, ib), "count", a._game_count, b._game_count)
