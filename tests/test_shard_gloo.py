"""The N>1 host logic (shard plan, metrics gather, max-over-ranks timing) under world_size-2 gloo
on the CPU.  The data path has no collective to test: shards are independent games."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from ml2048_amd.shard import ShardPlan, gather_counts, max_over_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_plan_partitions_exactly():
    for total, world in [(2097152, 8), (65536, 1), (1000, 3), (7, 7), (10, 4)]:
        plans = [ShardPlan.make(total, r, world, base_seed=100) for r in range(world)]
        assert sum(p.boards for p in plans) == total
        assert plans[0].idx_base == 0
        for a, b in zip(plans, plans[1:]):
            assert b.idx_base == a.idx_base + a.boards
        assert [p.seed for p in plans] == [100 + r for r in range(world)]
    assert ShardPlan.make(2097152, 3, 8).boards == 262144  # C4: 8 x 262144
    with pytest.raises(ValueError):
        ShardPlan.make(2, 2, 3)
    with pytest.raises(ValueError):
        ShardPlan.make(0, 0, 1)


def test_single_rank_passthrough():
    assert np.array_equal(gather_counts([1, 2, 3]), [1, 2, 3])
    assert max_over_ranks(0.25) == 0.25


WORKER = textwrap.dedent(
    """
    import os, sys, json
    sys.path.insert(0, sys.argv[1])
    import numpy as np, torch, torch.distributed as dist
    from ml2048_amd.shard import ShardPlan, gather_counts, max_over_ranks
    from ml2048_amd.runner import RunnerStats
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = ShardPlan.make(1001, rank, world, base_seed=5)
    counts = np.zeros(20, np.int64); counts[3 + rank] = plan.boards
    tot = gather_counts(counts, dist)
    t = max_over_ranks(1.0 + rank, dist)
    st = RunnerStats(); st.counts[5] = 10 * (rank + 1); st.terminated_count = rank + 1
    merged = gather_counts(st.counts, dist)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"tot": tot.tolist(), "t": t, "merged5": int(merged[5]), "boards": plan.boards}))
    dist.destroy_process_group()
    """
)


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json

    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["tot"][3] == 501 and r["tot"][4] == 500 and sum(r["tot"]) == 1001
    assert r["t"] == 2.0 and r["merged5"] == 30 and r["boards"] == 501
