"""Sharding across the GPUs of a node: boards are independent games, so a shard is just an
independent `VecGame` (SURVEY 8e "replica mode").  No data-path collective exists; the only
cross-rank traffic is a host-side gather of a few integers for metrics, done here with whatever
`torch.distributed` backend the job uses (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass(frozen=True)
class ShardPlan:
    """What rank `rank` of `world` runs when a job asks for `total_boards` boards."""

    rank: int
    world: int
    boards: int      # boards on this rank
    idx_base: int    # global index of this rank's first board
    seed: int        # replica seed (base_seed + rank)

    @staticmethod
    def make(total_boards: int, rank: int, world: int, base_seed: int = 0) -> "ShardPlan":
        if total_boards <= 0 or world <= 0 or not (0 <= rank < world):
            raise ValueError(f"total_boards={total_boards} rank={rank} world={world}")
        q, r = divmod(total_boards, world)
        boards = q + (1 if rank < r else 0)
        idx_base = rank * q + min(rank, r)
        if boards == 0:
            raise ValueError(f"rank {rank} would own no boards ({total_boards} boards over {world} ranks)")
        return ShardPlan(rank, world, boards, idx_base, base_seed + rank)


def host_thread_plan(cpu_share: int, local_world: int, requested: str = "auto") -> dict:
    """How one rank of `local_world` ranks on a node with `cpu_share` usable CPUs produces the bit-exact mode's
    PCG64 draw schedule: {"threads": producer workers (ML2048_HOST_THREADS), "schedule": VecGame(schedule=...),
    "cpus_per_rank": ...}.

    Every rank runs a Python thread, the HIP runtime's helper threads and (host-async) one stitcher besides the
    workers, hence `share - 4`; more than 9 workers crowd those out on a 16-CPU share (measured end to end with the
    0.068 ms kernel, five runs each: 9 workers 0.0698-0.0700 ms per rollout every time; 10 workers 0.0697-0.0734;
    8 workers 0.0699-0.0776, producer-bound in three of five; 12-14 slower and erratic).
    Measured per 128-step schedule on the GPU box's EPYC 9575F: 0.48 / 0.26 / 0.13 / 0.10 / 0.074 / 0.062 ms with
    1 / 2 / 4 / 6 / 8 / 10 workers, against a 0.068 ms rollout kernel at 65 536 boards - so fewer than 9 workers make
    the run host-bound, but it stays correct and pipelined.  "auto" never picks the schedule drawn on the GPU
    (`schedule="device"`, 0.128 ms per rollout with no host threads at all): that path is opt-in.  With fewer than
    3 CPUs per rank there is no room for producer threads: the synchronous native call."""
    per_rank = max(1, int(cpu_share) // max(1, int(local_world)))
    threads = max(1, min(9, per_rank - 4))
    if requested != "auto":
        schedule = requested
    elif per_rank < 3:
        schedule = "host"
    else:
        schedule = "host-async"
    return {"threads": threads, "schedule": schedule, "cpus_per_rank": per_rank}


def gather_counts(counts, dist=None) -> np.ndarray:
    """Sum a small integer vector (max-tile histogram, terminated count, ...) over all ranks.
    dist: the torch.distributed module of an initialised process group, or None for one rank."""
    t = torch.as_tensor(np.asarray(counts), dtype=torch.int64).clone()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def max_over_ranks(seconds: float, dist=None) -> float:
    """The bench contract's max-over-ranks of a wall-clock interval."""
    t = torch.tensor([seconds], dtype=torch.float64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class GlobalVecGame:
    """SURVEY 8e "global mode": K shards (one per device, or several on one device) that together are
    bit-exact with ONE reference `VecGame(total)` - same table rows (idx_base), same host draw per
    call, and the reference's ascending game ids across shard boundaries.

    The ids are the only cross-shard dependency: shard k's new games are numbered after those of
    shards 0..k-1, so `prepare()` first reads the K terminated counts (K small D2H reads), then
    gives every shard its id base and lets it deal.  Throughput runs should use independent replicas
    (`ShardPlan`); this class exists for exact reproduction of a single big game on several GPUs.
    """

    def __init__(self, total: int, reward_fn=None, *, devices, two_prob: float = 0.8):
        from .game import VecGame

        devices = list(devices)
        if total <= 0 or not devices:
            raise ValueError(f"total={total} devices={devices}")
        self._size = int(total)
        self.plans = [ShardPlan.make(total, k, len(devices)) for k in range(len(devices))]
        self.shards = [VecGame(p.boards, reward_fn, two_prob=two_prob, device=d, idx_base=p.idx_base)
                       for p, d in zip(self.plans, devices)]
        self._game_count = 0
        self.reset()

    def reset(self, seed=None) -> None:
        from .hostrng import HostRng

        master = HostRng(seed)
        for sh in self.shards:  # every shard replays the same host stream (the draws do not depend on game state)
            sh.reset(0)
            sh._rng = master.clone()
            sh._upload_tables()

    def prepare(self):
        counts = [int((sh.terminated != 0).sum().item()) for sh in self.shards]
        base = self._game_count
        out = []
        for sh, p, n in zip(self.shards, self.plans, counts):
            sh._game_count_dev.fill_(base)
            (idx,) = sh.prepare()
            assert idx.size == n
            out.append(idx + p.idx_base)
            base += n
        self._game_count = base
        return (np.concatenate(out),)

    def step(self, actions):
        actions = np.ascontiguousarray(actions)
        assert actions.shape == (self._size,), actions.shape
        res = [sh.step(actions[p.idx_base:p.idx_base + p.boards]) for sh, p in zip(self.shards, self.plans)]
        return {k: np.concatenate([r[k] for r in res]) for k in res[0]}

    @property
    def _data(self):
        return np.concatenate([sh._data for sh in self.shards])

    def summary(self):
        counts = sum(sh.histogram().cpu().numpy().astype(np.int64) for sh in self.shards)
        total = int(counts.sum())
        rows = [(2 ** k, int(c), int(c) / total) for k, c in enumerate(counts) if c]
        rows.sort(key=lambda r: r[0], reverse=True)
        return rows
