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
