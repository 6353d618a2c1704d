"""Host side of the spawn randomness, kept exactly as the reference keeps it.

The reference draws everything random from one host `np.random.default_rng(seed)` (PCG64) and the
jitted kernel only indexes pre-generated tables (src/ml2048/game_numba.py:577-580, 589-591,
606-611, 622-626, 670).  rocRAND cannot reproduce PCG64, so bit-exact ("table") mode keeps this
tiny piece on the host: it is O(1) per step (one float + one or two integers; a 20 KiB table
refresh on ~10 % of steps) and - because it never looks at game state - can be drawn ahead for a
whole fused rollout.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

RAND_SIZE = 1024  # VecGame._RAND_SIZE, game_numba.py:533


class HostRng:
    def __init__(self, seed: Optional[int] = None):
        self.rand = np.random.default_rng(seed)  # :607
        self.rand_step = 0  # :609
        self.randperm = np.empty((RAND_SIZE, 16), dtype=np.uint8)
        self.randfloat = np.empty((RAND_SIZE,), dtype=np.float32)
        self.randperm[:, :] = np.arange(16).reshape((1, 16))  # :610
        self.epoch = 0  # number of table regenerations so far (bookkeeping only)
        self._reset_rand()  # :611

    # game_numba.py:589-591 - permuted() works in place, so successive tables compose
    def _reset_rand(self) -> None:
        self.rand.permuted(self.randperm, axis=1, out=self.randperm)
        self.rand.random(dtype=self.randfloat.dtype, out=self.randfloat)
        self.epoch += 1

    # game_numba.py:622-626 - drawn on every prepare(), terminated games or not
    def draw_prepare(self) -> tuple[bool, int]:
        """-> (tables regenerated?, row_base = rand_step + rand_offset)"""
        regen = False
        if self.rand.random() >= 0.9 or self.rand_step >= RAND_SIZE:
            self.rand_step = 0
            self._reset_rand()
            regen = True
        rand_offset = int(self.rand.integers(0, RAND_SIZE))
        return regen, self.rand_step + rand_offset

    # game_numba.py:670, :681, :685
    def draw_step(self) -> int:
        """-> row_base = rand_step + rand_offset (rand_step before its increment)"""
        rand_offset = int(self.rand.integers(0, RAND_SIZE))
        row_base = self.rand_step + rand_offset
        self.rand_step += 1
        return row_base

    def two_mask(self, two_prob: float) -> int:
        """bit c <=> float64(randfloat[c]) < two_prob: the only part of randfloat the reference
        ever reads (it indexes by the CELL, game_numba.py:207)."""
        m = 0
        for c in range(16):
            if float(self.randfloat[c]) < two_prob:
                m |= 1 << c
        return m
