"""Host side of the spawn randomness, kept exactly as the reference keeps it.

The reference draws everything random from one host `np.random.default_rng(seed)` (PCG64) and the
jitted kernel only indexes pre-generated tables (src/ml2048/game_numba.py:577-580, 589-591,
606-611, 622-626, 670).  rocRAND cannot reproduce PCG64, so bit-exact ("table") mode keeps this
tiny piece on the host: it is O(1) per step (one float + one or two integers; a 20 KiB table
refresh on ~10 % of steps) and - because it never looks at game state - can be drawn ahead for a
whole fused rollout.
"""
from __future__ import annotations

import ctypes as C
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

    def clone(self) -> "HostRng":
        """Independent copy at the same point of the stream (used to draw a schedule ahead on a
        worker thread without touching this object)."""
        c = object.__new__(HostRng)
        c.rand = np.random.Generator(type(self.rand.bit_generator)())
        c.rand.bit_generator.state = self.rand.bit_generator.state
        c.rand_step = self.rand_step
        c.randperm = self.randperm.copy()
        c.randfloat = self.randfloat.copy()
        c.epoch = self.epoch
        return c

    # game_numba.py:589-591 - permuted() works in place, so successive tables compose
    def _reset_rand(self) -> None:
        self.epoch += 1
        if self._native_regen():
            return
        self.rand.permuted(self.randperm, axis=1, out=self.randperm)
        self.rand.random(dtype=self.randfloat.dtype, out=self.randfloat)

    def _native_regen(self) -> bool:
        """The same regeneration through ml2048_hostrng_regen (numpy's permuted() on the 1024x16 table
        is ~140 us; the native restatement ~40 us and is checked against numpy in the tests).
        Returns False (caller uses numpy) when the library or a PCG64 generator is not available."""
        try:
            from . import _lib

            L = _lib.load()
        except Exception:
            return False
        bg = self.rand.bit_generator
        st = bg.state
        if st["bit_generator"] != "PCG64":
            return False
        s, inc = st["state"]["state"], st["state"]["inc"]
        m64 = (1 << 64) - 1
        c = _lib.Pcg64State(s >> 64, s & m64, inc >> 64, inc & m64, int(st["has_uint32"]), int(st["uinteger"]))
        if L.ml2048_hostrng_regen(C.byref(c), self.randperm.ctypes.data, self.randfloat.ctypes.data) != 0:
            return False
        st["state"]["state"] = (int(c.state_hi) << 64) | int(c.state_lo)
        st["has_uint32"] = int(c.has_uint32)
        st["uinteger"] = int(c.uinteger)
        bg.state = st
        return True

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

    def draw_schedule(self, n_steps: int, two_prob: float):
        """Pre-draw n_steps x {draw_prepare; draw_step} in one native call
        (ml2048_hostrng_schedule).  Returns (schedule int32[N,3], perm_epochs uint8[E,1024,16],
        two_masks uint32[E]); epoch 0 is the table in force at entry.  Leaves this object (numpy
        generator state, tables, rand_step) exactly where N python-level draws would."""
        from . import _lib

        L = _lib.load()
        n = int(n_steps)
        bg = self.rand.bit_generator
        st = bg.state
        if st["bit_generator"] != "PCG64":  # default_rng is PCG64; anything else: python path
            return self._draw_schedule_py(n, two_prob)
        s, inc = st["state"]["state"], st["state"]["inc"]
        mask64 = (1 << 64) - 1
        c = _lib.Pcg64State(s >> 64, s & mask64, inc >> 64, inc & mask64, int(st["has_uint32"]), int(st["uinteger"]))
        # a regeneration happens with probability 0.1 per step; n + 1 epochs can never overflow
        cap = n + 1 if n <= 64 else max(16, n // 4 + 8)
        while True:
            sched = np.empty((n, 3), dtype=np.int32)
            epochs = np.empty((cap, RAND_SIZE, 16), dtype=np.uint8)
            masks = np.empty((cap,), dtype=np.uint32)
            rs = C.c_int32(self.rand_step)
            ne = C.c_int32(0)
            perm, flt = self.randperm.copy(), self.randfloat.copy()
            cc = _lib.Pcg64State.from_buffer_copy(c)
            rc = L.ml2048_hostrng_schedule(C.byref(cc), perm.ctypes.data, flt.ctypes.data, C.byref(rs), n,
                                           float(two_prob), sched.ctypes.data, epochs.ctypes.data,
                                           masks.ctypes.data, cap, C.byref(ne))
            if rc == -3 and cap < n + 1:  # ML2048_E_UNSUPPORTED: epoch buffer too small - retry larger
                cap = n + 1
                continue
            _lib.check(rc, "hostrng_schedule")
            break
        self.randperm[...] = perm
        self.randfloat[...] = flt
        self.rand_step = int(rs.value)
        self.epoch += int(ne.value) - 1
        st["state"]["state"] = (int(cc.state_hi) << 64) | int(cc.state_lo)
        st["has_uint32"] = int(cc.has_uint32)
        st["uinteger"] = int(cc.uinteger)
        bg.state = st
        k = int(ne.value)
        return sched, epochs[:k], masks[:k]

    def _draw_schedule_py(self, n: int, two_prob: float):
        """Pure numpy reference of draw_schedule (tests): python-level draws, numpy's own permuted()."""
        native, self._native_regen = self._native_regen, (lambda: False)
        try:
            return self._draw_schedule_py_inner(n, two_prob)
        finally:
            self._native_regen = native

    def _draw_schedule_py_inner(self, n: int, two_prob: float):
        epochs = [self.randperm.copy()]
        masks = [self.two_mask(two_prob)]
        sched = np.empty((n, 3), dtype=np.int32)
        for t in range(n):
            regen, row_p = self.draw_prepare()
            if regen:
                epochs.append(self.randperm.copy())
                masks.append(self.two_mask(two_prob))
            row_s = self.draw_step()
            sched[t] = (len(epochs) - 1, row_p, row_s)
        return sched, np.stack(epochs), np.array(masks, dtype=np.uint32)
