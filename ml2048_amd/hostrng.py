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

_LA_POOL = None


def _la_pool():
    """One worker thread for all lookahead blocks of the process (HostRng.next_step_prepare)."""
    global _LA_POOL
    if _LA_POOL is None:
        import concurrent.futures as cf

        _LA_POOL = cf.ThreadPoolExecutor(max_workers=1, thread_name_prefix="ml2048-hostrng-la")
    return _LA_POOL


class HostRng:
    def __init__(self, seed: Optional[int] = None):
        self.rand = np.random.default_rng(seed)  # :607
        self.rand_step = 0  # :609
        self.randperm = np.empty((RAND_SIZE, 16), dtype=np.uint8)
        self.randfloat = np.empty((RAND_SIZE,), dtype=np.float32)
        self.randperm[:, :] = np.arange(16).reshape((1, 16))  # :610
        self.epoch = 0  # number of table regenerations so far (bookkeeping only)
        self._la = None  # draws taken ahead for step_prepare loops (see next_step_prepare)
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
        c._la = None
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

    # ---- draws ahead for a loop of {step(t); prepare(t+1)} calls (VecGame.step_prepare_torch) ----------------------
    # numpy's Generator costs ~2.5 us per integers() call; a loop that issues one 8 us kernel per env step cannot afford
    # three of them.  The draw order is state-independent, so a block of LOOKAHEAD (prepare, step)
    # pairs is drawn by ONE native call (draw_schedule) and handed out pair by pair.  The object then runs AHEAD of
    # what the env has consumed; `sync()` puts it back to exactly the consumed position (by replaying the consumed part
    # of the block from a clone taken at its start) and must precede anything else that touches rand / rand_step / the
    # tables.  Consumption order per call: the pending step draw, then the next prepare's draws.
    LOOKAHEAD = 128

    @staticmethod
    def _la_block(start: "HostRng", n: int, two_prob: float):
        """One block of n (prepare, step) pairs drawn from a copy of `start` (which stays at the block's start for
        sync() to replay from).  Runs on the caller's thread or on the lookahead worker (the native call drops the GIL)."""
        after = start.clone()
        sched, epochs, masks = after.draw_schedule(n, two_prob)
        return after, sched.tolist(), epochs, masks.tolist()

    def next_step_prepare(self, two_prob: float):
        """Draws of step(t) followed by prepare(t+1): -> (row_base_step, row_base_prepare, e_step, e_prepare, block).
        `block` is the lookahead block the draws come from (a list; block[2] = its epoch tables uint8[E,1024,16],
        block[7] = their 2/4 masks - all the reference reads of randfloat, H1); e_step / e_prepare index the tables in
        force for the step and after the prepare (they differ when that prepare regenerated the tables).  A new block
        object means new tables: its epoch 0 is the table the previous block ended with."""
        la = self._la
        if la is None or la[4] == la[5]:
            if la is None:
                step_row = self.draw_step()
                start = self.clone()
                block = self._la_block(start, self.LOOKAHEAD, two_prob)
            else:  # block exhausted: its last draw - a step - is still pending; the next block was drawn meanwhile
                step_row, start = la[3], la[10]
                block = la[9].result()
            after, sched, epochs, masks = block
            # this object moves to the END of the block (sync() brings it back to what was consumed) ...
            self.rand, self.rand_step, self.epoch = after.rand, after.rand_step, after.epoch
            self.randperm[...] = after.randperm
            self.randfloat[...] = after.randfloat
            # ... and the block after it is drawn on the worker thread while this one is handed out
            nstart = after.clone()
            fut = _la_pool().submit(self._la_block, nstart, self.LOOKAHEAD, two_prob)
            #                0      1      2       3         4  5               6  7      8         9    10
            la = self._la = [start, sched, epochs, step_row, 0, self.LOOKAHEAD, 0, masks, two_prob, fut, nstart]
        row_s, e_s = la[3], la[6]
        e_p, row_p, la[3] = la[1][la[4]]
        la[4] += 1
        la[6] = e_p
        return row_s, row_p, e_s, e_p, la

    def sync(self) -> None:
        """Back to exactly what has been consumed (no-op without a lookahead)."""
        la = self._la
        if la is None:
            return
        self._la = None
        j = la[4]  # consumed of the block: pairs 1..j-1 and the prepare of pair j (its step is pending, unused)
        start = la[0]
        if j >= 2:
            start.draw_schedule(j - 1, la[8])
        if j >= 1:
            start.draw_prepare()
        else:  # nothing of the block consumed, but the pending step was drawn before its start: cannot happen after
            raise AssertionError("lookahead block with nothing consumed")  # a call (j >= 1), kept as a guard
        self.rand = start.rand
        self.rand_step = start.rand_step
        self.randperm[...] = start.randperm
        self.randfloat[...] = start.randfloat
        self.epoch = start.epoch

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
