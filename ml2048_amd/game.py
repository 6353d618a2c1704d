"""`VecGame`: the reference's vectorized 2048 environment API on MI355X HIP kernels.

Drop-in for `ml2048.game_numba.VecGame` (src/ml2048/game_numba.py:522-698): same constructor,
`reset / prepare / observations / step / summary`, same `VecStepResult` keys and dtypes, same
private attributes callers read (`_size`, `_data`, `_game_count`, `_rand_step`, ...).  All
per-board work runs in libml2048_hip.so through the C ABI of include/ml2048_hip.h; state lives in
HBM as PyTorch-ROCm tensors (caller-visible, structure-of-arrays).  Besides the numpy-returning
reference API there is a device-native one (`*_torch`, `onehot`, `rollout`) that never leaves the GPU.

Differences from the reference, all deliberate and documented in DESIGN.md:
  * `reward_fn` is one of the four reference reward functions, passed as the sentinel objects
    exported here (or their names): a Python callback cannot be fused into a HIP kernel.
  * the numpy API returns snapshots, not aliasing views of live state (hazard H12).
  * out-of-range actions are treated as invalid moves instead of undefined behaviour (H11).
"""
from __future__ import annotations

import collections
import collections.abc
import ctypes as C
import os
from typing import Any, Optional, TypedDict

import numpy as np
import torch

from . import _lib
from .hostrng import RAND_SIZE, HostRng

# src/ml2048/game.py:14-17
STEP_LEFT = 0
STEP_RIGHT = 1
STEP_UP = 2
STEP_DOWN = 3

# game_numba.py:23-45
ITEM_VALUES = np.array([0] + [2**k for k in range(1, 18)], dtype=np.int32)

# game_numba.py:537-551
RECORD_DTYPE = np.dtype(
    [
        ("id", np.int32, ()),
        ("step", np.int32, ()),
        ("score", np.float32, ()),
        ("reward", np.float32, ()),
        ("board", np.uint8, (16,)),
        ("merged", np.uint8, (16,)),
        ("valid_actions", np.uint8, (4,)),
        ("terminated", np.uint8, ()),
        ("invalid", np.uint8, ()),
        ("_padding", np.uint8, 10),
    ],
    align=True,
)
assert RECORD_DTYPE.itemsize == 64, RECORD_DTYPE.itemsize


class RewardFn:
    """Sentinel for one of the reference's reward functions (game_numba.py:408-504).

    Pass it as `VecGame(reward_fn=...)`; the kernel computes it on device.  Calling it evaluates
    the same formula on one (state, prev_state, merged) triple on the host (convenience only).
    """

    def __init__(self, name: str, kind: int):
        self.name, self.kind = name, kind
        self.__name__ = f"reward_fn_{name}"

    def __repr__(self) -> str:
        return f"<ml2048_amd.{self.__name__}>"

    def __call__(self, state, prev_state, merged) -> float:
        state = [int(v) for v in state]
        prev_state = [int(v) for v in prev_state]
        merged = [int(v) for v in merged][:16]
        normal = sum(m * (2 << k) for k, m in enumerate(merged))
        if self.kind == 0:
            return float(normal)
        if self.kind == 1:
            extra = 0
            if prev_state[0] != 0:
                extra -= 64 * (1 << prev_state[0])
            if state[0] != 0:
                extra += 64 * (1 << state[0])
            return float(normal) + float(extra)
        if self.kind == 2:
            return float(sum((k + 1) * m for k, m in enumerate(merged)))
        reward = 0.0
        if max(state) > max(prev_state):
            reward += float(2 ** max(state))
        return reward + float(sum(merged))


reward_fn_normal = RewardFn("normal", 0)
reward_fn_improved = RewardFn("improved", 1)
reward_fn_rank = RewardFn("rank", 2)
reward_fn_maxcell = RewardFn("maxcell", 3)
_REWARDS = {r.name: r for r in (reward_fn_normal, reward_fn_improved, reward_fn_rank, reward_fn_maxcell)}


def _reward_kind(reward_fn) -> int:
    if reward_fn is None:
        return 0
    if isinstance(reward_fn, RewardFn):
        return reward_fn.kind
    if isinstance(reward_fn, str) and reward_fn in _REWARDS:
        return _REWARDS[reward_fn].kind
    if isinstance(reward_fn, int) and 0 <= reward_fn < 4:
        return reward_fn
    name = getattr(reward_fn, "__name__", "")
    if name.startswith("reward_fn_") and name[len("reward_fn_"):] in _REWARDS:
        return _REWARDS[name[len("reward_fn_"):]].kind  # the reference's own function objects
    raise TypeError(
        "reward_fn must be one of ml2048_amd.reward_fn_{normal,improved,rank,maxcell} (or its name): "
        "an arbitrary Python callable cannot be compiled into the HIP step kernel"
    )


class VecStepResult(TypedDict):  # game_numba.py:507-519
    state: Any
    valid_actions: Any
    step: Any
    merged: Any
    reward: Any
    score: Any
    terminated: Any
    invalid: Any
    prev_state: Any
    prev_valid_actions: Any


_ACT_CODE = {torch.uint8: 0, torch.int32: 1, torch.int64: 2}


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()
_NO_NEXT = (None, None)


class _Event:
    """A hipEvent_t of the library (ml2048_event_*): record / query / synchronize / wait without torch's event objects -
    a plain rollout issues one ctypes call for the launch, its marker and the id scan (ml2048_rollout_chain_ids), and
    only these handles cross the boundary."""
    __slots__ = ("h", "_L")

    def __init__(self, L):
        h = C.c_void_p()
        _lib.check(L.ml2048_event_create(C.byref(h)), "event_create")
        self.h, self._L = h.value, L

    def record(self, stream_raw: int) -> None:
        _lib.check(self._L.ml2048_event_record(self.h, stream_raw), "event_record")

    def query(self) -> bool:
        r = self._L.ml2048_event_query(self.h)
        if r < 0:
            _lib.check(r, "event_query")
        return r == 1

    def synchronize(self) -> None:
        _lib.check(self._L.ml2048_event_synchronize(self.h), "event_synchronize")

    def __del__(self):
        try:
            self._L.ml2048_event_destroy(self.h)
        except Exception:
            pass


class _RolloutBuffers(collections.abc.Mapping):
    """What a plain rollout returns: the same keys as the dict of the general path, over ONE device allocation whose
    views are only built when somebody asks for them (a view costs more host time than the rollout has to spare)."""
    _KEYS = ("actions", "digests", "step_resets", "reset_flags", "last_reset")

    def __init__(self, buf, N, W, M, o_last, o_resets, has_resets):
        self._buf, self._dims, self._views = buf, (N, W, M, o_last, o_resets, has_resets), {}

    def __getitem__(self, key):
        if key not in self._KEYS:
            raise KeyError(key)
        if key not in self._views:
            N, W, M, o_last, o_resets, has_resets = self._dims
            b = self._buf
            if key == "reset_flags":
                v = b[:N * W * 8].view(torch.int64).view(N, W)
            elif key == "last_reset":
                v = b[o_last:o_last + M * 4].view(torch.int32)
            elif key == "step_resets":
                v = b[o_resets:o_resets + N * 4].view(torch.int32) if has_resets else None
            else:
                v = None
            self._views[key] = v
        return self._views[key]

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)


_LAZY_IDS = os.environ.get("ML2048_LAZY_IDS", "1") != "0"  # see VecGame._rollout_plain_chain
_ROW_LUTS: dict = {}  # device index -> the line table of the plain rollout kernels (constant data, 128 KiB)


def _row_lut(L, device: torch.device) -> int:
    """Device pointer of this device's line table (ml2048_row_lut_fill), built on first use.  ML2048_ROW_LUT=0 makes the
    plain rollouts compute the move arithmetically instead (A/B; same results)."""
    if os.environ.get("ML2048_ROW_LUT", "1") == "0":
        return 0
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _ROW_LUTS.get(key)
    if t is None:
        with torch.cuda.device(device):
            t = torch.empty((131072,), dtype=torch.uint8, device=device)
            _lib.check(L.ml2048_row_lut_fill(torch.cuda.current_stream(device).cuda_stream, t.data_ptr()), "row_lut_fill")
            torch.cuda.current_stream(device).synchronize()  # any stream may read it from now on
        _ROW_LUTS[key] = t
    return t.data_ptr()


class VecGame:
    """Vectorized 2048 environment on one MI355X (see module docstring)."""

    _RAND_SIZE: int = RAND_SIZE
    _DATA_DTYPE = RECORD_DTYPE

    def __init__(
        self,
        size: int,
        reward_fn=None,
        *,
        two_prob: float = 0.8,
        reuse_state: bool = False,
        device: torch.device | str | None = None,
        spawn: str = "table",
        spawn_seed: Optional[int] = None,
        idx_base: int = 0,
        step_mapping: str = "lane",
        schedule: str = "host",
    ):
        if size <= 0:
            raise ValueError(f"size={size}")  # game_numba.py:561-562
        if spawn not in ("table", "rocrand"):
            raise ValueError(f"spawn={spawn!r} (expected 'table' or 'rocrand')")
        self._L = _lib.load()  # raises when the HIP extension is missing: no fallback
        if not torch.cuda.is_available():
            raise _lib.Ml2048Error("ml2048_amd needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise ValueError(f"device={self.device}: VecGame state lives in GPU memory")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())

        self._size = int(size)
        self._two_prob = float(two_prob)
        self._reuse_state = reuse_state  # stored, never read - as in the reference (:569)
        self._reward_fn = reward_fn if reward_fn is not None else reward_fn_normal
        self._reward_kind = _reward_kind(reward_fn)
        self._spawn = spawn
        if schedule not in ("host", "host-async", "device"):
            raise ValueError(f"schedule={schedule!r} (expected 'host', 'host-async' or 'device')")
        # where the PCG64 draw schedule of fused rollouts is computed: on the host (native, parallel)
        # or by the single-workgroup GPU kernel of csrc/devrng.hip (no host work, same bits)
        # "host-async": a native producer thread keeps the next schedules drawn ahead (no GIL involved)
        self._schedule_mode = schedule
        self._sstream = None        # ml2048_sched_stream handle
        self._sstream_n = 0
        self._ids_stream = None     # rollout(finish_ids="deferred"): side stream of the id scan ...
        self._ids_done = None       # ... its completion event, and whether the launch stream still has to
        self._ids_pending = False   # wait for it before anything touches ids / the game counter
        self._lazy_scan = None      # the newest deferred plain rollout, whose id scan has not been issued yet
        self._side_scans = False    # scans are in flight on the side stream (its _ids_done event is meaningful)
        self._spin = [None, None, None, None]  # pinned staging ring: [tensor, event of the launch that reads it]
        self._spin_i = 0
        self._roll_ev = None        # ring of per-rollout events (ONE per rollout: the kernel's own completion event)
        self._plain_layouts = {}    # (M, N) -> offsets inside a plain rollout's one device allocation
        self._lut_ptr = _row_lut(self._L, self.device)  # the plain rollout kernels' line table (0: arithmetic move)
        self._roll_i = 0
        self._sdev = [None, None]   # device-side packed schedules (the resident one and the next one)
        self._sdev_i = 0
        self._la = None             # host-async: the NEXT schedule, already taken from the stream and uploaded
        self._ss_mem = None         # pinned memory the schedule stream's slots live in (the GPU reads them in place)
        self._ss_slots = None       # per slot: [pinned view, event of the launch that reads it, randperm after, randfloat after]
        self._ss_held = collections.deque()  # slots handed out by take_slot and not released yet, oldest first
        self._sstream_after = None  # generator / counter / tables after the last CONSUMED schedule
        self._rng_on_device = False  # True while the device copy of the generator is the current one
        if step_mapping not in ("lane", "wave"):
            raise ValueError(f"step_mapping={step_mapping!r} (expected 'lane' or 'wave')")
        # "wave" = the one-wavefront-per-board step kernel (same results; measured alternative)
        self._step_fn = self._L.ml2048_step_wave if step_mapping == "wave" else self._L.ml2048_step
        self._spawn_seed = 0x2048 if spawn_seed is None else int(spawn_seed)
        self._idx_base = int(idx_base)

        M, dev = self._size, self.device
        u8, i32, f32 = torch.uint8, torch.int32, torch.float32
        self.board = torch.empty((M, 16), dtype=u8, device=dev)
        self.merged = torch.empty((M, 16), dtype=u8, device=dev)
        self.valid_actions = torch.empty((M, 4), dtype=u8, device=dev)
        self.step_count = torch.empty((M,), dtype=i32, device=dev)
        self.score = torch.empty((M,), dtype=f32, device=dev)
        self.reward = torch.empty((M,), dtype=f32, device=dev)
        self.terminated = torch.empty((M,), dtype=u8, device=dev)
        self.invalid = torch.empty((M,), dtype=u8, device=dev)
        self._game_id = torch.empty((M,), dtype=i32, device=dev)  # public as the `game_id` property
        self.prev_state = torch.empty((M, 16), dtype=u8, device=dev)
        self.prev_valid_actions = torch.empty((M, 4), dtype=u8, device=dev)
        self._state = _lib.State(
            self.board.data_ptr(), self.merged.data_ptr(), self.valid_actions.data_ptr(),
            self.step_count.data_ptr(), self.score.data_ptr(), self.reward.data_ptr(),
            self.terminated.data_ptr(), self.invalid.data_ptr(), self._game_id.data_ptr(),
            self.prev_state.data_ptr(), self.prev_valid_actions.data_ptr(),
        )
        self._randperm_dev = torch.empty((RAND_SIZE, 16), dtype=u8, device=dev)
        self._randfloat_dev = torch.empty((RAND_SIZE,), dtype=f32, device=dev)
        self._game_count_dev = torch.zeros((1,), dtype=i32, device=dev)  # :582 - survives reset()
        self._n_reset_dev = torch.zeros((1,), dtype=i32, device=dev)
        self._reset_indices = torch.empty((M,), dtype=torch.int64, device=dev)
        self._scratch = torch.zeros((int(self._L.ml2048_scratch_ints(M)),), dtype=i32, device=dev)  # zeroed once
        self._hist = torch.zeros((20,), dtype=i32, device=dev)
        self._fast_counter = 0
        self._serial = 0            # bumped by every call that changes game state (callers cache things keyed on it)
        self._prepared = False      # True between a prepare (or step_prepare) and the next step
        self._devrng = torch.zeros((6,), dtype=torch.int64, device=dev)  # struct ml2048_devrng
        self._devrng_ws = None
        self._sched_stream = None
        self._sched_slots = None
        self.reset()  # :584 (unseeded, as the reference constructor does)

    # ------------------------------------------------------------------------------------------
    def _next_roll_event(self) -> "_Event":
        """The marker of the next rollout, from a ring longer than any ring / slot count that refers to its entries."""
        if self._roll_ev is None:
            self._roll_ev = [_Event(self._L) for _ in range(16)]
        ev = self._roll_ev[self._roll_i]
        self._roll_i = (self._roll_i + 1) % len(self._roll_ev)
        return ev

    def _ids_side_stream(self):
        if self._ids_stream is None:
            self._ids_stream = torch.cuda.Stream(device=self.device)
            self._ids_done = _Event(self._L)
        return self._ids_stream

    def _rollout_plain_chain(self, N, masked, act_seed, act_counter, finish_ids, kernel_timer, packed_ptr, o_mask, o_tab,
                             E, chain, staged):
        """The pipelined plain rollout (schedule="host-async", no optional outputs): ONE native call launches the kernel
        with the next schedule's upload chained to it, attaches the rollout's event to that dispatch and issues the id scan
        (ml2048_rollout_chain_ids); ONE device allocation holds the re-deal flags, the last-deal steps, the per-step
        re-deal counts and the scan's scratch.  Everything here is on the critical path of a 70 us kernel."""
        L, M, dev = self._L, self._size, self.device
        key = (M, N)
        lay = self._plain_layouts.get(key)
        if lay is None:
            W = int(L.ml2048_rollout_flag_words(M))
            r16 = lambda x: (x + 15) // 16 * 16
            o_last = r16(N * W * 8)
            o_resets = o_last + r16(M * 4)
            o_scratch = o_resets + r16(N * 4)
            total = o_scratch + r16(int(L.ml2048_rollout_scratch_ints(M, N)) * 4)
            lay = self._plain_layouts[key] = (W, o_last, o_resets, o_scratch, total)
        W, o_last, o_resets, o_scratch, total = lay
        with self._dev_ctx():
            buf = torch.empty((total,), dtype=torch.uint8, device=dev)
            base = buf.data_ptr()
            main = self._raw_stream()
            roll_ev = self._next_roll_event()
            ids_stream, done, scratch = main, None, base + o_scratch
            prev = None
            if finish_ids == "deferred":
                # the rollout kernel reads neither ids nor the game counter, so the id scan of this rollout may run
                # beside the NEXT rollout's kernel.  It is not even issued now: the next rollout() call issues it on a
                # side stream right behind ITS launch (nothing is ever queued between two rollout kernels), and whatever
                # touches ids first issues it on the launch stream itself (join_ids) - no cross-stream wait between the
                # last kernel of a run and its end (that wait, ~20 us, sat in every short run's tail).
                prev, scratch = self._lazy_scan, None
            elif not finish_ids:
                scratch = None
            elif self._ids_pending:
                # an inline scan behind deferred ones: the older rollouts' games are numbered first (their scans go onto
                # the launch stream / are waited for now, in front of this rollout's kernel and its scan)
                self.join_ids()
            if kernel_timer is not None:
                _lib.check(L.ml2048_timer_arm(kernel_timer), "timer_arm")
            if self._lut_ptr:
                L.ml2048_rollout_attach_row_lut(self._lut_ptr)
            _lib.check(L.ml2048_rollout_chain_ids(
                main, M, self._idx_base, self._st(), N, packed_ptr + o_tab, packed_ptr + o_mask, E, packed_ptr,
                self._reward_kind, int(masked), int(act_seed), int(act_counter), base, base + o_last,
                chain[0][0].data_ptr(), chain[1].data_ptr(), chain[2], chain[3], roll_ev.h, ids_stream,
                self._game_count_dev.data_ptr(), base + o_resets, scratch, done), "rollout_chain_ids")
            if finish_ids == "deferred":
                if prev is not None:
                    self._issue_scan(prev, on_side=True)
                self._lazy_scan = (buf, N, base, base + o_last, base + o_resets, base + o_scratch, roll_ev)
                if not _LAZY_IDS:  # A/B: this rollout's scan on the side stream at once (round 2's behaviour)
                    rec, self._lazy_scan = self._lazy_scan, None
                    self._issue_scan(rec, on_side=True)
                self._ids_pending = True
        for ps in staged:  # the pinned slots this launch read: free once its marker has passed
            ps[1] = roll_ev
        return _RolloutBuffers(buf, N, W, M, o_last, o_resets, bool(finish_ids))

    def _issue_scan(self, rec, on_side: bool) -> None:
        """Issue the ordered id scan of a deferred plain rollout (`rec`, see _rollout_plain_chain): on the side stream,
        behind that rollout's event and the earlier scans - or on the launch stream (join_ids)."""
        buf, N, p_flags, p_last, p_resets, p_scratch, roll_ev = rec
        L = self._L
        if on_side:
            side = self._ids_side_stream()
            buf.record_stream(side)  # allocated on the launch stream, used on `side`
            stream, done = side.cuda_stream, self._ids_done.h
            self._side_scans = True
        else:
            stream, done = self._raw_stream(), None
        _lib.check(L.ml2048_rollout_ids_after(stream, roll_ev.h, self._size, self._st(), N, p_flags, p_last,
                                              self._game_count_dev.data_ptr(), p_resets, p_scratch, done), "rollout_ids_after")

    def join_ids(self) -> None:
        """Order the launch stream after a deferred id scan (`rollout(finish_ids="deferred")`).  Every
        method of this class does it implicitly; call it before reading a deferred rollout's
        `step_resets` with your own torch code."""
        if self._ids_pending:
            with self._dev_ctx():
                if self._side_scans:  # scans already issued on the side stream: the launch stream continues behind them
                    _lib.check(self._L.ml2048_stream_wait_event(self._raw_stream(), self._ids_done.h), "stream_wait_event")
                    self._side_scans = False
                if self._lazy_scan is not None:  # the newest rollout's scan: on the launch stream itself
                    rec, self._lazy_scan = self._lazy_scan, None
                    self._issue_scan(rec, on_side=False)
            self._ids_pending = False

    @property
    def game_id(self) -> torch.Tensor:
        """(M,) int32 device tensor of game ids (the record's `id`, game_numba.py:641-644)."""
        self.join_ids()
        return self._game_id

    def _raw_stream(self) -> int:
        try:
            return torch._C._cuda_getCurrentRawStream(self.device.index)
        except AttributeError:  # pragma: no cover
            return torch.cuda.current_stream(self.device).cuda_stream

    def _stream(self, join: bool = True) -> int:
        """hipStream_t of torch's current stream on this device (raw accessor: ~0.3 us vs ~4 us).
        join=False only for launches that touch neither ids nor the game counter (the rollout kernel)."""
        if join and self._ids_pending:
            self.join_ids()
        try:
            return torch._C._cuda_getCurrentRawStream(self.device.index)
        except AttributeError:  # pragma: no cover - older/newer torch without the raw accessor
            return torch.cuda.current_stream(self.device).cuda_stream

    def _dev_ctx(self):
        """Device guard only when the current device differs (the guard costs a few us per call)."""
        if torch.cuda.current_device() == self.device.index:
            return _NULL_CTX
        return torch.cuda.device(self.device)

    def _st(self):
        return C.byref(self._state)

    def __del__(self):  # stop the producer thread of an open schedule stream
        try:
            if getattr(self, "_sstream", None) is not None:
                self._L.ml2048_sched_stream_destroy(self._sstream)
                self._sstream = None
        except Exception:
            pass

    # ---- host <-> device copies of the generator (schedule="device") ---------------------------
    def _push_rng(self) -> None:
        """Make the device generator current (before a device-side schedule)."""
        if self._rng_on_device:
            return
        self._drop_prefetch()
        self._close_sched_stream()
        st = self._rng.rand.bit_generator.state
        s, inc = st["state"]["state"], st["state"]["inc"]
        m64 = (1 << 64) - 1
        words = np.zeros(6, dtype=np.uint64)
        words[0], words[1], words[2], words[3] = s >> 64, s & m64, inc >> 64, inc & m64
        words[4] = (int(st["has_uint32"]) & 0xFFFFFFFF) | ((int(st["uinteger"]) & 0xFFFFFFFF) << 32)
        words[5] = self._rng.rand_step & 0xFFFFFFFF  # rand_step | error << 32
        self._devrng.copy_(torch.from_numpy(words.view(np.int64)))
        self._upload_tables()
        # the schedule kernels run on a side stream: they must not start before these copies land
        self._rng_push_event = torch.cuda.Event()
        self._rng_push_event.record(torch.cuda.current_stream(self.device))
        self._rng_on_device = True

    # ---- asynchronous host schedule stream (schedule="host-async") ------------------------------
    def _close_sched_stream(self) -> None:
        """Stop the producer thread and bring self._rng to exactly what has been consumed."""
        if self._sstream is None:
            return
        self._L.ml2048_sched_stream_destroy(self._sstream)
        self._sstream = None
        # launches that still read a slot in place: the memory is reused by the next stream
        for slot in self._ss_held:
            if slot[1] is not None:
                slot[1].synchronize()
        self._ss_held.clear()
        self._ss_slots = None
        self._la = None  # a schedule taken ahead of its rollout is simply dropped with the stream
        after = self._sstream_after
        self._sstream_after = None
        if after is not None:
            c, rand_step, perm, flt = after
            bg = self._rng.rand.bit_generator
            st = bg.state
            st["state"]["state"] = (int(c.state_hi) << 64) | int(c.state_lo)
            st["has_uint32"] = int(c.has_uint32)
            st["uinteger"] = int(c.uinteger)
            bg.state = st
            self._rng.rand_step = int(rand_step)
            self._rng.randperm[...] = perm
            self._rng.randfloat[...] = flt
            self._tables_dirty = True

    def _pack_pinned(self, sched, epochs, masks, N: int):
        """Pack [schedule | masks | epoch tables] (16-byte aligned sections: the layout of a schedule-stream slot) into
        the next pinned buffer of the ring.  Returns (slot, layout): slot = [pinned tensor, event]; layout = (E, o_mask,
        o_tab, nbytes)."""
        E = int(epochs.shape[0])
        o_mask = ((N * 12 + 15) // 16) * 16
        o_tab = o_mask + (((N + 1) * 4 + 15) // 16) * 16
        nbytes = o_tab + E * RAND_SIZE * 16
        i = self._spin_i
        self._spin_i = (i + 1) % len(self._spin)
        slot = self._spin[i]
        if slot is not None and slot[1] is not None:
            slot[1].synchronize()  # the launch that read this buffer (a ring length ago) has finished
        if slot is None or slot[0].numel() < nbytes:
            slot = [torch.empty((max(2 * nbytes, 1 << 19),), dtype=torch.uint8, pin_memory=True), None]
            self._spin[i] = slot
        packed = slot[0].numpy()
        packed[:N * 12] = sched.view(np.uint8).reshape(-1)
        packed[N * 12:o_mask] = 0
        packed[o_mask:o_mask + E * 4] = masks.view(np.uint8)
        packed[o_mask + E * 4:o_tab] = 0
        packed[o_tab:nbytes] = epochs.reshape(-1)
        return slot, (E, o_mask, o_tab, nbytes)

    def _dev_sched_buf(self, idx: int, nbytes: int) -> torch.Tensor:
        buf = self._sdev[idx]
        if buf is None or buf.numel() < nbytes:
            # only ever touched by launches on the current stream, so the caching allocator's
            # stream-ordered reuse of the old block is safe
            buf = torch.empty((max(2 * nbytes, 1 << 19),), dtype=torch.uint8, device=self.device)
            self._sdev[idx] = buf
        return buf

    def _open_sched_stream(self, N: int) -> None:
        L = self._L
        if self._sstream is not None and self._sstream_n != N:
            self._close_sched_stream()
        if self._sstream is None:
            self._drop_prefetch()
            st = self._rng.rand.bit_generator.state
            s, inc = st["state"]["state"], st["state"]["inc"]
            m64 = (1 << 64) - 1
            c = _lib.Pcg64State(s >> 64, s & m64, inc >> 64, inc & m64, int(st["has_uint32"]), int(st["uinteger"]))
            depth = int(os.environ.get("ML2048_STREAM_DEPTH", "8"))
            sb = (int(L.ml2048_sched_stream_slot_bytes(N)) + 63) // 64 * 64
            pinned = depth >= 4 and depth * sb <= (256 << 20) and os.environ.get("ML2048_STREAM_PINNED", "1") != "0"
            if pinned:
                # the stream's slots ARE the pinned staging buffers: the producer threads write a schedule where the
                # upload kernels read it, and handing it over copies nothing (ml2048_sched_stream_take_slot)
                if self._ss_mem is None or self._ss_mem.numel() < depth * sb:
                    self._ss_mem = torch.empty((depth * sb,), dtype=torch.uint8, pin_memory=True)
                mem = self._ss_mem
                ptrs = (C.c_void_p * depth)(*[mem.data_ptr() + k * sb for k in range(depth)])
                self._sstream = L.ml2048_sched_stream_create_pinned(
                    C.byref(c), self._rng.randperm.ctypes.data, self._rng.randfloat.ctypes.data, self._rng.rand_step, N,
                    self._two_prob, depth, ptrs, sb)
                self._ss_slots = [[mem[k * sb:(k + 1) * sb], None, np.empty((RAND_SIZE, 16), np.uint8),
                                   np.empty((RAND_SIZE,), np.float32)] for k in range(depth)]
                # slots the launch thread may hold (= rollouts it may run ahead of the GPU); the rest is the producer's room
                self._ss_hold = min(depth - 1, max(2, int(os.environ.get("ML2048_STREAM_HOLD", str(depth - 3)))))
                self._ss_taken = 0
            else:
                self._sstream = L.ml2048_sched_stream_create(C.byref(c), self._rng.randperm.ctypes.data,
                                                             self._rng.randfloat.ctypes.data, self._rng.rand_step, N,
                                                             self._two_prob, depth)
                self._ss_slots = None
            if not self._sstream:
                raise _lib.Ml2048Error("ml2048_sched_stream_create failed")
            self._sstream_n = N

    def _take_schedule(self, N: int):
        """The next schedule of the stream.  Returns (slot, (E, o_mask, o_tab, nbytes), after): slot[0] = the pinned
        buffer that holds it as [schedule | masks | tables], slot[1] = the event of the launch that reads it (set by the
        caller).  Pinned-slot streams hand the slot itself over (nothing copied but the 20 KB of after-state) and take
        it back, oldest first, once that event has passed; otherwise ONE native call copies the schedule into the next
        buffer of the staging ring."""
        L = self._L
        self._open_sched_stream(N)
        ne, rs = C.c_int32(0), C.c_int32(0)
        o_m, o_t, nb = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        after = _lib.Pcg64State()
        if self._ss_slots is not None:
            held = self._ss_held
            while held and (len(held) >= self._ss_hold or (held[0][1] is not None and held[0][1].query())):
                if held[0][1] is not None:
                    held[0][1].synchronize()
                held.popleft()[1] = None
                _lib.check(L.ml2048_sched_stream_release(self._sstream), "sched_stream_release")
            # slots are handed out round-robin; the after-state buffers belong to the slot INDEX (rewritten depth
            # schedules later, when this schedule has long been executed)
            slot = self._ss_slots[self._ss_taken % len(self._ss_slots)]
            k = C.c_int32(-1)
            rc = L.ml2048_sched_stream_take_slot(self._sstream, C.byref(k), C.byref(ne), C.byref(o_m), C.byref(o_t),
                                                 C.byref(nb), C.byref(after), C.byref(rs), slot[2].ctypes.data,
                                                 slot[3].ctypes.data)
            _lib.check(rc, "sched_stream_take_slot")
            assert self._ss_slots[k.value] is slot, (k.value, self._ss_taken)
            self._ss_taken += 1
            held.append(slot)
            return slot, (int(ne.value), int(o_m.value), int(o_t.value), int(nb.value)), (after, rs.value, slot[2], slot[3])
        i = self._spin_i
        self._spin_i = (i + 1) % len(self._spin)
        slot = self._spin[i]
        if slot is not None and slot[1] is not None:
            slot[1].synchronize()  # the launch that read this buffer (a ring length ago) has finished
        if slot is None:
            slot = [torch.empty((1 << 19,), dtype=torch.uint8, pin_memory=True), None]
            self._spin[i] = slot
        if len(slot) < 4:  # the generator / tables after this schedule, in buffers that live with the slot
            slot += [np.empty((RAND_SIZE, 16), np.uint8), np.empty((RAND_SIZE,), np.float32)]
        while True:
            rc = L.ml2048_sched_stream_take(self._sstream, slot[0].data_ptr(), slot[0].numel(), C.byref(ne), C.byref(o_m),
                                            C.byref(o_t), C.byref(nb), C.byref(after), C.byref(rs),
                                            slot[2].ctypes.data, slot[3].ctypes.data)
            if rc == -3 and nb.value > slot[0].numel():  # unusually many epochs: a bigger pinned buffer
                slot[0] = torch.empty((2 * nb.value,), dtype=torch.uint8, pin_memory=True)
                continue
            _lib.check(rc, "sched_stream_take")
            break
        return slot, (int(ne.value), int(o_m.value), int(o_t.value), int(nb.value)), (after, rs.value, slot[2], slot[3])

    def _pull_rng(self, sync_lookahead: bool = True) -> None:
        """Make the host generator current again (before any host-side draw or inspection)."""
        if sync_lookahead and self._rng._la is not None:
            # draws taken ahead by step_prepare_torch: back to what was consumed; the per-step API's device tables were
            # not kept up to date meanwhile (the fused kernel reads the lookahead block's own device copy)
            self._rng.sync()
            self._tables_dirty = True
        self._close_sched_stream()
        if not self._rng_on_device:
            return
        words = self._devrng.cpu().numpy().view(np.uint64)
        err = int(words[5] >> 32)
        if err:
            raise _lib.Ml2048Error(f"device schedule failed (code {err}): epoch buffer or word window too small")
        bg = self._rng.rand.bit_generator
        st = bg.state
        st["state"]["state"] = (int(words[0]) << 64) | int(words[1])
        st["has_uint32"] = int(words[4] & 0xFFFFFFFF)
        st["uinteger"] = int(words[4] >> 32)
        bg.state = st
        self._rng.rand_step = int(words[5] & 0xFFFFFFFF)
        self._rng.randperm[...] = self._randperm_dev.cpu().numpy()
        self._rng.randfloat[...] = self._randfloat_dev.cpu().numpy()
        self._tables_dirty = False
        self._rng_on_device = False

    def _upload_tables(self) -> None:
        self._randperm_dev.copy_(torch.from_numpy(self._rng.randperm))
        self._randfloat_dev.copy_(torch.from_numpy(self._rng.randfloat))
        self._tables_dirty = False

    # reference-visible private attributes
    @property
    def _rand_step(self) -> int:
        self._drop_prefetch()
        self._pull_rng()
        return self._rng.rand_step

    @property
    def _randperm(self) -> np.ndarray:
        self._pull_rng()
        return self._rng.randperm

    @property
    def _randfloat(self) -> np.ndarray:
        self._pull_rng()
        return self._rng.randfloat

    @property
    def _game_count(self) -> int:
        self.join_ids()
        return int(self._game_count_dev.item())

    @property
    def _prev_state(self) -> np.ndarray:
        return self.prev_state.cpu().numpy()

    @property
    def _prev_valid_actions(self) -> np.ndarray:
        return self.prev_valid_actions.cpu().numpy()

    @property
    def _data(self) -> np.ndarray:
        """Snapshot of the game records in the reference's 64-byte layout (`VecGame._data`,
        game_numba.py:571), e.g. for `game._data[slot]["id"]` (src/ml2048/replay.py:147-151)."""
        return self.records_torch().cpu().numpy().view(RECORD_DTYPE).reshape(self._size)

    def records_torch(self) -> torch.Tensor:
        rec = torch.empty((self._size, 64), dtype=torch.uint8, device=self.device)
        with self._dev_ctx():
            _lib.check(self._L.ml2048_pack_records(self._stream(), self._size, self._st(), rec.data_ptr()), "pack_records")
        return rec

    def load_records(self, records: np.ndarray) -> None:
        """Overwrite the device state from reference-layout records (tests / restoring a run)."""
        rec = np.ascontiguousarray(records).view(np.uint8).reshape(self._size, 64)
        self._serial += 1
        self._prepared = False
        dev = torch.from_numpy(rec).to(self.device)
        with self._dev_ctx():
            _lib.check(self._L.ml2048_unpack_records(self._stream(), self._size, self._st(), dev.data_ptr()), "unpack_records")

    # ------------------------------------------------------------------------------------------
    # ---- schedule prefetch (fused rollouts) --------------------------------------------------
    # A worker thread may draw the NEXT rollout's host schedule on a CLONE of the host RNG while
    # the GPU runs the current one (the native draw releases the GIL).  The clone replaces
    # self._rng only when that schedule is actually consumed, so dropping it is always safe.
    def _drop_prefetch(self) -> None:
        fut = getattr(self, "_prefetch", None)
        if fut is not None:
            fut[1].result()  # let the worker finish; its clone is simply discarded
            self._prefetch = None

    def _start_prefetch(self, n: int) -> None:
        import concurrent.futures as cf

        if getattr(self, "_pool", None) is None:
            self._pool = cf.ThreadPoolExecutor(max_workers=1, thread_name_prefix="ml2048-hostrng")
        clone = self._rng.clone()
        two_prob = self._two_prob

        def work():
            return clone, clone.draw_schedule(n, two_prob)

        self._prefetch = (n, self._pool.submit(work))

    def _next_schedule(self, n: int):
        fut = getattr(self, "_prefetch", None)
        self._prefetch = None
        if fut is not None:
            clone, drawn = fut[1].result()
            if fut[0] == n:
                self._rng = clone
                return drawn
        return self._rng.draw_schedule(n, self._two_prob)

    def reset(self, seed: Optional[int] = None) -> None:
        """game_numba.py:606-617.  `_game_count` is NOT reset (only the constructor sets it)."""
        self._drop_prefetch()
        if getattr(self, "_sstream", None) is not None:
            self._sstream_after = None  # a reset discards whatever was drawn ahead
            self._close_sched_stream()
        self._rng_on_device = False
        self._rng = HostRng(seed)
        self._fast_counter = 0
        self._serial += 1
        self._prepared = False
        self._upload_tables()
        with self._dev_ctx():
            _lib.check(self._L.ml2048_reset(self._stream(), self._size, self._st()), "reset")

    def observations_torch(self) -> tuple[torch.Tensor, torch.Tensor]:
        """(board (M,16) uint8, valid_actions (M,4) uint8): live device tensors (aliases)."""
        return self.board, self.valid_actions

    def observations(self) -> tuple[np.ndarray, np.ndarray]:
        """game_numba.py:586-587, as host snapshots."""
        return self.board.cpu().numpy(), self.valid_actions.cpu().numpy()

    def prepare_torch(self) -> tuple[torch.Tensor, torch.Tensor]:
        """prepare() without a host sync: (reset_indices buffer (M,) int64, n_reset (1,) int32);
        the first n_reset entries of the buffer are the ascending re-dealt slots."""
        L, M = self._L, self._size
        self._drop_prefetch()
        self._pull_rng()
        with self._dev_ctx():
            if self._spawn == "table":
                regen, row_base = self._rng.draw_prepare()  # :622-626, drawn unconditionally
                if regen or self._tables_dirty:
                    self._upload_tables()
                rc = L.ml2048_prepare(
                    self._stream(), M, self._idx_base, self._st(), self._randperm_dev.data_ptr(),
                    self._randfloat_dev.data_ptr(), self._two_prob, row_base, self._game_count_dev.data_ptr(),
                    self._n_reset_dev.data_ptr(), self._reset_indices.data_ptr(), self._scratch.data_ptr())
            else:
                rc = L.ml2048_prepare_fast(
                    self._stream(), M, self._idx_base, self._st(), self._two_prob, self._spawn_seed,
                    self._fast_counter, self._game_count_dev.data_ptr(), self._n_reset_dev.data_ptr(),
                    self._reset_indices.data_ptr(), self._scratch.data_ptr())
        _lib.check(rc, "prepare")
        self._serial += 1
        self._prepared = True
        return self._reset_indices, self._n_reset_dev

    def prepare(self) -> tuple[np.ndarray]:
        """game_numba.py:619-658: re-deal terminated games; returns (ascending indices,)."""
        idx, n = self.prepare_torch()
        k = int(n.item())
        return (idx[:k].cpu().numpy(),)

    def _as_device_actions(self, actions) -> torch.Tensor:
        if isinstance(actions, np.ndarray):
            assert actions.shape == (self._size,), actions.shape  # :668
            if actions.dtype not in (np.uint8, np.int32, np.int64):
                actions = actions.astype(np.int64)
            actions = torch.from_numpy(np.ascontiguousarray(actions))
        assert tuple(actions.shape) == (self._size,), tuple(actions.shape)
        if actions.dtype not in _ACT_CODE:
            actions = actions.to(torch.int64)
        if actions.device != self.device:
            actions = actions.to(self.device)
        return actions.contiguous()

    def step_torch(self, actions) -> VecStepResult:
        """step() on device tensors; the result aliases live state (like the reference's views)."""
        a = self._as_device_actions(actions)
        L, M = self._L, self._size
        self._drop_prefetch()
        self._pull_rng()
        with self._dev_ctx():
            if self._spawn == "table":
                if self._tables_dirty:
                    self._upload_tables()
                row_base = self._rng.draw_step()  # :670
                rc = self._step_fn(self._stream(), M, self._idx_base, self._st(), a.data_ptr(), _ACT_CODE[a.dtype],
                                   self._reward_kind, self._two_prob, row_base, self._randperm_dev.data_ptr(),
                                   self._randfloat_dev.data_ptr())
            else:
                rc = L.ml2048_step_fast(self._stream(), M, self._idx_base, self._st(), a.data_ptr(),
                                        _ACT_CODE[a.dtype], self._reward_kind, self._two_prob, self._spawn_seed,
                                        self._fast_counter)
                self._fast_counter += 1
        _lib.check(rc, "step")
        self._serial += 1
        self._prepared = False
        return {
            "state": self.board, "valid_actions": self.valid_actions, "merged": self.merged,
            "step": self.step_count, "reward": self.reward, "score": self.score,
            "terminated": self.terminated, "invalid": self.invalid,
            "prev_state": self.prev_state, "prev_valid_actions": self.prev_valid_actions,
        }

    def step(self, actions) -> VecStepResult:
        """game_numba.py:660-698, returning host snapshots of the ten result arrays."""
        res = self.step_torch(actions)
        return {k: v.cpu().numpy() for k, v in res.items()}  # type: ignore[return-value]

    # ------------------------------------------------------------------------------------------
    STEP_OUT_SPEC = {  # ml2048_step_out: name -> (trailing shape, accepted dtypes)
        "state": ((16,), (torch.uint8, torch.int8)), "valid_actions": ((4,), (torch.uint8, torch.bool)),
        "merged": ((16,), (torch.uint8,)), "step": ((), (torch.int32,)), "reward": ((), (torch.float32,)),
        "score": ((), (torch.float32,)), "terminated": ((), (torch.uint8, torch.bool)),
        "invalid": ((), (torch.uint8, torch.bool)), "id": ((), (torch.int32,)),
        "prev_state": ((16,), (torch.uint8, torch.int8)), "prev_valid_actions": ((4,), (torch.uint8, torch.bool)),
    }

    def new_step_out(self, names=None) -> dict:
        """Buffers for `step_prepare_torch(out=...)`: the reference's VecStepResult arrays (+ "id")."""
        names = list(self.STEP_OUT_SPEC) if names is None else list(names)
        return {n: torch.empty((self._size, *self.STEP_OUT_SPEC[n][0]), dtype=self.STEP_OUT_SPEC[n][1][0],
                               device=self.device) for n in names}

    def _sp_args(self):
        A = getattr(self, "_sp", None)
        if A is None:
            L = self._L
            A = _lib.StepPrepareArgs()
            A.M, A.idx_base, A.st = self._size, self._idx_base, self._state
            A.reward_kind, A.two_prob = self._reward_kind, self._two_prob
            A.spawn_seed = self._spawn_seed
            A.game_count, A.n_reset = self._game_count_dev.data_ptr(), self._n_reset_dev.data_ptr()
            A.reset_indices = self._reset_indices.data_ptr()
            nbytes = int(L.ml2048_step_prepare_scratch_bytes())
            self._sp_scratch = torch.zeros((nbytes // 8,), dtype=torch.int64, device=self.device)  # zeroed once
            A.scratch = self._sp_scratch.data_ptr()
            self._sp = A
            self._sp_ref = C.byref(A)
            self._sp_out = None      # the `out` dict whose pointers are in A.out
            self._sp_obs = (None, None, None)  # (onehot, board_i64, hist) whose pointers are in A
            self._sp_next = [None, None]  # ping-pong (actions, logp) buffers of the fused sampler
            self._sp_k = 0
            self._sp_sampling = False
            self._sp_block = None    # the lookahead block (HostRng.next_step_prepare) whose tables are on the device
            self._sp_block_dev = None
            self._sp_block_ptr = 0
            self._sp_stage = [None, None, None]  # pinned staging ring: [buffer, event of the copy kernel that read it]
            self._sp_stage_i = 0
        return A

    def _sp_upload_block(self, blk) -> None:
        """The epoch tables of a lookahead block (HostRng.next_step_prepare) -> one device array, by the copy kernel
        that reads pinned host memory (ml2048_upload): no hipMemcpyAsync (which can stall the host for milliseconds
        behind queued kernels) and no host-device synchronisation inside a step_prepare loop."""
        tabs = blk[2]  # uint8[E, 1024, 16]
        nbytes = int(tabs.size)
        ring = self._sp_stage
        i = self._sp_stage_i
        self._sp_stage_i = (i + 1) % len(ring)
        slot = ring[i]
        if slot is not None and slot[1] is not None:
            slot[1].synchronize()  # the copy kernel that read this staging buffer (three blocks ago) has finished
        if slot is None or slot[0].numel() < nbytes:
            slot = [torch.empty((max(nbytes, 48 * RAND_SIZE * 16),), dtype=torch.uint8, pin_memory=True), None]
            ring[i] = slot
        slot[0].numpy()[:nbytes] = tabs.reshape(-1)
        # the previous block's device array goes back to the caching allocator, whose reuse is ordered on this stream
        self._sp_block_dev = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.ml2048_upload(self._stream(), self._sp_block_dev.data_ptr(), slot[0].data_ptr(), nbytes, nbytes),
                   "upload")
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.device))
        self._sp_block_ptr = self._sp_block_dev.data_ptr()
        self._sp_block = blk

    def step_prepare_torch(self, actions, *, out: Optional[dict] = None, onehot: Optional[torch.Tensor] = None,
                           board_i64: Optional[torch.Tensor] = None, sample=None,
                           hist: Optional[torch.Tensor] = None):
        """step(actions) and the following prepare() as ONE launch (`ml2048_step_prepare`): same draws, same
        state, ids and index list as `step_torch(actions); prepare_torch()`.  Afterwards the live tensors
        (`board`, `valid_actions`, ...) show the re-dealt games, i.e. what the next policy call observes.

        out:       dict of device tensors keyed like VecStepResult (+ "id"), see `new_step_out()`: what step()
                   returned, as of before the re-deal (the reference's result views).  May hold slices of replay
                   buffers (int8 / bool views of the same bytes are accepted).
        onehot:    (M,16,4,4) float32 tensor to receive the observation of the post-prepare boards;
        board_i64: (M,16) int64 likewise.
        sample:    a RandomPolicy: its draw for the NEXT step on the post-prepare masks comes out of the same
                   launch (the policy's call counter advances as if sample_actions_u8 had been called).
        hist:      int32[20] accumulator of the max tile of the games this step finished (RunnerStats).
        Returns {"n_reset": (1,) int32, "reset_indices": (M,) int64 buffer, "actions": uint8 (M,) | None,
        "log_probs": float32 (M,) | None} - all device tensors, no host sync.
        """
        M, dev = self._size, self.device
        if torch.is_tensor(actions) and actions.dtype == torch.uint8 and actions.device == dev and \
                actions.ndim == 1 and actions.shape[0] == M and actions.is_contiguous():
            a = actions  # the common case (the fused sampler's own output): no conversions
        else:
            a = self._as_device_actions(actions)
        if getattr(self, "_prefetch", None) is not None:
            self._drop_prefetch()
        if self._sstream is not None or self._rng_on_device:
            self._pull_rng(sync_lookahead=False)
        A = self._sp_args()
        A.actions, A.action_dtype = a.data_ptr(), _ACT_CODE[a.dtype]
        with self._dev_ctx():
            if self._spawn == "table":
                # step(t)'s draw (:670), then the following prepare()'s (:622-626), from the block of draws taken
                # ahead; the block's epoch tables live on the device as one array, so a regeneration is a pointer change
                A.row_base_step, A.row_base_prepare, e_s, e_p, blk = self._rng.next_step_prepare(self._two_prob)
                if blk is not self._sp_block:
                    self._sp_upload_block(blk)
                masks = blk[7]
                A.randperm_step = self._sp_block_ptr + e_s * (RAND_SIZE * 16)
                A.two_mask_step = masks[e_s]
                A.randperm_prepare = self._sp_block_ptr + e_p * (RAND_SIZE * 16)
                A.two_mask_prepare = masks[e_p]
            else:
                A.randperm_step = None
                A.spawn_counter_step = self._fast_counter  # call counters (board_ops.hpp spawn_word)
                self._fast_counter += 1
                A.spawn_counter_prepare = self._fast_counter
            if out is not self._sp_out:
                for name, (shape, dts) in self.STEP_OUT_SPEC.items():
                    buf = None if out is None else out.get(name)
                    if buf is not None:
                        assert buf.is_contiguous() and buf.dtype in dts and buf.device == dev and \
                            tuple(buf.shape) == (M, *shape), (name, buf.dtype, tuple(buf.shape))
                    setattr(A.out, name, None if buf is None else buf.data_ptr())
                self._sp_out = out
            if onehot is not self._sp_obs[0] or board_i64 is not self._sp_obs[1] or hist is not self._sp_obs[2]:
                if onehot is not None:
                    assert onehot.shape == (M, 16, 4, 4) and onehot.dtype == torch.float32 and onehot.is_contiguous()
                if board_i64 is not None:
                    assert board_i64.shape == (M, 16) and board_i64.dtype == torch.int64 and board_i64.is_contiguous()
                A.onehot = None if onehot is None else onehot.data_ptr()
                A.board_i64 = None if board_i64 is None else board_i64.data_ptr()
                A.hist_counts = None if hist is None else hist.data_ptr()
                self._sp_obs = (onehot, board_i64, hist)
            nxt = _NO_NEXT
            if sample is not None:
                k = self._sp_k
                self._sp_k = k ^ 1
                nxt = self._sp_next[k]  # two sets: this launch writes set k while it reads set k^1 as `actions`
                if nxt is None:
                    nxt = (torch.empty((M,), dtype=torch.uint8, device=dev), torch.empty((M,), dtype=torch.float32, device=dev))
                    self._sp_next[k] = nxt
                    assert sample.idx_base == self._idx_base
                A.next_actions, A.next_logp = nxt[0].data_ptr(), nxt[1].data_ptr()
                A.act_seed, A.act_counter, A.act_masked = sample.seed, sample.counter, int(sample.masked)
                sample.counter += 1
                self._sp_sampling = True
            elif self._sp_sampling:
                A.next_actions = A.next_logp = None
                self._sp_sampling = False
            rc = self._L.ml2048_step_prepare(self._stream(), self._sp_ref)
        if rc:
            _lib.check(rc, "step_prepare")
        self._serial += 1
        self._prepared = True
        return {"n_reset": self._n_reset_dev, "reset_indices": self._reset_indices, "actions": nxt[0], "log_probs": nxt[1]}

    # ------------------------------------------------------------------------------------------
    def onehot(self, out: Optional[torch.Tensor] = None, *, with_long: bool = False, check: bool = False):
        """(M,16,4,4) float32 observation, what `CNNEncoder.forward` builds from the (M,16) long
        state (src/ml2048/policy/_network.py:91-95,105).  with_long: also the (M,16) int64 tensor.
        check: raise like F.one_hot would when a tile exponent >= 16 is present (H10)."""
        M = self._size
        if out is None:
            out = torch.empty((M, 16, 4, 4), dtype=torch.float32, device=self.device)
        assert out.shape == (M, 16, 4, 4) and out.dtype == torch.float32 and out.is_contiguous()
        long = torch.empty((M, 16), dtype=torch.int64, device=self.device) if with_long else None
        bad = torch.zeros((1,), dtype=torch.int32, device=self.device) if check else None
        with self._dev_ctx():
            _lib.check(self._L.ml2048_encode_onehot(
                self._stream(), M, self.board.data_ptr(), out.data_ptr(),
                long.data_ptr() if long is not None else None, bad.data_ptr() if bad is not None else None), "onehot")
        if check and int(bad.item()):
            raise RuntimeError("Class values must be smaller than num_classes.")  # F.one_hot's message
        return (out, long) if with_long else out

    def histogram(self, terminated_only: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """counts[max tile exponent] over all boards (or the terminated ones) added into `out`."""
        if out is None:
            out = torch.zeros((20,), dtype=torch.int32, device=self.device)
        with self._dev_ctx():
            _lib.check(self._L.ml2048_hist_maxcell(
                self._stream(), self._size, self.board.data_ptr(),
                self.terminated.data_ptr() if terminated_only else None, out.data_ptr()), "hist")
        return out

    def summary(self) -> list[Any]:
        """game_numba.py:593-604."""
        counts = self.histogram().cpu().numpy()
        total = int(counts.sum())
        entries = [(2 ** k, int(c), int(c) / total) for k, c in enumerate(counts) if c]
        entries.sort(key=lambda s: s[0], reverse=True)
        return entries

    def digest(self, with_id: bool = True) -> int:
        out = torch.zeros((1,), dtype=torch.int64, device=self.device)
        with self._dev_ctx():
            _lib.check(self._L.ml2048_digest(self._stream(), self._size, self._idx_base, self._st(), int(with_id),
                                             out.data_ptr()), "digest")
        return int(out.item()) & 0xFFFFFFFFFFFFFFFF

    # ------------------------------------------------------------------------------------------
    def _device_schedule(self, N: int, expanded: bool = False):
        """Enqueue the GPU computation of the next N-step draw schedule (csrc/devrng2.hip) and return
        device views (perm_epochs, schedule, two_masks, max_epochs).  The schedule kernels run on a
        side stream, one slot ahead is possible: the rollout stream only waits for its own slot.
        expanded: epoch tables as 48-byte expanded rows (ml2048_hip.h), for ml2048_rollout_chain."""
        L, dev = self._L, self.device
        if self._devrng_ws is None:
            # default: the data-parallel generation (csrc/devrng2.hip); ML2048_DEVRNG=1 selects the original
            # single-workgroup kernel (csrc/devrng.hip) for comparison
            self._devrng_v1 = os.environ.get("ML2048_DEVRNG", "2") == "1"
            nbytes = int(L.ml2048_devrng_workspace_bytes() if self._devrng_v1 else L.ml2048_devrng2_workspace_bytes())
            self._devrng_ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            with self._dev_ctx():
                init = L.ml2048_devrng_init if self._devrng_v1 else L.ml2048_devrng2_init
                _lib.check(init(self._stream(), self._devrng_ws.data_ptr()), "devrng_init")
        self._push_rng()
        cap = N + 1  # one epoch per step at most: can never overflow
        if self._sched_slots is None or self._sched_slots[0] != N:
            if self._sched_slots is not None and self._sched_stream is not None:
                for old in self._sched_slots[1]:  # written on the side stream: reuse only after it is done
                    for key in ("perm", "sched", "mask", "ne"):
                        old[key].record_stream(self._sched_stream)
            slots = []
            for _ in range(3):  # a ring of three: generation k+1 must not wait for the rollout that read slot k-1
                # (with two, each cross-stream hand-over - about 20 us - sat on the critical path twice per period)
                slots.append({
                    "perm": torch.empty((cap, RAND_SIZE, 48), dtype=torch.uint8, device=dev),  # room for the expanded form
                    "sched": torch.empty((N, 3), dtype=torch.int32, device=dev),
                    "mask": torch.empty((cap,), dtype=torch.int32, device=dev),
                    "ne": torch.zeros((1,), dtype=torch.int32, device=dev),
                    "free": None,  # event: the rollout that consumed this slot has finished
                })
            self._sched_slots = (N, slots, 0)
            if self._sched_stream is None:
                self._sched_stream = torch.cuda.Stream(device=dev)
            self._sched_stream.wait_stream(torch.cuda.current_stream(dev))  # workspace init, fresh buffers
        _, slots, k = self._sched_slots
        slot = slots[k]
        self._sched_slots = (N, slots, (k + 1) % len(slots))
        main = torch.cuda.current_stream(dev)
        side = self._sched_stream
        if slot["free"] is not None:
            side.wait_event(slot["free"])  # its previous consumer (three rollouts ago) has finished
        push = getattr(self, "_rng_push_event", None)
        if push is not None:
            side.wait_event(push)
            self._rng_push_event = None
        with self._dev_ctx():
            args = (side.cuda_stream, self._devrng_ws.data_ptr(), self._devrng.data_ptr(), self._randperm_dev.data_ptr(),
                    self._randfloat_dev.data_ptr(), N, self._two_prob, slot["sched"].data_ptr(), slot["perm"].data_ptr(),
                    slot["mask"].data_ptr(), cap, slot["ne"].data_ptr())
            if self._devrng_v1:
                expanded = False
                _lib.check(L.ml2048_devrng_schedule(*args), "devrng_schedule")
            else:
                _lib.check(L.ml2048_devrng2_schedule(*args, int(expanded)), "devrng2_schedule")
        ready = torch.cuda.Event()
        ready.record(side)
        main.wait_event(ready)
        self._sched_pending_slot = slot
        perm = slot["perm"] if expanded else slot["perm"].view(-1)[:cap * RAND_SIZE * 16].view(cap, RAND_SIZE, 16)
        return perm, slot["sched"], slot["mask"], cap, expanded

    def rollout(self, n_steps: int, *, masked: bool = True, act_seed: int = 0, act_counter: int = 0,
                record_actions: bool = False, digests: bool = False, finish_ids=True,
                kernel_events: Optional[tuple] = None, prefetch_next: bool = False,
                trajectory: Optional[dict] = None, kernel_timer=None):
        """N x {prepare; uniform-random actions; step} fused into one kernel launch, bit-exact
        with the per-step API on the same seeds (both spawn modes; spawn="rocrand" needs no host
        schedule and is purely GPU-bound).

        Returns a dict: actions (N,M) uint8 | None, digests (N,) int64 (two's complement of the
        uint64) | None, step_resets (N,) int32 | None.
        kernel_events: optional (start, end) torch.cuda.Event pair recorded on the launch stream
        immediately around the rollout kernel (bench.py times the dominant kernel with it).
        kernel_timer: a handle from ml2048_timer_create; its event pair is attached to the rollout kernel's
        dispatch itself (exact kernel duration, no marker packets; read it with ml2048_timer_elapsed_ms).
        finish_ids: True = number the re-dealt games (ordered ids, game counter, step_resets) on the launch
        stream right after the kernel; "deferred" = the same scan on a side stream, overlapped with whatever
        the launch stream does next (every method of this class waits for it before it touches ids; read
        `step_resets` from your own torch code only after join_ids()); False = skip it.
        prefetch_next: draw the host schedule of the next rollout(n_steps) on a worker thread while
        this one runs on the GPU (results are identical either way).
        trajectory: dict of [N, M, ...] device tensors keyed like REPLAY_SPEC (train_ops.new_buffers((N, M),
        ...)); every step's transition is written into them by the same launch (table spawn mode).
        """
        N, M, dev = int(n_steps), self._size, self.device
        if N <= 0:
            raise ValueError(f"n_steps={n_steps}")
        L = self._L
        self._serial += 1
        self._prepared = False
        if self._rng._la is not None:  # draws taken ahead by step_prepare_torch: back to what was consumed
            self._rng.sync()
            self._tables_dirty = True
        fast = self._spawn != "table"
        on_device = (not fast) and self._schedule_mode == "device"
        dev_expanded = False
        cur_inv = False  # the host schedule's tables are on the device as expanded rows
        if on_device:
            plain = trajectory is None and not record_actions and not digests
            perm_dev, sched_dev, mask_dev, E, dev_expanded = self._device_schedule(N, expanded=plain)
        elif not fast:
            # The host schedule does not depend on game state: it is packed as [epoch tables | schedule |
            # masks] in pinned memory and brought to the device by a kernel on THIS stream - either a
            # small copy kernel in front of the rollout, or (host-async, plain rollouts) by the previous
            # rollout's own launch (ml2048_rollout_chain), which leaves nothing between two rollout kernels.
            asyn = self._schedule_mode == "host-async"
            plain = trajectory is None and not record_actions and not digests
            # The pipelined form (host-async, plain rollouts) keeps EXPANDED tables on the device: every
            # 16-byte permutation row as 48 bytes of nibbles + dot-product weights (what the upload kernels
            # produce), for the eight-instruction spawn of ml2048_rollout_chain.  A schedule taken ahead for another length,
            # or in front of a rollout with optional outputs (which reads the plain layout), is dropped with
            # the stream: the generator is re-synchronised to what was executed and the schedule redrawn.
            if asyn and self._la is not None and (self._la["N"] != N or not plain):
                self._close_sched_stream()
            cur = self._la if asyn else None
            self._la = None
            staged = []  # pinned slots read by this rollout's launches: released by its event
            # (round 3: a trajectory-writing rollout reads expanded tables too - ml2048_rollout_traj_expanded; only the
            # chained upload of the NEXT schedule stays with the pipelined plain form)
            expand = (asyn and plain) or (trajectory is not None and os.environ.get("ML2048_TRAJ_EXPANDED", "1") != "0")
            chain_next = asyn and plain

            def dev_layout(E, o_mask, o_tab, nbytes):
                # device copy: [schedule | masks | tables], the tables as 48-byte expanded rows when `expand`
                return {"E": E, "o_mask": o_mask, "o_tab": o_tab, "host_nbytes": nbytes,
                        "nbytes": nbytes + (2 * (nbytes - o_tab) if expand else 0),
                        "tab": o_tab if expand else nbytes, "inv": expand}

            with self._dev_ctx():
                if cur is None:
                    if asyn:
                        pslot, lay, after_state = self._take_schedule(N)
                    else:
                        sched, epochs, masks = self._next_schedule(N)
                        after_state = None
                        if prefetch_next:
                            self._start_prefetch(N)  # the NEXT schedule, on a worker thread (sync mode)
                        pslot, lay = self._pack_pinned(sched, epochs, masks, N)
                    cur = dev_layout(*lay)
                    self._sdev_i ^= 1
                    dbuf = self._dev_sched_buf(self._sdev_i, cur["nbytes"])
                    _lib.check(L.ml2048_upload(self._stream(join=False), dbuf.data_ptr(), pslot[0].data_ptr(),
                                               cur["host_nbytes"], cur["tab"]), "upload")
                    staged.append(pslot)
                    cur.update({"N": N, "idx": self._sdev_i, "after": after_state})
                E, o_mask, o_tab = cur["E"], cur["o_mask"], cur["o_tab"]
                packed_dev = self._sdev[cur["idx"]]
                chain = None
                if chain_next:
                    # take the next schedule now and let this launch upload (and expand) it
                    nslot, lay2, after_next = self._take_schedule(N)
                    nxt = dev_layout(*lay2)
                    ndev = self._dev_sched_buf(cur["idx"] ^ 1, nxt["nbytes"])
                    chain = (nslot, ndev, nxt["host_nbytes"], nxt["tab"])
                    staged.append(nslot)
                    nxt.update({"N": N, "idx": cur["idx"] ^ 1, "after": after_next})
                    self._la = nxt
            if asyn:
                self._sstream_after = cur["after"]  # what _close_sched_stream re-synchronises the generator to
            if E > 1:
                self._tables_dirty = True  # device copies of the tables are refreshed lazily (per-step API)
            if chain is not None and kernel_events is None:
                return self._rollout_plain_chain(N, masked, act_seed, act_counter, finish_ids, kernel_timer,
                                                 packed_dev.data_ptr(), o_mask, o_tab, E, chain, staged)
            perm_dev = packed_dev[o_tab:cur["nbytes"]]
            cur_inv = bool(cur["inv"])
            sched_dev = packed_dev[:N * 12]
            mask_dev = packed_dev[o_mask:o_mask + E * 4]

        W = int(L.ml2048_rollout_flag_words(M))
        flags = torch.empty((N, W), dtype=torch.int64, device=dev)
        last = torch.empty((M,), dtype=torch.int32, device=dev)
        actions = torch.empty((N, M), dtype=torch.uint8, device=dev) if record_actions else None
        dig = torch.zeros((N,), dtype=torch.int64, device=dev) if digests else None
        a_ptr = actions.data_ptr() if actions is not None else None
        d_ptr = dig.data_ptr() if dig is not None else None
        with self._dev_ctx():
            if kernel_events is not None:
                kernel_events[0].record(torch.cuda.current_stream(dev))
            if kernel_timer is not None:
                _lib.check(L.ml2048_timer_arm(kernel_timer), "timer_arm")
            roll_ev = None
            if (not fast and not on_device and staged) or finish_ids == "deferred":
                roll_ev = self._next_roll_event()
                if kernel_events is None:  # (markers around the kernel are wanted: then this one is a marker too)
                    _lib.check(L.ml2048_rollout_attach_event(roll_ev.h), "rollout_attach_event")
            if self._lut_ptr and trajectory is None and a_ptr is None and d_ptr is None:
                L.ml2048_rollout_attach_row_lut(self._lut_ptr)
            if trajectory is not None:
                if fast or record_actions or digests:
                    raise ValueError("trajectory= is for table spawn mode, without record_actions/digests")
                tr = _lib.Traj()
                for name, dt in (("state", torch.int8), ("valid_actions", torch.bool), ("action", torch.int8),
                                 ("action_log_prob", torch.float32), ("reward", torch.float32),
                                 ("next_state", torch.int8), ("next_valid_actions", torch.bool),
                                 ("step", torch.int32), ("terminated", torch.bool)):
                    buf = trajectory.get(name)
                    if buf is None:
                        continue
                    assert buf.is_contiguous() and buf.dtype == dt and buf.device == dev and \
                        tuple(buf.shape[:2]) == (N, M), (name, buf.dtype, tuple(buf.shape))
                    setattr(tr, name, buf.data_ptr())
                _lib.check((L.ml2048_rollout_traj_expanded if cur_inv else L.ml2048_rollout_traj)(
                    self._stream(join=False), M, self._idx_base, self._st(), N, perm_dev.data_ptr(), mask_dev.data_ptr(),
                    E, sched_dev.data_ptr(), self._reward_kind, int(masked), int(act_seed), int(act_counter),
                    flags.data_ptr(), last.data_ptr(), C.byref(tr)), "rollout_traj")
            elif fast:  # rocRAND spawn: no host schedule at all
                _lib.check(L.ml2048_rollout_fast(
                    self._stream(join=False), M, self._idx_base, self._st(), N, self._two_prob, self._spawn_seed,
                    self._fast_counter, self._reward_kind, int(masked), int(act_seed), int(act_counter),
                    a_ptr, d_ptr, flags.data_ptr(), last.data_ptr()), "rollout_fast")
                self._fast_counter += N
            elif on_device and dev_expanded:  # expanded tables straight from the device schedule
                _lib.check(L.ml2048_rollout_chain(
                    self._stream(join=False), M, self._idx_base, self._st(), N, perm_dev.data_ptr(), mask_dev.data_ptr(),
                    E, sched_dev.data_ptr(), self._reward_kind, int(masked), int(act_seed), int(act_counter),
                    flags.data_ptr(), last.data_ptr(), None, None, 0, 0), "rollout_chain")
            elif not on_device and chain is not None:
                _lib.check(L.ml2048_rollout_chain(
                    self._stream(join=False), M, self._idx_base, self._st(), N, perm_dev.data_ptr(), mask_dev.data_ptr(),
                    E, sched_dev.data_ptr(), self._reward_kind, int(masked), int(act_seed), int(act_counter),
                    flags.data_ptr(), last.data_ptr(), chain[0][0].data_ptr(), chain[1].data_ptr(), chain[2], chain[3]),
                    "rollout_chain")
            else:
                _lib.check(L.ml2048_rollout(
                    self._stream(join=False), M, self._idx_base, self._st(), N, perm_dev.data_ptr(), mask_dev.data_ptr(),
                    E, sched_dev.data_ptr(), self._reward_kind, int(masked), int(act_seed), int(act_counter),
                    a_ptr, d_ptr, flags.data_ptr(), last.data_ptr()), "rollout")
            if on_device:  # (every launch variant) the slot may be overwritten once this rollout has consumed it
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                self._sched_pending_slot["free"] = ev
            if kernel_events is not None:
                kernel_events[1].record(torch.cuda.current_stream(dev))
            if roll_ev is not None:
                # ONE marker per rollout: it frees the pinned buffers the launch read and is what a deferred id scan
                # waits for.  It rode on the kernel's own dispatch (see above) unless that dispatch was a timed one.
                if not L.ml2048_rollout_event_attached():
                    roll_ev.record(self._raw_stream())
                if not fast and not on_device:
                    for ps in staged:
                        ps[1] = roll_ev
            step_resets = None
            if finish_ids:
                step_resets = torch.empty((N,), dtype=torch.int32, device=dev)
                scratch = torch.empty((int(L.ml2048_rollout_scratch_ints(M, N)),), dtype=torch.int32, device=dev)
                if finish_ids == "deferred":
                    # the rollout kernel reads neither ids nor the game counter, so the id scan of this
                    # rollout may run beside the NEXT rollout's kernel: a side stream, in order with the
                    # previous scans, joined by whatever touches ids next (join_ids)
                    main = torch.cuda.current_stream(dev)
                    side = self._ids_side_stream()
                    if self._lazy_scan is not None:  # an older plain rollout's scan goes first (the game counter)
                        rec, self._lazy_scan = self._lazy_scan, None
                        self._issue_scan(rec, on_side=True)
                    self._side_scans = True
                    _lib.check(L.ml2048_stream_wait_event(side.cuda_stream, roll_ev.h), "stream_wait_event")
                    for buf in (flags, last, step_resets, scratch):  # allocated on `main`, used on `side`
                        buf.record_stream(side)
                    ids_stream = side.cuda_stream
                else:
                    ids_stream = self._stream()
                _lib.check(L.ml2048_rollout_finish_ids(
                    ids_stream, M, self._st(), N, flags.data_ptr(), last.data_ptr(),
                    self._game_count_dev.data_ptr(), step_resets.data_ptr(), scratch.data_ptr()), "rollout_finish_ids")
                if finish_ids == "deferred":
                    self._ids_done.record(side.cuda_stream)
                    self._ids_pending = True
        # every temporary above was allocated on the launch stream, so the caching allocator's
        # stream-ordered reuse keeps them valid until the kernels have consumed them
        return {"actions": actions, "digests": dig, "step_resets": step_resets,
                "reset_flags": flags, "last_reset": last}
