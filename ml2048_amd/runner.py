"""Rollout driver: the reference's `VecRunner` / `RunnerStats` contract
(src/ml2048/runner.py:28-189, src/ml2048/event.py) with observations, actions and results kept
on the GPU (the reference crosses host<->device 2+1+7 times per step, runner.py:91-104 and
run_train3.py:143-149)."""
from __future__ import annotations

import collections
from typing import Any, Callable, Sequence

import numpy as np
import torch

from . import _lib
from .game import VecGame, VecStepResult
from .policy import Policy, RandomPolicy


class EventEmitter:
    """Named callback lists (the contract of src/ml2048/event.py:7-38: add_listener / emit)."""

    def __init__(self):
        self._hooks: dict[str, list[Callable[..., Any]]] = collections.defaultdict(list)

    def add_listener(self, name: str, fn: Callable[..., Any], prepend: bool = False) -> None:
        hooks = self._hooks[name]
        hooks.insert(0 if prepend else len(hooks), fn)

    def has_listeners(self, name: str) -> bool:
        return len(self._hooks.get(name, ())) > 0

    def emit(self, name: str, args: tuple[Any, ...], kwargs: dict[str, Any] | None = None) -> None:
        extra = kwargs or {}
        for hook in tuple(self._hooks.get(name, ())):
            hook(*args, **extra)


class VecRunner:
    """prepare -> observations -> policy.sample_actions -> step -> emit, runner.py:74-117.

    Events: "prepared"(env, new_indices) and "stepped"(env, result, actions, log_probs), as in the
    reference (runner.py:35-43, :80-109).  `new_indices` is a numpy array (it costs one 4-byte
    D2H sync, paid only when a "prepared" listener is registered); `result` holds device tensors.
    """

    EVENT_PREPARED = "prepared"
    EVENT_STEPPED = "stepped"

    def __init__(self, env: VecGame, capacity: int = 0, *, sample_device=None, long_state: bool = True,
                 fused: bool = False, out_names=None):
        """fused=True: one launch per env step (`VecGame.step_prepare_torch`: step(t) and prepare(t+1) in one kernel;
        a RandomPolicy's next draw comes out of the same launch).  Same events in the same order with the same
        contents; `result` then holds the step's outcome in buffers owned by the runner (the reference's views are
        clobbered by the next prepare() just the same), plus "id" (the games' ids as of the step).
        out_names: which of VecGame.STEP_OUT_SPEC's arrays `result` needs (default: all of them, 74 B/board)."""
        self.env = env
        self.sample_device = sample_device if sample_device is not None else env.device
        self._emitter = EventEmitter()
        self._vec_size = env._size  # runner.py:57
        self._capacity = capacity
        self._long_state = long_state
        self._fused = fused
        self._out = env.new_step_out(out_names) if fused else None
        self._next = None  # fused + RandomPolicy: (actions, log_probs, env serial, policy, policy counter)

    def add_callback(self, event: str, fn: Callable[..., Any]) -> None:
        assert event in {self.EVENT_STEPPED, self.EVENT_PREPARED}
        self._emitter.add_listener(event, fn)

    def _step_once_fused(self, policy: Policy) -> VecStepResult:
        env = self.env
        want_prepared = self._emitter.has_listeners(self.EVENT_PREPARED)
        if not env._prepared:  # first step (or another API stepped the env): deal as prepare() does
            if want_prepared:
                (new_indices,) = env.prepare()
                self._emitter.emit(self.EVENT_PREPARED, (env, new_indices))
            else:
                env.prepare_torch()
        board, valid = env.observations_torch()
        is_random = isinstance(policy, RandomPolicy) and policy.idx_base == env._idx_base
        with torch.no_grad():
            nx = self._next
            if is_random and nx is not None and nx[2] == env._serial and nx[3] is policy and nx[4] == policy.counter:
                actions, log_probs = nx[0], nx[1]  # drawn by the previous launch on these very masks
            elif isinstance(policy, RandomPolicy):
                actions, log_probs = policy.sample_actions_u8(valid)
            else:
                state = board.to(torch.long) if self._long_state else board
                actions, log_probs = policy.sample_actions(state, valid.to(torch.bool))
        res = env.step_prepare_torch(actions, out=self._out, sample=policy if is_random else None)
        self._next = (res["actions"], res["log_probs"], env._serial, policy, policy.counter) if is_random else None
        result = dict(self._out)
        self._emitter.emit(self.EVENT_STEPPED, (env, result, actions, log_probs))
        if want_prepared:  # the prepare() of the NEXT step already ran: same event, same place in the sequence
            k = int(res["n_reset"].item())
            self._emitter.emit(self.EVENT_PREPARED, (env, res["reset_indices"][:k].cpu().numpy()))
        return result  # type: ignore[return-value]

    def step_once(self, policy: Policy) -> VecStepResult:
        if self._fused:
            return self._step_once_fused(policy)
        env = self.env
        if self._emitter.has_listeners(self.EVENT_PREPARED):
            (new_indices,) = env.prepare()
            self._emitter.emit(self.EVENT_PREPARED, (env, new_indices))
        else:
            env.prepare_torch()
        board, valid = env.observations_torch()
        with torch.no_grad():
            if isinstance(policy, RandomPolicy):
                actions, log_probs = policy.sample_actions_u8(valid)
            else:
                # the reference hands the policy (N,16) long / (N,4) bool (runner.py:91-95)
                state = board.to(torch.long) if self._long_state else board
                actions, log_probs = policy.sample_actions(state, valid.to(torch.bool))
        result = env.step_torch(actions)
        self._emitter.emit(self.EVENT_STEPPED, (env, result, actions, log_probs))
        return result

    def step_many(self, policy: Policy, count: int) -> None:
        """runner.py:111-117.  With a RandomPolicy and no listeners nothing observes the individual steps, so
        the `count` prepare/sample/step triples run as ONE fused launch (`VecGame.rollout`): same sampler
        stream (seed, call counter), same final state, ids and generator state as the loop
        (`tests/test_gpu_parity.py::test_runner_step_many_fuses_random_rollouts`)."""
        if (count > 0 and isinstance(policy, RandomPolicy) and policy.idx_base == self.env._idx_base
                and not self._emitter.has_listeners(self.EVENT_PREPARED)
                and not self._emitter.has_listeners(self.EVENT_STEPPED)):
            self.env.rollout(count, masked=policy.masked, act_seed=policy.seed, act_counter=policy.counter)
            policy.counter += count
            return
        for _ in range(count):
            self.step_once(policy)


class RunnerStats:
    """Max-tile histogram of terminated games (runner.py:139-189), accumulated on device."""

    def __init__(self):
        self.counts = np.zeros((20,), dtype=np.int32)
        self.terminated_count = 0
        self._dev: torch.Tensor | None = None

    def reset(self) -> None:
        self.counts.fill(0)
        self.terminated_count = 0
        if self._dev is not None:
            self._dev.zero_()

    def device_counts(self, game: VecGame) -> torch.Tensor:
        """The int32[20] device accumulator (pass it as `VecGame.step_prepare_torch(hist=...)`: the fused step adds
        the finished games' max tiles itself, no extra launch)."""
        if self._dev is None:
            self._dev = torch.zeros((20,), dtype=torch.int32, device=game.device)
        return self._dev

    def on_stepped(self, game: VecGame, result, actions, action_log_probs) -> None:
        # _update_count(counts, result["state"], result["terminated"]), runner.py:120-136, on the result's own arrays
        # (the live state in the per-step mode; the runner's outcome buffers in the fused mode)
        acc = self.device_counts(game)
        state, term = result["state"], result["terminated"]
        with game._dev_ctx():
            _lib.check(game._L.ml2048_hist_maxcell(game._stream(), game._size, state.data_ptr(), term.data_ptr(),
                                                   acc.data_ptr()), "hist")

    def sync(self) -> None:
        """Fold the device accumulator into `counts` / `terminated_count` (one small D2H).  Every terminated game is
        counted in exactly one bin (runner.py:131-136, :162), so the terminated count is the histogram's total."""
        if self._dev is not None:
            delta = self._dev.cpu().numpy()
            self.counts += delta
            self.terminated_count += int(delta.sum())
            self._dev.zero_()

    def summary(self) -> list[tuple]:
        """[(tile value, games, share)] from the largest tile down, skipping empty bins
        (same rows as the reference's RunnerStats.summary, runner.py:168-179)."""
        self.sync()
        total = int(self.counts.sum())
        rows = [(1 << e, int(self.counts[e])) for e in range(16, 0, -1) if self.counts[e]]
        return [(tile, n, n / total) for tile, n in rows]

    @classmethod
    def combine(cls, seq: Sequence["RunnerStats"]) -> "RunnerStats":
        """Host-side gather over shards / GPUs (no collective needed: 20 ints per shard)."""
        merged = cls()
        for part in seq:
            part.sync()
            merged.counts += part.counts
            merged.terminated_count += part.terminated_count
        return merged
