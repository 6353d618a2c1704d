"""Episode statistics of the FIRST N games, on the GPU: the evaluation loop of the reference's
`eval_perf.py` (:64-102) with `ReplayRecorder` (src/ml2048/replay.py:110-232) as its collector.

The reference records every game from its deal to its termination in per-game Python buffers and, for
the games whose id < rounds, adds (1, steps, score) to the entry of their max tile - "the first N
terminated games is not the first N games" (README.md:112-118).  Here one small kernel after each step
(`ml2048_episode_stats`) keeps a per-slot step counter and folds a game into the table in the step that
terminates it; nothing is recorded or copied per game.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .game import VecGame


class EpisodeStats:
    """Register `on_stepped` as a `VecRunner.EVENT_STEPPED` callback (or call it after every
    `env.step*`), starting with a freshly constructed or freshly reset env; `completed()` == rounds ends the
    evaluation.  "The first `rounds` games" are counted from the env's game counter at construction of this
    object: `_game_count` survives `reset()` (game_numba.py:582 vs :606), so a re-used env deals ids that
    start above zero."""

    def __init__(self, env: VecGame, rounds: int):
        if rounds <= 0:
            raise ValueError(f"rounds={rounds}")
        self.env, self.rounds = env, int(rounds)
        self.id_base = int(env._game_count)
        dev, M = env.device, env._size
        self._age = torch.zeros((M,), dtype=torch.int32, device=dev)
        self._seen = torch.full((M,), -(2**31), dtype=torch.int32, device=dev)
        self._acc = torch.zeros((3, 18), dtype=torch.int64, device=dev)  # count, step_sum, score_sum

    def on_stepped(self, game=None, result=None, actions=None, action_log_probs=None) -> None:
        env = self.env
        st = env._st()
        if result is not None and "id" in result:
            # fused runner (VecRunner(fused=True)): the live state already shows the re-dealt games; the step's
            # outcome - ids included - is in the result's own arrays
            key = tuple(result[k].data_ptr() for k in ("id", "terminated", "state", "score"))
            if key != getattr(self, "_res_key", None):
                s2 = _lib.State.from_buffer_copy(env._state)
                s2.id, s2.terminated, s2.board, s2.score = key
                self._res_key, self._res_state = key, s2
            st = C.byref(self._res_state)
        with env._dev_ctx():
            _lib.check(env._L.ml2048_episode_stats(
                env._stream(), env._size, st, self._age.data_ptr(), self._seen.data_ptr(), self.id_base, self.rounds,
                self._acc[0].data_ptr(), self._acc[1].data_ptr(), self._acc[2].data_ptr()), "episode_stats")

    def completed(self) -> int:
        """Games with id < rounds that have terminated so far (one 8-byte D2H read)."""
        return int(self._acc[0].sum().item())

    @property
    def done(self) -> bool:
        return self.completed() >= self.rounds

    def table(self) -> dict[int, dict[str, float]]:
        """{max tile: count, step_sum, score_sum, steps (mean), score (mean), share} as eval_perf.py prints it."""
        acc = self._acc.cpu().numpy()
        total = int(acc[0].sum())
        out = {}
        for k in range(18):
            n = int(acc[0, k])
            if n:
                out[2 ** k] = {"count": n, "step_sum": int(acc[1, k]), "score_sum": float(acc[2, k]),
                               "steps": acc[1, k] / n, "score": acc[2, k] / n, "share": n / total}
        return out


def evaluate(env: VecGame, policy, rounds: int, *, check_every: int = 16, max_steps: int = 1_000_000,
             fused: bool = False):
    """Play `policy` until the first `rounds` games have all terminated (eval_perf.py:77-99); returns
    (table, env steps played).  The completion test costs a device sync, so it runs every
    `check_every` steps - the table itself only ever contains games with id < rounds."""
    from .runner import VecRunner

    stats = EpisodeStats(env, rounds)
    runner = VecRunner(env, env._size, fused=fused, out_names=("id", "terminated", "state", "score") if fused else None)
    runner.add_callback(VecRunner.EVENT_STEPPED, stats.on_stepped)
    steps = 0
    while steps < max_steps:
        runner.step_once(policy)
        steps += 1
        if steps % check_every == 0 and stats.done:
            break
    else:
        raise RuntimeError(f"evaluation did not finish within {max_steps} steps")
    return stats.table(), steps
