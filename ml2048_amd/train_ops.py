"""Device-side consumers of the rollout: transition recording into the trainer's
`[use, step, game]` buffers and the GAE scan - each one HIP launch instead of a chain of torch ops."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .game import VecGame

# src/ml2048/replay.py:10-20 (REPLAY_SPEC) + the advantage column (actor_critic.py ADV_SPEC)
REPLAY_SPEC = {
    "state": ((16,), torch.int8),
    "valid_actions": ((4,), torch.bool),
    "action": ((), torch.int8),
    "action_log_prob": ((), torch.float32),
    "reward": ((), torch.float32),
    "next_state": ((16,), torch.int8),
    "next_valid_actions": ((4,), torch.bool),
    "step": ((), torch.int32),
    "terminated": ((), torch.bool),
}
ADV_SPEC = {"adv": ((), torch.float32)}


def new_buffers(batch_shape: tuple[int, ...], device, spec=None) -> dict[str, torch.Tensor]:
    """`new_tensors(REPLAY_SPEC | ADV_SPEC, batch_shape, device=...)` (src/ml2048/util.py:16)."""
    spec = spec if spec is not None else {**REPLAY_SPEC, **ADV_SPEC}
    return {k: torch.zeros(tuple(batch_shape) + shape, dtype=dt, device=device) for k, (shape, dt) in spec.items()}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def record_transition(env: VecGame, actions: Optional[torch.Tensor], log_probs: Optional[torch.Tensor],
                      buffers: dict[str, torch.Tensor], ui: int, si: int) -> None:
    """What `Trainer.on_stepped` does with nine copies (run_train3.py:125-155), as one kernel:
    write the post-step transition of every game into buffers[name][ui, si]."""
    L = _lib.load()
    M = env._size

    def sl(name):
        b = buffers.get(name)
        if b is None:
            return None
        v = b[ui, si]
        assert v.is_contiguous() and v.shape[0] == M and v.device == env.device, (name, v.shape)
        return v

    a = actions
    if a is not None:
        if a.dtype != torch.uint8:
            a = a.to(torch.uint8)
        a = a.contiguous()
    lp = log_probs.detach().to(torch.float32).contiguous() if log_probs is not None else None
    outs = [sl(n) for n in ("state", "valid_actions", "action", "action_log_prob", "reward", "next_state",
                            "next_valid_actions", "step", "terminated")]
    with env._dev_ctx():
        _lib.check(L.ml2048_record_transition(env._stream(), M, env._st(), _ptr(a), _ptr(lp), *[_ptr(o) for o in outs]),
                   "record_transition")


def gae(v0: torch.Tensor, v1: torch.Tensor, reward: torch.Tensor, terminated: torch.Tensor, *, gamma: float,
        lambda_: float, adv: Optional[torch.Tensor] = None, want_delta: bool = False):
    """compute_gae's arithmetic (src/ml2048/gae.py:50, :58-68) on [use, step, game] tensors, one launch;
    bit-identical to the reference's torch expression.  Returns adv (and delta)."""
    L = _lib.load()
    assert v0.ndim == 3 and v0.shape == v1.shape == reward.shape == terminated.shape, (v0.shape, terminated.shape)
    U, S, G = v0.shape
    v0, v1, reward = (t.to(torch.float32).contiguous() for t in (v0, v1, reward))
    term = terminated.contiguous()
    term = term.view(torch.uint8) if term.dtype == torch.bool else term.to(torch.uint8)
    if adv is None:
        adv = torch.empty_like(v0)
    assert adv.is_contiguous() and adv.dtype == torch.float32 and adv.shape == v0.shape
    delta = torch.empty_like(v0) if want_delta else None
    with torch.cuda.device(v0.device):
        stream = torch.cuda.current_stream(v0.device).cuda_stream
        _lib.check(L.ml2048_gae(stream, U, S, G, v0.data_ptr(), v1.data_ptr(), reward.data_ptr(), term.data_ptr(),
                                float(gamma), float(lambda_), adv.data_ptr(), _ptr(delta)), "gae")
    return (adv, delta) if want_delta else adv
