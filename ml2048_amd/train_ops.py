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


def encoder_front(board: torch.Tensor, dw_full, dw_rows, dw_cols, *, negative_slope: float = 0.01, out=None):
    """leaky_relu of the reference encoder's three depthwise convolutions (`CNNEncoder.forward`,
    src/ml2048/policy/_network.py:98-116) computed straight from the (N,16) board - the one-hot is never
    materialised.  `dw_full` / `dw_rows` / `dw_cols` are the reference's modules
    (`_depthwise_full` Conv1d(16,256,16,groups=16), `_depthwise_hori` Conv2d (1,4), `_depthwise_vert` Conv2d (4,1))
    or anything with `.weight` / `.bias` of those shapes.  Inference only (no autograd through the kernel).

    Returns (full (N,256), rows (N,256,4), cols (N,256,4)) float32: the inputs of `_pointwise_full`,
    `_pointwise_hori` (as (N,256,4,1)) and `_pointwise_vert` (as (N,256,1,4))."""
    assert board.is_cuda and board.ndim == 2 and board.shape[1] == 16, board.shape
    b = board if board.dtype == torch.uint8 else board.to(torch.uint8)
    b = b.contiguous()
    N, dev = b.shape[0], b.device

    def flat(t, shape):
        t = t.detach().to(device=dev, dtype=torch.float32).reshape(shape).contiguous()
        return t

    wf, bf = flat(dw_full.weight, (256, 16)), flat(dw_full.bias, (256,))
    wr, br = flat(dw_rows.weight, (256, 4)), flat(dw_rows.bias, (256,))
    wc, bc = flat(dw_cols.weight, (256, 4)), flat(dw_cols.bias, (256,))
    if out is None:
        out = (torch.empty((N, 256), dtype=torch.float32, device=dev),
               torch.empty((N, 256, 4), dtype=torch.float32, device=dev),
               torch.empty((N, 256, 4), dtype=torch.float32, device=dev))
    L = _lib.load()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        _lib.check(L.ml2048_encoder_front(torch.cuda.current_stream(dev).cuda_stream, N, b.data_ptr(), wf.data_ptr(),
                                          bf.data_ptr(), wr.data_ptr(), br.data_ptr(), wc.data_ptr(), bc.data_ptr(),
                                          float(negative_slope), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()),
                   "encoder_front")
    return out
