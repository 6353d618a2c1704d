"""Policies with the reference's `Policy.sample_actions` contract
(src/ml2048/policy/__init__.py:10-19): (state, valid_actions) -> (actions long, log_probs float)."""
from __future__ import annotations

import os

import torch

from . import _lib


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


class Policy:
    def sample_actions(self, state, valid_actions, *, generator=None):
        raise NotImplementedError


class RandomPolicy(Policy):
    """Uniform over the valid actions (src/ml2048/policy/random.py:8-27), sampled by the HIP
    kernel with a counter-based Philox stream (rocRAND device generator): reproducible on any
    device from (seed, call counter, board index), unlike torch.multinomial."""

    def __init__(self, seed: int | None = None, *, masked: bool = True, idx_base: int = 0):
        self._L = _lib.load()
        # an unseeded policy draws its own seed; torch's global generators are left alone (the reference seeds a
        # private torch.Generator, policy/random.py:12-15), so a caller's torch.manual_seed() stays in force
        self.seed = int(int.from_bytes(os.urandom(8), "little") if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.masked = masked
        self.counter = 0
        self.idx_base = idx_base

    def sample_actions_u8(self, valid_actions: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """Device-native form: (actions uint8 (M,), log_probs float32 (M,)); no dtype round trips."""
        v = valid_actions
        if v.dtype == torch.bool:
            v = v.view(torch.uint8)
        assert v.is_cuda and v.dtype == torch.uint8 and v.ndim == 2 and v.shape[1] == 4, (v.dtype, v.shape)
        v = v.contiguous()
        M = v.shape[0]
        actions = torch.empty((M,), dtype=torch.uint8, device=v.device)
        logp = torch.empty((M,), dtype=torch.float32, device=v.device)
        idx = v.device.index if v.device.index is not None else torch.cuda.current_device()
        try:
            stream = torch._C._cuda_getCurrentRawStream(idx)
        except AttributeError:  # pragma: no cover
            stream = torch.cuda.current_stream(v.device).cuda_stream
        guard = torch.cuda.device(idx) if torch.cuda.current_device() != idx else _NO_GUARD
        with guard:
            _lib.check(self._L.ml2048_sample_actions(stream, M, self.idx_base, v.data_ptr(), self.seed, self.counter,
                                                     int(self.masked), actions.data_ptr(), logp.data_ptr()), "sample")
        self.counter += 1
        return actions, logp

    def sample_actions(self, state, valid_actions, *, generator=None):
        scalar = valid_actions.ndim == 1
        v = valid_actions.reshape(-1, 4)
        actions, logp = self.sample_actions_u8(v)
        actions = actions.long()
        if scalar:
            return actions.reshape(()), logp.reshape(())
        return actions, logp
