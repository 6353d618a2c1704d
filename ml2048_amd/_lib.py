"""ctypes binding of libml2048_hip.so (include/ml2048_hip.h).

The product has exactly one compute path: the HIP library.  If it is missing or cannot be loaded
this module raises - there is no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

LIB_PATH = _build.LIB_PATH


class State(C.Structure):
    """struct ml2048_state (include/ml2048_hip.h)."""

    _fields_ = [(n, C.c_void_p) for n in (
        "board", "merged", "valid", "step", "score", "reward", "terminated", "invalid", "id",
        "prev_board", "prev_valid")]


class Ml2048Error(RuntimeError):
    pass


class Traj(C.Structure):
    """struct ml2048_traj (include/ml2048_hip.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("state", "valid_actions", "action", "action_log_prob", "reward",
                                           "next_state", "next_valid_actions", "step", "terminated")]


class StepOut(C.Structure):
    """struct ml2048_step_out (include/ml2048_hip.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("state", "valid_actions", "merged", "step", "reward", "score", "terminated",
                                           "invalid", "id", "prev_state", "prev_valid_actions")]


class StepPrepareArgs(C.Structure):
    """struct ml2048_step_prepare_args (include/ml2048_hip.h)."""

    _fields_ = [("M", C.c_int64), ("idx_base", C.c_int64), ("st", State), ("actions", C.c_void_p),
                ("action_dtype", C.c_int32), ("reward_kind", C.c_int32), ("two_prob", C.c_double),
                ("randperm_step", C.c_void_p), ("row_base_step", C.c_int64),
                ("randperm_prepare", C.c_void_p), ("row_base_prepare", C.c_int64),
                ("two_mask_step", C.c_uint32), ("two_mask_prepare", C.c_uint32),
                ("spawn_seed", C.c_uint64), ("spawn_counter_step", C.c_uint64), ("spawn_counter_prepare", C.c_uint64),
                ("game_count", C.c_void_p), ("n_reset", C.c_void_p), ("reset_indices", C.c_void_p),
                ("scratch", C.c_void_p), ("out", StepOut), ("onehot", C.c_void_p), ("board_i64", C.c_void_p),
                ("next_actions", C.c_void_p), ("next_logp", C.c_void_p), ("act_seed", C.c_uint64),
                ("act_counter", C.c_uint64), ("act_masked", C.c_int32), ("reserved", C.c_int32),
                ("hist_counts", C.c_void_p)]


class Pcg64State(C.Structure):
    """struct ml2048_pcg64 (include/ml2048_hip.h)."""

    _fields_ = [("state_hi", C.c_uint64), ("state_lo", C.c_uint64), ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64),
                ("has_uint32", C.c_int32), ("uinteger", C.c_uint32)]


_lib = None

_SIGNATURES = {
    # name: (restype, argtypes)
    "ml2048_abi_version": (C.c_int, []),
    "ml2048_last_error": (C.c_char_p, []),
    "ml2048_scratch_ints": (C.c_int64, [C.c_int64]),
    "ml2048_reset": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(State)]),
    "ml2048_prepare": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_void_p, C.c_void_p,
                                 C.c_double, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_step": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_void_p, C.c_int, C.c_int,
                              C.c_double, C.c_int64, C.c_void_p, C.c_void_p]),
    "ml2048_step_prepare_scratch_bytes": (C.c_int64, []),
    "ml2048_step_prepare": (C.c_int, [C.c_void_p, C.POINTER(StepPrepareArgs)]),
    "ml2048_step_wave": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_void_p, C.c_int, C.c_int,
                                   C.c_double, C.c_int64, C.c_void_p, C.c_void_p]),
    "ml2048_prepare_fast": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_double, C.c_uint64,
                                      C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_step_fast": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_void_p, C.c_int, C.c_int,
                                   C.c_double, C.c_uint64, C.c_uint64]),
    "ml2048_sample_actions": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_uint64, C.c_uint64,
                                        C.c_int, C.c_void_p, C.c_void_p]),
    "ml2048_encode_onehot": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_hist_maxcell": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_pack_records": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(State), C.c_void_p]),
    "ml2048_unpack_records": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(State), C.c_void_p]),
    "ml2048_digest": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_int, C.c_void_p]),
    "ml2048_debug_philox": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "ml2048_hostrng_schedule": (C.c_int, [C.POINTER(Pcg64State), C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32,
                                          C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.POINTER(C.c_int32)]),
    "ml2048_record_transition": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(State)] + [C.c_void_p] * 11),
    "ml2048_gae": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "ml2048_sched_stream_create": (C.c_void_p, [C.POINTER(Pcg64State), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                               C.c_double, C.c_int32]),
    "ml2048_sched_stream_next": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(Pcg64State),
                                           C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "ml2048_sched_stream_release": (C.c_int, [C.c_void_p]),
    "ml2048_sched_stream_slot_bytes": (C.c_int64, [C.c_int32]),
    "ml2048_sched_stream_create_pinned": (C.c_void_p, [C.POINTER(Pcg64State), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                      C.c_double, C.c_int32, C.POINTER(C.c_void_p), C.c_int64]),
    "ml2048_sched_stream_take_slot": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                                C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(Pcg64State),
                                                C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "ml2048_sched_stream_take": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(Pcg64State),
                                           C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "ml2048_sched_stream_destroy": (None, [C.c_void_p]),
    "ml2048_hostrng_regen": (C.c_int, [C.POINTER(Pcg64State), C.c_void_p, C.c_void_p]),
    "ml2048_devrng_workspace_bytes": (C.c_int64, []),
    "ml2048_devrng_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ml2048_devrng_schedule": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ml2048_devrng2_workspace_bytes": (C.c_int64, []),
    "ml2048_devrng2_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ml2048_devrng2_schedule": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int]),
    "ml2048_rollout_flag_words": (C.c_int64, [C.c_int64]),
    "ml2048_rollout_scratch_ints": (C.c_int64, [C.c_int64, C.c_int32]),
    "ml2048_timer_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ml2048_timer_arm": (C.c_int, [C.c_void_p]),
    "ml2048_timer_elapsed_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ml2048_timer_destroy": (None, [C.c_void_p]),
    "ml2048_encoder_front": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_episode_stats": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(State), C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_rollout_attach_event": (C.c_int, [C.c_void_p]),
    "ml2048_rollout_event_attached": (C.c_int, []),
    "ml2048_row_lut_fill": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ml2048_rollout_attach_row_lut": (C.c_int, [C.c_void_p]),
    "ml2048_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ml2048_event_destroy": (None, [C.c_void_p]),
    "ml2048_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ml2048_event_query": (C.c_int, [C.c_void_p]),
    "ml2048_event_synchronize": (C.c_int, [C.c_void_p]),
    "ml2048_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ml2048_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "ml2048_rollout": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_rollout_chain": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "ml2048_rollout_chain_ids": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_rollout_traj": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                      C.c_void_p, C.c_void_p, C.POINTER(Traj)]),
    "ml2048_rollout_traj_expanded": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_int32, C.c_void_p,
                                               C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                               C.c_void_p, C.c_void_p, C.POINTER(Traj)]),
    "ml2048_rollout_fast": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(State), C.c_int32, C.c_double,
                                      C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_rollout_finish_ids": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(State), C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ml2048_rollout_ids_after": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(State), C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)


def load(path: str | None = None) -> C.CDLL:
    """Load (once) and bind the HIP library; raises Ml2048Error when it is not there."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # ML2048_HIP_LIB: alternative build of the same ABI (A/B experiments); the default is the in-tree library
    p = path or os.environ.get("ML2048_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise Ml2048Error(
            f"{p} not found: build the HIP extension first (python -m ml2048_amd.build). "
            "ml2048_amd has no CPU fallback."
        )
    try:
        L = C.CDLL(p)
    except OSError as e:  # pragma: no cover - depends on the box
        raise Ml2048Error(f"cannot load {p}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if L.ml2048_abi_version() != 1:
        raise Ml2048Error(f"ABI version {L.ml2048_abi_version()} != 1")
    if path is None:
        _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    """Map the C ABI's error convention onto Python exceptions (ValueError for bad arguments,
    like the reference's own argument checks; RuntimeError for HIP failures)."""
    if rc == 0:
        return
    msg = load().ml2048_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError(f"{what}: {msg} (code {rc})")
    raise Ml2048Error(f"{what}: {msg} (hipError {rc})")
