"""ml2048_amd - MI355X-native vectorized 2048 environment (drop-in for ml2048.game_numba.VecGame).

Only the environment hot path lives here: HIP kernels + C ABI (csrc/, include/ml2048_hip.h) and
the Python host that mirrors the reference interface.  Importing the package does not touch the
GPU; constructing a VecGame does, and raises if the HIP library or a GPU is missing.
"""
from .game import (  # noqa: F401
    ITEM_VALUES,
    RECORD_DTYPE,
    STEP_DOWN,
    STEP_LEFT,
    STEP_RIGHT,
    STEP_UP,
    RewardFn,
    VecGame,
    VecStepResult,
    reward_fn_improved,
    reward_fn_maxcell,
    reward_fn_normal,
    reward_fn_rank,
)
from .evalstats import EpisodeStats, evaluate  # noqa: F401
from .hostrng import HostRng  # noqa: F401
from .policy import Policy, RandomPolicy  # noqa: F401
from .runner import EventEmitter, RunnerStats, VecRunner  # noqa: F401

__version__ = "0.1.0"
