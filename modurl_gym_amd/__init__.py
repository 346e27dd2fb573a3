"""modurl_gym_amd — MI355X-native batched environment engine for the ModuRL_Gym hot path.

Host-side mirror of the reference's Gym interface (CartPoleV1 / MountainCarV0 / LunarLanderV3:
reset(), step()) over the C ABI of libmgym.so (include/mgym.h), whose kernels are hand-written
HIP for gfx950.  There is no CPU fallback: importing the env classes requires the built library.
"""
import os as _os

# A LunarLander step runs its contact kernel, its free-flight kernel and the preparation of the next resets side by side on three
# streams.  The HIP runtime gives a process 4 hardware queues per device by default and lets further streams SHARE them; streams
# that share a queue run one after the other (262 144 envs: 1.93-1.99 ms per step instead of 1.45 in a process that already holds
# four or five streams).  The knob is read once, when the runtime initialises, so it is set here — before this package loads
# libmgym.so (and, through it, torch / HIP).  An embedder's own setting wins (setdefault); C / C++ / Rust embedders call
# setenv("GPU_MAX_HW_QUEUES", "8", 0) before their first HIP call; `mgym_get_info` reports what a handle got.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import (CARTPOLE, LUNARLANDER, MOUNTAINCAR, MOUNTAINCAR_CONT, LIB_PATH, PROTOTYPES)  # noqa: F401
from .envs import (BadConfigError, CartPoleV1, DeviceArray, InvalidActionError, LunarLanderV3,  # noqa: F401
                   MgymError, MountainCarContinuousV0, MountainCarV0, NotResetError, StepInfo, VecEnv,
                   device_count, get_spec)
from .shard import Shard, all_reduce_episode_count, mixed_population, population_plan, shard_range  # noqa: F401
from .torch_env import TorchVecEnv  # noqa: F401

__all__ = ["CartPoleV1", "MountainCarV0", "MountainCarContinuousV0", "LunarLanderV3", "VecEnv", "StepInfo",
           "DeviceArray", "MgymError", "InvalidActionError", "NotResetError", "BadConfigError", "device_count",
           "get_spec", "shard_range", "mixed_population", "population_plan", "all_reduce_episode_count", "Shard", "TorchVecEnv"]
