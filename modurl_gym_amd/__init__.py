"""modurl_gym_amd — MI355X-native batched environment engine for the ModuRL_Gym hot path.

Host-side mirror of the reference's Gym interface (CartPoleV1 / MountainCarV0 / LunarLanderV3:
reset(), step()) over the C ABI of libmgym.so (include/mgym.h), whose kernels are hand-written
HIP for gfx950.  There is no CPU fallback: importing the env classes requires the built library.
"""
from ._lib import (CARTPOLE, LUNARLANDER, MOUNTAINCAR, MOUNTAINCAR_CONT, LIB_PATH, PROTOTYPES)  # noqa: F401
from .envs import (BadConfigError, CartPoleV1, DeviceArray, InvalidActionError, LunarLanderV3,  # noqa: F401
                   MgymError, MountainCarContinuousV0, MountainCarV0, NotResetError, StepInfo, VecEnv,
                   device_count, get_spec)
from .shard import Shard, all_reduce_episode_count, mixed_population, population_plan, shard_range  # noqa: F401
from .torch_env import TorchVecEnv  # noqa: F401

__all__ = ["CartPoleV1", "MountainCarV0", "MountainCarContinuousV0", "LunarLanderV3", "VecEnv", "StepInfo",
           "DeviceArray", "MgymError", "InvalidActionError", "NotResetError", "BadConfigError", "device_count",
           "get_spec", "shard_range", "mixed_population", "population_plan", "all_reduce_episode_count", "Shard", "TorchVecEnv"]
