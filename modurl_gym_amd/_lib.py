"""Loader and ctypes prototypes for libmgym.so (the C ABI in include/mgym.h).

The engine has no CPU fallback: if the shared library is missing this module raises at
import, and if no HIP device is visible `mgym_create` returns MGYM_ERR_NO_DEVICE, which the
host classes turn into `MgymError`.  PyTorch is optional plumbing (device tensors, streams,
torch.distributed); when it is importable it is imported BEFORE libmgym so both share one
HIP runtime (torch bundles libamdhip64.so.7 under the same SONAME).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmgym.so")

OK, ERR_INVALID_ACTION, ERR_NOT_RESET, ERR_BAD_CONFIG, ERR_HIP, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_CAPACITY = range(8)
CARTPOLE, MOUNTAINCAR, MOUNTAINCAR_CONT, LUNARLANDER = range(4)
FLAG_AUTO_RESET = 1


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("n_envs", C.c_uint64), ("env_id_base", C.c_uint64), ("seed", C.c_uint64),
                ("sutton_barto_reward", C.c_int32), ("is_euler", C.c_int32), ("goal_velocity", C.c_float),
                ("gravity", C.c_float), ("enable_wind", C.c_int32), ("wind_power", C.c_float),
                ("turbulence_power", C.c_float), ("reserved", C.c_uint32)]


class Spec(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("n_actions", C.c_int32), ("action_is_float", C.c_int32),
                ("state_cols", C.c_int32), ("obs_low", C.c_float * 8), ("obs_high", C.c_float * 8),
                ("action_low", C.c_float), ("action_high", C.c_float)]


# every symbol include/mgym.h declares: name -> (restype, argtypes)
_vp, _u64p = C.c_void_p, C.POINTER(C.c_uint64)
PROTOTYPES = {
    "mgym_abi_version": (C.c_int, []),
    "mgym_default_config": (C.c_int, [C.c_int, C.POINTER(Config)]),
    "mgym_create": (C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    "mgym_destroy": (C.c_int, [_vp]),
    "mgym_set_stream": (C.c_int, [_vp, _vp]),
    "mgym_get_stream": (_vp, [_vp]),
    "mgym_reset": (C.c_int, [_vp, _vp, _vp]),
    "mgym_reset_done": (C.c_int, [_vp, _vp, _vp, _vp]),
    "mgym_reset_deterministic": (C.c_int, [_vp, _vp]),
    "mgym_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mgym_rollout": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "mgym_rollout_uniform": (C.c_int, [_vp, C.c_uint64, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "mgym_rollout_linear": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "mgym_episode_count": (C.c_int, [_vp, _u64p]),
    "mgym_observation": (C.c_int, [_vp, C.POINTER(_vp), _u64p]),
    "mgym_observation_aos": (C.c_int, [_vp, _vp]),
    "mgym_get_state": (C.c_int, [_vp, _vp]),
    "mgym_set_state": (C.c_int, [_vp, _vp]),
    "mgym_set_dispersion_override": (C.c_int, [_vp, _vp]),
    "mgym_get_spec": (C.c_int, [C.c_int, C.POINTER(Spec)]),
    "mgym_sync": (C.c_int, [_vp]),
    "mgym_last_error": (C.c_char_p, []),
    "mgym_get_info": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "mgym_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_vp)]),
    "mgym_free": (C.c_int, [C.c_int, _vp]),
    "mgym_memcpy_h2d": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t]),
    "mgym_memcpy_d2h": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t]),
    "mgym_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mgym_timer_start": (C.c_int, [_vp]),
    "mgym_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "mgym_selftest_cartpole_math": (C.c_int, [C.c_int, _u64p]),
    "mgym_graph_begin": (C.c_int, [_vp]),
    "mgym_graph_end": (C.c_int, [_vp, C.POINTER(_vp)]),
    "mgym_graph_launch": (C.c_int, [_vp, _vp]),
    "mgym_graph_destroy": (C.c_int, [_vp]),
}

_lib = None


def load():
    """dlopen libmgym.so; raises OSError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C modurl_gym_amd/csrc` (needs hipcc). There is no CPU fallback.")
    try:  # share torch's HIP runtime when torch is around (plumbing only)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.mgym_abi_version() != 4:
        raise OSError("libmgym.so ABI version mismatch")
    _lib = lib
    return lib
