"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (modurl_gym_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

CARTPOLE, MOUNTAINCAR, MOUNTAINCAR_CONT, LUNARLANDER = 0, 1, 2, 3
OK, INVALID_ACTION, NOT_RESET, BAD_CONFIG = 0, 1, 2, 3


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


class StepInfo(C.Structure):
    _fields_ = [("reward", C.c_float), ("done", C.c_uint8), ("truncated", C.c_uint8)]


class CartPole(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("gravity", "masspole", "total_mass", "length", "polemass_length",
                                         "force_mag", "tau", "x_threshold", "theta_threshold_radians")] + [
        ("is_euler", C.c_int), ("sbt_is_some", C.c_int), ("sbt", C.c_uint64), ("state", C.c_float * 4),
        ("steps_since_reset", C.c_uint64), ("sutton_barto_reward", C.c_int)]


class MountainCar(C.Structure):
    _fields_ = [("state", C.c_float * 2)] + [(n, C.c_float) for n in (
        "min_position", "max_position", "max_speed", "goal_position", "goal_velocity", "force", "gravity")]


class MountainCarCont(C.Structure):
    _fields_ = [("state", C.c_float * 2)] + [(n, C.c_float) for n in (
        "min_action", "max_action", "min_position", "max_position", "max_speed", "goal_position",
        "goal_velocity", "power")]


class VecConfig(C.Structure):
    _fields_ = [("kind", C.c_int), ("n_envs", C.c_uint64), ("env_id_base", C.c_uint64), ("seed", C.c_uint64),
                ("sutton_barto_reward", C.c_int), ("is_euler", C.c_int), ("goal_velocity", C.c_float),
                ("gravity", C.c_float), ("enable_wind", C.c_int), ("wind_power", C.c_float),
                ("turbulence_power", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ora_u53.restype = C.c_double
        L.ora_u53.argtypes = [C.c_uint32, C.c_uint32]
        L.ora_u23.restype = C.c_float
        L.ora_u23.argtypes = [C.c_uint32]
        L.ora_lunarlander_new.restype = C.c_void_p
        L.ora_lunarlander_new.argtypes = [C.c_float, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int)]
        L.ora_lunarlander_free.argtypes = [C.c_void_p]
        fp = C.POINTER(C.c_float)
        L.ora_lunarlander_reset.argtypes = [C.c_void_p, fp, fp, C.c_int32, C.c_int32, fp, fp]
        L.ora_lunarlander_reset_deterministic.argtypes = [C.c_void_p, fp]
        L.ora_lunarlander_step.argtypes = [C.c_void_p, C.c_uint32, fp, fp, C.POINTER(StepInfo)]
        L.ora_lunarlander_set_state.argtypes = [C.c_void_p, fp, C.c_int, C.c_int]
        L.ora_lunarlander_export.argtypes = [C.c_void_p, fp]
        L.ora_lunarlander_import.argtypes = [C.c_void_p, fp]
        L.ora_cartpole_step.argtypes = [C.POINTER(CartPole), C.c_uint32, C.POINTER(StepInfo)]
        L.ora_mountaincar_step.argtypes = [C.POINTER(MountainCar), C.c_uint32, C.POINTER(StepInfo)]
        L.ora_mountaincar_cont_step.argtypes = [C.POINTER(MountainCarCont), C.c_float, C.POINTER(StepInfo)]
        L.ora_mountaincar_new.argtypes = [C.POINTER(MountainCar), C.c_float]
        L.ora_mountaincar_cont_new.argtypes = [C.POINTER(MountainCarCont), C.c_float]
        L.ora_mountaincar_reset.argtypes = [C.POINTER(MountainCar), C.c_double]
        L.ora_mountaincar_cont_reset.argtypes = [C.POINTER(MountainCarCont), C.c_double]
        L.ora_vec_new.restype = C.c_void_p
        L.ora_set_contact_order_variant.argtypes = [C.c_int]
        L.ora_vec_new.argtypes = [C.POINTER(VecConfig), C.POINTER(C.c_int)]
        L.ora_vec_free.argtypes = [C.c_void_p]
        L.ora_vec_obs_dim.argtypes = [C.c_void_p]
        L.ora_vec_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_vec_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
        L.ora_vec_state_cols.argtypes = [C.c_void_p]
        L.ora_vec_get_state.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_vec_set_state.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_vec_set_dispersion.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_vec_reset_deterministic.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_vec_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int]
        L.ora_vec_run.restype = C.c_long
        _lib = L
    return _lib


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().ora_philox4x32_10(c, k, o)
    return list(o)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleVec:
    """`for env in envs: env.step(a)` over n scalar restatements, SoA numpy in/out."""

    def __init__(self, kind, n_envs, seed=0, env_id_base=0, sutton_barto_reward=False, is_euler=True,
                 goal_velocity=0.0, gravity=-10.0, enable_wind=False, wind_power=15.0, turbulence_power=1.5):
        self.kind, self.n = kind, int(n_envs)
        cfg = VecConfig(kind, self.n, env_id_base, seed, int(sutton_barto_reward), int(is_euler), goal_velocity,
                        gravity, int(enable_wind), wind_power, turbulence_power)
        st = C.c_int(0)
        self._h = lib().ora_vec_new(C.byref(cfg), C.byref(st))
        if not self._h:
            raise ValueError(f"ora_vec_new failed: status {st.value}")
        self.obs_dim = lib().ora_vec_obs_dim(self._h)
        self._disp = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ora_vec_free(self._h)
            self._h = None

    def reset(self, mask=None, nthreads=1):
        obs = np.zeros((self.obs_dim, self.n), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        st = lib().ora_vec_reset(self._h, _ptr(m), _ptr(obs), nthreads)
        if st != OK:
            raise RuntimeError(f"oracle reset status {st}")
        return obs

    def reset_deterministic(self):
        obs = np.zeros((self.obs_dim, self.n), np.float32)
        st = lib().ora_vec_reset_deterministic(self._h, _ptr(obs))
        if st != OK:
            raise RuntimeError(f"oracle reset_deterministic status {st}")
        return obs

    def step(self, actions, nthreads=1, out=None):
        a = np.ascontiguousarray(actions, np.float32 if self.kind == MOUNTAINCAR_CONT else np.uint32)
        assert a.shape == (self.n,)
        if out is None:
            out = (np.zeros((self.obs_dim, self.n), np.float32), np.zeros(self.n, np.float32),
                   np.zeros(self.n, np.uint8), np.zeros(self.n, np.uint8))
        obs, rew, done, trunc = out
        st = lib().ora_vec_step(self._h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done), _ptr(trunc), nthreads)
        if st == INVALID_ACTION:
            raise ValueError("invalid action")
        if st == NOT_RESET:
            raise RuntimeError("You forgot to call reset()")
        if st != OK:
            raise RuntimeError(f"oracle step status {st}")
        return obs, rew, done, trunc

    def run(self, actions, K, nthreads=1):
        """K steps with reset-on-finish inside the C library (actions [ring, n], cycled); returns episodes finished."""
        a = np.ascontiguousarray(actions, np.float32 if self.kind == MOUNTAINCAR_CONT else np.uint32)
        assert a.ndim == 2 and a.shape[1] == self.n
        r = lib().ora_vec_run(self._h, _ptr(a), a.shape[0], int(K), nthreads)
        if r < 0:
            raise RuntimeError(f"oracle run status {-r}")
        return r

    @property
    def state_cols(self):
        return lib().ora_vec_state_cols(self._h)

    def get_state(self):
        s = np.zeros((self.state_cols, self.n), np.float32)
        lib().ora_vec_get_state(self._h, _ptr(s))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, np.float32)
        assert s.shape == (self.state_cols, self.n)
        lib().ora_vec_set_state(self._h, _ptr(s))

    def set_dispersion(self, disp):
        self._disp = None if disp is None else np.ascontiguousarray(disp, np.float32)
        lib().ora_vec_set_dispersion(self._h, _ptr(self._disp))
