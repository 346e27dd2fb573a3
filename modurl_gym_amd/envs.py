"""Host-side mirror of the reference's environment interface over the libmgym C ABI.

Reference surface (ModuRL/ModuRL_Gym, `impl Gym for ...`):
    CartPoleV1::builder().sutton_barto_reward(..).is_euler(..).build()   cartpole.rs:34-44
    MountainCarV0::builder().goal_velocity(..).build()                   mountain_car.rs:25-34
    LunarLanderV3::builder().gravity(..).enable_wind(..)...build()       lunar_lander.rs:278-291
    reset() -> Tensor, step(action) -> StepInfo{state, reward, done, truncated}
The single-env classes below keep those names, argument meanings and error behaviour
(panics become exceptions).  `VecEnv` is the batched superset the GPU engine exists for:
n_envs environments, SoA buffers, one kernel launch per reset()/step().

Python is test/bench plumbing here (the reference is Rust; its toolchain is absent from this
image, INTEGRATION.md shows the Rust shim).  The compute path is HIP only: nothing in this
package falls back to a CPU implementation.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib as L

StepInfo = namedtuple("StepInfo", ["state", "reward", "done", "truncated"])  # modurl::gym::StepInfo

KIND_NAMES = {L.CARTPOLE: "CartPoleV1", L.MOUNTAINCAR: "MountainCarV0",
              L.MOUNTAINCAR_CONT: "MountainCarContinuousV0", L.LUNARLANDER: "LunarLanderV3"}


class MgymError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"mgym status {status}: {msg}")
        self.status = status


class InvalidActionError(MgymError, ValueError):
    """assert!(self.action_space.contains(&action)) — cartpole.rs:252, mountain_car.rs:294"""


class NotResetError(MgymError):
    """assert!(self.lander.is_some(), "You forgot to call reset()") — lunar_lander.rs:920"""


class BadConfigError(MgymError, ValueError):
    """assert!(-12.0 < gravity && gravity < 0.0) — lunar_lander.rs:292-296"""


_EXC = {L.ERR_INVALID_ACTION: InvalidActionError, L.ERR_NOT_RESET: NotResetError, L.ERR_BAD_CONFIG: BadConfigError}


def _check(st):
    if st != L.OK:
        msg = L.load().mgym_last_error().decode(errors="replace")
        raise _EXC.get(st, MgymError)(st, msg)


def device_count():
    n = C.c_int(0)
    st = L.load().mgym_device_count(C.byref(n))
    return n.value if st == L.OK else 0


def get_spec(kind):
    s = L.Spec()
    _check(L.load().mgym_get_spec(kind, C.byref(s)))
    return s


class DeviceArray:
    """A typed device allocation made through mgym_malloc (for callers without torch)."""

    def __init__(self, shape, dtype, device=0):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        _check(L.load().mgym_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                L.load().mgym_free(self.device, self.ptr)
            except Exception:
                pass
            self.ptr = None

    def data_ptr(self):
        return self.ptr

    def copy_from(self, host):
        a = np.ascontiguousarray(host, self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        _check(L.load().mgym_memcpy_h2d(self.device, self.ptr, a.ctypes.data, self.nbytes))
        return self

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        _check(L.load().mgym_memcpy_d2h(self.device, out.ctypes.data, self.ptr, self.nbytes))
        return out

    @classmethod
    def from_numpy(cls, a, device=0):
        a = np.ascontiguousarray(a)
        return cls(a.shape, a.dtype, device).copy_from(a)


def _ptr(x):
    """device pointer of None / int / DeviceArray / torch tensor"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


class VecEnv:
    """n_envs environments of one family on one GPU (one mgym_env handle, one stream)."""

    def __init__(self, kind, n_envs=1, device=0, seed=0, env_id_base=0, auto_reset=False, stream=None,
                 sutton_barto_reward=False, is_euler=True, goal_velocity=0.0, gravity=-10.0, enable_wind=False,
                 wind_power=15.0, turbulence_power=1.5):
        lib = L.load()
        cfg = L.Config()
        _check(lib.mgym_default_config(kind, C.byref(cfg)))
        cfg.device, cfg.n_envs, cfg.env_id_base, cfg.seed = device, int(n_envs), int(env_id_base), int(seed)
        cfg.flags = L.FLAG_AUTO_RESET if auto_reset else 0
        cfg.sutton_barto_reward, cfg.is_euler = int(sutton_barto_reward), int(is_euler)
        cfg.goal_velocity = goal_velocity
        cfg.gravity, cfg.enable_wind = gravity, int(enable_wind)
        cfg.wind_power, cfg.turbulence_power = wind_power, turbulence_power
        h = C.c_void_p()
        self._h = None
        _check(lib.mgym_create(C.byref(cfg), C.byref(h)))
        self._h, self._lib = h, lib
        self.kind, self.n, self.device = kind, int(n_envs), device
        spec = get_spec(kind)
        self.obs_dim, self.n_actions, self.action_is_float = spec.obs_dim, spec.n_actions, bool(spec.action_is_float)
        self.action_dtype = np.float32 if self.action_is_float else np.uint32
        if stream is not None:
            self.set_stream(stream)
        self._bufs = None
        self._disp = None

    # ---- lifetime -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mgym_destroy(self._h)
            self._h = None

    __del__ = close

    def set_stream(self, hip_stream):
        _check(self._lib.mgym_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def sync(self):
        """hipStreamSynchronize + sticky device status (raises InvalidActionError / NotResetError)."""
        _check(self._lib.mgym_sync(self._h))

    def info(self):
        """mgym_get_info as a dict: kind, n_envs, GPU_MAX_HW_QUEUES as seen at creation; LunarLander adds its launch
        structure and how many of its streams were seen running at the same time (a probe launch on each)."""
        buf = C.create_string_buffer(1024)
        _check(self._lib.mgym_get_info(self._h, buf, len(buf)))
        return dict(line.split("=", 1) for line in buf.value.decode().splitlines() if "=" in line)

    # ---- raw device-pointer path (what bench.py and a Rust/C caller use) --------------
    def reset_device(self, mask=None, obs_out=None):
        _check(self._lib.mgym_reset(self._h, _ptr(mask), _ptr(obs_out)))

    def reset_done_device(self, done, trunc, obs_out=None):
        _check(self._lib.mgym_reset_done(self._h, _ptr(done), _ptr(trunc), _ptr(obs_out)))

    def step_device(self, actions, obs_out=None, reward_out=None, done_out=None, trunc_out=None):
        _check(self._lib.mgym_step(self._h, _ptr(actions), _ptr(obs_out), _ptr(reward_out), _ptr(done_out),
                                   _ptr(trunc_out)))

    def rollout_device(self, actions, K, obs_out=None, reward_out=None, done_out=None, trunc_out=None):
        _check(self._lib.mgym_rollout(self._h, _ptr(actions), int(K), _ptr(obs_out), _ptr(reward_out), _ptr(done_out),
                                      _ptr(trunc_out)))

    def rollout_uniform_device(self, policy_seed, K, actions_out=None, obs_out=None, reward_out=None, done_out=None, trunc_out=None):
        """K fused steps under the on-device uniform random policy (mgym_rollout_uniform; CartPole, MountainCar, MountainCarContinuous)."""
        _check(self._lib.mgym_rollout_uniform(self._h, int(policy_seed), int(K), _ptr(actions_out), _ptr(obs_out), _ptr(reward_out),
                                              _ptr(done_out), _ptr(trunc_out)))

    def rollout_linear_device(self, policy, K, actions_out=None, obs_out=None, reward_out=None, done_out=None, trunc_out=None):
        """K fused steps under the on-device linear policy (mgym_rollout_linear): policy = obs_dim weights + bias (CartPole: action 1 if the score is
        positive; MountainCarContinuous: the score is the force), or one such row per action (MountainCar: the index of the largest score).  Host floats."""
        n_pol = 3 * (self.obs_dim + 1) if self.kind == L.MOUNTAINCAR else self.obs_dim + 1   # MountainCar: a row of (weights, bias) per action
        pol_in = np.asarray(policy, np.float32).ravel()
        if pol_in.size != n_pol:
            raise ValueError(f"policy must hold {n_pol} floats for this family, got {pol_in.size}")
        pol = (C.c_float * n_pol)(*[float(v) for v in pol_in])
        _check(self._lib.mgym_rollout_linear(self._h, pol, int(K), _ptr(actions_out), _ptr(obs_out), _ptr(reward_out), _ptr(done_out),
                                             _ptr(trunc_out)))

    def rollout_linear(self, policy, K):
        """host-array form: -> (actions [K, n], obs [K, obs_dim, n], reward [K, n], done, trunc)"""
        n = max(self.n, 1)
        da = DeviceArray((K, n), np.uint32, self.device)
        do = DeviceArray((K, self.obs_dim, n), np.float32, self.device)
        dr = DeviceArray((K, n), np.float32, self.device)
        dd, dt = DeviceArray((K, n), np.uint8, self.device), DeviceArray((K, n), np.uint8, self.device)
        self.rollout_linear_device(policy, K, da, do, dr, dd, dt)
        self.sync()
        sh = lambda x, shape: x.numpy().reshape(-1)[: int(np.prod(shape))].reshape(shape)
        return (sh(da, (K, self.n)), sh(do, (K, self.obs_dim, self.n)), sh(dr, (K, self.n)), sh(dd, (K, self.n)), sh(dt, (K, self.n)))

    def rollout_uniform(self, policy_seed, K):
        """host-array form: -> (actions [K, n], obs [K, obs_dim, n], reward [K, n], done, trunc)"""
        n = max(self.n, 1)
        da = DeviceArray((K, n), np.uint32, self.device)
        do = DeviceArray((K, self.obs_dim, n), np.float32, self.device)
        dr = DeviceArray((K, n), np.float32, self.device)
        dd, dt = DeviceArray((K, n), np.uint8, self.device), DeviceArray((K, n), np.uint8, self.device)
        self.rollout_uniform_device(policy_seed, K, da, do, dr, dd, dt)
        self.sync()
        sh = lambda x, shape: x.numpy().reshape(-1)[: int(np.prod(shape))].reshape(shape)
        return (sh(da, (K, self.n)), sh(do, (K, self.obs_dim, self.n)), sh(dr, (K, self.n)), sh(dd, (K, self.n)), sh(dt, (K, self.n)))

    def episode_count(self):
        """env-steps of this handle that returned done or truncated since creation (mgym_episode_count; synchronises)"""
        c = C.c_uint64()
        _check(self._lib.mgym_episode_count(self._h, C.byref(c)))
        return c.value

    def _coerce_actions(self, actions):
        """Discrete envs take integer actions only (reference: `action_space.contains(&action)` needs a u32 tensor,
        cartpole.rs:252,377): floats and bools are rejected instead of being truncated; negative integers become
        out-of-range u32 values and fail on the device like any invalid action."""
        a = np.asarray(actions)
        if not self.action_is_float and a.dtype.kind not in "iu":
            raise InvalidActionError(L.ERR_INVALID_ACTION, f"discrete actions must be integers, got dtype {a.dtype}")
        return np.ascontiguousarray(a, self.action_dtype)

    def rollout(self, actions):
        """K fused steps on host arrays: actions [K, n] -> (obs [K, obs_dim, n], reward [K, n], done, trunc)."""
        a = self._coerce_actions(actions)
        K = a.shape[0]
        assert a.shape == (K, self.n)
        n = max(self.n, 1)
        da = DeviceArray.from_numpy(a, self.device) if a.size else DeviceArray(1, self.action_dtype, self.device)
        do = DeviceArray((K, self.obs_dim, n), np.float32, self.device)
        dr = DeviceArray((K, n), np.float32, self.device)
        dd, dt = DeviceArray((K, n), np.uint8, self.device), DeviceArray((K, n), np.uint8, self.device)
        self.rollout_device(da, K, do, dr, dd, dt)
        self.sync()
        sh = lambda x, shape: x.numpy().reshape(-1)[: int(np.prod(shape))].reshape(shape)
        return (sh(do, (K, self.obs_dim, self.n)), sh(dr, (K, self.n)), sh(dd, (K, self.n)), sh(dt, (K, self.n)))

    def observation_device(self):
        p, stride = C.c_void_p(), C.c_uint64()
        _check(self._lib.mgym_observation(self._h, C.byref(p), C.byref(stride)))
        return p.value, stride.value

    def timer_start(self):
        _check(self._lib.mgym_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        _check(self._lib.mgym_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def graph_capture(self, fn):
        """Capture the launches fn() issues on this env's stream into a hipGraphExec."""
        _check(self._lib.mgym_graph_begin(self._h))
        try:
            fn()
        finally:
            g = C.c_void_p()
            st = self._lib.mgym_graph_end(self._h, C.byref(g))
        _check(st)
        return g

    def graph_launch(self, g):
        _check(self._lib.mgym_graph_launch(self._h, g))

    def graph_destroy(self, g):
        _check(self._lib.mgym_graph_destroy(g))

    # ---- host-array convenience path (tests) ----------------------------------------
    def _buffers(self):
        if self._bufs is None:
            n = max(self.n, 1)
            self._bufs = dict(act=DeviceArray(n, self.action_dtype, self.device),
                              obs=DeviceArray((self.obs_dim, n), np.float32, self.device),
                              rew=DeviceArray(n, np.float32, self.device), done=DeviceArray(n, np.uint8, self.device),
                              trunc=DeviceArray(n, np.uint8, self.device), mask=DeviceArray(n, np.uint8, self.device))
        return self._bufs

    def _host(self, key, shape):
        return self._bufs[key].numpy().reshape(-1)[: int(np.prod(shape))].reshape(shape)

    def reset(self, mask=None):
        b = self._buffers()
        if mask is not None:
            m = np.ascontiguousarray(mask, np.uint8)
            assert m.shape == (self.n,)
            if self.n:
                b["mask"].copy_from(m)
            self.reset_device(b["mask"], b["obs"])
        else:
            self.reset_device(None, b["obs"])
        self.sync()
        return self._host("obs", (self.obs_dim, self.n))

    def reset_deterministic(self):
        """Testable::reset_deterministic of the reference's test modules, on every env."""
        b = self._buffers()
        _check(self._lib.mgym_reset_deterministic(self._h, b["obs"].ptr))
        self.sync()
        return self._host("obs", (self.obs_dim, self.n))

    def step(self, actions):
        b = self._buffers()
        a = self._coerce_actions(actions)
        assert a.shape == (self.n,), f"actions must be [{self.n}], got {a.shape}"
        if self.n:
            b["act"].copy_from(a)
        self.step_device(b["act"], b["obs"], b["rew"], b["done"], b["trunc"])
        self.sync()
        return (self._host("obs", (self.obs_dim, self.n)), self._host("rew", (self.n,)),
                self._host("done", (self.n,)), self._host("trunc", (self.n,)))

    def observation(self):
        """copy of the engine-owned observation (zero-copy pointer: observation_device())"""
        p, stride = self.observation_device()
        out = np.empty((self.obs_dim, self.n), np.float32)
        for k in range(self.obs_dim):
            if self.n:
                _check(self._lib.mgym_memcpy_d2h(self.device, out[k].ctypes.data, p + 4 * k * stride, 4 * self.n))
        return out

    def observation_aos(self):
        """row-major [n, obs_dim] copy of the current observation (mgym_observation_aos)"""
        d = DeviceArray((max(self.n, 1), self.obs_dim), np.float32, self.device)
        _check(self._lib.mgym_observation_aos(self._h, d.ptr))
        self.sync()
        return d.numpy()[: self.n]

    @property
    def state_cols(self):
        return get_spec(self.kind).state_cols

    def get_state(self):
        cols = self.state_cols
        d = DeviceArray((cols, max(self.n, 1)), np.float32, self.device)
        _check(self._lib.mgym_get_state(self._h, d.ptr))
        self.sync()
        return d.numpy().reshape(-1)[: cols * self.n].reshape(cols, self.n)

    def set_state(self, blob):
        b = np.ascontiguousarray(blob, np.float32)
        assert b.shape == (self.state_cols, self.n)
        if self.n == 0:
            return
        d = DeviceArray.from_numpy(b, self.device)
        _check(self._lib.mgym_set_state(self._h, d.ptr))
        self.sync()

    def set_dispersion(self, disp):
        if disp is None:
            self._disp = None
            _check(self._lib.mgym_set_dispersion_override(self._h, None))
        else:
            a = np.ascontiguousarray(disp, np.float32)
            assert a.shape == (2, self.n)
            self._disp = DeviceArray.from_numpy(a, self.device)
            _check(self._lib.mgym_set_dispersion_override(self._h, self._disp.ptr))


class _SingleEnv:
    """n_envs = 1 view with the reference's scalar signature: reset() -> state, step(a) -> StepInfo."""

    KIND = None

    def __init__(self, device=0, seed=0, **builder_args):
        self.vec = VecEnv(self.KIND, 1, device=device, seed=seed, **builder_args)

    def reset(self):
        return self.vec.reset()[:, 0].copy()

    def step(self, action):
        a = np.asarray(action)
        if a.ndim != 0:  # cartpole.rs:392-403: a [1]-shaped action tensor is rejected (rank-0 required)
            raise InvalidActionError(L.ERR_INVALID_ACTION, f"action must be a scalar, got shape {a.shape}")
        obs, rew, done, trunc = self.vec.step(a.reshape(1))
        return StepInfo(obs[:, 0].copy(), float(rew[0]), bool(done[0]), bool(trunc[0]))

    def observation_space(self):
        s = get_spec(self.KIND)
        return np.array(s.obs_low[: s.obs_dim], np.float32), np.array(s.obs_high[: s.obs_dim], np.float32)

    def action_space(self):
        s = get_spec(self.KIND)
        return ("Box", s.action_low, s.action_high) if s.action_is_float else ("Discrete", s.n_actions)

    def close(self):
        self.vec.close()


class CartPoleV1(_SingleEnv):
    """cartpole.rs:13-357"""
    KIND = L.CARTPOLE

    def __init__(self, device=0, sutton_barto_reward=False, is_euler=True, seed=0):
        super().__init__(device, seed, sutton_barto_reward=sutton_barto_reward, is_euler=is_euler)


class MountainCarV0(_SingleEnv):
    """mountain_car.rs:10-339"""
    KIND = L.MOUNTAINCAR

    def __init__(self, device=0, goal_velocity=0.0, seed=0):
        super().__init__(device, seed, goal_velocity=goal_velocity)


class MountainCarContinuousV0(_SingleEnv):
    """Not in the reference (BASELINE config 3); gymnasium semantics, parity unpinned."""
    KIND = L.MOUNTAINCAR_CONT

    def __init__(self, device=0, goal_velocity=0.0, seed=0):
        super().__init__(device, seed, goal_velocity=goal_velocity)


class LunarLanderV3(_SingleEnv):
    """lunar_lander.rs:232-1201"""
    KIND = L.LUNARLANDER

    def __init__(self, device=0, gravity=-10.0, enable_wind=False, wind_power=15.0, turbulence_power=1.5, seed=0):
        super().__init__(device, seed, gravity=gravity, enable_wind=enable_wind, wind_power=wind_power,
                         turbulence_power=turbulence_power)
