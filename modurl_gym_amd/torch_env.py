"""Torch-ROCm interop for the batched engine (the step either side of the path).

A policy network consumes observations and produces `action[n]`; this module lets that loop run
without leaving the GPU or the torch stream:

    env = TorchVecEnv(mg.CARTPOLE, 1 << 20, auto_reset=True)
    obs = env.reset()                                  # torch.float32 [obs_dim, n]  (SoA, like the engine)
    for _ in range(T):
        actions = policy(obs)                          # torch.int32 [n]  (float32 for MountainCarContinuous)
        obs, reward, done, truncated = env.step(actions)
    env.check()                                        # surfaces the sticky device status (invalid action ...)

Nothing here computes environment physics: tensors are handed to the C ABI by `data_ptr()` and
the engine's launches are enqueued on torch's *current* stream, so they order with the policy's
kernels exactly like torch ops do (no host synchronisation in `step`).  Torch is plumbing (device
memory + streams), as in bench.py.  The reference has no counterpart: its `Gym::step` takes and
returns host `candle` tensors one environment at a time (cartpole.rs:251-305).

Layouts: `obs_layout="soa"` returns `[obs_dim, n]` (the engine's native column layout, what
`mgym_step` writes); `"aos"` additionally runs `mgym_observation_aos` and returns `[n, obs_dim]`
row-major, the shape an `nn.Linear` policy wants.  `observation_view()` is a zero-copy tensor
over the engine-owned columns (`mgym_observation`): valid until the next call that advances them.
"""
import numpy as np

from . import envs as E


class _DeviceSpan:
    """Engine-owned device memory exposed through __cuda_array_interface__ (torch reads it under ROCm too)."""

    def __init__(self, ptr, shape, strides_bytes, owner):
        self._owner = owner  # keep the env alive while views exist
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": tuple(strides_bytes)}


class TorchVecEnv:
    """`VecEnv` whose inputs/outputs are torch tensors on the env's GPU, stream-ordered with torch."""

    def __init__(self, kind, n_envs, device=None, seed=0, env_id_base=0, auto_reset=True, obs_layout="soa", **params):
        import torch
        if not torch.cuda.is_available():
            raise E.MgymError(E.L.ERR_NO_DEVICE, "TorchVecEnv needs a GPU (there is no CPU path)")
        if obs_layout not in ("soa", "aos"):
            raise ValueError("obs_layout must be 'soa' or 'aos'")
        self._torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.env = E.VecEnv(kind, n_envs, device=self.device.index, seed=seed, env_id_base=env_id_base,
                            auto_reset=auto_reset, **params)
        self.kind, self.n, self.obs_dim = kind, self.env.n, self.env.obs_dim
        self.n_actions, self.action_is_float = self.env.n_actions, self.env.action_is_float
        self.obs_layout = obs_layout
        self._stream = None
        f32, u8 = torch.float32, torch.uint8
        self._obs = torch.empty((self.obs_dim, self.n), dtype=f32, device=self.device)
        self._obs_aos = torch.empty((self.n, self.obs_dim), dtype=f32, device=self.device) if obs_layout == "aos" else None
        self._rew = torch.empty(self.n, dtype=f32, device=self.device)
        self._done = torch.empty(self.n, dtype=u8, device=self.device)
        self._trunc = torch.empty(self.n, dtype=u8, device=self.device)

    # ---- plumbing -----------------------------------------------------------------------
    def _bind_stream(self):
        s = self._torch.cuda.current_stream(self.device).cuda_stream
        if s != self._stream:
            self.env.set_stream(s)
            self._stream = s

    def _actions(self, a, lead=()):
        torch = self._torch
        if not isinstance(a, torch.Tensor):
            raise TypeError("actions must be a torch tensor on the env's device")
        if a.device != self.device:
            raise ValueError(f"actions are on {a.device}, the env is on {self.device}")
        if tuple(a.shape) != tuple(lead) + (self.n,):
            raise ValueError(f"actions must have shape {tuple(lead) + (self.n,)}, got {tuple(a.shape)}")
        if self.action_is_float:
            if a.dtype != torch.float32:
                a = a.to(torch.float32)
        elif a.dtype not in (torch.int32, torch.uint32):
            if a.dtype.is_floating_point or a.dtype == torch.bool:
                raise TypeError(f"discrete actions must be an integer tensor, got {a.dtype}")
            a = a.to(torch.int32)  # one extra elementwise kernel; pass int32 to avoid it
        return a.contiguous()

    def _obs_out(self):
        if self._obs_aos is None:
            return self._obs
        E._check(self.env._lib.mgym_observation_aos(self.env._h, self._obs_aos.data_ptr()))
        return self._obs_aos

    def _flags(self):
        return self._done.view(self._torch.bool), self._trunc.view(self._torch.bool)

    # ---- the Gym-shaped surface ---------------------------------------------------------
    def reset(self, mask=None):
        """Reset all envs, or those with mask != 0 (torch.uint8/bool [n]); returns the observation tensor."""
        self._bind_stream()
        if mask is not None:
            torch = self._torch
            if mask.dtype == torch.bool:
                mask = mask.view(torch.uint8)
            if mask.dtype != torch.uint8 or tuple(mask.shape) != (self.n,) or mask.device != self.device:
                raise ValueError("mask must be a uint8/bool tensor of shape [n] on the env's device")
            mask = mask.contiguous()
        self.env.reset_device(mask, self._obs)
        return self._obs_out()

    def step(self, actions, check=False):
        """One step of every env.  Returns (obs, reward, done, truncated); the tensors are reused by the
        next call (clone what must outlive it).  check=True synchronises and raises on invalid actions."""
        self._bind_stream()
        a = self._actions(actions)
        self.env.step_device(a, self._obs, self._rew, self._done, self._trunc)
        if check:
            self.check()
        d, t = self._flags()
        return self._obs_out(), self._rew, d, t

    def rollout(self, actions):
        """K fused steps for a caller-supplied action tensor [K, n] (mgym_rollout): returns
        (obs [K, obs_dim, n], reward [K, n], done [K, n], truncated [K, n]) as fresh tensors."""
        torch = self._torch
        self._bind_stream()
        K = int(actions.shape[0])
        a = self._actions(actions, (K,))
        obs = torch.empty((K, self.obs_dim, self.n), dtype=torch.float32, device=self.device)
        rew = torch.empty((K, self.n), dtype=torch.float32, device=self.device)
        done = torch.empty((K, self.n), dtype=torch.uint8, device=self.device)
        trunc = torch.empty((K, self.n), dtype=torch.uint8, device=self.device)
        self.env.rollout_device(a, K, obs, rew, done, trunc)
        return obs, rew, done.view(torch.bool), trunc.view(torch.bool)

    def observation_view(self):
        """Zero-copy [obs_dim, n] tensor over the engine-owned observation columns (column stride n_pad)."""
        ptr, stride = self.env.observation_device()
        span = _DeviceSpan(ptr, (self.obs_dim, self.n), (4 * stride, 4), self)
        return self._torch.as_tensor(span, device=self.device)

    def check(self):
        """hipStreamSynchronize + sticky device status (raises InvalidActionError / NotResetError)."""
        self.env.sync()

    def close(self):
        self.env.close()

    def state_numpy(self):
        return np.array(self.env.get_state())
