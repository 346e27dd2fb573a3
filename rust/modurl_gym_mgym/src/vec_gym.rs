//! Batched superset of `Gym`: n_envs environments behind one handle, device buffers only.
//! The reference cannot express a batch (StepInfo carries scalars, cartpole.rs:300-305); this is the
//! surface the GPU engine exists for.  Semantics of each call: include/mgym.h.
use crate::sys::*;
use std::ffi::CStr;
use std::os::raw::c_void;

#[derive(Debug)]
pub struct MgymError {
    pub status: i32,
    pub message: String,
}

impl std::fmt::Display for MgymError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "mgym status {}: {}", self.status, self.message)
    }
}
impl std::error::Error for MgymError {}

pub(crate) fn check(status: i32) -> Result<(), MgymError> {
    if status == MGYM_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(mgym_last_error()) }.to_string_lossy().into_owned();
    Err(MgymError { status, message })
}

/// A typed allocation in the env's GPU memory (mgym_malloc / mgym_free).
pub struct DeviceBuf<T: Copy> {
    ptr: *mut T,
    len: usize,
    device: i32,
}

impl<T: Copy> DeviceBuf<T> {
    pub fn new(device: i32, len: usize) -> Result<Self, MgymError> {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(unsafe { mgym_malloc(device, len.max(1) * std::mem::size_of::<T>(), &mut p) })?;
        Ok(Self { ptr: p as *mut T, len, device })
    }
    pub fn as_ptr(&self) -> *mut T {
        self.ptr
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn upload(&mut self, host: &[T]) -> Result<(), MgymError> {
        assert_eq!(host.len(), self.len);
        check(unsafe { mgym_memcpy_h2d(self.device, self.ptr as _, host.as_ptr() as _, std::mem::size_of_val(host)) })
    }
    pub fn download(&self, host: &mut [T]) -> Result<(), MgymError> {
        assert_eq!(host.len(), self.len);
        check(unsafe { mgym_memcpy_d2h(self.device, host.as_mut_ptr() as _, self.ptr as _, std::mem::size_of_val(host)) })
    }
}

impl<T: Copy> Drop for DeviceBuf<T> {
    fn drop(&mut self) {
        unsafe { mgym_free(self.device, self.ptr as _) };
    }
}

/// Output buffers of one step: SoA observation `[obs_dim][n]`, reward, done, truncated.
pub struct StepBuffers {
    pub obs: DeviceBuf<f32>,
    pub reward: DeviceBuf<f32>,
    pub done: DeviceBuf<u8>,
    pub truncated: DeviceBuf<u8>,
}

pub struct VecGym {
    env: *mut mgym_env,
    pub cfg: mgym_config,
    pub spec: mgym_spec,
}

impl VecGym {
    pub fn default_config(kind: i32) -> mgym_config {
        let mut cfg: mgym_config = unsafe { std::mem::zeroed() };
        check(unsafe { mgym_default_config(kind, &mut cfg) }).expect("mgym_default_config");
        cfg
    }

    pub fn new(cfg: mgym_config) -> Result<Self, MgymError> {
        assert_eq!(unsafe { mgym_abi_version() }, MGYM_ABI_VERSION, "libmgym ABI mismatch");
        let mut env: *mut mgym_env = std::ptr::null_mut();
        check(unsafe { mgym_create(&cfg, &mut env) })?;
        let mut spec: mgym_spec = unsafe { std::mem::zeroed() };
        check(unsafe { mgym_get_spec(cfg.kind, &mut spec) })?;
        Ok(Self { env, cfg, spec })
    }

    pub fn n_envs(&self) -> usize {
        self.cfg.n_envs as usize
    }

    pub fn buffers(&self) -> Result<StepBuffers, MgymError> {
        let (n, d) = (self.n_envs(), self.cfg.device);
        Ok(StepBuffers {
            obs: DeviceBuf::new(d, n * self.spec.obs_dim as usize)?,
            reward: DeviceBuf::new(d, n)?,
            done: DeviceBuf::new(d, n)?,
            truncated: DeviceBuf::new(d, n)?,
        })
    }

    /// Enqueue later launches on a caller-owned HIP stream (e.g. candle's / the trainer's).
    pub fn set_stream(&mut self, hip_stream: *mut c_void) -> Result<(), MgymError> {
        check(unsafe { mgym_set_stream(self.env, hip_stream) })
    }

    /// `mask`: device `u8[n]` (non-zero = reset) or None for all.
    pub fn reset(&mut self, mask: Option<&DeviceBuf<u8>>, obs_out: Option<&mut DeviceBuf<f32>>) -> Result<(), MgymError> {
        let m = mask.map_or(std::ptr::null(), |b| b.as_ptr() as *const u8);
        let o = obs_out.map_or(std::ptr::null_mut(), |b| b.as_ptr());
        check(unsafe { mgym_reset(self.env, m, o) })
    }

    /// Reset exactly the envs the last step reported finished (done | truncated).
    pub fn reset_done(&mut self, out: &StepBuffers) -> Result<(), MgymError> {
        check(unsafe { mgym_reset_done(self.env, out.done.as_ptr(), out.truncated.as_ptr(), std::ptr::null_mut()) })
    }

    /// `actions`: device `u32[n]` (f32[n] for MountainCarContinuous).  Asynchronous on the env's stream.
    pub fn step(&mut self, actions: *const c_void, out: &mut StepBuffers) -> Result<(), MgymError> {
        check(unsafe {
            mgym_step(self.env, actions, out.obs.as_ptr(), out.reward.as_ptr(), out.done.as_ptr(), out.truncated.as_ptr())
        })
    }

    /// K fused steps for a caller-supplied action table `[K][n]`; outputs are `[K]`-leading.
    pub fn rollout(
        &mut self,
        actions: *const c_void,
        k: i32,
        obs: *mut f32,
        reward: *mut f32,
        done: *mut u8,
        truncated: *mut u8,
    ) -> Result<(), MgymError> {
        check(unsafe { mgym_rollout(self.env, actions, k, obs, reward, done, truncated) })
    }

    /// K fused steps under an on-device linear policy (`mgym_rollout_linear`): the loop `let a = policy(&obs); env.step(a)` of
    /// cartpole.rs:251-348 / mountain_car.rs:293-330 without leaving the GPU.  `policy`: CartPole 4 weights + bias (action 1 if the score
    /// is positive); MountainCar three rows of (w_position, w_velocity, bias), the action is the index of the largest score;
    /// MountainCarContinuous one such row, the score is the force.
    /// # Safety
    /// the output pointers are device pointers of `[K][n]` (`[K][obs_dim][n]` for `obs`) elements, or null.
    pub unsafe fn rollout_linear(
        &mut self,
        policy: &[f32],
        k: i32,
        actions_out: *mut c_void,
        obs: *mut f32,
        reward: *mut f32,
        done: *mut u8,
        truncated: *mut u8,
    ) -> Result<(), MgymError> {
        let want: usize = if self.cfg.kind == MGYM_MOUNTAINCAR { 9 } else { (self.spec.obs_dim + 1) as usize };
        if policy.len() != want {
            return Err(MgymError { status: MGYM_ERR_BAD_ARG, message: format!("rollout_linear: {} policy floats for this family, got {}", want, policy.len()) });
        }
        check(unsafe { mgym_rollout_linear(self.env, policy.as_ptr(), k, actions_out, obs, reward, done, truncated) })
    }
    /// K fused steps under the on-device uniform random policy (`mgym_rollout_uniform`; CartPole, MountainCar, MountainCarContinuous): no action table.
    /// `actions_out` (`[K][n]` u32, may be null) receives the drawn actions.
    #[allow(clippy::too_many_arguments)]
    pub fn rollout_uniform(
        &mut self,
        policy_seed: u64,
        k: i32,
        actions_out: *mut c_void,
        obs: *mut f32,
        reward: *mut f32,
        done: *mut u8,
        truncated: *mut u8,
    ) -> Result<(), MgymError> {
        check(unsafe { mgym_rollout_uniform(self.env, policy_seed, k, actions_out, obs, reward, done, truncated) })
    }

    /// Env-steps that returned `done` or `truncated` since creation (ballot/popcount reduction inside the step
    /// kernels); synchronises the stream.
    pub fn episode_count(&mut self) -> Result<u64, MgymError> {
        let mut c = 0u64;
        check(unsafe { mgym_episode_count(self.env, &mut c) })?;
        Ok(c)
    }

    /// Zero-copy view of the engine-owned observation columns: (pointer, column stride in floats).
    pub fn observation(&self) -> Result<(*const f32, u64), MgymError> {
        let (mut p, mut stride) = (std::ptr::null(), 0u64);
        check(unsafe { mgym_observation(self.env, &mut p, &mut stride) })?;
        Ok((p, stride))
    }

    /// Wait for the stream and surface the sticky device status (invalid action, step before reset).
    pub fn sync(&mut self) -> Result<(), MgymError> {
        check(unsafe { mgym_sync(self.env) })
    }

    // test seams (≙ `Testable`, testing.rs:15-18; `deterministic_mode`, lunar_lander.rs:967-970)
    pub fn set_state(&mut self, blob: &DeviceBuf<f32>) -> Result<(), MgymError> {
        check(unsafe { mgym_set_state(self.env, blob.as_ptr() as _) })
    }
    pub fn get_state(&mut self, blob: &mut DeviceBuf<f32>) -> Result<(), MgymError> {
        check(unsafe { mgym_get_state(self.env, blob.as_ptr() as _) })
    }
    pub fn reset_deterministic(&mut self, obs_out: &mut DeviceBuf<f32>) -> Result<(), MgymError> {
        check(unsafe { mgym_reset_deterministic(self.env, obs_out.as_ptr()) })
    }
    pub fn set_dispersion_override(&mut self, disp: Option<&DeviceBuf<f32>>) -> Result<(), MgymError> {
        check(unsafe { mgym_set_dispersion_override(self.env, disp.map_or(std::ptr::null(), |b| b.as_ptr() as *const f32)) })
    }
}

impl Drop for VecGym {
    fn drop(&mut self) {
        unsafe { mgym_destroy(self.env) };
    }
}
