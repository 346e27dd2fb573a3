//! `modurl::gym::Gym` for CartPoleV1 / MountainCarV0 / LunarLanderV3, backed by libmgym (MI355X HIP kernels).
//!
//! Drop-in for the reference's environments: same constructor parameters, same `reset()` / `step()`
//! signatures and `StepInfo`, same panics (the engine reports integer statuses; this shim turns the
//! ones the reference asserts on back into panics).  A single `Gym` is a batch of one — correct but
//! slow (one kernel launch + one small copy per call); training loops that want the GPU's
//! throughput use `VecGym` directly.
//!
//! Random streams differ from the reference (counter-based Philox keyed by (seed, env id, episode)
//! instead of ThreadRng/StdRng): reset states are drawn from the same distributions but are not the
//! same numbers.  The reference's tests never pin a drawn value.
pub mod sys;
pub mod vec_gym;

use candle_core::{Device, Tensor};
use modurl::{
    gym::{Gym, StepInfo},
    spaces::{self, Space},
};
use std::os::raw::c_void;
use sys::*;
pub use vec_gym::{DeviceBuf, MgymError, StepBuffers, VecGym};

/// State shared by the three single-environment wrappers.
struct Single {
    vec: VecGym,
    out: StepBuffers,
    action: DeviceBuf<u32>,
}

impl Single {
    fn new(cfg: mgym_config) -> Self {
        // mgym_create → MGYM_ERR_BAD_CONFIG is the reference's constructor assert (lunar_lander.rs:292-296)
        let vec = VecGym::new(cfg).unwrap_or_else(|e| panic!("{e}"));
        let out = vec.buffers().expect("device buffers");
        let action = DeviceBuf::new(cfg.device, 1).expect("device buffers");
        Self { vec, out, action }
    }

    fn obs_tensor(&self) -> Result<Tensor, candle_core::Error> {
        let dim = self.vec.spec.obs_dim as usize;
        let mut host = vec![0f32; dim];
        self.out.obs.download(&mut host).map_err(candle_core::Error::wrap)?;
        Tensor::from_vec(host, vec![dim], &Device::Cpu)
    }

    fn reset(&mut self) -> Result<Tensor, candle_core::Error> {
        self.vec.reset(None, Some(&mut self.out.obs)).map_err(candle_core::Error::wrap)?;
        self.vec.sync().map_err(candle_core::Error::wrap)?;
        self.obs_tensor()
    }

    fn step(&mut self, action: Tensor) -> Result<StepInfo, candle_core::Error> {
        // rank-0 u32, like the reference: a [1]-shaped action is not in Discrete(n) (cartpole.rs:392-403)
        assert!(action.rank() == 0, "action is not contained in the action space");
        let a = action.to_vec0::<u32>()?;
        self.action.upload(&[a]).map_err(candle_core::Error::wrap)?;
        self.vec.step(self.action.as_ptr() as *const c_void, &mut self.out).map_err(candle_core::Error::wrap)?;
        match self.vec.sync() {
            Ok(()) => {}
            // cartpole.rs:252, mountain_car.rs:294: assert!(self.action_space.contains(&action))
            Err(e) if e.status == MGYM_ERR_INVALID_ACTION => panic!("action is not contained in the action space"),
            // lunar_lander.rs:920
            Err(e) if e.status == MGYM_ERR_NOT_RESET => panic!("You forgot to call reset()"),
            Err(e) => return Err(candle_core::Error::wrap(e)),
        }
        let (mut reward, mut done, mut truncated) = ([0f32], [0u8], [0u8]);
        self.out.reward.download(&mut reward).map_err(candle_core::Error::wrap)?;
        self.out.done.download(&mut done).map_err(candle_core::Error::wrap)?;
        self.out.truncated.download(&mut truncated).map_err(candle_core::Error::wrap)?;
        Ok(StepInfo { state: self.obs_tensor()?, reward: reward[0], done: done[0] != 0, truncated: truncated[0] != 0 })
    }

    fn observation_space(&self) -> spaces::BoxSpace {
        let dim = self.vec.spec.obs_dim as usize;
        let low = Tensor::from_vec(self.vec.spec.obs_low[..dim].to_vec(), vec![dim], &Device::Cpu).expect("Failed to create tensor.");
        let high = Tensor::from_vec(self.vec.spec.obs_high[..dim].to_vec(), vec![dim], &Device::Cpu).expect("Failed to create tensor.");
        spaces::BoxSpace::new(low, high)
    }

    fn action_space(&self) -> spaces::Discrete {
        spaces::Discrete::new(self.vec.spec.n_actions as usize)
    }
}

macro_rules! impl_gym {
    ($name:ident) => {
        impl Gym for $name {
            type Error = candle_core::Error;
            type SpaceError = candle_core::Error;

            fn reset(&mut self) -> Result<Tensor, Self::Error> {
                self.inner.reset()
            }
            fn step(&mut self, action: Tensor) -> Result<StepInfo, Self::Error> {
                self.inner.step(action)
            }
            fn observation_space(&self) -> Box<dyn Space<Error = Self::SpaceError>> {
                Box::new(self.inner.observation_space())
            }
            fn action_space(&self) -> Box<dyn Space<Error = Self::SpaceError>> {
                Box::new(self.inner.action_space())
            }
        }
    };
}

/// CartPole-v1 (reference: src/classic_control/cartpole.rs:13-357).
pub struct CartPoleV1 {
    inner: Single,
}

impl CartPoleV1 {
    /// `gpu`: HIP device ordinal.  Defaults of the reference's builder: sutton_barto_reward = false, is_euler = true.
    pub fn new(gpu: i32, sutton_barto_reward: bool, is_euler: bool, seed: u64) -> Self {
        let mut cfg = VecGym::default_config(MGYM_CARTPOLE);
        cfg.device = gpu;
        cfg.n_envs = 1;
        cfg.seed = seed;
        cfg.sutton_barto_reward = sutton_barto_reward as i32;
        cfg.is_euler = is_euler as i32;
        Self { inner: Single::new(cfg) }
    }
}
impl_gym!(CartPoleV1);

/// MountainCar-v0 (reference: src/classic_control/mountain_car.rs:10-339).
pub struct MountainCarV0 {
    inner: Single,
}

impl MountainCarV0 {
    pub fn new(gpu: i32, goal_velocity: f32, seed: u64) -> Self {
        let mut cfg = VecGym::default_config(MGYM_MOUNTAINCAR);
        cfg.device = gpu;
        cfg.n_envs = 1;
        cfg.seed = seed;
        cfg.goal_velocity = goal_velocity;
        Self { inner: Single::new(cfg) }
    }
}
impl_gym!(MountainCarV0);

/// LunarLander-v3 (reference: src/box_2d/lunar_lander.rs:232-1201).
pub struct LunarLanderV3 {
    inner: Single,
}

impl LunarLanderV3 {
    /// Reference defaults: gravity −10, wind off, wind_power 15, turbulence_power 1.5 (lunar_lander.rs:278-291).
    /// Panics when gravity is outside (−12, 0), like the reference (lunar_lander.rs:292-296).
    pub fn new(gpu: i32, gravity: f32, enable_wind: bool, wind_power: f32, turbulence_power: f32, seed: u64) -> Self {
        let mut cfg = VecGym::default_config(MGYM_LUNARLANDER);
        cfg.device = gpu;
        cfg.n_envs = 1;
        cfg.seed = seed;
        cfg.gravity = gravity;
        cfg.enable_wind = enable_wind as i32;
        cfg.wind_power = wind_power;
        cfg.turbulence_power = turbulence_power;
        Self { inner: Single::new(cfg) }
    }
}
impl_gym!(LunarLanderV3);

#[cfg(test)]
mod tests {
    //! The reference's unit tests, re-stated against the shim (cartpole.rs:365-434, mountain_car.rs:347-400,
    //! lunar_lander.rs:1557-1606).  They need an MI355X and libmgym.so on the loader path.
    use super::*;

    #[test]
    fn cartpole_first_step_reward_one_not_done() {
        let mut env = CartPoleV1::new(0, false, true, 0);
        let state = env.reset().unwrap();
        assert_eq!(state.shape().dims(), &[4]);
        let info = env.step(Tensor::new(0u32, &Device::Cpu).unwrap()).unwrap();
        assert_eq!(info.state.shape().dims(), &[4]);
        assert_eq!(info.reward, 1.0);
        assert!(!info.done);
    }

    #[test]
    #[should_panic]
    fn cartpole_rejects_vector_action() {
        let mut env = CartPoleV1::new(0, false, true, 0);
        env.reset().unwrap();
        let _ = env.step(Tensor::from_vec(vec![0u32], vec![1], &Device::Cpu).unwrap());
    }

    #[test]
    fn cartpole_constant_push_ends_episode() {
        let mut env = CartPoleV1::new(0, false, true, 0);
        env.reset().unwrap();
        let mut done = false;
        for _ in 0..=50 {
            done = env.step(Tensor::new(1u32, &Device::Cpu).unwrap()).unwrap().done;
            if done {
                break;
            }
        }
        assert!(done);
    }

    #[test]
    #[should_panic]
    fn mountain_car_rejects_action_three() {
        let mut env = MountainCarV0::new(0, 0.0, 0);
        env.reset().unwrap();
        let _ = env.step(Tensor::new(3u32, &Device::Cpu).unwrap());
    }

    #[test]
    fn lunar_lander_shapes_and_finite_rewards() {
        let mut env = LunarLanderV3::new(0, -10.0, true, 15.0, 1.5, 7);
        assert_eq!(env.reset().unwrap().shape().dims(), &[8]);
        for a in 0..4u32 {
            let info = env.step(Tensor::new(a, &Device::Cpu).unwrap()).unwrap();
            assert!(info.reward.is_finite());
        }
    }

    #[test]
    #[should_panic(expected = "You forgot to call reset()")]
    fn lunar_lander_step_before_reset_panics() {
        let mut env = LunarLanderV3::new(0, -10.0, false, 15.0, 1.5, 0);
        let _ = env.step(Tensor::new(0u32, &Device::Cpu).unwrap());
    }
}
