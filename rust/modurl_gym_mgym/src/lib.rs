//! `modurl::gym::Gym` for CartPoleV1 / MountainCarV0 / LunarLanderV3, backed by libmgym (MI355X HIP kernels).
//!
//! Drop-in for the reference's environments: the same `bon` builders with the same argument names and
//! defaults (`CartPoleV1::builder().sutton_barto_reward(true).build()`, `LunarLanderV3::builder().seed(42).build()`,
//! `impl Default`), same `reset()` / `step()` signatures and `StepInfo`, same panics (the engine reports integer
//! statuses; this shim turns the ones the reference asserts on back into panics).  A single `Gym` is a batch of
//! one — correct but slow: every `step()` is an upload, a launch, a synchronise and four small downloads
//! (tens of microseconds, against ~18 ns per step of the CPU reference); training loops that
//! want the GPU's throughput use `VecGym` directly.
//!
//! Random streams differ from the reference (counter-based Philox keyed by (seed, env id, episode)
//! instead of ThreadRng/StdRng): reset states are drawn from the same distributions but are not the
//! same numbers.  The reference's tests never pin a drawn value.
pub mod sys;
pub mod vec_gym;

use bon::bon;
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

    /// the state tensor lives on the builder's `device`, like the reference's (cartpole.rs:285-290)
    fn obs_tensor(&self, device: &Device) -> Result<Tensor, candle_core::Error> {
        let dim = self.vec.spec.obs_dim as usize;
        let mut host = vec![0f32; dim];
        self.out.obs.download(&mut host).map_err(candle_core::Error::wrap)?;
        Tensor::from_vec(host, vec![dim], device)
    }

    fn reset(&mut self, device: &Device) -> Result<Tensor, candle_core::Error> {
        self.vec.reset(None, Some(&mut self.out.obs)).map_err(candle_core::Error::wrap)?;
        self.vec.sync().map_err(candle_core::Error::wrap)?;
        self.obs_tensor(device)
    }

    fn step(&mut self, action: Tensor, device: &Device) -> Result<StepInfo, candle_core::Error> {
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
        Ok(StepInfo { state: self.obs_tensor(device)?, reward: reward[0], done: done[0] != 0, truncated: truncated[0] != 0 })
    }

    fn observation_space(&self, device: &Device) -> spaces::BoxSpace {
        let dim = self.vec.spec.obs_dim as usize;
        let low = Tensor::from_vec(self.vec.spec.obs_low[..dim].to_vec(), vec![dim], device).expect("Failed to create tensor.");
        let high = Tensor::from_vec(self.vec.spec.obs_high[..dim].to_vec(), vec![dim], device).expect("Failed to create tensor.");
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
                self.inner.reset(&self.device)
            }
            fn step(&mut self, action: Tensor) -> Result<StepInfo, Self::Error> {
                self.inner.step(action, &self.device)
            }
            fn observation_space(&self) -> Box<dyn Space<Error = Self::SpaceError>> {
                Box::new(self.inner.observation_space(&self.device))
            }
            fn action_space(&self) -> Box<dyn Space<Error = Self::SpaceError>> {
                Box::new(self.inner.action_space())
            }
        }
    };
}

/// Seed for the engine's counter-based streams: the caller's, or — like the reference's thread-local generator
/// (`EnvRng::default()`, lunar_lander.rs:71-75; candle's CPU `Tensor::rand`, cartpole.rs:240) — fresh entropy.
fn seed_or_entropy(seed: Option<u64>) -> u64 {
    seed.unwrap_or_else(|| rand::Rng::random(&mut rand::rng()))
}

/// CartPole-v1 (reference: src/classic_control/cartpole.rs:13-357).
pub struct CartPoleV1 {
    inner: Single,
    device: Device,
}

#[bon]
impl CartPoleV1 {
    /// Same builder as the reference (cartpole.rs:34-44): `device` (where the returned tensors live, default CPU),
    /// `sutton_barto_reward` (default false), `is_euler` (default true), `render` under the `rendering` feature.
    /// Superset arguments, both optional: `gpu` = HIP ordinal that runs the kernels (default 0), `seed`.
    #[builder]
    pub fn new(
        #[builder(default = &Device::Cpu)] device: &Device,
        #[builder(default = false)] sutton_barto_reward: bool,
        #[builder(default = true)] is_euler: bool,
        #[cfg(feature = "rendering")]
        #[builder(default = false)]
        render: bool,
        #[builder(default = 0)] gpu: i32,
        seed: Option<u64>,
    ) -> Self {
        #[cfg(feature = "rendering")]
        if render {
            eprintln!("modurl_gym_mgym: rendering is not part of the GPU engine; `render` is ignored");
        }
        let mut cfg = VecGym::default_config(MGYM_CARTPOLE);
        cfg.device = gpu;
        cfg.n_envs = 1;
        cfg.seed = seed_or_entropy(seed);
        cfg.sutton_barto_reward = sutton_barto_reward as i32;
        cfg.is_euler = is_euler as i32;
        Self { inner: Single::new(cfg), device: device.clone() }
    }
}

impl Default for CartPoleV1 {
    fn default() -> Self {
        CartPoleV1::builder().build() // cartpole.rs:228-231
    }
}
impl_gym!(CartPoleV1);

/// MountainCar-v0 (reference: src/classic_control/mountain_car.rs:10-339).
pub struct MountainCarV0 {
    inner: Single,
    device: Device,
}

#[bon]
impl MountainCarV0 {
    /// Same builder as the reference (mountain_car.rs:25-34): `device`, `render` (feature `rendering`),
    /// `goal_velocity` (default 0.0); superset: `gpu`, `seed`.
    #[builder]
    pub fn new(
        #[builder(default = &Device::Cpu)] device: &Device,
        #[cfg(feature = "rendering")]
        #[builder(default = false)]
        render: bool,
        #[builder(default = 0.0)] goal_velocity: f32,
        #[builder(default = 0)] gpu: i32,
        seed: Option<u64>,
    ) -> Self {
        #[cfg(feature = "rendering")]
        if render {
            eprintln!("modurl_gym_mgym: rendering is not part of the GPU engine; `render` is ignored");
        }
        let mut cfg = VecGym::default_config(MGYM_MOUNTAINCAR);
        cfg.device = gpu;
        cfg.n_envs = 1;
        cfg.seed = seed_or_entropy(seed);
        cfg.goal_velocity = goal_velocity;
        Self { inner: Single::new(cfg), device: device.clone() }
    }
}

impl Default for MountainCarV0 {
    fn default() -> Self {
        MountainCarV0::builder().build() // mountain_car.rs:269-273
    }
}
impl_gym!(MountainCarV0);

/// LunarLander-v3 (reference: src/box_2d/lunar_lander.rs:232-1201).
pub struct LunarLanderV3 {
    inner: Single,
    device: Device,
}

#[bon]
impl LunarLanderV3 {
    /// Same builder as the reference (lunar_lander.rs:278-291): gravity −10, enable_wind false, wind_power 15,
    /// turbulence_power 1.5, `device` (by value, default CPU), `render` (feature `rendering`), `seed: Option<u64>`;
    /// superset: `gpu`.  Panics with the reference's message when gravity is outside (−12, 0) (lunar_lander.rs:292-296)
    /// and prints its two range recommendations (:298-310).
    #[builder]
    pub fn new(
        #[builder(default = -10.0)] gravity: f32,
        #[builder(default = false)] enable_wind: bool,
        #[builder(default = 15.0)] wind_power: f32,
        #[builder(default = 1.5)] turbulence_power: f32,
        #[builder(default = Device::Cpu)] device: Device,
        #[cfg(feature = "rendering")]
        #[builder(default = false)]
        render: bool,
        seed: Option<u64>,
        #[builder(default = 0)] gpu: i32,
    ) -> Self {
        assert!(
            -12.0 < gravity && gravity < 0.0,
            "gravity (current value: {}) must be between -12 and 0",
            gravity
        );
        if wind_power < 0.0 || wind_power > 20.0 {
            eprintln!("wind_power value is recommended to be between 0.0 and 20.0, (current value: {})", wind_power);
        }
        if turbulence_power < 0.0 || turbulence_power > 2.0 {
            eprintln!("turbulence_power value is recommended to be between 0.0 and 2.0, (current value: {})", turbulence_power);
        }
        #[cfg(feature = "rendering")]
        if render {
            eprintln!("modurl_gym_mgym: rendering is not part of the GPU engine; `render` is ignored");
        }
        let mut cfg = VecGym::default_config(MGYM_LUNARLANDER);
        cfg.device = gpu;
        cfg.n_envs = 1;
        cfg.seed = seed_or_entropy(seed);
        cfg.gravity = gravity;
        cfg.enable_wind = enable_wind as i32;
        cfg.wind_power = wind_power;
        cfg.turbulence_power = turbulence_power;
        Self { inner: Single::new(cfg), device }
    }
}

impl Default for LunarLanderV3 {
    fn default() -> Self {
        LunarLanderV3::builder().build() // lunar_lander.rs:717-721
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
        let mut env = CartPoleV1::builder().build(); // as in cartpole.rs:366
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
        let mut env = CartPoleV1::default();
        env.reset().unwrap();
        let _ = env.step(Tensor::from_vec(vec![0u32], vec![1], &Device::Cpu).unwrap());
    }

    #[test]
    fn cartpole_constant_push_ends_episode() {
        let mut env = CartPoleV1::builder().sutton_barto_reward(false).is_euler(true).build();
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
        let mut env = MountainCarV0::builder().build(); // mountain_car.rs:376
        env.reset().unwrap();
        let _ = env.step(Tensor::new(3u32, &Device::Cpu).unwrap());
    }

    #[test]
    fn lunar_lander_shapes_and_finite_rewards() {
        let mut env = LunarLanderV3::builder().enable_wind(true).seed(7).build(); // lunar_lander.rs:1596
        assert_eq!(env.reset().unwrap().shape().dims(), &[8]);
        for a in 0..4u32 {
            let info = env.step(Tensor::new(a, &Device::Cpu).unwrap()).unwrap();
            assert!(info.reward.is_finite());
        }
    }

    #[test]
    #[should_panic(expected = "gravity (current value: -12) must be between -12 and 0")]
    fn lunar_lander_gravity_out_of_range_panics() {
        let _ = LunarLanderV3::builder().gravity(-12.0).build(); // lunar_lander.rs:292-296
    }

    #[test]
    fn lunar_lander_seeded_envs_agree() {
        // lunar_lander.rs:1689-1761: two envs, seed 42, wind on, 50 steps, actions step % 4, observations equal
        let mut a = LunarLanderV3::builder().enable_wind(true).seed(42).build();
        let mut b = LunarLanderV3::builder().enable_wind(true).seed(42).build();
        assert_eq!(a.reset().unwrap().to_vec1::<f32>().unwrap(), b.reset().unwrap().to_vec1::<f32>().unwrap());
        for step in 0..50u32 {
            let sa = a.step(Tensor::new(step % 4, &Device::Cpu).unwrap()).unwrap();
            let sb = b.step(Tensor::new(step % 4, &Device::Cpu).unwrap()).unwrap();
            assert_eq!(sa.state.to_vec1::<f32>().unwrap(), sb.state.to_vec1::<f32>().unwrap());
        }
    }

    #[test]
    #[should_panic(expected = "You forgot to call reset()")]
    fn lunar_lander_step_before_reset_panics() {
        let mut env = LunarLanderV3::builder().build();
        let _ = env.step(Tensor::new(0u32, &Device::Cpu).unwrap());
    }
}
