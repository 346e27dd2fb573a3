//! Raw FFI: one declaration per symbol of include/mgym.h (ABI version 3), same order as the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct mgym_env {
    _opaque: [u8; 0],
}

pub const MGYM_ABI_VERSION: c_int = 4;

pub const MGYM_OK: c_int = 0;
pub const MGYM_ERR_INVALID_ACTION: c_int = 1;
pub const MGYM_ERR_NOT_RESET: c_int = 2;
pub const MGYM_ERR_BAD_CONFIG: c_int = 3;
pub const MGYM_ERR_HIP: c_int = 4;
pub const MGYM_ERR_BAD_ARG: c_int = 5;
pub const MGYM_ERR_NO_DEVICE: c_int = 6;
pub const MGYM_ERR_CAPACITY: c_int = 7;

pub const MGYM_CARTPOLE: c_int = 0;
pub const MGYM_MOUNTAINCAR: c_int = 1;
pub const MGYM_MOUNTAINCAR_CONT: c_int = 2;
pub const MGYM_LUNARLANDER: c_int = 3;

pub const MGYM_FLAG_AUTO_RESET: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mgym_config {
    pub struct_size: u32,
    pub kind: i32,
    pub device: i32,
    pub flags: u32,
    pub n_envs: u64,
    pub env_id_base: u64,
    pub seed: u64,
    pub sutton_barto_reward: i32,
    pub is_euler: i32,
    pub goal_velocity: f32,
    pub gravity: f32,
    pub enable_wind: i32,
    pub wind_power: f32,
    pub turbulence_power: f32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mgym_spec {
    pub obs_dim: i32,
    pub n_actions: i32,
    pub action_is_float: i32,
    pub state_cols: i32,
    pub obs_low: [f32; 8],
    pub obs_high: [f32; 8],
    pub action_low: f32,
    pub action_high: f32,
}

unsafe extern "C" {
    pub fn mgym_abi_version() -> c_int;
    pub fn mgym_default_config(kind: c_int, cfg: *mut mgym_config) -> c_int;
    pub fn mgym_create(cfg: *const mgym_config, out: *mut *mut mgym_env) -> c_int;
    pub fn mgym_destroy(env: *mut mgym_env) -> c_int;
    pub fn mgym_set_stream(env: *mut mgym_env, hip_stream: *mut c_void) -> c_int;
    pub fn mgym_get_stream(env: *mut mgym_env) -> *mut c_void;
    pub fn mgym_reset(env: *mut mgym_env, mask: *const u8, obs_out: *mut f32) -> c_int;
    pub fn mgym_reset_done(env: *mut mgym_env, done: *const u8, truncated: *const u8, obs_out: *mut f32) -> c_int;
    pub fn mgym_step(
        env: *mut mgym_env,
        actions: *const c_void,
        obs_out: *mut f32,
        reward_out: *mut f32,
        done_out: *mut u8,
        trunc_out: *mut u8,
    ) -> c_int;
    pub fn mgym_rollout(
        env: *mut mgym_env,
        actions: *const c_void,
        k: i32,
        obs_out: *mut f32,
        reward_out: *mut f32,
        done_out: *mut u8,
        trunc_out: *mut u8,
    ) -> c_int;
    pub fn mgym_rollout_uniform(
        env: *mut mgym_env,
        policy_seed: u64,
        k: i32,
        actions_out: *mut c_void,
        obs_out: *mut f32,
        reward_out: *mut f32,
        done_out: *mut u8,
        trunc_out: *mut u8,
    ) -> c_int;
    pub fn mgym_rollout_linear(
        env: *mut mgym_env,
        policy: *const f32,
        k: i32,
        actions_out: *mut c_void,
        obs_out: *mut f32,
        reward_out: *mut f32,
        done_out: *mut u8,
        trunc_out: *mut u8,
    ) -> c_int;
    pub fn mgym_episode_count(env: *mut mgym_env, finished: *mut u64) -> c_int;
    pub fn mgym_selftest_cartpole_math(device: c_int, mismatches: *mut u64) -> c_int;
    pub fn mgym_observation(env: *mut mgym_env, obs: *mut *const f32, col_stride: *mut u64) -> c_int;
    pub fn mgym_observation_aos(env: *mut mgym_env, out_aos: *mut f32) -> c_int;
    pub fn mgym_get_state(env: *mut mgym_env, blob: *mut c_void) -> c_int;
    pub fn mgym_set_state(env: *mut mgym_env, blob: *const c_void) -> c_int;
    pub fn mgym_reset_deterministic(env: *mut mgym_env, obs_out: *mut f32) -> c_int;
    pub fn mgym_set_dispersion_override(env: *mut mgym_env, disp: *const f32) -> c_int;
    pub fn mgym_get_spec(kind: c_int, spec: *mut mgym_spec) -> c_int;
    pub fn mgym_sync(env: *mut mgym_env) -> c_int;
    pub fn mgym_last_error() -> *const c_char;
    pub fn mgym_get_info(env: *mut mgym_env, buf: *mut c_char, cap: usize) -> c_int;
    pub fn mgym_malloc(device: c_int, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn mgym_free(device: c_int, ptr: *mut c_void) -> c_int;
    pub fn mgym_memcpy_h2d(device: c_int, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn mgym_memcpy_d2h(device: c_int, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn mgym_device_count(count: *mut c_int) -> c_int;
    pub fn mgym_timer_start(env: *mut mgym_env) -> c_int;
    pub fn mgym_timer_stop(env: *mut mgym_env, elapsed_ms: *mut f32) -> c_int;
    pub fn mgym_graph_begin(env: *mut mgym_env) -> c_int;
    pub fn mgym_graph_end(env: *mut mgym_env, graph_exec_out: *mut *mut c_void) -> c_int;
    pub fn mgym_graph_launch(env: *mut mgym_env, graph_exec: *mut c_void) -> c_int;
    pub fn mgym_graph_destroy(graph_exec: *mut c_void) -> c_int;
}
