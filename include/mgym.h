/* mgym.h — C ABI of libmgym: the MI355X (gfx950) batched environment-step engine.
 *
 * This is the drop-in boundary for the hot path of ModuRL/ModuRL_Gym: the bodies of
 * `impl Gym for CartPoleV1 / MountainCarV0 / LunarLanderV3` — reset() and step() —
 * re-expressed as struct-of-arrays HIP kernels over n_envs independent environments.
 * Plain C: opaque handle, raw device pointers, sizes and status codes; no C++ or
 * PyTorch types.  Each entry point cites the reference interface it replaces
 * (paths relative to the reference repo).  A Rust `Gym` shim / cgo / ctypes stub binds
 * exactly these symbols.
 *
 * Conventions
 *  - All `*_out` / `actions` / `mask` / `blob` pointers are DEVICE pointers on the
 *    env's HIP device.  The engine owns only its internal state (its layout is private: columns for
 *    CartPole / MountainCar, one record per env for LunarLander — DESIGN.md section 3); callers own
 *    every buffer they pass.  No allocation or synchronisation happens inside
 *    mgym_reset/mgym_step/mgym_reset_done (they are hipGraph-capturable).
 *  - Observations are SoA: obs_out[k * n_envs + i] is component k of env i
 *    (reference: rank-1 f32 tensor of 4 / 2 / 8 per env — cartpole.rs:285-290,
 *    mountain_car.rs:315, lunar_lander.rs:1112-1123).
 *  - Discrete actions are uint32_t[n_envs] (reference: rank-0 u32 tensor,
 *    cartpole.rs:257,377); MountainCarContinuous takes float[n_envs].
 *  - reward f32, done/truncated uint8 (StepInfo, cartpole.rs:300-305).
 *  - Episode semantics are the reference's: NO auto-reset; stepping a finished env
 *    keeps integrating (cartpole.rs:330-346).  MGYM_FLAG_AUTO_RESET opts into a fused
 *    same-step reset (a superset feature, off by default).
 *  - Where the reference panics (assert!/expect) the engine returns a status code;
 *    errors detected on the device (invalid action) are sticky and reported by the
 *    next mgym_sync().
 *  - One handle = one device + one stream; calls on one handle are not thread-safe,
 *    different handles are independent (reference: &mut self, single-threaded).
 */
#ifndef MGYM_H
#define MGYM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGYM_ABI_VERSION 4

typedef struct mgym_env mgym_env;

typedef enum mgym_status {
    MGYM_OK = 0,
    MGYM_ERR_INVALID_ACTION = 1, /* assert!(action_space.contains(&action)): cartpole.rs:252, mountain_car.rs:294 */
    MGYM_ERR_NOT_RESET = 2,      /* assert!(self.lander.is_some(), "You forgot to call reset()"): lunar_lander.rs:920 */
    MGYM_ERR_BAD_CONFIG = 3,     /* assert!(-12.0 < gravity && gravity < 0.0): lunar_lander.rs:292-296 */
    MGYM_ERR_HIP = 4,            /* HIP runtime failure (reference: candle_core::Error from tensor ops) */
    MGYM_ERR_BAD_ARG = 5,        /* null/misaligned pointer, wrong kind */
    MGYM_ERR_NO_DEVICE = 6,      /* no gfx950 device / HIP runtime unusable: the engine has no CPU fallback */
    MGYM_ERR_CAPACITY = 7        /* LunarLander: more body/ground-edge pairs than the per-env contact cache holds (12: reachable only
                                    at horizontal speeds > 8 m/s next to the ground) or more touching contacts than the island solver
                                    holds (9 = the geometric bound of the scene); mgym_last_error() says which.  The affected
                                    environment's results are undefined until it is reset.  The reference has no such limit. */
} mgym_status;

typedef enum mgym_kind {
    MGYM_CARTPOLE = 0,         /* CartPoleV1      — src/classic_control/cartpole.rs */
    MGYM_MOUNTAINCAR = 1,      /* MountainCarV0   — src/classic_control/mountain_car.rs */
    MGYM_MOUNTAINCAR_CONT = 2, /* MountainCarContinuous-v0 — NOT in the reference (parity unpinned) */
    MGYM_LUNARLANDER = 3       /* LunarLanderV3   — src/box_2d/lunar_lander.rs */
} mgym_kind;

enum {
    MGYM_FLAG_AUTO_RESET = 1u /* fused same-step reset of finished envs inside mgym_step (superset; default off) */
};

/* Builder arguments of the three reference constructors, plus batching/sharding fields.
 *   CartPoleV1::builder():  sutton_barto_reward=false, is_euler=true      (cartpole.rs:36-44)
 *   MountainCarV0::builder(): goal_velocity=0.0                           (mountain_car.rs:27-34)
 *   LunarLanderV3::builder(): gravity=-10, enable_wind=false, wind_power=15,
 *                             turbulence_power=1.5, seed                  (lunar_lander.rs:280-291)
 * `device` is the HIP ordinal (reference: candle Device).  `env_id_base` is the global index
 * of this handle's env 0: per-env random streams are keyed by (seed, env_id_base + i), so
 * results do not depend on how a population is sharded over GPUs. */
typedef struct mgym_config {
    uint32_t struct_size; /* = sizeof(mgym_config) */
    int32_t kind;         /* mgym_kind */
    int32_t device;
    uint32_t flags;
    uint64_t n_envs;
    uint64_t env_id_base;
    uint64_t seed;
    int32_t sutton_barto_reward;
    int32_t is_euler;
    float goal_velocity;
    float gravity;
    int32_t enable_wind;
    float wind_power;
    float turbulence_power;
    uint32_t reserved;
} mgym_config;

/* Space metadata (reference: observation_space()/action_space(), cartpole.rs:58-69,350-356;
 * mountain_car.rs:42-48; lunar_lander.rs:1169-1200). */
typedef struct mgym_spec {
    int32_t obs_dim;
    int32_t n_actions;      /* Discrete(n); 0 for a continuous (Box) action space */
    int32_t action_is_float;
    int32_t state_cols;     /* columns of the get/set_state blob */
    float obs_low[8], obs_high[8];
    float action_low, action_high; /* continuous only */
} mgym_spec;

int mgym_abi_version(void);

/* Fill `cfg` with the reference builder defaults for `kind` (n_envs = 1, device 0, seed 0). */
int mgym_default_config(int kind, mgym_config *cfg);

/* Replaces {CartPoleV1,MountainCarV0,LunarLanderV3}::builder()...build()
 * (cartpole.rs:37-96, mountain_car.rs:27-69, lunar_lander.rs:281-352).
 * Initial state as constructed by the reference: zeros; CartPole steps_beyond_terminated=Some(0). */
int mgym_create(const mgym_config *cfg, mgym_env **out);
int mgym_destroy(mgym_env *env);

/* Launch stream (hipStream_t).  Default: a stream created by the engine.  Every call is ordered on this stream as far
 * as the caller can tell.  A LunarLander step may run part of its kernels on an engine-owned helper stream, forked
 * from and joined back into this stream by events inside the call (so it is also capturable by mgym_graph_begin). */
int mgym_set_stream(mgym_env *env, void *hip_stream);
void *mgym_get_stream(mgym_env *env);

/* Gym::reset (cartpole.rs:238-249, mountain_car.rs:279-291, lunar_lander.rs:727-917) for the
 * envs with mask[i] != 0 (mask == NULL: all).  obs_out may be NULL. */
int mgym_reset(mgym_env *env, const uint8_t *mask, float *obs_out);

/* mgym_reset with mask = done | truncated (either may be NULL): the "reset what just finished"
 * call a rollout loop issues after every step. */
int mgym_reset_done(mgym_env *env, const uint8_t *done, const uint8_t *truncated, float *obs_out);

/* Gym::step (cartpole.rs:251-348, mountain_car.rs:293-330, lunar_lander.rs:919-1167).
 * Any of obs_out / reward_out / done_out / trunc_out may be NULL (not written). */
int mgym_step(mgym_env *env, const void *actions, float *obs_out, float *reward_out,
              uint8_t *done_out, uint8_t *trunc_out);

/* K consecutive Gym::step calls on every env in ONE launch (fused rollout).  Results are
 * identical to K calls of mgym_step with the same per-step action columns, including the
 * MGYM_FLAG_AUTO_RESET behaviour between steps; for CartPole/MountainCar the state stays in registers
 * across the K steps, so per-step HBM traffic is the action read and the reward/flag writes only.
 * actions: [K][n_envs]; reward_out/done_out/trunc_out: [K][n_envs] or NULL; obs_out: [K][obs_dim][n_envs] or NULL.
 * LunarLander: ONE persistent launch in which every environment advances as soon as IT is ready (free-flight environments stay in
 * registers from step to step, environments on the contact path travel through device queues in batches of their own kind, whatever step
 * index each is at) — a step of mgym_step ends with its slowest environment, K steps of mgym_rollout cost the mean one.  Per-environment
 * results are those of K mgym_step calls, word for word (the loop being fused: lunar_lander.rs:919-1167 called K times).  Which form runs is the
 * engine's choice by K and population (mgym_get_info: rollout, rollout_min_k, rollout_waves, rollout_helper_blocks): K >= 8 (K >= 12 below 163 840 envs) below 491 520 envs per handle
 * the persistent launch, otherwise K steps — and always K steps while the stream is being captured (mgym_graph_begin or an external capture): replays
 * of a captured persistent launch aborted inside the HIP runtime at the end of round 4, unexplained; the words are the same either way. */
int mgym_rollout(mgym_env *env, const void *actions, int32_t K, float *obs_out, float *reward_out,
                 uint8_t *done_out, uint8_t *trunc_out);

/* mgym_rollout under an on-device uniform random policy (SURVEY §8f-1 "policy hook": random), so a rollout needs no
 * [K][n] action table: env with global id g takes, at step t of this call, bit (t % 32) of word (g % 4) of
 * Philox4x32-10(key = policy_seed; counter = (g & ~3, number of earlier mgym_rollout_uniform calls on this handle,
 * 0x40000000 + t / 32)).  actions_out ([K][n_envs] uint32, may be NULL) receives the drawn actions.  n_envs and env_id_base must be
 * multiples of 4.  Reference loop it fuses: a trainer calling `action_space().sample()` then `step()` (cartpole.rs:251-348, 350-356;
 * mountain_car.rs:293-330).
 * MountainCar (Discrete(3)): same key and the first two counter words, fourth word 0x40000000 + t / 2; env g takes half (t % 2) (low half first)
 * of word (g % 4), h, and the action (3 h) >> 16 — every action within 2^-16 of probability 1/3.  MountainCarContinuous (Box(-1, 1)): fourth
 * word 0x40000000 + t, the top 24 bits u of word (g % 4) give the force u * 2^-23 - 1 (actions_out then holds f32 words).
 * LunarLander: MGYM_ERR_BAD_ARG. */
int mgym_rollout_uniform(mgym_env *env, uint64_t policy_seed, int32_t K, void *actions_out, float *obs_out,
                         float *reward_out, uint8_t *done_out, uint8_t *trunc_out);

/* mgym_rollout under an on-device LINEAR policy (SURVEY §8f-1 "policy hook"; §8f-4: the step either side of the path — policy inference consumes the
 * observation, produces the action): at every step env i takes action 1 if ((w[0]*x + w[1]*x_dot) + w[2]*theta) + w[3]*theta_dot + b > 0 (f32, evaluated in
 * exactly that order, nothing fused), else 0, on the observation it holds BEFORE the step — the loop a trainer's rollout collector runs
 * (`let a = policy(&obs); let info = env.step(a)`, cartpole.rs:251-348) without leaving the GPU's registers.  `policy` is a HOST pointer to obs_dim weights
 * followed by the bias (5 floats, read during the call).  actions_out ([K][n_envs] uint32, may be NULL) receives the actions taken.
 * MountainCar (Discrete(3), mountain_car.rs:293-330): `policy` = three rows of (w_position, w_velocity, bias), 9 floats; the action is the index of the largest
 * score s_j = (w_j0 * position + w_j1 * velocity) + b_j, the first of equal ones.  MountainCarContinuous: one row, 3 floats; the score itself is the force
 * (actions_out then holds f32 words).  n_envs a multiple of 4.  LunarLander: MGYM_ERR_BAD_ARG (its rollout is a persistent launch of its own, ll_roll.h). */
int mgym_rollout_linear(mgym_env *env, const float *policy, int32_t K, void *actions_out, float *obs_out,
                        float *reward_out, uint8_t *done_out, uint8_t *trunc_out);

/* Number of env-steps of this handle that returned done or truncated since mgym_create (StepInfo.done / .truncated,
 * cartpole.rs:300-305): accumulated inside the step kernels by __ballot/popcount reductions of the done mask,
 * one fire-and-forget atomic per wave.  Synchronises the stream. */
int mgym_episode_count(mgym_env *env, uint64_t *finished);

/* Zero-copy view of the engine-owned current observation, SoA with column stride *col_stride
 * floats (for CartPole/MountainCar the state columns ARE the observation — `self.state.clone()`,
 * cartpole.rs:301).  Valid until the handle is destroyed; contents follow the stream order. */
int mgym_observation(mgym_env *env, const float **obs, uint64_t *col_stride);

/* Row-major copy of the current observation for consumers that want `[n_envs][obs_dim]` (one row per env,
 * the shape of the reference's per-env state tensor stacked over envs): out_aos[i * obs_dim + k]. */
int mgym_observation_aos(mgym_env *env, float *out_aos);

/* Test seam / checkpoint ≙ Testable::set_state (src/testing.rs:15-18).  blob = [state_cols][n_envs]
 * 4-byte words (integer columns as bit patterns); column meaning per kind in DESIGN.md.
 * LunarLander: the blob carries the bodies, legs, counters and the episode number, not the terrain — as in the reference,
 * where set_state moves the lander of an existing world (lunar_lander.rs Testable impl) — so the handle must have been reset
 * before a state is imported; an environment without a world keeps none and mgym_step reports MGYM_ERR_NOT_RESET for it.
 *
 * LIMITS of the CartPole counters (the engine keeps steps_since_reset, steps_beyond_terminated and the per-env
 * episode number in ONE 32-bit word per env, cartpole_step.h):
 *  - blob column 4, steps_since_reset, saturates at 1023 and column 5, steps_beyond_terminated, at Some(2): the
 *    reference only ever tests "steps >= 500" and None/Some (cartpole.rs:297, 319, 330), so no result depends on more;
 *    larger imported values are clamped.
 *  - blob column 6, the episode number (= resets of this env so far), is kept MODULO 2^20, on import too.  It is word 2
 *    of the Philox counter of the env's next reset draw, so after 1 048 576 resets of ONE env its initial states
 *    repeat from the first one (same seed, same env id).  With a uniform random policy (~22-step episodes) that is
 *    ~2.3e7 steps of that env — about four minutes at this engine's CartPole rate.  The reference draws from an RNG
 *    that never cycles in practice (cartpole.rs:240); a trainer that steps one handle past that horizon and needs
 *    fresh initial states should create the next handle with a new `seed`.  Nothing else depends on the wrap.
 *    MountainCar and LunarLander keep a full 32-bit episode counter. */
int mgym_get_state(mgym_env *env, void *blob);
int mgym_set_state(mgym_env *env, const void *blob);

/* Test seam ≙ Testable::reset_deterministic (cartpole.rs:437-442: reset() then state = 0;
 * mountain_car.rs:403-408: state = 0; lunar_lander.rs:1249-1442: flat terrain, lander at
 * (W/2, 0.8 H) falling at 1 m/s, deterministic_mode = true) applied to every env. obs_out may be NULL. */
int mgym_reset_deterministic(mgym_env *env, float *obs_out);

/* LunarLander: deterministic_mode (lunar_lander.rs:967-970) generalised — disp = [2][n_envs]
 * raw U(-1,1) draws used instead of the per-env generator; NULL restores the generator. */
int mgym_set_dispersion_override(mgym_env *env, const float *disp);

int mgym_get_spec(int kind, mgym_spec *spec);

/* Wait for the stream and return (then clear) the sticky device-side status. */
int mgym_sync(mgym_env *env);

/* Thread-local message for the last failing call on this thread. */
const char *mgym_last_error(void);

/* Describes how this handle runs, as NUL-terminated "key=value" lines written to buf (at most cap bytes, always terminated;
 * returns MGYM_ERR_BAD_ARG if cap is too small).  Keys every family reports: kind, n_envs, device, auto_reset,
 * GPU_MAX_HW_QUEUES (the HIP runtime knob as this process's environment held it when the handle was created, or "unset").
 * LunarLander adds its launch structure — contact_block, launch_order, staged_resets — and `concurrent_streams`: the
 * reference is single-threaded (lunar_lander.rs:240-249), this engine steps one population on up to three streams
 * (caller's stream, two helpers), and they only run side by side when the runtime put them on different hardware queues.
 * The call launches a ~50 us probe kernel on each of them at once and reports how many were seen running at the same
 * time (3 of 3 is the measured configuration; fewer: set GPU_MAX_HW_QUEUES=8 before the first HIP call of the
 * process).  Synchronises the stream. */
int mgym_get_info(mgym_env *env, char *buf, size_t cap);

/* Plain device-memory helpers for callers without their own HIP allocator (C/C++/ctypes). */
int mgym_malloc(int device, size_t bytes, void **out);
int mgym_free(int device, void *ptr);
int mgym_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes);
int mgym_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes);
int mgym_device_count(int *count);

/* Stream timing helpers (hipEvent on the env's stream) so harnesses can time launches
 * on the stream the kernels actually run on. */
int mgym_timer_start(mgym_env *env);
int mgym_timer_stop(mgym_env *env, float *elapsed_ms); /* synchronises */

/* Test seam: evaluates, on the device, the exhaustive bit-identity checks of the cheap CartPole instruction sequences
 * against the reference-form arithmetic (cartpole.rs:264-271 with IEEE divide and glibc-equal sin/cos); see
 * modurl_gym_amd/csrc/selftest.hip.  mismatches[5] = {CartPole sincos, x / total_mass, n / d, whole CartPole step, fused lock-step
 * sincos / cos of the LunarLander and MountainCar kernels}; all must be 0. */
int mgym_selftest_cartpole_math(int device, uint64_t *mismatches);

/* hipGraph capture of a caller-issued launch sequence on the env's stream.  What a captured launch freezes and what it does not:
 *  - environment STATE is read when the graph runs: mgym_set_state, mgym_reset (any form) and eager steps between replays are fine.
 *    (LunarLander: a captured step always carries the launches that reset a finished environment WITHOUT a prepared next episode —
 *    empty when every environment has one — because only the run can tell; reset semantics: lunar_lander.rs:727-917 incl. the implicit
 *    step(0) at :911-916.)
 *  - the handle's CONFIGURATION is frozen into kernel arguments: after mgym_set_dispersion_override (deterministic_mode,
 *    lunar_lander.rs:967-970) mgym_graph_launch refuses graphs captured before it with MGYM_ERR_BAD_ARG — capture again.
 *    The pointers passed to the captured calls are frozen too, of course.
 * mgym_get_info may not be called between mgym_graph_begin and mgym_graph_end (it synchronises): MGYM_ERR_BAD_ARG. */
int mgym_graph_begin(mgym_env *env);
int mgym_graph_end(mgym_env *env, void **graph_exec_out);
int mgym_graph_launch(mgym_env *env, void *graph_exec);
int mgym_graph_destroy(void *graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* MGYM_H */
