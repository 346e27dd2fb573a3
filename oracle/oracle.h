/* oracle.h — CPU restatement of the ModuRL_Gym step()/reset() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under modurl_gym_amd/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / the timed CPU baseline.
 *
 * Plain C99, scalar, one struct per environment (array-of-structs), written to
 * follow the Rust reference statement by statement.  Compile with
 *   -O2 -ffp-contract=off -fno-fast-math
 * (Rust never contracts a*b+c; flags/rewards must be bit-comparable).
 * libm's sinf/cosf/tanhf/sqrtf are used exactly where Rust's f32::sin/cos/tanh/sqrt
 * are (Rust lowers those to the platform libm).
 *
 * Parity pin: the three golden trajectories of the reference
 * (python_tests/{cartpole,mountain_car,lunar_lander}/{inputs,output}.json, re-encoded
 * under tests/golden/) replayed with the reference's own protocol and tolerances
 * (src/testing.rs:65-134).  MountainCarContinuous is NOT in the reference:
 * "parity unpinned" (gymnasium semantics restated from memory, see mountain_car.c).
 * Reset RNG streams (candle Tensor::rand, rand::StdRng) are "parity unpinned":
 * draws are supplied by the caller (ora_rng), distributions follow the reference.
 */
#ifndef MGYM_ORACLE_H
#define MGYM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* modurl::gym::StepInfo (cartpole.rs:300-305): state, reward, done, truncated. */
typedef struct {
    float reward;
    uint8_t done;
    uint8_t truncated;
} ora_stepinfo;

enum { ORA_OK = 0, ORA_INVALID_ACTION = 1, ORA_NOT_RESET = 2, ORA_BAD_CONFIG = 3 };

/* ------------------------------------------------------------------ RNG --- */
/* Counter-based Philox4x32-10 (Salmon et al., SC'11), restated from the paper.
 * key = 64-bit seed, counter = (env_id lo, env_id hi, episode, slot).  The
 * reference's RNG streams cannot be reproduced (see header); the oracle takes
 * this generator so that GPU-vs-oracle reset parity can be checked draw by draw. */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* uniform double in [0,1) from two words (53 bits), uniform float in [0,1) from one (23 bits:
 * rand 0.9 UniformFloat<f32> uses the top 23 bits, value1_2 - 1.0). */
double ora_u53(uint32_t hi, uint32_t lo);
float ora_u23(uint32_t w);

/* slots of the per-(env, episode) counter space */
enum {
    ORA_SLOT_RESET0 = 0,     /* reset draws, words 0..3  */
    ORA_SLOT_RESET1 = 1,     /* reset draws, words 4..7  */
    ORA_SLOT_RESET2 = 2,     /* lunar lander: words 8..11 */
    ORA_SLOT_RESET3 = 3,     /* lunar lander: words 12..15 */
    ORA_SLOT_STEP_BASE = 16  /* lunar lander dispersion of step s uses slot 16+s */
};

/* -------------------------------------------------------------- CartPole --- */
/* struct CartPoleV1, cartpole.rs:13-32 */
typedef struct {
    float gravity, masspole, total_mass, length, polemass_length, force_mag, tau;
    float x_threshold, theta_threshold_radians;
    int is_euler;
    int sbt_is_some;          /* steps_beyond_terminated: Option<usize> */
    uint64_t sbt;
    float state[4];
    uint64_t steps_since_reset;
    int sutton_barto_reward;
} ora_cartpole;

void ora_cartpole_new(ora_cartpole *e, int sutton_barto_reward, int is_euler); /* cartpole.rs:37-96 */
/* reset with 4 uniform draws u[i] in [0,1): state = (-0.05 + 0.1*u) as f32, cartpole.rs:238-249 */
void ora_cartpole_reset(ora_cartpole *e, const double u[4]);
int ora_cartpole_step(ora_cartpole *e, uint32_t action, ora_stepinfo *out); /* cartpole.rs:251-348 */

/* ----------------------------------------------------------- MountainCar --- */
/* struct MountainCarV0, mountain_car.rs:10-23 */
typedef struct {
    float state[2];
    float min_position, max_position, max_speed, goal_position, goal_velocity, force, gravity;
} ora_mountaincar;

void ora_mountaincar_new(ora_mountaincar *e, float goal_velocity);            /* mountain_car.rs:27-69 */
void ora_mountaincar_reset(ora_mountaincar *e, double u);                     /* mountain_car.rs:279-291 */
int ora_mountaincar_step(ora_mountaincar *e, uint32_t action, ora_stepinfo *out); /* mountain_car.rs:293-330 */

/* MountainCarContinuous-v0: NOT in the reference — parity unpinned. */
typedef struct {
    float state[2];
    float min_action, max_action, min_position, max_position, max_speed, goal_position, goal_velocity, power;
} ora_mountaincar_cont;

void ora_mountaincar_cont_new(ora_mountaincar_cont *e, float goal_velocity);
void ora_mountaincar_cont_reset(ora_mountaincar_cont *e, double u);
int ora_mountaincar_cont_step(ora_mountaincar_cont *e, float action, ora_stepinfo *out);

/* ----------------------------------------------------------- LunarLander --- */
typedef struct ora_lunarlander ora_lunarlander; /* opaque: lunar_lander.c */

ora_lunarlander *ora_lunarlander_new(float gravity, int enable_wind, float wind_power,
                                     float turbulence_power, int *status); /* lunar_lander.rs:281-352 */
void ora_lunarlander_free(ora_lunarlander *e);
/* reset() with caller-supplied draws (lunar_lander.rs:727-917):
 * u_height[12] in [0,1) -> heights, u_force[2] in [0,1) -> initial force,
 * wind_idx/torque_idx in [-9999,9998] (used only if enable_wind),
 * dispersion of the implicit first step(0) in disp0[2] (raw U(-1,1) draws, before /SCALE). */
int ora_lunarlander_reset(ora_lunarlander *e, const float u_height[12], const float u_force[2],
                          int32_t wind_idx, int32_t torque_idx, const float disp0[2], float obs[8]);
/* Testable::reset_deterministic (lunar_lander.rs:1249-1442); sets deterministic_mode */
int ora_lunarlander_reset_deterministic(ora_lunarlander *e, float obs[8]);
/* step(); disp = the two raw U(-1,1) draws of lunar_lander.rs:973-974 (ignored in deterministic_mode) */
int ora_lunarlander_step(ora_lunarlander *e, uint32_t action, const float disp[2], float obs[8],
                         ora_stepinfo *out);
/* Testable::set_state (lunar_lander.rs:1444-1554): raw[18] = lander{x,y,angle,vx,vy,w}, leg0{..}, leg1{..} */
void ora_lunarlander_set_state(ora_lunarlander *e, const float raw[18], int leg0_contact, int leg1_contact);
/* full-state export/import for GPU parity tests (layout documented in lunar_lander.c) */
int ora_lunarlander_state_floats(void);
void ora_lunarlander_export(const ora_lunarlander *e, float *blob);
void ora_lunarlander_import(ora_lunarlander *e, const float *blob);

/* ------------------------------------------------------ batched drivers --- */
/* Array-of-envs drivers used by the parity tests and the cpu_baseline timing.
 * kind: 0 cartpole, 1 mountaincar, 2 mountaincar_cont, 3 lunarlander.  SoA in/out
 * ([col][n]) so buffers compare 1:1 with the HIP engine's.  nthreads<=1 -> scalar loop. */
typedef struct ora_vec ora_vec;
typedef struct {
    int kind;
    uint64_t n_envs, env_id_base, seed;
    int sutton_barto_reward, is_euler;
    float goal_velocity;
    float gravity;
    int enable_wind;
    float wind_power, turbulence_power;
} ora_vec_config;

ora_vec *ora_vec_new(const ora_vec_config *cfg, int *status);
void ora_vec_free(ora_vec *v);
int ora_vec_obs_dim(const ora_vec *v);
/* masked reset (mask NULL = all); draws from Philox keyed (seed; env_id_base+i, episode[i]) */
int ora_vec_reset(ora_vec *v, const uint8_t *mask, float *obs_soa, int nthreads);
int ora_vec_step(ora_vec *v, const void *actions, float *obs_soa, float *reward, uint8_t *done,
                 uint8_t *trunc, int nthreads);
long ora_vec_run(ora_vec *v, const void *actions, int ring, long K, int nthreads); /* K steps + reset-on-finish */
int ora_vec_state_cols(const ora_vec *v);
void ora_vec_get_state(const ora_vec *v, float *soa);
void ora_vec_set_state(ora_vec *v, const float *soa);
/* lunar lander only: Testable::reset_deterministic on every env (lunar_lander.rs:1249-1442) */
int ora_vec_reset_deterministic(ora_vec *v, float *obs_soa);
/* lunar lander only: per-env dispersion override ([2][n] raw draws), NULL = generator */
void ora_vec_set_dispersion(ora_vec *v, const float *disp_soa);
/* what-if switch of the contact-creation order (b2mini.c b2mini_order_variant; tools/ll_contact_order_probe.py): process-wide, 0 = normal */
void ora_set_contact_order_variant(int variant);

#ifdef __cplusplus
}
#endif
#endif
