// cartpole_step.h — CartPoleV1::step for ONE environment held in registers, in two forms with identical results:
//   cartpole_step_ref  : the reference's statements in the written association (cartpole.rs:251-348), IEEE divide,
//                        glibc-equal sin/cos (mgym_math.h); handles every input incl. invalid actions.
//   cartpole_step_fast : branch-free, cheaper instruction sequences (cartpole_math.h), valid under cartpole_fast_ok();
//                        bit-identical to the reference form there (exhaustive + differential proofs in
//                        tests/native/cartpole_fast_check.cpp, and on the MI355X in tests/test_gpu_classic.py).
// Compiles for the device (hipcc) and for the host test (g++, MG_HD expands to `static inline`).
//
// Per-env counter word `ctr` (one u32 column, vector-loaded with the state):
//   bits  0..9   steps_since_reset, saturating at 1023 (only ">= 500" is observable: cartpole.rs:297)
//   bits 10..11  steps_beyond_terminated: 0 = None, 1 = Some(0), 2 = Some(1), 3 = Some(>= 2) (only None/Some is
//                observable: cartpole.rs:319,330)
//   bits 12..31  episode = number of resets of this env mod 2^20: word 2 of the Philox counter of the next reset
#pragma once
#include "cartpole_math.h"

namespace mgym {

struct CartPoleParams {
    float gravity, masspole, total_mass, length, polemass_length, force_mag, tau;
    float x_threshold, theta_threshold_radians;
    float inv_total_mass;  // RN(1 / total_mass), for cp_div_const
    int is_euler, sutton_barto;
};

constexpr uint32_t kCpStepsMask = 0x3ffu, kCpSbtShift = 10, kCpSbtMask = 3u, kCpEpShift = 12, kCpEpMask = 0xfffffu;

MG_HD uint32_t cp_ctr_pack(uint32_t steps, uint32_t sbt_code, uint32_t episode) {
    return (steps > kCpStepsMask ? kCpStepsMask : steps) | ((sbt_code > 3u ? 3u : sbt_code) << kCpSbtShift) | ((episode & kCpEpMask) << kCpEpShift);
}
MG_HD uint32_t cp_ctr_after_reset(uint32_t ctr) {  // steps = 0, sbt = None, episode + 1
    return (((ctr >> kCpEpShift) + 1u) & kCpEpMask) << kCpEpShift;
}

// constructor constants, cartpole.rs:45-56, f32 arithmetic in the written order.  None of them is a builder argument
// of the reference (only sutton_barto_reward and is_euler are, cartpole.rs:36-44), so the kernels take them as
// compile-time constants (constexpr evaluation is IEEE f32, identical to the run-time evaluation the reference does).
constexpr CartPoleParams cartpole_params(int is_euler, int sutton_barto) {
    CartPoleParams p{};
    const float masscart = 1.0f;
    p.gravity = 9.8f;
    p.masspole = 0.1f;
    p.total_mass = p.masspole + masscart;
    p.length = 0.5f;
    p.polemass_length = p.masspole * p.length;
    p.force_mag = 10.0f;
    p.tau = 0.02f;
    p.theta_threshold_radians = 12.0f * 2.0f * 3.14159265358979323846f / 360.0f;
    p.x_threshold = 2.4f;
    p.inv_total_mass = 1.0f / p.total_mass;
    p.is_euler = is_euler;
    p.sutton_barto = sutton_barto;
    return p;
}

// ---- reference form ------------------------------------------------------------------------------------------------
// returns false (state untouched) for an invalid action: cartpole.rs:252 assert!(action_space.contains(&action))
MG_HD bool cartpole_step_ref(const CartPoleParams& p, float& sx, float& sxd, float& sth, float& sthd, uint32_t& ctr, uint32_t action,
                             float& reward, uint32_t& done, uint32_t& trunc) {
    if (action >= 2u) {
        reward = 0.0f; done = 0u; trunc = 0u;
        return false;
    }
    float x = sx, x_dot = sxd, theta = sth, theta_dot = sthd;
    float force = (action == 0u) ? -p.force_mag : p.force_mag;  // :258-262

    float sintheta, costheta;
    mg_sincosf(theta, &sintheta, &costheta);  // :264-265

    // :267-271
    float temp = (force + p.polemass_length * theta_dot * theta_dot * sintheta) / p.total_mass;
    float thetaacc = (p.gravity * sintheta - costheta * temp) /
                     (p.length * (4.0f / 3.0f - p.masspole * costheta * costheta / p.total_mass));
    float xacc = temp - p.polemass_length * thetaacc * costheta / p.total_mass;

    if (p.is_euler) {  // :273-277
        x += p.tau * x_dot;
        x_dot += p.tau * xacc;
        theta += p.tau * theta_dot;
        theta_dot += p.tau * thetaacc;
    } else {  // :279-282 verbatim
        x_dot += 0.5f * p.tau * (xacc + temp);
        theta_dot += 0.5f * p.tau * (thetaacc + temp);
        theta += p.tau * theta_dot + 0.5f * p.tau * p.tau * thetaacc;
        theta_dot += 0.5f * p.tau * (thetaacc + temp);
    }
    sx = x; sxd = x_dot; sth = theta; sthd = theta_dot;  // :285-290

    bool terminated = x < -p.x_threshold || x > p.x_threshold || theta < -p.theta_threshold_radians ||
                      theta > p.theta_threshold_radians;  // :291-294

    uint32_t steps = ctr & kCpStepsMask;
    uint32_t sbt = (ctr >> kCpSbtShift) & kCpSbtMask;  // 0 = None
    steps = steps < kCpStepsMask ? steps + 1u : steps;  // :296

    if (steps >= 500u) {  // :297-306 truncation takes precedence
        sbt = 1u;
        reward = 1.0f; done = 0u; trunc = 1u;
    } else if (!terminated) {  // :310-318
        reward = p.sutton_barto ? 0.0f : 1.0f; done = 0u; trunc = 0u;
    } else if (sbt == 0u) {  // :319-329
        sbt = 1u;
        reward = p.sutton_barto ? -1.0f : 1.0f; done = 1u; trunc = 0u;
    } else {  // :330-346
        reward = p.sutton_barto ? -1.0f : 0.0f;
        sbt = sbt < 3u ? sbt + 1u : sbt;
        done = 1u; trunc = 0u;
    }
    ctr = (ctr & ~(kCpStepsMask | (kCpSbtMask << kCpSbtShift))) | steps | (sbt << kCpSbtShift);
    return true;
}

// ---- fast form -----------------------------------------------------------------------------------------------------
// Guard: theta on the small-argument path of sincos, theta_dot small enough that every quotient's numerator stays
// in the range the cheap divisions are proven for (see the range argument in DESIGN.md §4), action valid.
// NaN fails the compares.
MG_HD bool cartpole_fast_ok(float theta, float theta_dot, uint32_t action) {
    return cp_sincos_small_ok(theta) && (as_u32(theta_dot) & 0x7fffffffu) < 0x53800000u /* 2^40 */ && action < 2u;
}

template <bool EULER, bool SB>
MG_HD void cartpole_step_fast(const CartPoleParams& p, float& sx, float& sxd, float& sth, float& sthd, uint32_t& ctr, uint32_t action,
                              float& reward, uint32_t& done, uint32_t& trunc) {
    float x = sx, x_dot = sxd, theta = sth, theta_dot = sthd;
    const float force = action ? p.force_mag : -p.force_mag;
    float sintheta, costheta;
    cp_sincos_small(theta, &sintheta, &costheta);
    const float temp = cp_div_const(force + p.polemass_length * theta_dot * theta_dot * sintheta, p.total_mass, p.inv_total_mass);
    const float den = p.length * (4.0f / 3.0f - cp_div_const(p.masspole * costheta * costheta, p.total_mass, p.inv_total_mass));
    const float thetaacc = cp_div(p.gravity * sintheta - costheta * temp, den);
    const float xacc = temp - cp_div_const(p.polemass_length * thetaacc * costheta, p.total_mass, p.inv_total_mass);
    if (EULER) {
        x += p.tau * x_dot;
        x_dot += p.tau * xacc;
        theta += p.tau * theta_dot;
        theta_dot += p.tau * thetaacc;
    } else {
        x_dot += 0.5f * p.tau * (xacc + temp);
        theta_dot += 0.5f * p.tau * (thetaacc + temp);
        theta += p.tau * theta_dot + 0.5f * p.tau * p.tau * thetaacc;
        theta_dot += 0.5f * p.tau * (thetaacc + temp);
    }
    sx = x; sxd = x_dot; sth = theta; sthd = theta_dot;
    // |x| > thr  ==  x < -thr || x > thr (false for NaN either way)
    const bool terminated = as_f32(as_u32(x) & 0x7fffffffu) > p.x_threshold || as_f32(as_u32(theta) & 0x7fffffffu) > p.theta_threshold_radians;
    uint32_t steps = ctr & kCpStepsMask;
    const uint32_t sbt = (ctr >> kCpSbtShift) & kCpSbtMask;
    steps = steps + (steps < kCpStepsMask ? 1u : 0u);
    const bool tr = steps >= 500u;
    const uint32_t sbt_next = sbt + (sbt < 3u ? 1u : 0u);
    const uint32_t sbt_new = tr ? 1u : (terminated ? sbt_next : sbt);
    if (SB) reward = tr ? 1.0f : (terminated ? -1.0f : 0.0f);
    else reward = (tr || !terminated || sbt == 0u) ? 1.0f : 0.0f;
    done = (terminated && !tr) ? 1u : 0u;
    trunc = tr ? 1u : 0u;
    ctr = (ctr & (kCpEpMask << kCpEpShift)) | steps | (sbt_new << kCpSbtShift);
}

}  // namespace mgym
