/* classic_control.c — CPU restatement of CartPoleV1 and MountainCarV0 (+ the unpinned
 * MountainCarContinuous).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows /root/reference/src/classic_control/cartpole.rs and mountain_car.rs
 * statement by statement; every block cites the lines it restates.
 */
#include "oracle.h"

#include <math.h>

/* ===================================================================== CartPole */

/* cartpole.rs:37-96 (constructor).  All literals are f32 in the reference because the
 * struct fields are f32; `total_mass`, `polemass_length` and the threshold are computed
 * with f32 arithmetic in the written order. */
void ora_cartpole_new(ora_cartpole *e, int sutton_barto_reward, int is_euler) {
    const float gravity = 9.8f;
    const float masscart = 1.0f;
    const float masspole = 0.1f;
    const float total_mass = masspole + masscart;           /* :48 */
    const float length = 0.5f;                              /* :49 */
    const float polemass_length = masspole * length;        /* :50 */
    const float force_mag = 10.0f;
    const float tau = 0.02f;
    /* :55  12.0 * 2.0 * PI / 360.0, left to right in f32 */
    const float pi = 3.14159265358979323846f;
    const float theta_threshold_radians = 12.0f * 2.0f * pi / 360.0f;
    const float x_threshold = 2.4f;

    e->gravity = gravity;
    e->masspole = masspole;
    e->total_mass = total_mass;
    e->length = length;
    e->polemass_length = polemass_length;
    e->force_mag = force_mag;
    e->tau = tau;
    e->x_threshold = x_threshold;
    e->theta_threshold_radians = theta_threshold_radians;
    e->sbt_is_some = 1; /* :81 steps_beyond_terminated: Some(0) */
    e->sbt = 0;
    e->is_euler = is_euler;
    e->state[0] = e->state[1] = e->state[2] = e->state[3] = 0.0f; /* :85 */
    e->steps_since_reset = 0;
    e->sutton_barto_reward = sutton_barto_reward;
}

/* cartpole.rs:238-249.  Tensor::rand(-0.05, 0.05, [4]) samples f64 then .to_dtype(F32). */
void ora_cartpole_reset(ora_cartpole *e, const double u[4]) {
    e->sbt_is_some = 0; /* :239 None */
    for (int i = 0; i < 4; ++i) {
        double v = u[i] * (0.05 - (-0.05)) + (-0.05);
        e->state[i] = (float)v;
    }
    e->steps_since_reset = 0; /* :243 */
}

/* cartpole.rs:251-348 */
int ora_cartpole_step(ora_cartpole *e, uint32_t action, ora_stepinfo *out) {
    if (action >= 2) return ORA_INVALID_ACTION; /* :252 assert!(action_space.contains) */
    float x = e->state[0], x_dot = e->state[1], theta = e->state[2], theta_dot = e->state[3]; /* :253-255 */

    float force = (action == 0) ? -e->force_mag : e->force_mag; /* :258-262 */

    float costheta = cosf(theta); /* :264 */
    float sintheta = sinf(theta); /* :265 */

    /* :267-271, association exactly as written */
    float temp = (force + e->polemass_length * theta_dot * theta_dot * sintheta) / e->total_mass;
    float thetaacc = (e->gravity * sintheta - costheta * temp) /
                     (e->length * (4.0f / 3.0f - e->masspole * costheta * costheta / e->total_mass));
    float xacc = temp - e->polemass_length * thetaacc * costheta / e->total_mass;

    if (e->is_euler) { /* :273-277 */
        x += e->tau * x_dot;
        x_dot += e->tau * xacc;
        theta += e->tau * theta_dot;
        theta_dot += e->tau * thetaacc;
    } else { /* :279-282, reproduced verbatim (x is never advanced; theta_dot advanced twice) */
        x_dot += 0.5f * e->tau * (xacc + temp);
        theta_dot += 0.5f * e->tau * (thetaacc + temp);
        theta += e->tau * theta_dot + 0.5f * e->tau * e->tau * thetaacc;
        theta_dot += 0.5f * e->tau * (thetaacc + temp);
    }

    e->state[0] = x; /* :285-290 */
    e->state[1] = x_dot;
    e->state[2] = theta;
    e->state[3] = theta_dot;

    int terminated = x < -e->x_threshold || x > e->x_threshold ||
                     theta < -e->theta_threshold_radians || theta > e->theta_threshold_radians; /* :291-294 */

    e->steps_since_reset += 1; /* :296 */
    if (e->steps_since_reset >= 500) { /* :297-306 truncation beats termination */
        e->sbt_is_some = 1;
        e->sbt = 0;
        out->reward = 1.0f;
        out->done = 0;
        out->truncated = 1;
        return ORA_OK;
    }

    if (!terminated) { /* :310-318 */
        out->reward = e->sutton_barto_reward ? 0.0f : 1.0f;
        out->done = 0;
        out->truncated = 0;
    } else if (!e->sbt_is_some) { /* :319-329 pole just fell */
        e->sbt_is_some = 1;
        e->sbt = 0;
        out->reward = e->sutton_barto_reward ? -1.0f : 1.0f;
        out->done = 1;
        out->truncated = 0;
    } else { /* :330-346 */
        out->reward = e->sutton_barto_reward ? -1.0f : 0.0f;
        e->sbt += 1;
        out->done = 1;
        out->truncated = 0;
    }
    return ORA_OK;
}

/* ================================================================== MountainCar */

/* f32::clamp (Rust core): if self < min {min} else if self > max {max} else {self} */
static float clampf(float v, float lo, float hi) {
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* mountain_car.rs:27-69 */
void ora_mountaincar_new(ora_mountaincar *e, float goal_velocity) {
    e->min_position = -1.2f;
    e->max_position = 0.6f;
    e->max_speed = 0.07f;
    e->goal_position = 0.5f;
    e->force = 0.001f;
    e->gravity = 0.0025f;
    e->goal_velocity = goal_velocity;
    e->state[0] = e->state[1] = 0.0f; /* :51 */
}

/* mountain_car.rs:279-291: position ~ U[-0.6,-0.4) sampled f64 -> f32, velocity 0 */
void ora_mountaincar_reset(ora_mountaincar *e, double u) {
    double p = u * (-0.4 - (-0.6)) + (-0.6);
    e->state[0] = (float)p;
    e->state[1] = 0.0f;
}

/* mountain_car.rs:293-330 */
int ora_mountaincar_step(ora_mountaincar *e, uint32_t action, ora_stepinfo *out) {
    if (action >= 3) return ORA_INVALID_ACTION; /* :294 */
    float position = e->state[0], velocity = e->state[1]; /* :296-297 */

    /* :301-302 */
    velocity += ((float)action - 1.0f) * e->force + cosf(3.0f * position) * (-e->gravity);
    velocity = clampf(velocity, -e->max_speed, e->max_speed); /* :304 */
    position += velocity;                                     /* :306 */
    position = clampf(position, e->min_position, e->max_position); /* :308 */
    if (position == e->min_position && velocity < 0.0f) velocity = 0.0f; /* :311-313 */

    e->state[0] = position; /* :315 */
    e->state[1] = velocity;

    int terminated = position >= e->goal_position && velocity >= e->goal_velocity; /* :318 */
    out->reward = -1.0f;                                                           /* :319 */
    out->done = (uint8_t)terminated;
    out->truncated = 0; /* :328 — no 200-step TimeLimit in the reference */
    return ORA_OK;
}

/* ======================================================= MountainCarContinuous */
/* NOT IN THE REFERENCE — parity unpinned.  gymnasium Continuous_MountainCarEnv
 * semantics restated in f32, in the style of MountainCarV0::step above:
 *   force = clamp(a, -1, 1); v += force*0.0015 + cos(3p)*(-0.0025); clamp v; p += v;
 *   clamp p; left wall; done = p >= 0.45 && v >= goal_velocity;
 *   reward = (done ? 100 : 0) - 0.1*a*a   (a un-clamped, as gymnasium does). */
void ora_mountaincar_cont_new(ora_mountaincar_cont *e, float goal_velocity) {
    e->min_action = -1.0f;
    e->max_action = 1.0f;
    e->min_position = -1.2f;
    e->max_position = 0.6f;
    e->max_speed = 0.07f;
    e->goal_position = 0.45f;
    e->goal_velocity = goal_velocity;
    e->power = 0.0015f;
    e->state[0] = e->state[1] = 0.0f;
}

void ora_mountaincar_cont_reset(ora_mountaincar_cont *e, double u) {
    double p = u * (-0.4 - (-0.6)) + (-0.6);
    e->state[0] = (float)p;
    e->state[1] = 0.0f;
}

int ora_mountaincar_cont_step(ora_mountaincar_cont *e, float action, ora_stepinfo *out) {
    if (action != action) return ORA_INVALID_ACTION; /* NaN is outside Box(-1,1) */
    float position = e->state[0], velocity = e->state[1];
    float force = clampf(action, e->min_action, e->max_action);

    velocity += force * e->power + cosf(3.0f * position) * (-0.0025f);
    velocity = clampf(velocity, -e->max_speed, e->max_speed);
    position += velocity;
    position = clampf(position, e->min_position, e->max_position);
    if (position == e->min_position && velocity < 0.0f) velocity = 0.0f;

    e->state[0] = position;
    e->state[1] = velocity;

    int terminated = position >= e->goal_position && velocity >= e->goal_velocity;
    float reward = 0.0f;
    if (terminated) reward = 100.0f;
    reward -= action * action * 0.1f;
    out->reward = reward;
    out->done = (uint8_t)terminated;
    out->truncated = 0;
    return ORA_OK;
}
