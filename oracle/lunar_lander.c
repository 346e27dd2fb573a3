/* lunar_lander.c — CPU restatement of LunarLanderV3 (reference src/box_2d/lunar_lander.rs) on top
 * of b2mini (the Box2D subset it drives).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Every block cites the reference lines it restates. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "b2mini.h"
#include "oracle.h"

/* lunar_lander.rs:31-58 */
#define FPS 50.0f
#define SCALE 30.0f
#define MAIN_ENGINE_POWER 13.0f
#define SIDE_ENGINE_POWER 0.6f
#define INITIAL_RANDOM 1000.0f
static const float LANDER_POLY[6][2] = {{-14.0f, 17.0f}, {-17.0f, 0.0f}, {-17.0f, -10.0f}, {17.0f, -10.0f}, {17.0f, 0.0f}, {14.0f, 17.0f}};
#define LEG_AWAY 20.0f
#define LEG_DOWN 18.0f
#define LEG_W 2.0f
#define LEG_H 8.0f
#define LEG_SPRING_TORQUE 40.0f
#define SIDE_ENGINE_HEIGHT 14.0f
#define SIDE_ENGINE_AWAY 12.0f
#define MAIN_ENGINE_Y_LOCATION 4.0f
#define VIEWPORT_W 600.0f
#define VIEWPORT_H 400.0f
#define PI_F32 3.14159265358979323846f /* std::f32::consts::PI */

/* struct LunarLanderV3, lunar_lander.rs:232-276 (+ ContactDetector :139-143) */
struct ora_lunarlander {
    float gravity; int enable_wind; float wind_power, turbulence_power;
    b2world world; int has_world;      /* world/lander: Option<..> */
    int det_game_over, det_legs[2];    /* ContactDetector */
    int game_over;
    int prev_shaping_some; float prev_shaping;
    float helipad_x1, helipad_x2, helipad_y;
    int32_t wind_idx, torque_idx;
    int deterministic_mode;
};

/* ContactDetector::begin_contact / end_contact, lunar_lander.rs:154-205 */
static void on_begin(void *ud, int a, int b) {
    ora_lunarlander *e = (ora_lunarlander *)ud;
    if ((a == 1 && b == 0) || (b == 1 && a == 0)) e->det_game_over = 1;
    for (int i = 0; i < 2; ++i) {
        int leg = 2 + i;
        if ((a == leg && b == 0) || (b == leg && a == 0)) e->det_legs[i] = 1;
    }
}
static void on_end(void *ud, int a, int b) {
    ora_lunarlander *e = (ora_lunarlander *)ud;
    for (int i = 0; i < 2; ++i) {
        int leg = 2 + i;
        if ((a == leg && b == 0) || (b == leg && a == 0)) e->det_legs[i] = 0;
    }
}

/* lunar_lander.rs:281-352 */
ora_lunarlander *ora_lunarlander_new(float gravity, int enable_wind, float wind_power, float turbulence_power, int *status) {
    if (!(-12.0f < gravity && gravity < 0.0f)) { /* :292-296 */
        if (status) *status = ORA_BAD_CONFIG;
        return NULL;
    }
    ora_lunarlander *e = (ora_lunarlander *)calloc(1, sizeof(*e));
    e->gravity = gravity; e->enable_wind = enable_wind; e->wind_power = wind_power; e->turbulence_power = turbulence_power;
    if (status) *status = ORA_OK;
    return e;
}
void ora_lunarlander_free(ora_lunarlander *e) { free(e); }

static b2v bv(float x, float y) { b2v r = {x, y}; return r; }

/* shared by reset() (:733-908) and reset_deterministic() (:1256-1438): world, terrain, lander, legs */
static void build_scene(ora_lunarlander *e, const float height_in[12], float initial_y, int random_force, const float u_force[2],
                        int32_t wind_idx, int32_t torque_idx, int deterministic) {
    b2w_init(&e->world, bv(0.0f, e->gravity)); /* :734-735 */
    e->world.listener.begin = on_begin; e->world.listener.end = on_end; e->world.listener.ud = e; /* :738-741 */
    e->has_world = 1;
    e->det_game_over = 0; e->det_legs[0] = e->det_legs[1] = 0;
    e->game_over = 0;          /* :745 */
    e->prev_shaping_some = 0;  /* :746 */

    const float w = VIEWPORT_W / SCALE, h = VIEWPORT_H / SCALE; /* :748-749 */
    enum { CHUNKS = 11 };
    float height[CHUNKS + 1];
    for (int i = 0; i <= CHUNKS; ++i) height[i] = height_in[i]; /* :753-756 / :1276-1280 */
    float chunk_x[CHUNKS];
    for (int i = 0; i < CHUNKS; ++i) chunk_x[i] = w / (float)(CHUNKS - 1) * (float)i; /* :758-760 */
    e->helipad_x1 = chunk_x[CHUNKS / 2 - 1];
    e->helipad_x2 = chunk_x[CHUNKS / 2 + 1];
    e->helipad_y = h / 4.0f;
    height[CHUNKS / 2 - 2] = e->helipad_y; /* :766-770 */
    height[CHUNKS / 2 - 1] = e->helipad_y;
    height[CHUNKS / 2] = e->helipad_y;
    height[CHUNKS / 2 + 1] = e->helipad_y;
    height[CHUNKS / 2 + 2] = e->helipad_y;
    float smooth_y[CHUNKS];
    for (int i = 0; i < CHUNKS; ++i) { /* :772-774, saturating_sub(1) */
        int im1 = i > 0 ? i - 1 : 0;
        smooth_y[i] = 0.33f * (height[im1] + height[i] + height[i + 1]);
    }
    b2w_add_ground_edge(&e->world, bv(0.0f, 0.0f), bv(w, 0.0f), 0.1f); /* :783-792 */
    for (int i = 0; i < CHUNKS - 1; ++i)                               /* :795-808 */
        b2w_add_ground_edge(&e->world, bv(chunk_x[i], smooth_y[i]), bv(chunk_x[i + 1], smooth_y[i + 1]), 0.1f);

    const float initial_x = VIEWPORT_W / SCALE / 2.0f; /* :816 */
    int lander = b2w_add_body(&e->world, bv(initial_x, initial_y), 0.0f, 1); /* :818-824 */
    b2v verts[6];
    for (int i = 0; i < 6; ++i) verts[i] = bv(LANDER_POLY[i][0] / SCALE, LANDER_POLY[i][1] / SCALE); /* :828-832 */
    b2w_set_polygon(&e->world, lander, verts, 6, 5.0f, 0.1f); /* :834-842 */

    if (random_force) { /* :845-849  rng.random_range(-INITIAL_RANDOM..INITIAL_RANDOM) */
        float force_x = u_force[0] * (INITIAL_RANDOM - (-INITIAL_RANDOM)) + (-INITIAL_RANDOM);
        float force_y = u_force[1] * (INITIAL_RANDOM - (-INITIAL_RANDOM)) + (-INITIAL_RANDOM);
        b2b_apply_force_to_center(&e->world, lander, bv(force_x, force_y), 1);
    } else { /* :1371-1374 */
        b2b_set_linear_velocity(&e->world, lander, bv(0.0f, -1.0f));
        b2b_set_angular_velocity(&e->world, lander, 0.0f);
    }
    if (e->enable_wind) { e->wind_idx = wind_idx; e->torque_idx = torque_idx; } /* :854-857 / :1379-1382 */

    for (int leg_index = 0; leg_index < 2; ++leg_index) { /* :862-908 */
        float i_f = leg_index == 0 ? -1.0f : 1.0f;
        int leg = b2w_add_body(&e->world, bv(initial_x - i_f * LEG_AWAY / SCALE, initial_y), i_f * 0.05f, 2 + leg_index);
        b2w_set_box(&e->world, leg, LEG_W / SCALE, LEG_H / SCALE, 1.0f, 0.2f /* b2FixtureDef default friction */);
        float lower, upper;
        if (i_f == -1.0f) { lower = 0.9f - 0.5f; upper = 0.9f; } else { lower = -0.9f; upper = -0.9f + 0.5f; }
        b2w_add_revolute(&e->world, leg, bv(0.0f, 0.0f), bv(i_f * LEG_AWAY / SCALE, LEG_DOWN / SCALE), lower, upper,
                         LEG_SPRING_TORQUE, 0.3f * i_f);
        if (deterministic) { /* :1434-1435 */
            b2b_set_linear_velocity(&e->world, leg, bv(0.0f, -1.0f));
            b2b_set_angular_velocity(&e->world, leg, 0.0f);
        }
    }
}

/* observation, lunar_lander.rs:1095-1121 (= get_current_state :1209-1245) */
static void observe(const ora_lunarlander *e, float state[8]) {
    const b2body *lander = &e->world.bodies[1];
    b2v pos = lander->xf.p;      /* get_position: body origin */
    b2v vel = lander->linvel;    /* get_linear_velocity: centre-of-mass velocity */
    float angle = lander->sw.a, angular_vel = lander->angvel;
    state[0] = (pos.x - VIEWPORT_W / SCALE / 2.0f) / (VIEWPORT_W / SCALE / 2.0f);
    state[1] = (pos.y - (e->helipad_y + LEG_DOWN / SCALE)) / (VIEWPORT_H / SCALE / 2.0f);
    state[2] = vel.x * (VIEWPORT_W / SCALE / 2.0f) / FPS;
    state[3] = vel.y * (VIEWPORT_H / SCALE / 2.0f) / FPS;
    state[4] = angle;
    state[5] = 20.0f * angular_vel / FPS;
    state[6] = e->det_legs[0] ? 1.0f : 0.0f;
    state[7] = e->det_legs[1] ? 1.0f : 0.0f;
}

/* step(), lunar_lander.rs:919-1167 */
int ora_lunarlander_step(ora_lunarlander *e, uint32_t action, const float disp[2], float obs[8], ora_stepinfo *out) {
    if (!e->has_world) return ORA_NOT_RESET; /* :920 */
    b2world *world = &e->world;
    const int lander = 1;

    if (e->enable_wind) { /* :927-959 */
        int legs_contact = e->det_legs[0] || e->det_legs[1];
        if (!legs_contact) {
            float wind_mag = tanhf(sinf(0.02f * (float)e->wind_idx) + sinf(PI_F32 * 0.01f * (float)e->wind_idx)) * e->wind_power;
            e->wind_idx += 1;
            b2b_apply_force_to_center(world, lander, bv(wind_mag, 0.0f), 1);
            float torque_mag = tanhf(sinf(0.02f * (float)e->torque_idx) + sinf(PI_F32 * 0.01f * (float)e->torque_idx)) * e->turbulence_power;
            e->torque_idx += 1;
            b2b_apply_torque(world, lander, torque_mag, 1);
        }
    }

    float lander_angle = world->bodies[lander].sw.a; /* :962-964 */
    float tip0 = sinf(lander_angle), tip1 = cosf(lander_angle);
    float side0 = -tip1, side1 = tip0;

    float dispersion[2]; /* :967-976 */
    if (e->deterministic_mode) { dispersion[0] = 0.0f; dispersion[1] = 0.0f; }
    else { dispersion[0] = disp[0] / SCALE; dispersion[1] = disp[1] / SCALE; }

    float m_power = 0.0f;
    if (action == 2) { /* :979-1003 */
        m_power = 1.0f;
        b2v lander_pos = world->bodies[lander].xf.p;
        float ox = tip0 * (MAIN_ENGINE_Y_LOCATION / SCALE + 2.0f * dispersion[0]) + side0 * dispersion[1];
        float oy = -tip1 * (MAIN_ENGINE_Y_LOCATION / SCALE + 2.0f * dispersion[0]) - side1 * dispersion[1];
        b2v impulse_pos = bv(lander_pos.x + ox, lander_pos.y + oy);
        b2v impulse_force = bv(-ox * MAIN_ENGINE_POWER * m_power, -oy * MAIN_ENGINE_POWER * m_power);
        b2b_apply_linear_impulse(world, lander, impulse_force, impulse_pos, 1);
    }
    float s_power = 0.0f;
    if (action == 1 || action == 3) { /* :1019-1048 */
        float direction = (float)((int32_t)action - 2);
        s_power = 1.0f;
        b2v lander_pos = world->bodies[lander].xf.p;
        float ox = tip0 * dispersion[0] + side0 * (3.0f * dispersion[1] + direction * SIDE_ENGINE_AWAY / SCALE);
        float oy = -tip1 * dispersion[0] - side1 * (3.0f * dispersion[1] + direction * SIDE_ENGINE_AWAY / SCALE);
        b2v impulse_pos = bv(lander_pos.x + ox - tip0 * 17.0f / SCALE, lander_pos.y + oy + tip1 * SIDE_ENGINE_HEIGHT / SCALE);
        b2v impulse_force = bv(-ox * SIDE_ENGINE_POWER * s_power, -oy * SIDE_ENGINE_POWER * s_power);
        b2b_apply_linear_impulse(world, lander, impulse_force, impulse_pos, 1);
    }

    b2w_step(world, 1.0f / FPS, 6 * 30, 2 * 30); /* :1066 */

    float state[8];
    observe(e, state); /* :1095-1121 */
    for (int k = 0; k < 8; ++k) obs[k] = state[k];

    float reward = 0.0f; /* :1126-1139 */
    float shaping = -100.0f * sqrtf(state[0] * state[0] + state[1] * state[1]) -
                    100.0f * sqrtf(state[2] * state[2] + state[3] * state[3]) - 100.0f * fabsf(state[4]) +
                    10.0f * state[6] + 10.0f * state[7];
    if (e->prev_shaping_some) reward = shaping - e->prev_shaping;
    e->prev_shaping_some = 1; e->prev_shaping = shaping;
    reward -= m_power * 0.30f;
    reward -= s_power * 0.03f;

    int terminated = 0; /* :1142-1156 */
    if (e->game_over || e->det_game_over || fabsf(state[0]) >= 1.0f) {
        terminated = 1; reward = -100.0f;
    } else if (!world->bodies[lander].awake) {
        terminated = 1; reward = 100.0f;
    }
    out->reward = reward; out->done = (uint8_t)terminated; out->truncated = 0;
    return ORA_OK;
}

/* reset(), lunar_lander.rs:727-917 */
int ora_lunarlander_reset(ora_lunarlander *e, const float u_height[12], const float u_force[2], int32_t wind_idx,
                          int32_t torque_idx, const float disp0[2], float obs[8]) {
    e->deterministic_mode = 0; /* :731 */
    const float h = VIEWPORT_H / SCALE;
    float height[12];
    for (int i = 0; i < 12; ++i) height[i] = u_height[i] * (h / 2.0f - 0.0f) + 0.0f; /* :755 random_range(0.0..h/2.0) */
    build_scene(e, height, VIEWPORT_H / SCALE, 1, u_force, wind_idx, torque_idx, 0);
    ora_stepinfo si;
    return ora_lunarlander_step(e, 0, disp0, obs, &si); /* :911-916 */
}

/* Testable::reset_deterministic, lunar_lander.rs:1249-1442 */
int ora_lunarlander_reset_deterministic(ora_lunarlander *e, float obs[8]) {
    e->deterministic_mode = 1; /* :1254 */
    const float h = VIEWPORT_H / SCALE;
    float height[12];
    for (int i = 0; i < 12; ++i) height[i] = h / 8.0f; /* :1278-1280 */
    build_scene(e, height, VIEWPORT_H / SCALE * 0.8f, 0, NULL, 0, 0, 1);
    observe(e, obs); /* :1441 */
    return ORA_OK;
}

/* Testable::set_state, lunar_lander.rs:1444-1554 */
void ora_lunarlander_set_state(ora_lunarlander *e, const float raw[18], int leg0_contact, int leg1_contact) {
    if (!e->has_world) return;
    b2world *w = &e->world;
    for (int b = 0; b < 3; ++b) { /* lander, then legs[0], legs[1] */
        const float *r = raw + 6 * b;
        b2b_set_transform(w, 1 + b, bv(r[0], r[1]), r[2]);
        b2b_set_linear_velocity(w, 1 + b, bv(r[3], r[4]));
        b2b_set_angular_velocity(w, 1 + b, r[5]);
    }
    for (int b = 0; b < 3; ++b) b2b_set_awake(w, 1 + b, 1); /* :1532-1535 */
    e->det_legs[0] = leg0_contact; e->det_legs[1] = leg1_contact; /* :1539-1553 */
}

/* state blob (25 words): raw {x,y,angle,vx,vy,w} of lander, leg0, leg1 (body-origin position, COM
 * velocity: what lunar_lander.py custom_info records), leg0_contact, leg1_contact, game_over,
 * lander awake, prev_shaping (NaN = None), wind_idx, torque_idx (i32 bit patterns).
 * import applies Testable::set_state to words 0..19 and ignores the rest. */
int ora_lunarlander_state_floats(void) { return 25; }

void ora_lunarlander_export(const ora_lunarlander *e, float *blob) {
    memset(blob, 0, 25 * sizeof(float));
    if (!e->has_world) return;
    for (int b = 0; b < 3; ++b) {
        const b2body *bd = &e->world.bodies[1 + b];
        float *r = blob + 6 * b;
        r[0] = bd->xf.p.x; r[1] = bd->xf.p.y; r[2] = bd->sw.a; r[3] = bd->linvel.x; r[4] = bd->linvel.y; r[5] = bd->angvel;
    }
    blob[18] = (float)e->det_legs[0]; blob[19] = (float)e->det_legs[1];
    blob[20] = (float)(e->game_over || e->det_game_over);
    blob[21] = (float)e->world.bodies[1].awake;
    blob[22] = e->prev_shaping_some ? e->prev_shaping : NAN;
    memcpy(&blob[23], &e->wind_idx, 4); memcpy(&blob[24], &e->torque_idx, 4);
}

void ora_lunarlander_import(ora_lunarlander *e, const float *blob) {
    ora_lunarlander_set_state(e, blob, blob[18] > 0.5f, blob[19] > 0.5f);
}
