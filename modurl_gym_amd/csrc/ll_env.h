// ll_env.h — LunarLanderV3 environment logic around the per-lane world (ll_world.h): state columns,
// load/store, step(), reset(), the Testable seam, and the host-side constants.  Everything here is
// plain C++ over ll_b2.h/ll_world.h so that, besides the HIP kernels (lunar_lander.hip), the CPU test
// tests/native/ll_host_check.cpp can compile the SAME source for the host and compare it with the
// oracle without a GPU (a test of the kernel logic, not a product CPU path).
#pragma once
#include <math.h>
#include <string.h>

#include "ll_world.h"
#include "philox.h"

namespace mgym {
using namespace ll;

// State words of one environment.  LL_RECORD_LAYOUT 1 (default): one RECORD of kRec words per environment, 128-byte aligned, the words the
// free-flight role touches in its first two cache lines (0-55), terrain and contact cache behind them — every role reads and writes whole
// lines of its own environments, however scattered those are (the contact role's list holds ~9 % of the population: with one COLUMN per
// word each of its 4-byte accesses moved a 64-byte line, ~350 MB of the 612 MB counted per step of 262 144 envs).  0: the column layout
// of rounds 1-3 ([word][n_pad]), kept for A/B runs.
#ifndef LL_RECORD_LAYOUT
#define LL_RECORD_LAYOUT 1
#endif
enum Col : int {
    C_BODY = 0,        // + 9*i : PX PY CX CY A VX VY W SLEEP
    C_JOINT = 27,      // + 5*j : IMPX IMPY MOTOR LOWER UPPER
    C_FAT = 37,        // + 4*i : lo.x lo.y hi.x hi.y
    C_PREV = 49,
    C_FLAGS = 50,
    C_WIND = 51,
    C_TORQUE = 52,
    C_STEP = 53,
    C_EPISODE = 54,
    C_SEQ = 55,        // (56-63 unused: the third line starts with the terrain)
    C_SMOOTH = 64,     // 11 (75-79 unused)
    C_CONTACT = 80,    // + 16*s : KEY SEQ LNX LNY LPX LPY P0X P0Y P0N P0T P1X P1Y P1N P1T IDS TOI
                       //   KEY bits: 0 exists, 1 touching, 2 enabled, 3-4 body, 5-8 edge, 9-10 manifold type, 11-12 pointCount,
                       //             13 toiFlag, 14-17 toiCount (the last two and TOI only matter inside a step, see C_MID)
                       //   IDS = contact-feature ids of both manifold points, 16 bits each
    C_MID = C_CONTACT + 16 * kSlots,  // state of an unfinished SolveTOI between the launches of one step (solve_toi_part):
                               //   + 4*i : sweep c0.x c0.y a0 alpha0 of body i; + 12 : gA
    C_COUNT = C_MID + 13
};
constexpr int kRec = 288;   // words per record (9 lines of 128 bytes)
static_assert(C_COUNT <= kRec, "record too small");

enum Flag : uint32_t {
    F_AWAKE0 = 1u << 0, F_GAME_OVER = 1u << 3, F_LEG0 = 1u << 4, F_LEG1 = 1u << 5, F_HAS_WORLD = 1u << 6,
    F_DETERMINISTIC = 1u << 7, F_NEW_CONTACTS = 1u << 8, F_PENDING0 = 1u << 9, F_STEPPED = 1u << 12,
    F_PREV_SOME = 1u << 13, F_TOUCHING = 1u << 14 /* some cached contact is touching (worklist bucketing) */,
    F_RESUME = 1u << 15 /* only between two kernels of one step: the free-flight kernel has done the island solve, the contact kernel continues at
                           FindNewContacts (ll_free_store_resume); bits 9-11 then say which proxies moved, C_MID holds the sweeps' start */,
    F_NCONTACT_SHIFT = 16
};

struct LLDev {
    uint32_t* st;       // state: records [n_pad][kRec] (LL_RECORD_LAYOUT 0: columns [C_COUNT][n_pad])
    float* obs;         // engine-owned observation [8][n_pad]
    const float* disp;  // dispersion override [2][n] or nullptr
    uint64_t n, n_pad, seed, env_id_base;
    uint32_t* err;
    unsigned long long* done_count;  // kDoneShards partial counts of finished env-steps (mgym_episode_count)
    // compacted env-index lists built on the device (n words each) and their lengths (count[0..2]):
    //   0 = general steps, 1 = resets (fast path), 2 = resets the fast path declined (general path)
    uint32_t* work_list;
    uint32_t* work_count;
    LLConst k;          // host copy (builder arguments folded into constants)
    const LLConst* kd;  // the same in device memory: kernels read the constants through this pointer (scalar loads), so the
                        // by-value kernel argument never has its address taken — otherwise every lane keeps a private copy
                        // of it in scratch (~400 B/lane), and scratch size caps the number of concurrent waves
    VConstraint* vc_far;  // workspace for the velocity constraints that do not fit a block's LDS: [kSolverCap - near][contact-kernel lanes]
    VConstraint* vc_far_late;  // ... of the late contact launch of the overlapped order (it runs beside the main one: a slice of its own)
    int vc_near;          // constraints per lane kept in LDS (<= the kernel's LDS array; lower only as a test knob)
    int auto_reset;
    int bucket;  // worklist bucketing by F_TOUCHING
    const uint32_t* episode_src;  // staged resets: this LLDev addresses the SHADOW columns; the episode counter comes from the live ones
    int prep;            // staged resets: ll_reset_kernel lists the envs it has reset (L_PREP), their next reset is then prepared
    uint8_t* env_class;  // overlapped launch order: class byte per env for this step (ll_classify_kernel)
    int fused_tail;  // fused order of step(): lists and counters are maintained by ll_epilogue_kernel (lunar_lander.hip)
    int resume;  // the free-flight kernel stores the post-solve state of an env it has to hand over (F_RESUME); the contact kernel continues from it
    int split;   // 0: sequential launch order; 1: overlapped (contact list built by ll_classify_kernel; the free-flight kernel lists only what it declines)
    int contact_blocks;  // single-launch step: how many contact blocks the list is dealt out over (each takes list / contact_blocks lanes, 8..32); 0: 32 lanes per block
#ifdef LL_WHATIF_BUILD
    uint32_t whatif;  // tools/ll_whatif.hip only
#endif
};

// lunar_lander.rs:31-58
constexpr float FPS = 50.0f, SCALE = 30.0f, MAIN_ENGINE_POWER = 13.0f, SIDE_ENGINE_POWER = 0.6f, INITIAL_RANDOM = 1000.0f;
constexpr float LEG_AWAY = 20.0f, LEG_DOWN = 18.0f, SIDE_ENGINE_HEIGHT = 14.0f, SIDE_ENGINE_AWAY = 12.0f, MAIN_ENGINE_Y_LOCATION = 4.0f;
constexpr float VIEWPORT_W = 600.0f, VIEWPORT_H = 400.0f;
constexpr float PI_F32 = 3.14159265358979323846f;

struct EnvRegs {  // LunarLanderV3 fields beside the world (lunar_lander.rs:232-276)
    float prev_shaping; bool prev_some;
    int32_t wind_idx, torque_idx;
    uint32_t step, episode;
    bool has_world, deterministic;
};

#if LL_RECORD_LAYOUT
#define ST_AT(base, col) (base)[(uint64_t)i * kRec + (uint64_t)(col)]
#define LL_CT_STRIDE(d) 1u
constexpr uint64_t ll_state_words(uint64_t n_pad) { return (uint64_t)kRec * n_pad; }
#else
#define ST_AT(base, col) (base)[(uint64_t)(col) * d.n_pad + i]
#define LL_CT_STRIDE(d) (d).n_pad
constexpr uint64_t ll_state_words(uint64_t n_pad) { return (uint64_t)C_COUNT * n_pad; }
#endif
#define ST(col) ST_AT(d.st, col)
#define LLK(d) (*(d).kd)

// mid = true: continue an unfinished SolveTOI (the env was stored by ll_store(..., mid = true) earlier in this step)
__device__ __forceinline__ void ll_load(const LLDev& d, uint64_t i, World& w, EnvRegs& e, CtHot hot, bool mid_in = false) {
    const uint32_t flags = ST(C_FLAGS);
    const bool mid = mid_in || (flags & F_RESUME);   // a resumed env carries its sweeps' start in the C_MID columns too
    w.resume = flags & F_RESUME;
    for (int b = 0; b < 3; ++b) {
        Body& bd = w.b[b];
        const int c = C_BODY + 9 * b;
        bd.xf.p = mk(as_f32(ST(c + 0)), as_f32(ST(c + 1)));
        bd.sw.c = mk(as_f32(ST(c + 2)), as_f32(ST(c + 3)));
        bd.sw.a = as_f32(ST(c + 4));
        bd.v = mk(as_f32(ST(c + 5)), as_f32(ST(c + 6)));
        bd.w = as_f32(ST(c + 7));
        bd.sleepTime = as_f32(ST(c + 8));
        bd.xf.q = rot_set(bd.sw.a);
        bd.sw.localCenter = LLK(d).localCenter[b == 0 ? 0 : 1];
        bd.sw.c0 = bd.sw.c; bd.sw.a0 = bd.sw.a; bd.sw.alpha0 = 0.0f;
        if (mid) {
            bd.sw.c0 = mk(as_f32(ST(C_MID + 4 * b + 0)), as_f32(ST(C_MID + 4 * b + 1)));
            bd.sw.a0 = as_f32(ST(C_MID + 4 * b + 2)); bd.sw.alpha0 = as_f32(ST(C_MID + 4 * b + 3));
        }
        bd.force = mk(0.0f, 0.0f); bd.torque = 0.0f;
        bd.awake = (flags >> b) & 1u;
        bd.islandFlag = false;
        w.fat[b].lo = mk(as_f32(ST(C_FAT + 4 * b + 0)), as_f32(ST(C_FAT + 4 * b + 1)));
        w.fat[b].hi = mk(as_f32(ST(C_FAT + 4 * b + 2)), as_f32(ST(C_FAT + 4 * b + 3)));
    }
    for (int j = 0; j < 2; ++j) {
        const int c = C_JOINT + 5 * j;
        w.jt[j].impulse = mk(as_f32(ST(c + 0)), as_f32(ST(c + 1)));
        w.jt[j].motorImpulse = as_f32(ST(c + 2));
        w.jt[j].lowerImpulse = as_f32(ST(c + 3));
        w.jt[j].upperImpulse = as_f32(ST(c + 4));
    }
    for (int q = 0; q < kEdges; ++q) w.smooth[q] = as_f32(ST(C_SMOOTH + q));
    w.next_seq = ST(C_SEQ);
    w.gA = mid ? as_f32(ST(C_MID + 12)) : 0.0f;
    w.pending = (flags >> 9) & 7u;
    w.newContacts = flags & F_NEW_CONTACTS;
    w.stepped_once = flags & F_STEPPED;
    w.game_over = flags & F_GAME_OVER;
    w.legs[0] = flags & F_LEG0; w.legs[1] = flags & F_LEG1;
    w.overflow = 0u; w.terrain_dirty = false;
#ifdef LL_WHATIF_BUILD
    w.whatif = d.whatif;
#endif
    // the contact cache: manifold words in place, KEY / SEQ / TOI staged in `hot` for the step (ll_b2.h)
    w.cs.p = &ST(C_CONTACT); w.cs.stride = LL_CT_STRIDE(d); w.cs.hot = hot;
    for (int s = 0; s < kSlots; ++s) {
        ct_set_key(w.cs, s, ST(C_CONTACT + 16 * s + 0));
        ct_set_seq(w.cs, s, ST(C_CONTACT + 16 * s + 1));
        if (mid && hot.toi_staged) ct_set_toi(w.cs, s, as_f32(ST(C_CONTACT + 16 * s + 15)));
    }
    e.prev_shaping = as_f32(ST(C_PREV));
    e.prev_some = flags & F_PREV_SOME;
    e.wind_idx = (int32_t)ST(C_WIND); e.torque_idx = (int32_t)ST(C_TORQUE);
    e.step = ST(C_STEP); e.episode = ST(C_EPISODE);
    e.has_world = flags & F_HAS_WORLD;
    e.deterministic = flags & F_DETERMINISTIC;
}

__device__ __forceinline__ uint32_t ll_store(const LLDev& d, uint64_t i, const World& w, const EnvRegs& e, bool mid = false) {   // -> the flag word it stored
    uint32_t flags = 0, ncont = 0, touching = 0;
    for (int b = 0; b < 3; ++b) {
        const Body& bd = w.b[b];
        const int c = C_BODY + 9 * b;
        ST(c + 0) = as_u32(bd.xf.p.x); ST(c + 1) = as_u32(bd.xf.p.y);
        ST(c + 2) = as_u32(bd.sw.c.x); ST(c + 3) = as_u32(bd.sw.c.y);
        ST(c + 4) = as_u32(bd.sw.a);
        ST(c + 5) = as_u32(bd.v.x); ST(c + 6) = as_u32(bd.v.y);
        ST(c + 7) = as_u32(bd.w);
        ST(c + 8) = as_u32(bd.sleepTime);
        if (bd.awake) flags |= 1u << b;
        ST(C_FAT + 4 * b + 0) = as_u32(w.fat[b].lo.x); ST(C_FAT + 4 * b + 1) = as_u32(w.fat[b].lo.y);
        ST(C_FAT + 4 * b + 2) = as_u32(w.fat[b].hi.x); ST(C_FAT + 4 * b + 3) = as_u32(w.fat[b].hi.y);
        if (mid) {
            ST(C_MID + 4 * b + 0) = as_u32(bd.sw.c0.x); ST(C_MID + 4 * b + 1) = as_u32(bd.sw.c0.y);
            ST(C_MID + 4 * b + 2) = as_u32(bd.sw.a0); ST(C_MID + 4 * b + 3) = as_u32(bd.sw.alpha0);
        }
    }
    if (mid) ST(C_MID + 12) = as_u32(w.gA);
    for (int j = 0; j < 2; ++j) {
        const int c = C_JOINT + 5 * j;
        ST(c + 0) = as_u32(w.jt[j].impulse.x); ST(c + 1) = as_u32(w.jt[j].impulse.y);
        ST(c + 2) = as_u32(w.jt[j].motorImpulse); ST(c + 3) = as_u32(w.jt[j].lowerImpulse); ST(c + 4) = as_u32(w.jt[j].upperImpulse);
    }
    if (w.terrain_dirty)
        for (int q = 0; q < kEdges; ++q) ST(C_SMOOTH + q) = as_u32(w.smooth[q]);
    ST(C_SEQ) = w.next_seq;
    for (int s = 0; s < kSlots; ++s) {  // the manifolds are already in their columns; the staged words go back
        uint32_t key = ct_key(w.cs, s);
        if (!mid) key &= kCkRestMask;  // in-step bits (toiFlag, toiCount, islandFlag) do not persist
        ST(C_CONTACT + 16 * s + 0) = key;
        if (!(key & CK_EXISTS)) continue;
        ST(C_CONTACT + 16 * s + 1) = ct_seq(w.cs, s);
        if (mid && w.cs.hot.toi_staged) ST(C_CONTACT + 16 * s + 15) = as_u32(ct_toi(w.cs, s));
        ++ncont;
        touching |= (key & CK_TOUCHING) ? 1u : 0u;
    }
    flags |= (w.game_over ? F_GAME_OVER : 0u) | (w.legs[0] ? F_LEG0 : 0u) | (w.legs[1] ? F_LEG1 : 0u) |
             (e.has_world ? F_HAS_WORLD : 0u) | (e.deterministic ? F_DETERMINISTIC : 0u) |
             (w.newContacts ? F_NEW_CONTACTS : 0u) | ((w.pending & 7u) << 9) | (w.stepped_once ? F_STEPPED : 0u) |
             (e.prev_some ? F_PREV_SOME : 0u) | (touching ? F_TOUCHING : 0u) | (ncont << F_NCONTACT_SHIFT);
    ST(C_FLAGS) = flags;
    ST(C_PREV) = as_u32(e.prev_shaping);
    ST(C_WIND) = (uint32_t)e.wind_idx; ST(C_TORQUE) = (uint32_t)e.torque_idx;
    ST(C_STEP) = e.step; ST(C_EPISODE) = e.episode;
    return flags;
}

// b2Body::ApplyForceToCenter / ApplyTorque / ApplyLinearImpulse on the lander (wake = true)
__device__ __forceinline__ void apply_force_to_center(Body& b, V2 f) {
    if (!b.awake) body_set_awake(b, true);
    b.force = b.force + f;
}
__device__ __forceinline__ void apply_torque(Body& b, float t) {
    if (!b.awake) body_set_awake(b, true);
    b.torque += t;
}
__device__ __forceinline__ void apply_linear_impulse(Body& b, const LLConst& k, V2 impulse, V2 point) {
    if (!b.awake) body_set_awake(b, true);
    b.v = b.v + k.invMass[0] * impulse;
    b.w += k.invI[0] * cross(point - b.sw.c, impulse);
}

// observation, lunar_lander.rs:1095-1121
__device__ __forceinline__ void ll_observe(const Body& lander, bool leg0, bool leg1, float state[8]) {
    V2 pos = lander.xf.p, vel = lander.v;
    const float helipad_y = (VIEWPORT_H / SCALE) / 4.0f;
    state[0] = (pos.x - VIEWPORT_W / SCALE / 2.0f) / (VIEWPORT_W / SCALE / 2.0f);
    state[1] = (pos.y - (helipad_y + LEG_DOWN / SCALE)) / (VIEWPORT_H / SCALE / 2.0f);
    state[2] = vel.x * (VIEWPORT_W / SCALE / 2.0f) / FPS;
    state[3] = vel.y * (VIEWPORT_H / SCALE / 2.0f) / FPS;
    state[4] = lander.sw.a;
    state[5] = 20.0f * lander.w / FPS;
    state[6] = leg0 ? 1.0f : 0.0f;
    state[7] = leg1 ? 1.0f : 0.0f;
}

// step() before the world step: wind and engine impulses on the lander, lunar_lander.rs:926-1048
__device__ __forceinline__ void ll_pre_step(Body& lander, bool leg0, bool leg1, EnvRegs& e, const LLConst& k, uint32_t action, float disp0,
                                            float disp1, float& m_power, float& s_power) {
    if (k.enable_wind) {  // :927-959
        bool legs_contact = leg0 || leg1;
        if (!legs_contact) {
            float wind_mag = mg_tanhf(mg_sinf(0.02f * (float)e.wind_idx) + mg_sinf(PI_F32 * 0.01f * (float)e.wind_idx)) * k.wind_power;
            e.wind_idx += 1;
            apply_force_to_center(lander, mk(wind_mag, 0.0f));
            float torque_mag = mg_tanhf(mg_sinf(0.02f * (float)e.torque_idx) + mg_sinf(PI_F32 * 0.01f * (float)e.torque_idx)) * k.turbulence_power;
            e.torque_idx += 1;
            apply_torque(lander, torque_mag);
        }
    }
    float tip0, tip1;
    mg_sincosf(lander.sw.a, &tip0, &tip1);  // :962-964
    float side0 = -tip1, side1 = tip0;
    float dispersion0, dispersion1;  // :967-976
    if (e.deterministic) { dispersion0 = 0.0f; dispersion1 = 0.0f; }
    else { dispersion0 = disp0 / SCALE; dispersion1 = disp1 / SCALE; }

    m_power = 0.0f;
    if (action == 2u) {  // :979-1003
        m_power = 1.0f;
        V2 lander_pos = lander.xf.p;
        float ox = tip0 * (MAIN_ENGINE_Y_LOCATION / SCALE + 2.0f * dispersion0) + side0 * dispersion1;
        float oy = -tip1 * (MAIN_ENGINE_Y_LOCATION / SCALE + 2.0f * dispersion0) - side1 * dispersion1;
        V2 impulse_pos = mk(lander_pos.x + ox, lander_pos.y + oy);
        V2 impulse_force = mk(-ox * MAIN_ENGINE_POWER * m_power, -oy * MAIN_ENGINE_POWER * m_power);
        apply_linear_impulse(lander, k, impulse_force, impulse_pos);
    }
    s_power = 0.0f;
    if (action == 1u || action == 3u) {  // :1019-1048
        float direction = (float)((int32_t)action - 2);
        s_power = 1.0f;
        V2 lander_pos = lander.xf.p;
        float ox = tip0 * dispersion0 + side0 * (3.0f * dispersion1 + direction * SIDE_ENGINE_AWAY / SCALE);
        float oy = -tip1 * dispersion0 - side1 * (3.0f * dispersion1 + direction * SIDE_ENGINE_AWAY / SCALE);
        V2 impulse_pos = mk(lander_pos.x + ox - tip0 * 17.0f / SCALE, lander_pos.y + oy + tip1 * SIDE_ENGINE_HEIGHT / SCALE);
        V2 impulse_force = mk(-ox * SIDE_ENGINE_POWER * s_power, -oy * SIDE_ENGINE_POWER * s_power);
        apply_linear_impulse(lander, k, impulse_force, impulse_pos);
    }
}

// step() after the world step: observation, shaping reward, termination, lunar_lander.rs:1094-1167
__device__ __forceinline__ void ll_post_step(const Body& lander, bool game_over, bool leg0, bool leg1, EnvRegs& e, float m_power, float s_power,
                                             float state[8], float& reward, uint32_t& done) {
    ll_observe(lander, leg0, leg1, state);
    reward = 0.0f;  // :1126-1139
    float shaping = -100.0f * sqrtf(state[0] * state[0] + state[1] * state[1]) -
                    100.0f * sqrtf(state[2] * state[2] + state[3] * state[3]) - 100.0f * fabsf(state[4]) +
                    10.0f * state[6] + 10.0f * state[7];
    if (e.prev_some) reward = shaping - e.prev_shaping;
    e.prev_some = true; e.prev_shaping = shaping;
    reward -= m_power * 0.30f;
    reward -= s_power * 0.03f;
    done = 0u;  // :1142-1156
    if (game_over || fabsf(state[0]) >= 1.0f) { done = 1u; reward = -100.0f; }
    else if (!lander.awake) { done = 1u; reward = 100.0f; }
    e.step += 1u;
}

// The step of an env the free-flight kernel stopped after the island solve (F_RESUME, ll_free.h): b2Island::Solve's last act —
// FindNewContacts over the proxies that moved, in body-list order — then on with SolveTOI as for everyone else.
__device__ __forceinline__ void ll_resume_after_island(World& w) {
    int moved[3], nm = 0;
    for (int b = 2; b >= 0; --b)
        if (w.pending & (1u << b)) moved[nm++] = b;
    find_new_contacts(w, moved, nm);
    w.pending = 0; w.resume = false;
}

// step(), lunar_lander.rs:919-1167, in resumable pieces (disp = the two raw U(-1,1) draws of :973-974):
//   ll_step_begin    : wind / engines (:926-1048) and world.step (:1066) up to and including the first `toi_budget`
//                      sub-steps of its continuous-collision phase; returns whether world.step is complete
//   ll_step_continue : `toi_budget` more sub-steps (< 0: all that remain)
//   ll_step_finish   : the tail of world.step, observation, reward, termination (:1094-1167)
// ll_env_step = begin(all) + finish.  m_power / s_power are functions of the action alone (:979-981, :1019-1024).
__device__ __forceinline__ bool ll_step_begin(World& w, EnvRegs& e, const PolyTab& tab, const LLConst& k, const CSolverMem& mem, uint32_t action, float disp0,
                                              float disp1, int toi_budget) {
    float m_power, s_power;
    ll_pre_step(w.b[0], w.legs[0], w.legs[1], e, k, action, disp0, disp1, m_power, s_power);
    world_step_begin(w, tab, k, mem);
    return solve_toi_part(w, tab, k, mem, kStepDt, true, toi_budget);
}
__device__ __forceinline__ bool ll_step_continue(World& w, const PolyTab& tab, const LLConst& k, const CSolverMem& mem, int toi_budget) {
    return solve_toi_part(w, tab, k, mem, kStepDt, false, toi_budget);
}
__device__ __forceinline__ void ll_step_finish(World& w, EnvRegs& e, uint32_t action, float state[8], float& reward, uint32_t& done) {
    world_step_end(w);
    const float m_power = action == 2u ? 1.0f : 0.0f, s_power = (action == 1u || action == 3u) ? 1.0f : 0.0f;
    ll_post_step(w.b[0], w.game_over, w.legs[0], w.legs[1], e, m_power, s_power, state, reward, done);
}
__device__ __forceinline__ void ll_env_step(World& w, EnvRegs& e, const PolyTab& tab, const LLConst& k, const CSolverMem& mem, uint32_t action, float disp0,
                            float disp1, float state[8], float& reward, uint32_t& done) {
    (void)ll_step_begin(w, e, tab, k, mem, action, disp0, disp1, -1);  // :1066
    ll_step_finish(w, e, action, state, reward, done);
}

// world/terrain/lander/legs construction shared by reset() (:733-908) and reset_deterministic() (:1256-1438)
__device__ __forceinline__ void ll_build_scene(World& w, EnvRegs& e, const PolyTab& tab, const LLConst& k, const float height_in[12], float initial_y,
                               bool random_force, float u_force0, float u_force1, int32_t wind_idx, int32_t torque_idx, bool deterministic) {
    const float W = VIEWPORT_W / SCALE, H = VIEWPORT_H / SCALE;
    float height[12];
    for (int q = 0; q < 12; ++q) height[q] = height_in[q];
    const float helipad_y = H / 4.0f;
    height[3] = helipad_y; height[4] = helipad_y; height[5] = helipad_y; height[6] = helipad_y; height[7] = helipad_y;  // :766-770
    for (int q = 0; q < kEdges; ++q) {  // :772-774
        int im1 = q > 0 ? q - 1 : 0;
        w.smooth[q] = 0.33f * (height[im1] + height[q] + height[q + 1]);
    }
    (void)W;
    w.terrain_dirty = true;
    const float initial_x = VIEWPORT_W / SCALE / 2.0f;
    for (int b = 0; b < 3; ++b) {  // b2World::CreateBody + CreateFixture + ResetMassData
        Body& bd = w.b[b];
        const float i_f = b == 1 ? -1.0f : 1.0f;
        V2 pos = b == 0 ? mk(initial_x, initial_y) : mk(initial_x - i_f * LEG_AWAY / SCALE, initial_y);
        float angle = b == 0 ? 0.0f : i_f * 0.05f;
        bd.xf.p = pos; bd.xf.q = rot_set(angle);
        bd.sw.localCenter = k.localCenter[poly_of(b)];
        bd.sw.a0 = bd.sw.a = angle; bd.sw.alpha0 = 0.0f;
        bd.sw.c0 = bd.sw.c = xmul(bd.xf, bd.sw.localCenter);
        bd.v = mk(0.0f, 0.0f); bd.w = 0.0f; bd.force = mk(0.0f, 0.0f); bd.torque = 0.0f;
        bd.sleepTime = 0.0f; bd.awake = true; bd.islandFlag = false;
        AABB a = poly_aabb(tab, poly_of(b), bd.xf);  // b2DynamicTree::CreateProxy
        w.fat[b].lo = mk(a.lo.x - b2_aabbExtension, a.lo.y - b2_aabbExtension);
        w.fat[b].hi = mk(a.hi.x + b2_aabbExtension, a.hi.y + b2_aabbExtension);
    }
    for (int j = 0; j < 2; ++j) { w.jt[j].impulse = mk(0.0f, 0.0f); w.jt[j].motorImpulse = 0.0f; w.jt[j].lowerImpulse = 0.0f; w.jt[j].upperImpulse = 0.0f; }
    for (int s = 0; s < kSlots; ++s) ct_set_key(w.cs, s, 0u);  // (ll_load pointed w.cs at the env's contact storage)
    w.next_seq = 1u; w.pending = 7u; w.newContacts = true; w.stepped_once = false;
    w.game_over = false; w.legs[0] = w.legs[1] = false; w.overflow = 0u;
    if (random_force) {  // :845-849
        float force_x = u_force0 * (INITIAL_RANDOM - (-INITIAL_RANDOM)) + (-INITIAL_RANDOM);
        float force_y = u_force1 * (INITIAL_RANDOM - (-INITIAL_RANDOM)) + (-INITIAL_RANDOM);
        apply_force_to_center(w.b[0], mk(force_x, force_y));
    } else {  // :1371-1374, :1434-1435: set_linear_velocity((0,-1)), set_angular_velocity(0)
        for (int b = 0; b < 3; ++b) { w.b[b].v = mk(0.0f, -1.0f); w.b[b].w = 0.0f; }
    }
    if (k.enable_wind) { e.wind_idx = wind_idx; e.torque_idx = torque_idx; }
    e.prev_some = false; e.prev_shaping = 0.0f;
    e.has_world = true; e.deterministic = deterministic;
}

__device__ __forceinline__ void ll_dispersion(const LLDev& d, uint64_t i, const EnvRegs& e, float& d0, float& d1) {
    if (d.disp) { d0 = d.disp[i]; d1 = d.disp[d.n + i]; return; }
    Philox4 r = env_draw(d.seed, d.env_id_base + i, e.episode, SLOT_STEP_BASE + e.step);
    d0 = u23(r.w[0]) * 2.0f + -1.0f;  // rng.random_range(-1.0..1.0), :973-974
    d1 = u23(r.w[1]) * 2.0f + -1.0f;
}

// reset() up to (not including) its implicit step(0): lunar_lander.rs:727-908
__device__ __forceinline__ void ll_reset_scene(const LLDev& d, uint64_t i, World& w, EnvRegs& e, const PolyTab& tab) {
    const uint64_t gid = d.env_id_base + i;
    uint32_t r[16];
    for (uint32_t s = 0; s < 4; ++s) {
        Philox4 p = env_draw(d.seed, gid, e.episode, SLOT_RESET0 + s);
        for (int q = 0; q < 4; ++q) r[4 * s + q] = p.w[q];
    }
    const float H = VIEWPORT_H / SCALE;
    float height[12];
    for (int q = 0; q < 12; ++q) height[q] = u23(r[q]) * (H / 2.0f - 0.0f) + 0.0f;  // :755
    int32_t wi = -9999 + (int32_t)(((uint64_t)r[14] * 19998u) >> 32);               // :855-856
    int32_t ti = -9999 + (int32_t)(((uint64_t)r[15] * 19998u) >> 32);
    ll_build_scene(w, e, tab, LLK(d), height, VIEWPORT_H / SCALE, true, u23(r[12]), u23(r[13]), wi, ti, false);
    e.step = 0u;
}

// reset(), lunar_lander.rs:727-917 (episode counter = Philox counter word 2)
__device__ inline void ll_env_reset(const LLDev& d, uint64_t i, World& w, EnvRegs& e, const PolyTab& tab, const CSolverMem& mem, float state[8]) {
    ll_reset_scene(d, i, w, e, tab);
    float d0, d1, reward; uint32_t done;
    ll_dispersion(d, i, e, d0, d1);
    ll_env_step(w, e, tab, LLK(d), mem, 0u, d0, d1, state, reward, done);  // :911-916
    e.episode += 1u;
}

// Testable::set_state (lunar_lander.rs:1444-1554): raw = {x, y, angle, vx, vy, w} of lander, leg0, leg1
__device__ inline void ll_apply_set_state(World& w, const PolyTab& tab, const float raw[18], bool leg0, bool leg1) {
    for (int b = 0; b < 3; ++b) {
        Body& bd = w.b[b];
        const float* r = raw + 6 * b;
        V2 v = mk(r[3], r[4]);
        float wv = r[5];
        bd.xf.q = rot_set(r[2]);  // b2Body::SetTransform
        bd.xf.p = mk(r[0], r[1]);
        bd.sw.c = xmul(bd.xf, bd.sw.localCenter);
        bd.sw.a = r[2]; bd.sw.c0 = bd.sw.c; bd.sw.a0 = r[2];
        if (fixture_sync(w, tab, b, bd.xf, bd.xf)) w.pending |= 1u << b;
        w.newContacts = true;
        if (dot(v, v) > 0.0f) body_set_awake(bd, true);  // SetLinearVelocity
        bd.v = v;
        if (wv * wv > 0.0f) body_set_awake(bd, true);    // SetAngularVelocity
        bd.w = wv;
    }
    for (int b = 0; b < 3; ++b) body_set_awake(w.b[b], true);  // :1532-1535
    w.legs[0] = leg0; w.legs[1] = leg1;                          // :1539-1553
}

// ---- host side: polygon/mass constants exactly as Box2D derives them -------------------------------------
namespace host {
struct HV { float x, y; };
inline HV hv(float x, float y) { return HV{x, y}; }
inline HV sub(HV a, HV b) { return hv(a.x - b.x, a.y - b.y); }
inline HV add(HV a, HV b) { return hv(a.x + b.x, a.y + b.y); }
inline HV mul(float s, HV a) { return hv(s * a.x, s * a.y); }
inline float hdot(HV a, HV b) { return a.x * b.x + a.y * b.y; }
inline float hcross(HV a, HV b) { return a.x * b.y - a.y * b.x; }
inline float hlen2(HV a) { return a.x * a.x + a.y * a.y; }

// b2PolygonShape::Set (weld, gift-wrapped hull, normals)
inline int poly_set(const HV* in, int count, HV* v, HV* nrm) {
    HV ps[8]; int n = 0;
    for (int i = 0; i < count; ++i) {
        bool unique = true;
        for (int j = 0; j < n; ++j)
            if (hlen2(sub(in[i], ps[j])) < ((0.5f * ll::b2_linearSlop) * (0.5f * ll::b2_linearSlop))) { unique = false; break; }
        if (unique) ps[n++] = in[i];
    }
    int i0 = 0; float x0 = ps[0].x;
    for (int i = 1; i < n; ++i) {
        float x = ps[i].x;
        if (x > x0 || (x == x0 && ps[i].y < ps[i0].y)) { i0 = i; x0 = x; }
    }
    int hull[8], m = 0, ih = i0;
    for (;;) {
        hull[m] = ih;
        int ie = 0;
        for (int j = 1; j < n; ++j) {
            if (ie == ih) { ie = j; continue; }
            HV r = sub(ps[ie], ps[hull[m]]), vv = sub(ps[j], ps[hull[m]]);
            float c = hcross(r, vv);
            if (c < 0.0f) ie = j;
            if (c == 0.0f && hlen2(vv) > hlen2(r)) ie = j;
        }
        ++m; ih = ie;
        if (ie == i0) break;
    }
    for (int i = 0; i < m; ++i) v[i] = ps[hull[i]];
    for (int i = 0; i < m; ++i) {
        int i2 = i + 1 < m ? i + 1 : 0;
        HV e = sub(v[i2], v[i]);
        HV nn = hv(1.0f * e.y, -1.0f * e.x);
        float l = sqrtf(nn.x * nn.x + nn.y * nn.y);
        if (l >= FLT_EPSILON) { float inv = 1.0f / l; nn.x *= inv; nn.y *= inv; }
        nrm[i] = nn;
    }
    return m;
}
// b2PolygonShape::ComputeMass + b2Body::ResetMassData (single fixture)
inline void mass_data(const HV* v, int count, float density, float& mass, float& invMass, float& invI, HV& localCenter) {
    HV center = hv(0.0f, 0.0f); float area = 0.0f, I = 0.0f;
    HV s = v[0];
    const float k_inv3 = 1.0f / 3.0f;
    for (int i = 0; i < count; ++i) {
        HV e1 = sub(v[i], s), e2 = i + 1 < count ? sub(v[i + 1], s) : sub(v[0], s);
        float D = hcross(e1, e2);
        float triangleArea = 0.5f * D;
        area += triangleArea;
        center = add(center, mul(triangleArea * k_inv3, add(e1, e2)));
        float intx2 = e1.x * e1.x + e2.x * e1.x + e2.x * e2.x;
        float inty2 = e1.y * e1.y + e2.y * e1.y + e2.y * e2.y;
        I += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    mass = density * area;
    center = mul(1.0f / area, center);
    HV mcenter = add(center, s);
    float mI = density * I;
    mI += mass * (hdot(mcenter, mcenter) - hdot(center, center));
    HV lc = mul(mass, mcenter);
    invMass = 1.0f / mass;
    lc = mul(invMass, lc);
    float bI = mI - mass * hdot(lc, lc);
    invI = 1.0f / bI;
    localCenter = lc;
}
}  // namespace host


// LLConst for one handle: polygon tables, masses, joint parameters (host code)
inline V2 mk_host(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
inline void ll_make_const(LLConst& k, float gravity, int enable_wind, float wind_power, float turbulence_power) {
    memset(&k, 0, sizeof k);
    k.gravity = gravity; k.wind_power = wind_power; k.turbulence_power = turbulence_power;
    k.enable_wind = enable_wind;
        // lander hexagon (lunar_lander.rs:39-46,828-832) and leg box (:875)
        const float LP[6][2] = {{-14.0f, 17.0f}, {-17.0f, 0.0f}, {-17.0f, -10.0f}, {17.0f, -10.0f}, {17.0f, 0.0f}, {14.0f, 17.0f}};
        host::HV in[6], v[8], nr[8];
        for (int i = 0; i < 6; ++i) in[i] = host::hv(LP[i][0] / SCALE, LP[i][1] / SCALE);
        int cnt = host::poly_set(in, 6, v, nr);
        k.poly_count[0] = cnt;
        for (int i = 0; i < cnt; ++i) { k.poly_v[0][i] = mk_host(v[i].x, v[i].y); k.poly_n[0][i] = mk_host(nr[i].x, nr[i].y); }
        host::HV lc;
        host::mass_data(v, cnt, 5.0f, k.mass[0], k.invMass[0], k.invI[0], lc);  // density 5 (:836)
        k.localCenter[0] = mk_host(lc.x, lc.y);
        const float hx = 2.0f / SCALE, hy = 8.0f / SCALE;  // SetAsBox(LEG_W/SCALE, LEG_H/SCALE)
        host::HV bv[4] = {host::hv(-hx, -hy), host::hv(hx, -hy), host::hv(hx, hy), host::hv(-hx, hy)};
        const float bn[4][2] = {{0.0f, -1.0f}, {1.0f, 0.0f}, {0.0f, 1.0f}, {-1.0f, 0.0f}};
        k.poly_count[1] = 4;
        for (int i = 0; i < 4; ++i) { k.poly_v[1][i] = mk_host(bv[i].x, bv[i].y); k.poly_n[1][i] = mk_host(bn[i][0], bn[i][1]); }
        host::mass_data(bv, 4, 1.0f, k.mass[1], k.invMass[1], k.invI[1], lc);  // density 1 (:879)
        k.localCenter[1] = mk_host(lc.x, lc.y);
        k.friction[0] = sqrtf(0.1f * 0.1f);  // b2MixFriction(ground 0.1, lander 0.1) (:790,837)
        k.friction[1] = sqrtf(0.1f * 0.2f);  // legs keep b2FixtureDef's default friction 0.2
        for (int j = 0; j < 2; ++j) {        // :887-902
            const float i_f = j == 0 ? -1.0f : 1.0f;
            k.localAnchorB[j] = mk_host(i_f * LEG_AWAY / SCALE, LEG_DOWN / SCALE);
            k.motorSpeed[j] = 0.3f * i_f;
            if (i_f == -1.0f) { k.lowerAngle[j] = 0.9f - 0.5f; k.upperAngle[j] = 0.9f; }
            else { k.lowerAngle[j] = -0.9f; k.upperAngle[j] = -0.9f + 0.5f; }
        }
        k.maxMotorTorque = 40.0f;  // LEG_SPRING_TORQUE
}

}  // namespace mgym
