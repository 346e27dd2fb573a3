// ll_free.h — contact-free fast path of the LunarLander step.
//
// Most environment-steps are free flight: no cached ground contact, all three bodies awake.  For those,
// b2World::Step (ll_world.h) reduces to: integrate velocities -> 180 sweeps over the two revolute joints
// -> integrate positions -> a few joint position iterations -> sleep bookkeeping -> fat-AABB update.
// This file restates exactly that reduced step with every quantity in VGPRs (no contact cache, no local
// arrays), using the SAME primitives as the general path (rj_*, integrate_*, sleep_update, fixture_sync),
// in the same order, so results are bit-identical to world_step().  It never creates a contact: if the
// updated fat AABBs start to overlap a ground edge it stops after the island solve and the contact kernel resumes the
// environment from there (FREE_RESUME); if a precondition fails at the start it reports FREE_DECLINED and the caller hands the
// untouched environment to the general path through the worklist.
#pragma once
#include "ll_env.h"

namespace mgym {

// columns a contact-free environment needs (no contact slots, no terrain unless a proxy moved)
struct FreeRegs {
    Body b[3];
    Joint jt[2];
    AABB fat[3];
    uint32_t flags;
};

__device__ __forceinline__ void ll_free_load(const LLDev& d, uint64_t i, FreeRegs& f, EnvRegs& e) {
    const uint32_t flags = ST(C_FLAGS);
    f.flags = flags;
    for (int b = 0; b < 3; ++b) {
        Body& bd = f.b[b];
        const int c = C_BODY + 9 * b;
        bd.xf.p = mk(as_f32(ST(c + 0)), as_f32(ST(c + 1)));
        bd.sw.c = mk(as_f32(ST(c + 2)), as_f32(ST(c + 3)));
        bd.sw.a = as_f32(ST(c + 4));
        bd.v = mk(as_f32(ST(c + 5)), as_f32(ST(c + 6)));
        bd.w = as_f32(ST(c + 7));
        bd.sleepTime = as_f32(ST(c + 8));
        bd.sw.localCenter = LLK(d).localCenter[b == 0 ? 0 : 1];
        bd.sw.c0 = bd.sw.c; bd.sw.a0 = bd.sw.a; bd.sw.alpha0 = 0.0f;
        bd.force = mk(0.0f, 0.0f); bd.torque = 0.0f;
        bd.awake = true; bd.islandFlag = false;
        f.fat[b].lo = mk(as_f32(ST(C_FAT + 4 * b + 0)), as_f32(ST(C_FAT + 4 * b + 1)));
        f.fat[b].hi = mk(as_f32(ST(C_FAT + 4 * b + 2)), as_f32(ST(C_FAT + 4 * b + 3)));
    }
    for (int j = 0; j < 2; ++j) {
        const int c = C_JOINT + 5 * j;
        f.jt[j].impulse = mk(as_f32(ST(c + 0)), as_f32(ST(c + 1)));
        f.jt[j].motorImpulse = as_f32(ST(c + 2));
        f.jt[j].lowerImpulse = as_f32(ST(c + 3));
        f.jt[j].upperImpulse = as_f32(ST(c + 4));
    }
    e.prev_shaping = as_f32(ST(C_PREV));
    e.prev_some = flags & F_PREV_SOME;
    e.wind_idx = (int32_t)ST(C_WIND); e.torque_idx = (int32_t)ST(C_TORQUE);
    e.step = ST(C_STEP); e.episode = ST(C_EPISODE);
    e.has_world = true;
    e.deterministic = flags & F_DETERMINISTIC;
}

__device__ __forceinline__ void ll_free_store(const LLDev& d, uint64_t i, const FreeRegs& f, const EnvRegs& e) {
    uint32_t flags = f.flags & (F_GAME_OVER | F_LEG0 | F_LEG1 | F_HAS_WORLD | F_DETERMINISTIC);
    for (int b = 0; b < 3; ++b) {
        const Body& bd = f.b[b];
        const int c = C_BODY + 9 * b;
        ST(c + 0) = as_u32(bd.xf.p.x); ST(c + 1) = as_u32(bd.xf.p.y);
        ST(c + 2) = as_u32(bd.sw.c.x); ST(c + 3) = as_u32(bd.sw.c.y);
        ST(c + 4) = as_u32(bd.sw.a);
        ST(c + 5) = as_u32(bd.v.x); ST(c + 6) = as_u32(bd.v.y);
        ST(c + 7) = as_u32(bd.w);
        ST(c + 8) = as_u32(bd.sleepTime);
        if (bd.awake) flags |= 1u << b;
        ST(C_FAT + 4 * b + 0) = as_u32(f.fat[b].lo.x); ST(C_FAT + 4 * b + 1) = as_u32(f.fat[b].lo.y);
        ST(C_FAT + 4 * b + 2) = as_u32(f.fat[b].hi.x); ST(C_FAT + 4 * b + 3) = as_u32(f.fat[b].hi.y);
    }
    for (int j = 0; j < 2; ++j) {
        const int c = C_JOINT + 5 * j;
        ST(c + 0) = as_u32(f.jt[j].impulse.x); ST(c + 1) = as_u32(f.jt[j].impulse.y);
        ST(c + 2) = as_u32(f.jt[j].motorImpulse); ST(c + 3) = as_u32(f.jt[j].lowerImpulse); ST(c + 4) = as_u32(f.jt[j].upperImpulse);
    }
    flags |= F_STEPPED | (e.prev_some ? F_PREV_SOME : 0u);  // newContacts = false, pending = 0, no contacts
    ST(C_FLAGS) = flags;
    ST(C_PREV) = as_u32(e.prev_shaping);
    ST(C_WIND) = (uint32_t)e.wind_idx; ST(C_TORQUE) = (uint32_t)e.torque_idx;
    ST(C_STEP) = e.step; ST(C_EPISODE) = e.episode;
}

// fat AABB of ground edge `e` of env i, terrain read from HBM on demand (same arithmetic as edge_fat)
__device__ __forceinline__ AABB ll_edge_fat_global(const LLDev& d, uint64_t i, int e) {
    V2 v1, v2;
    if (e == 0) { v1 = mk(0.0f, 0.0f); v2 = mk(kW, 0.0f); }
    else {
        const float cw = kW / 10.0f;
        v1 = mk(cw * (float)(e - 1), as_f32(ST(C_SMOOTH + e - 1)));
        v2 = mk(cw * (float)e, as_f32(ST(C_SMOOTH + e)));
    }
    V2 lower = mk(fmin2(v1.x, v2.x), fmin2(v1.y, v2.y)), upper = mk(fmax2(v1.x, v2.x), fmax2(v1.y, v2.y));
    AABB a;
    a.lo = mk((lower.x - b2_polygonRadius) - b2_aabbExtension, (lower.y - b2_polygonRadius) - b2_aabbExtension);
    a.hi = mk((upper.x + b2_polygonRadius) + b2_aabbExtension, (upper.y + b2_polygonRadius) + b2_aabbExtension);
    return a;
}
__device__ __forceinline__ bool ll_any_ground_overlap(const LLDev& d, uint64_t i, AABB box) {
    for (int e = 0; e < kEdges; ++e)
        if (aabb_overlap(box, ll_edge_fat_global(d, i, e))) return true;
    return false;
}
// Does the fat AABB of any body in `mask` overlap a ground edge's?  Same answer as ll_any_ground_overlap per body, without its chain
// of dependent loads (11 edges x 3 bodies, each edge's two heights fetched only after the previous edge's test: ~30 round trips per wave):
//  * every terrain vertex lies below kTerrainTop — smooth_y = 0.33 (h0 + h1 + h2) with h < H/2 (lunar_lander.rs:753-774), at most
//    6.6 m, and an edge's fat box reaches polygonRadius + aabbExtension = 0.11 m above its higher vertex — so a body whose box starts
//    above that overlaps nothing (aabb_overlap would say so for every edge: b.lo.y - a.hi.y > 0) and needs no terrain at all;
//  * otherwise the 11 heights are fetched together (independent loads, one round trip) and tested from registers.
constexpr float kTerrainTop = 6.75f;
#if defined(__HIPCC__)
#define LL_LOAD_AGENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define LL_LOAD_AGENT(p) (*(p))   // (the host build of this source, tests/native/ll_host_check.cpp)
#endif
// COH (ll_roll.h): the record may have been written by another compute unit during THIS launch — the terrain is fetched with
// agent-scope loads (they bypass this compute unit's vector L1, which another unit's stores never refresh)
template <bool COH = false>
__device__ __forceinline__ bool ll_ground_overlap_any_of(const LLDev& d, uint64_t i, const AABB fat[3], uint32_t mask) {
    uint32_t need = 0u;
#pragma unroll
    for (int b = 0; b < 3; ++b)
        if (((mask >> b) & 1u) && !(fat[b].lo.y > kTerrainTop)) need |= 1u << b;
    if (need == 0u) return false;
    float sm[kEdges];
#pragma unroll
    for (int q = 0; q < kEdges; ++q) sm[q] = as_f32(COH ? LL_LOAD_AGENT(&ST(C_SMOOTH + q)) : ST(C_SMOOTH + q));
    bool any = false;
#pragma unroll
    for (int e = 0; e < kEdges; ++e) {
        V2 v1, v2;
        if (e == 0) { v1 = mk(0.0f, 0.0f); v2 = mk(kW, 0.0f); }
        else { const float cw = kW / 10.0f; v1 = mk(cw * (float)(e - 1), sm[e - 1]); v2 = mk(cw * (float)e, sm[e]); }
        V2 lower = mk(fmin2(v1.x, v2.x), fmin2(v1.y, v2.y)), upper = mk(fmax2(v1.x, v2.x), fmax2(v1.y, v2.y));
        AABB a;   // (same arithmetic as edge_fat)
        a.lo = mk((lower.x - b2_polygonRadius) - b2_aabbExtension, (lower.y - b2_polygonRadius) - b2_aabbExtension);
        a.hi = mk((upper.x + b2_polygonRadius) + b2_aabbExtension, (upper.y + b2_polygonRadius) + b2_aabbExtension);
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if ((need >> b) & 1u) any = any || aabb_overlap(fat[b], a);
    }
    return any;
}

// eligibility from the flag word alone: world exists, all three bodies awake, no cached contacts
__device__ __forceinline__ bool ll_free_eligible(uint32_t flags) {
    return (flags & F_HAS_WORLD) && ((flags & 7u) == 7u) && (((flags >> F_NCONTACT_SHIFT) & 15u) == 0u);
}

// MoveProxy on a register-resident fat AABB (same rule as move_proxy in ll_world.h)
__device__ __forceinline__ bool ll_free_move_proxy(AABB& tree, AABB aabb, V2 displacement) {
    AABB fatAABB;
    fatAABB.lo = mk(aabb.lo.x - b2_aabbExtension, aabb.lo.y - b2_aabbExtension);
    fatAABB.hi = mk(aabb.hi.x + b2_aabbExtension, aabb.hi.y + b2_aabbExtension);
    V2 dd = b2_aabbMultiplier * displacement;
    if (dd.x < 0.0f) fatAABB.lo.x += dd.x; else fatAABB.hi.x += dd.x;
    if (dd.y < 0.0f) fatAABB.lo.y += dd.y; else fatAABB.hi.y += dd.y;
    if (aabb_contains(tree, aabb)) {
        AABB huge;
        huge.lo = mk(fatAABB.lo.x - 4.0f * b2_aabbExtension, fatAABB.lo.y - 4.0f * b2_aabbExtension);
        huge.hi = mk(fatAABB.hi.x + 4.0f * b2_aabbExtension, fatAABB.hi.y + 4.0f * b2_aabbExtension);
        if (aabb_contains(huge, tree)) return false;
    }
    tree = fatAABB;
    return true;
}

// An env whose step ends with a contact being CREATED (its new fat AABB overlaps a ground edge's) has done everything
// b2World::Step does before FindNewContacts exactly as the general path would have — the same primitives in the same
// order — so its post-solve state is stored (bodies, joints, fat AABBs, the sweeps' start in the C_MID columns, which
// proxies moved in the pending bits, F_RESUME) and the contact kernel RESUMES it there: FindNewContacts, SolveTOI, the tail
// of the step (ll_resume_after_island, ll_env.h) instead of redoing the 180-sweep island solve from the old state.
__device__ __forceinline__ void ll_free_store_resume(const LLDev& d, uint64_t i, const FreeRegs& f, const EnvRegs& e, uint32_t moved) {
    uint32_t flags = f.flags & (F_GAME_OVER | F_LEG0 | F_LEG1 | F_HAS_WORLD | F_DETERMINISTIC | F_STEPPED | F_PREV_SOME);
    for (int b = 0; b < 3; ++b) {
        const Body& bd = f.b[b];
        const int c = C_BODY + 9 * b;
        ST(c + 0) = as_u32(bd.xf.p.x); ST(c + 1) = as_u32(bd.xf.p.y);
        ST(c + 2) = as_u32(bd.sw.c.x); ST(c + 3) = as_u32(bd.sw.c.y);
        ST(c + 4) = as_u32(bd.sw.a);
        ST(c + 5) = as_u32(bd.v.x); ST(c + 6) = as_u32(bd.v.y);
        ST(c + 7) = as_u32(bd.w);
        ST(c + 8) = as_u32(bd.sleepTime);
        if (bd.awake) flags |= 1u << b;
        ST(C_FAT + 4 * b + 0) = as_u32(f.fat[b].lo.x); ST(C_FAT + 4 * b + 1) = as_u32(f.fat[b].lo.y);
        ST(C_FAT + 4 * b + 2) = as_u32(f.fat[b].hi.x); ST(C_FAT + 4 * b + 3) = as_u32(f.fat[b].hi.y);
        ST(C_MID + 4 * b + 0) = as_u32(bd.sw.c0.x); ST(C_MID + 4 * b + 1) = as_u32(bd.sw.c0.y);   // SolveTOI sweeps from here
        ST(C_MID + 4 * b + 2) = as_u32(bd.sw.a0); ST(C_MID + 4 * b + 3) = as_u32(0.0f);
    }
    ST(C_MID + 12) = as_u32(0.0f);
    for (int j = 0; j < 2; ++j) {
        const int c = C_JOINT + 5 * j;
        ST(c + 0) = as_u32(f.jt[j].impulse.x); ST(c + 1) = as_u32(f.jt[j].impulse.y);
        ST(c + 2) = as_u32(f.jt[j].motorImpulse); ST(c + 3) = as_u32(f.jt[j].lowerImpulse); ST(c + 4) = as_u32(f.jt[j].upperImpulse);
    }
    ST(C_FLAGS) = flags | F_RESUME | ((moved & 7u) << 9);   // no cached contacts yet (count 0), newContacts = false
    ST(C_WIND) = (uint32_t)e.wind_idx; ST(C_TORQUE) = (uint32_t)e.torque_idx;   // advanced by the wind of this step (:927-959)
}

// One full Gym::step of a contact-free environment.  FREE_DONE: stepped.  FREE_DECLINED: nothing may be stored, the env
// must go through the general path from its old state.  FREE_RESUME: the island solve is done and a contact has to be
// created: store with ll_free_store_resume(moved) and hand the env to the contact kernel.
// In three pieces — begin, 180 x sweep, finish — so that a lane can take TWO environments through the sweeps side by side
// (lunar_lander.hip ll_free_pass2: two independent dependent chains in one instruction stream).
enum { FREE_DECLINED = 0, FREE_DONE = 1, FREE_RESUME = 2 };
#ifndef LL_FREE_POS_STAT
#define LL_FREE_POS_STAT(iters)   // host statistics builds count the joint position iterations of a free-flight step (61: ran out of its 60)
#endif
struct FreeSolve { Pos pos[3]; Vel vel[3]; float m_power, s_power; };
constexpr float kFreeDt = 1.0f / 50.0f;

// wind / engines and b2Island::Solve up to the velocity sweeps; false: FREE_DECLINED (nothing was changed that matters: f, e are the caller's copies)
template <bool COH = false>
__device__ __forceinline__ bool ll_free_begin(const LLDev& d, uint64_t i, FreeRegs& f, EnvRegs& e, uint32_t action, float disp0, float disp1, FreeSolve& s) {
    const LLConst& k = LLK(d);
    if (f.flags & F_NEW_CONTACTS) {  // b2World::Step: pending FindNewContacts (after reset / set_state)
        if (ll_ground_overlap_any_of<COH>(d, i, f.fat, (f.flags >> 9) & 7u)) return false;
    }
    ll_pre_step(f.b[0], f.flags & F_LEG0, f.flags & F_LEG1, e, k, action, disp0, disp1, s.m_power, s.s_power);

    // ---- b2Island::Solve, island = {leg1, lander, leg0}, joints in DFS order [joint(leg1), joint(leg0)] ----
    const float dt = kFreeDt, inv_dt = 1.0f / dt, h = dt;
    const float dtRatio = ((f.flags & F_STEPPED) ? inv_dt : 0.0f) * dt;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        Body& bd = f.b[b];
        bd.sw.c0 = bd.sw.c; bd.sw.a0 = bd.sw.a;
        s.pos[b].c = bd.sw.c; s.pos[b].a = bd.sw.a;
        integrate_velocity(bd, b == 0 ? 0 : 1, k, h, s.vel[b].v, s.vel[b].w);
    }
    rj_init_velocity(f.jt[1], 2, k, s.pos[0].a, s.pos[2].a, s.vel[0], s.vel[2], dtRatio);
    rj_init_velocity(f.jt[0], 1, k, s.pos[0].a, s.pos[1].a, s.vel[0], s.vel[1], dtRatio);
    return true;
}
// one of the 180 velocity sweeps: the two joints
__device__ __forceinline__ void ll_free_sweep(FreeRegs& f, const LLConst& k, FreeSolve& s) {
    const float dt = kFreeDt, inv_dt = 1.0f / dt;
    rj_solve_velocity(f.jt[1], 1, k, s.vel[0].v, s.vel[0].w, s.vel[2].v, s.vel[2].w, dt, inv_dt);
    rj_solve_velocity(f.jt[0], 0, k, s.vel[0].v, s.vel[0].w, s.vel[1].v, s.vel[1].w, dt, inv_dt);
}
// positions, sleep, broad phase, observation / reward / termination
template <bool COH = false>
__device__ __forceinline__ int ll_free_finish(const LLDev& d, uint64_t i, FreeRegs& f, EnvRegs& e, const PolyTab& tab, FreeSolve& s,
                                              float state[8], float& reward, uint32_t& done, uint32_t& moved) {
    const LLConst& k = LLK(d);
    const float h = kFreeDt;
    moved = 0u;
    Pos* const pos = s.pos; Vel* const vel = s.vel;
#pragma unroll
    for (int b = 0; b < 3; ++b) integrate_position(pos[b].c, pos[b].a, vel[b].v, vel[b].w, h);
    bool positionSolved = false;
    for (int it = 0; it < 60; ++it) {
        bool ok1 = rj_solve_position(f.jt[1], 1, k, pos[0], pos[2]);
        bool ok0 = rj_solve_position(f.jt[0], 0, k, pos[0], pos[1]);
        if (ok1 && ok0) { positionSolved = true; LL_FREE_POS_STAT(it + 1); break; }
        if (it == 59) LL_FREE_POS_STAT(61);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        Body& bd = f.b[b];
        bd.sw.c = pos[b].c; bd.sw.a = pos[b].a; bd.v = vel[b].v; bd.w = vel[b].w;
        body_sync_transform(bd);
    }
    float minSleepTime = FLT_MAX;
    minSleepTime = sleep_update(f.b[2], h, minSleepTime);
    minSleepTime = sleep_update(f.b[0], h, minSleepTime);
    minSleepTime = sleep_update(f.b[1], h, minSleepTime);
    if (minSleepTime >= b2_timeToSleep && positionSolved)
        for (int b = 0; b < 3; ++b) body_set_awake(f.b[b], false);

    // b2Body::SynchronizeFixtures in body-list order (leg1, leg0, lander), then FindNewContacts
    bool create = false;
    for (int b = 2; b >= 0; --b) {
        Body& bd = f.b[b];
        Xf xf1 = bd.xf;
        if (bd.awake) {
            xf1.q = rot_set(bd.sw.a0);
            xf1.p = bd.sw.c0 - rmul(xf1.q, bd.sw.localCenter);
        }
        AABB a1 = poly_aabb(tab, poly_of(b), xf1), a2 = poly_aabb(tab, poly_of(b), bd.xf);
        AABB aabb;
        aabb.lo = mk(fmin2(a1.lo.x, a2.lo.x), fmin2(a1.lo.y, a2.lo.y));
        aabb.hi = mk(fmax2(a1.hi.x, a2.hi.x), fmax2(a1.hi.y, a2.hi.y));
        V2 c1 = 0.5f * (a1.lo + a1.hi), c2 = 0.5f * (a2.lo + a2.hi);
        if (ll_free_move_proxy(f.fat[b], aabb, c2 - c1)) moved |= 1u << b;
    }
    create = ll_ground_overlap_any_of<COH>(d, i, f.fat, moved);  // a contact would be created: the contact path takes over from here
    if (create) return FREE_RESUME;
    // SolveTOI: no contacts.  ClearForces; inv_dt0 = inv_dt.
    ll_post_step(f.b[0], f.flags & F_GAME_OVER, f.flags & F_LEG0, f.flags & F_LEG1, e, s.m_power, s.s_power, state, reward, done);
    return FREE_DONE;
}

__device__ __forceinline__ int ll_free_env_step(const LLDev& d, uint64_t i, FreeRegs& f, EnvRegs& e, const PolyTab& tab, uint32_t action,
                                                 float disp0, float disp1, float state[8], float& reward, uint32_t& done, uint32_t& moved) {
    FreeSolve s;
    moved = 0u;
    if (!ll_free_begin(d, i, f, e, action, disp0, disp1, s)) return FREE_DECLINED;
    for (int it = 0; it < 180; ++it) ll_free_sweep(f, LLK(d), s);
    return ll_free_finish(d, i, f, e, tab, s, state, reward, done, moved);
}

}  // namespace mgym
