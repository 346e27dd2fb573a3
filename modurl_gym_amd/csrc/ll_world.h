// ll_world.h — per-lane "world" of one LunarLander environment: contact manager, island solver,
// revolute joints, continuous (TOI) solver and b2World::Step, restated for gfx950 (see ll_b2.h).
// Follows Box2D v2.4.1's b2ContactManager / b2Island / b2RevoluteJoint / b2World; the call site being
// replaced is `world.borrow_mut().step(1.0 / FPS, 6 * 30, 2 * 30)` (reference lunar_lander.rs:1066).
#pragma once
#include "ll_b2.h"

#ifndef LL_STAMP
#define LL_STAMP(id)  // diagnostic builds (tools/ll_phase_prof.hip) define this to stamp wave cycles per phase
#endif

#ifndef LL_DIAG_SWEEP
#define LL_DIAG_SWEEP_BEGIN(kind)
#define LL_DIAG_SWEEP(kind, it, ...)   // diagnostic builds hash the sweep state to look for short cycles
#define LL_DIAG_SWEEP_END(kind)
#endif
// What-if switches for tools/ll_whatif.hip (NOT product code: they change the physics).  A diagnostic build defines
// LL_WHATIF(w, bit) to test a run-time mask; the product build compiles every one of them out.
#ifndef LL_WHATIF
#define LL_WHATIF(w, bit) false
#endif
enum : uint32_t { WI_NO_COLLIDE_UPDATE = 1u, WI_NO_TOI_EVAL = 2u, WI_NO_SUBSTEPS = 4u, WI_ONE_POSITION_ITER = 8u, WI_ISLAND_SWEEPS_10 = 16u,
                  WI_TOI_SWEEPS_12 = 32u, WI_NO_TOI_OTHER_UPDATES = 64u, WI_NO_FIND_CONTACTS = 128u,
                  WI_SUB_NO_POSITION = 256u, WI_SUB_NO_FIND = 512u, WI_SUB_NO_SYNC = 1024u, WI_SUB_NO_REEVAL = 2048u, WI_ONE_SUBSTEP = 4096u };
#ifdef LL_WHATIF_BUILD  // iteration limits of the out-of-line loops travel as an extra argument in what-if builds only
#define LL_WI_PARAM , int wi_limit
#define LL_WI_ARG(w, bit, normal, limited) , (LL_WHATIF(w, bit) ? (limited) : (normal))
#define LL_WI_LIMIT(normal) wi_limit
#else
#define LL_WI_PARAM
#define LL_WI_ARG(w, bit, normal, limited)
#define LL_WI_LIMIT(normal) (normal)
#endif
#ifndef LL_TOI_SWEEP_STAT
#define LL_TOI_SWEEP_STAT(count, done)  // host statistics builds count the sweeps a sub-step really ran
#endif
#ifndef LL_POS_ITER_STAT
#define LL_POS_ITER_STAT(kind, count, iters, solved)  // ... and the position iterations (kind 0: island, 1: sub-step)
#endif

namespace mgym {
namespace ll {

struct Joint {  // b2RevoluteJoint, bodyA = lander (0), bodyB = leg (1 + j)
    V2 impulse; float motorImpulse, lowerImpulse, upperImpulse;
    // solver temporaries
    V2 rA, rB;
    float K11, K12, K21, K22, angle, axialMass;
    float invDetK;  // 1/det(K) (or det when it is 0) exactly as b2Mat22::Solve forms it: K is fixed for the whole velocity solve
};

// Working storage of a step that is indexed at run time or handed to an out-of-line function by reference — as local
// variables these would be scratch memory (>= 500 cycles per dependent access); every kernel that runs the contact path
// keeps one per lane in LDS (224 B), whatever it does with the World record itself.
struct WorldTmp {
    Contact tmp;                // the contact being updated (b2Contact::Update works on it through a reference)
    V2 poly_tmp[2 * kMaxPoly];  // collide_edge_polygon's transformed polygon (see there)
    uint8_t idx[3][kSlots];     // slot lists: [0] contact order, [1] evaluations / updates / island contacts, [2] sub-step island
};

struct World {
    Body b[3];
    CtStore cs;             // the env's contact-cache slots, in place in the engine's columns (ll_b2.h)
    Joint jt[2];
    float smooth[kEdges];   // terrain heights smooth_y[0..10] (lunar_lander.rs:772-774)
    AABB fat[3];            // broad-phase AABBs of the three polygon proxies
    uint32_t next_seq;
    float gA;               // b2World::SolveTOI: alpha0 of the static ground's sweep (its c0/a0 never move; only alpha0 ratchets)
    uint32_t pending;       // bits 0..2: proxy of body i waits in the move buffer (test seam only)
    bool newContacts, stepped_once;
    int n_collide;          // Collide: length of the contact list between collide_list and collide_apply
    bool resume;            // the free-flight kernel has done this step's island solve; continue at FindNewContacts (ll_env.h F_RESUME)
    bool game_over, legs[2];  // ContactDetector (lunar_lander.rs:139-205)
    uint32_t overflow;        // capacity exhausted, reported through the sticky status: bit 0 contact cache (kSlots pairs), bit 1 island solver (kSolverCap touching contacts)
    bool terrain_dirty;       // smooth[] was regenerated (reset): store it back
    LL_LDS WorldTmp* t;       // per-lane working storage in LDS (below)
#ifdef LL_WHATIF_BUILD
    uint32_t whatif;
#endif
};

constexpr float kW = 600.0f / 30.0f;  // VIEWPORT_W / SCALE

LLD void edge_verts(const World& w, int e, V2& v1, V2& v2) {
    if (e == 0) { v1 = mk(0.0f, 0.0f); v2 = mk(kW, 0.0f); return; }  // lunar_lander.rs:785
    const float cw = kW / 10.0f;                                     // :758-760
    v1 = mk(cw * (float)(e - 1), w.smooth[e - 1]);                   // :796-800
    v2 = mk(cw * (float)e, w.smooth[e]);
}
LLD AABB edge_fat(const World& w, int e) {  // b2EdgeShape::ComputeAABB + b2DynamicTree::CreateProxy
    V2 v1, v2;
    edge_verts(w, e, v1, v2);
    V2 lower = mk(fmin2(v1.x, v2.x), fmin2(v1.y, v2.y)), upper = mk(fmax2(v1.x, v2.x), fmax2(v1.y, v2.y));
    AABB a;
    a.lo = mk((lower.x - b2_polygonRadius) - b2_aabbExtension, (lower.y - b2_polygonRadius) - b2_aabbExtension);
    a.hi = mk((upper.x + b2_polygonRadius) + b2_aabbExtension, (upper.y + b2_polygonRadius) + b2_aabbExtension);
    return a;
}
LLD AABB poly_aabb(const PolyTab& tab, int pi, Xf xf) {  // b2PolygonShape::ComputeAABB
    V2 lower = xmul(xf, tab.v[pi][0]), upper = lower;
    for (int i = 1; i < tab.count[pi]; ++i) {
        V2 v = xmul(xf, tab.v[pi][i]);
        lower = mk(fmin2(lower.x, v.x), fmin2(lower.y, v.y));
        upper = mk(fmax2(upper.x, v.x), fmax2(upper.y, v.y));
    }
    AABB a;
    a.lo = mk(lower.x - b2_polygonRadius, lower.y - b2_polygonRadius);
    a.hi = mk(upper.x + b2_polygonRadius, upper.y + b2_polygonRadius);
    return a;
}
LLD bool aabb_contains(AABB a, AABB b) { return a.lo.x <= b.lo.x && a.lo.y <= b.lo.y && b.hi.x <= a.hi.x && b.hi.y <= a.hi.y; }
LLD bool aabb_overlap(AABB a, AABB b) {
    V2 d1 = b.lo - a.hi, d2 = a.lo - b.hi;
    if (d1.x > 0.0f || d1.y > 0.0f) return false;
    if (d2.x > 0.0f || d2.y > 0.0f) return false;
    return true;
}
LLD int poly_of(int body) { return body == 0 ? 0 : 1; }

// b2DynamicTree::MoveProxy; returns true when the proxy was re-inserted (=> buffered move)
LLD bool move_proxy(World& w, int body, AABB aabb, V2 displacement) {
    AABB fatAABB;
    fatAABB.lo = mk(aabb.lo.x - b2_aabbExtension, aabb.lo.y - b2_aabbExtension);
    fatAABB.hi = mk(aabb.hi.x + b2_aabbExtension, aabb.hi.y + b2_aabbExtension);
    V2 d = b2_aabbMultiplier * displacement;
    if (d.x < 0.0f) fatAABB.lo.x += d.x; else fatAABB.hi.x += d.x;
    if (d.y < 0.0f) fatAABB.lo.y += d.y; else fatAABB.hi.y += d.y;
    AABB treeAABB = w.fat[body];
    if (aabb_contains(treeAABB, aabb)) {
        AABB huge;
        huge.lo = mk(fatAABB.lo.x - 4.0f * b2_aabbExtension, fatAABB.lo.y - 4.0f * b2_aabbExtension);
        huge.hi = mk(fatAABB.hi.x + 4.0f * b2_aabbExtension, fatAABB.hi.y + 4.0f * b2_aabbExtension);
        if (aabb_contains(huge, treeAABB)) return false;
    }
    w.fat[body] = fatAABB;
    return true;
}
LLD bool fixture_sync(World& w, const PolyTab& tab, int body, Xf xf1, Xf xf2) {  // b2Fixture::Synchronize
    AABB a1 = poly_aabb(tab, poly_of(body), xf1), a2 = poly_aabb(tab, poly_of(body), xf2);
    AABB aabb;
    aabb.lo = mk(fmin2(a1.lo.x, a2.lo.x), fmin2(a1.lo.y, a2.lo.y));
    aabb.hi = mk(fmax2(a1.hi.x, a2.hi.x), fmax2(a1.hi.y, a2.hi.y));
    V2 c1 = 0.5f * (a1.lo + a1.hi), c2 = 0.5f * (a2.lo + a2.hi);
    return move_proxy(w, body, aabb, c2 - c1);
}
LLD bool body_sync_fixtures(World& w, const PolyTab& tab, int body) {  // b2Body::SynchronizeFixtures
    Body& b = w.b[body];
    if (b.awake) {
        Xf xf1;
        xf1.q = rot_set(b.sw.a0);
        xf1.p = b.sw.c0 - rmul(xf1.q, b.sw.localCenter);
        return fixture_sync(w, tab, body, xf1, b.xf);
    }
    return fixture_sync(w, tab, body, b.xf, b.xf);
}

LLD int find_slot(const World& w, int body, int edge) {
    const uint32_t want = CK_EXISTS | ((uint32_t)body << 3) | ((uint32_t)edge << 5), mask = CK_EXISTS | (3u << 3) | (15u << 5);
    for (int s = 0; s < kSlots; ++s)
        if ((ct_key(w.cs, s) & mask) == want) return s;
    return -1;
}
// b2ContactManager::AddPair (ground edge, dynamic polygon)
LLD void add_pair(World& w, int edge, int body) {
    if (find_slot(w, body, edge) >= 0) return;
    int s = -1;
    for (int k = 0; k < kSlots; ++k)
        if (!(ct_key(w.cs, k) & CK_EXISTS)) { s = k; break; }
    if (s < 0) { w.overflow |= 1u; return; }
    // a fresh contact: only the key, the sequence number and the cached time of impact are meaningful (pointCount = 0)
    ct_set_key(w.cs, s, CK_EXISTS | CK_ENABLED | ((uint32_t)body << 3) | ((uint32_t)edge << 5));
    ct_set_seq(w.cs, s, w.next_seq++);
    ct_set_toi(w.cs, s, 1.0f);
}
// b2BroadPhase::UpdatePairs for the buffered moves `order[0..n)` (dynamic proxies; partners = ground edges
// by ascending proxy id — the brute-force stand-in for the dynamic-tree query, as in the CPU oracle)
LLD void find_new_contacts(World& w, const int* order, int n) {
    for (int i = 0; i < n; ++i) {
        int body = order[i];
        // no ground proxy can overlap a box that lies entirely above the highest terrain vertex
        for (int e = 0; e < kEdges; ++e)
            if (aabb_overlap(w.fat[body], edge_fat(w, e))) add_pair(w, e, body);
    }
}

// ContactDetector::begin_contact / end_contact (lunar_lander.rs:154-205); body ids: lander 1, legs 2/3, ground 0
LLD void on_begin(World& w, int body) {
    if (body == 0) w.game_over = true;
    else w.legs[body - 1] = true;
}
LLD void on_end(World& w, int body) {
    if (body != 0) w.legs[body - 1] = false;
}

LLD void contact_update(World& w, const PolyTab& tab, Contact& c) {  // b2Contact::Update
    Manifold oldManifold = c.m;
    c.enabled = true;
    bool wasTouching = c.touching;
    V2 v1, v2;
    edge_verts(w, c.edge, v1, v2);
    collide_edge_polygon(c.m, v1, v2, tab, poly_of(c.body), w.b[c.body].xf, (V2*)w.t->poly_tmp);
    bool touching = c.m.pointCount > 0;
    for (int i = 0; i < c.m.pointCount; ++i) {
        MPoint& mp2 = c.m.points[i];
        mp2.normalImpulse = 0.0f;
        mp2.tangentImpulse = 0.0f;
        for (int j = 0; j < oldManifold.pointCount; ++j) {
            const MPoint& mp1 = oldManifold.points[j];
            if (cf_equal(mp1.id, mp2.id)) {
                mp2.normalImpulse = mp1.normalImpulse;
                mp2.tangentImpulse = mp1.tangentImpulse;
                break;
            }
        }
    }
    if (touching != wasTouching) body_set_awake(w.b[c.body], true);
    c.touching = touching;
    if (!wasTouching && touching) on_begin(w, c.body);
    if (wasTouching && !touching) on_end(w, c.body);
}

// contact list order: newest first
template <typename I>
LLD int contact_order(const World& w, I* order) {
    // rank of a contact = the number of contacts created after it; keys and sequence numbers are read once with
    // static indices (registers), so nothing here waits on a chain of dependent loads
    uint32_t key[kSlots], seq[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) { key[s] = ct_key(w.cs, s); seq[s] = ct_seq(w.cs, s); }
    int n = 0;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        if (!(key[s] & CK_EXISTS)) continue;
        int rank = 0;
#pragma unroll
        for (int t = 0; t < kSlots; ++t) rank += ((key[t] & CK_EXISTS) && seq[t] > seq[s]) ? 1 : 0;
        order[rank] = (I)s;
        ++n;
    }
    return n;
}

// b2ContactManager::Collide in three pieces, so that the kernels can run the manifold refreshes of a whole wave side by side (as SolveTOI's
// evaluations): the list walk is split into (1) collide_list — which contacts go away (fat boxes stopped overlapping) and which are
// refreshed, in list order; (2) collide_refresh — b2Contact::Update of ONE contact without its callbacks: the new manifold
// (b2CollideEdgeAndPolygon, the expensive part), the warm-start impulses matched by feature id, the key bits; a function of the contact, its
// body's transform and its edge only; (3) collide_apply — the wake-ups and ContactDetector callbacks of (1) and (2), in list order.
// Same operations as the single walk: a contact only ever wakes its OWN body and is skipped when that body sleeps, so whether a contact is
// visited does not depend on the contacts before it, and the callbacks (which set and clear the same leg flags) keep their order.
enum : uint32_t { CK_WAS_TOUCHING = 1u << 19, CK_DESTROY = 1u << 20, CK_REFRESH = 1u << 21 };   // (within Collide only)
LLD int collide_list(World& w) {   // -> number of refreshes; their slots in w.t->idx[1], the whole list in w.t->idx[0] with its length in w.n_collide
    uint8_t* const order = (uint8_t*)w.t->idx[0];
    uint8_t* const refresh = (uint8_t*)w.t->idx[1];
    const int n = contact_order(w, order);
    int n_ref = 0;
    for (int k = 0; k < n; ++k) {
        const uint32_t key = ct_key(w.cs, order[k]);
        const int body = ck_body(key), edge = ck_edge(key);
        if (!w.b[body].awake) continue;
        if (!aabb_overlap(edge_fat(w, edge), w.fat[body])) { ct_set_key(w.cs, order[k], key | CK_DESTROY); continue; }  // b2ContactManager::Destroy, in collide_apply
        if (LL_WHATIF(w, WI_NO_COLLIDE_UPDATE)) continue;
        ct_set_key(w.cs, order[k], key | CK_REFRESH);
        refresh[n_ref++] = order[k];
    }
    w.n_collide = n;
    return n_ref;
}
// (`wo` may be another lane's World record, in LDS; `poly_tmp`: 2 * kMaxPoly vectors of the EXECUTING lane)
LLD void collide_refresh(World& wo, const PolyTab& tab, int slot, V2* poly_tmp) {
    Contact c = ct_get(wo.cs, slot);
    const Manifold oldManifold = c.m;
    c.enabled = true;
    const bool wasTouching = c.touching;
    V2 v1, v2;
    edge_verts(wo, c.edge, v1, v2);
    collide_edge_polygon(c.m, v1, v2, tab, poly_of(c.body), wo.b[c.body].xf, poly_tmp);
    for (int i = 0; i < c.m.pointCount; ++i) {
        MPoint& mp2 = c.m.points[i];
        mp2.normalImpulse = 0.0f;
        mp2.tangentImpulse = 0.0f;
        for (int j = 0; j < oldManifold.pointCount; ++j) {
            const MPoint& mp1 = oldManifold.points[j];
            if (cf_equal(mp1.id, mp2.id)) {
                mp2.normalImpulse = mp1.normalImpulse;
                mp2.tangentImpulse = mp1.tangentImpulse;
                break;
            }
        }
    }
    c.touching = c.m.pointCount > 0;
    ct_put(wo.cs, slot, c);
    ct_set_key(wo.cs, slot, ct_key(wo.cs, slot) | CK_REFRESH | (wasTouching ? CK_WAS_TOUCHING : 0u));   // (ct_put encodes the persistent bits)
}
LLD void collide_apply(World& w) {
    const uint8_t* const order = (const uint8_t*)w.t->idx[0];
    for (int k = 0; k < w.n_collide; ++k) {
        const uint32_t key = ct_key(w.cs, order[k]);
        const int body = ck_body(key);
        if (key & CK_DESTROY) {
            if (key & CK_TOUCHING) on_end(w, body);
            if (((key >> 11) & 3u) > 0u) body_set_awake(w.b[body], true);
            ct_set_key(w.cs, order[k], 0u);
        } else if (key & CK_REFRESH) {
            const bool was = key & CK_WAS_TOUCHING, now = key & CK_TOUCHING;
            if (now != was) body_set_awake(w.b[body], true);
            if (!was && now) on_begin(w, body);
            if (was && !now) on_end(w, body);
            ct_set_key(w.cs, order[k], key & ~(CK_REFRESH | CK_WAS_TOUCHING));
        }
    }
}
LLD void collide(World& w, const PolyTab& tab) {  // one lane runs every piece itself (host check, the kernels that do not deal the refreshes out)
    const int n_ref = collide_list(w);
    for (int j = 0; j < n_ref; ++j) collide_refresh(w, tab, ((const uint8_t*)w.t->idx[1])[j], (V2*)w.t->poly_tmp);
    collide_apply(w);
}

// ---- contact solver over a list of slots --------------------------------------------------------------
// Constraint storage is supplied by the caller.  On the GPU the first `vc_near` velocity constraints of a lane live in
// LDS, one column per lane (element i of lane l at vc[i * stride], stride = lanes per block), the rest — rarely more
// than a few exist — in `vc_far` (a global workspace, one column per lane of the grid): LDS is worth more to the
// run-time-indexed per-lane records (lunar_lander.hip) than to constraints the sweeps hold in registers anyway
// (any split works: every access goes through cs_vc).  The position constraints are local memory.
// The host test passes plain arrays with stride 1.
struct CSolverMem { VConstraint* vc; int vc_stride; int vc_near; VConstraint* vc_far; int vc_far_stride; PConstraint* pc; int pc_stride; int cap; };
struct CSolver {
    VConstraint* vc; int vs;
    int nl; VConstraint* vo; int vos;  // constraint i >= nl is vo[(i - nl) * vos]
    PConstraint* pc; int ps;
    int count;
};
LLD VConstraint& cs_vc(const CSolver& s, int i) { return i < s.nl ? s.vc[i * s.vs] : s.vo[(i - s.nl) * s.vos]; }
// body-indexed access to three register-resident velocity / position records
struct Vel3 { Vel b0, b1, b2; };
LLD Vel vel_get(const Vel3& v, int i) {
    Vel r = v.b0;
    switch (i) { case 1: r = v.b1; break; case 2: r = v.b2; break; default: break; }
    return r;
}
LLD void vel_set(Vel3& v, int i, Vel x) {
    switch (i) { case 0: v.b0 = x; break; case 1: v.b1 = x; break; default: v.b2 = x; break; }
}
// branch-free forms for the sweep loop: lanes of one wave hold constraints on different bodies, and a switch
// would run the whole constraint solve once per distinct body (single-level selects only, one word at a time)
LLD float sel3f(int i, float a0, float a1, float a2) {
    float r = a2;
    r = (i == 1) ? a1 : r;
    r = (i == 0) ? a0 : r;
    return r;
}
LLD Vel vel_pick(const Vel3& v, int i) {
    Vel r;
    r.v.x = sel3f(i, v.b0.v.x, v.b1.v.x, v.b2.v.x);
    r.v.y = sel3f(i, v.b0.v.y, v.b1.v.y, v.b2.v.y);
    r.w = sel3f(i, v.b0.w, v.b1.w, v.b2.w);
    return r;
}
LLD void vel_put1(Vel& dst, bool take, const Vel& x) {
    dst.v.x = take ? x.v.x : dst.v.x;
    dst.v.y = take ? x.v.y : dst.v.y;
    dst.w = take ? x.w : dst.w;
}
LLD void vel_put(Vel3& v, int i, const Vel& x) {
    vel_put1(v.b0, i == 0, x);
    vel_put1(v.b1, i == 1, x);
    vel_put1(v.b2, i != 0 && i != 1, x);
}

struct Pos3 { Pos b0, b1, b2; };
LLD Pos pos_pick(const Pos3& p, int i) {
    Pos r;
    r.c.x = sel3f(i, p.b0.c.x, p.b1.c.x, p.b2.c.x);
    r.c.y = sel3f(i, p.b0.c.y, p.b1.c.y, p.b2.c.y);
    r.a = sel3f(i, p.b0.a, p.b1.a, p.b2.a);
    return r;
}
LLD void pos_put1(Pos& dst, bool take, const Pos& x) {
    dst.c.x = take ? x.c.x : dst.c.x;
    dst.c.y = take ? x.c.y : dst.c.y;
    dst.a = take ? x.a : dst.a;
}
LLD void pos_put(Pos3& p, int i, const Pos& x) {
    pos_put1(p.b0, i == 0, x);
    pos_put1(p.b1, i == 1, x);
    pos_put1(p.b2, i != 0 && i != 1, x);
}
LLD bool pos_same(const Pos& a, const Pos& b) {
    return as_u32(a.c.x) == as_u32(b.c.x) && as_u32(a.c.y) == as_u32(b.c.y) && as_u32(a.a) == as_u32(b.a);
}

template <typename I>
LLD void cs_init(CSolver& s, const CSolverMem& mem, World& w, const LLConst& k, const I* slots, int count, bool warmStarting, float dtRatio) {
    s.vc = mem.vc; s.vs = mem.vc_stride; s.nl = mem.vc_near; s.vo = mem.vc_far; s.vos = mem.vc_far_stride; s.pc = mem.pc; s.ps = mem.pc_stride;
    if (count > mem.cap) { w.overflow |= 2u; count = mem.cap; }
    s.count = count;
    for (int i = 0; i < count; ++i) {
        const Contact contact = ct_get(w.cs, slots[i]);
        const Manifold& manifold = contact.m;
        const int t = poly_of(contact.body);
        VConstraint& vc = cs_vc(s, i);
        vc.friction = k.friction[t];
        vc.indexB = contact.body;
        vc.invMassB = k.invMass[t]; vc.invIB = k.invI[t];
        vc.slot = slots[i]; vc.pointCount = manifold.pointCount;
        vc.k11 = vc.k12 = vc.k21 = vc.k22 = 0.0f; vc.nm11 = vc.nm12 = vc.nm21 = vc.nm22 = 0.0f;
        PConstraint& pc = s.pc[i * s.ps];
        pc.indexB = contact.body; pc.invMassB = k.invMass[t]; pc.invIB = k.invI[t];
        pc.localCenterB = k.localCenter[t];
        pc.localNormal = manifold.localNormal; pc.localPoint = manifold.localPoint;
        pc.pointCount = manifold.pointCount; pc.type = manifold.type;
        for (int j = 0; j < manifold.pointCount; ++j) {
            const MPoint& cp = manifold.points[j];
            VCPoint& vcp = vc.points[j];
            if (warmStarting) {
                vcp.normalImpulse = dtRatio * cp.normalImpulse;
                vcp.tangentImpulse = dtRatio * cp.tangentImpulse;
            } else {
                vcp.normalImpulse = 0.0f; vcp.tangentImpulse = 0.0f;
            }
            vcp.rB = mk(0.0f, 0.0f);
            vcp.normalMass = 0.0f; vcp.tangentMass = 0.0f; vcp.velocityBias = 0.0f;
            pc.localPoints[j] = cp.localPoint;
        }
    }
}

// b2ContactSolver::InitializeVelocityConstraints.  restitution = 0: velocityBias = -0 * vRel (a signed
// zero when vRel < -threshold), which only ever enters `vn - velocityBias` and is therefore left at 0.
LLD void cs_init_velocity(CSolver& s, const World& w, const Pos* pos, const Vel3& vel) {
    for (int i = 0; i < s.count; ++i) {
        VConstraint& vc = cs_vc(s, i);
        const PConstraint& pc = s.pc[i * s.ps];
        const Manifold manifold = ct_get(w.cs, vc.slot).m;
        const float mB = vc.invMassB, iB = vc.invIB;
        V2 cB = pos[vc.indexB].c; float aB = pos[vc.indexB].a;
        const Vel velB = vel_get(vel, vc.indexB);
        V2 vB = velB.v; float wB = velB.w;
        Xf xfB;
        xfB.q = rot_set(aB);
        xfB.p = cB - rmul(xfB.q, pc.localCenterB);
        V2 wm_normal = mk(0.0f, 0.0f), wm_points[2];
        world_manifold(manifold, xfB, wm_normal, wm_points);
        vc.normal = wm_normal;
        for (int j = 0; j < vc.pointCount; ++j) {
            VCPoint& vcp = vc.points[j];
            vcp.rB = wm_points[j] - cB;
            float rnB = cross(vcp.rB, vc.normal);
            float kNormal = mB + iB * rnB * rnB;
            vcp.normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
            V2 tangent = cross_vs(vc.normal, 1.0f);
            float rtB = cross(vcp.rB, tangent);
            float kTangent = mB + iB * rtB * rtB;
            vcp.tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
            float vRel = dot(vc.normal, vB + cross_sv(wB, vcp.rB));
            vcp.velocityBias = 0.0f;
            if (vRel < -1.0f) vcp.velocityBias = -0.0f * vRel;
        }
        if (vc.pointCount == 2) {
            float rn1B = cross(vc.points[0].rB, vc.normal);
            float rn2B = cross(vc.points[1].rB, vc.normal);
            float k11 = mB + iB * rn1B * rn1B;
            float k22 = mB + iB * rn2B * rn2B;
            float k12 = mB + iB * rn1B * rn2B;
            const float k_maxConditionNumber = 1000.0f;
            if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {
                vc.k11 = k11; vc.k21 = k12; vc.k12 = k12; vc.k22 = k22;
                float a = k11, b = k12, c = k12, d = k22;
                float det = a * d - b * c;
                if (det != 0.0f) det = 1.0f / det;
                vc.nm11 = det * d; vc.nm12 = -det * b; vc.nm21 = -det * c; vc.nm22 = det * a;
            } else {
                vc.pointCount = 1;
            }
        }
    }
}

LLD void cs_warm_start(const CSolver& s, Vel3& vel) {  // b2ContactSolver::WarmStart
    for (int i = 0; i < s.count; ++i) {
        const VConstraint& vc = cs_vc(s, i);
        const float mB = vc.invMassB, iB = vc.invIB;
        const Vel velB = vel_get(vel, vc.indexB);
        V2 vB = velB.v; float wB = velB.w;
        V2 normal = vc.normal, tangent = cross_vs(normal, 1.0f);
        for (int j = 0; j < vc.pointCount; ++j) {
            const VCPoint& vcp = vc.points[j];
            V2 P = vcp.normalImpulse * normal + vcp.tangentImpulse * tangent;
            wB += iB * cross(vcp.rB, P);
            vB = vB + mB * P;
        }
        Vel outB; outB.v = vB; outB.w = wB;
        vel_set(vel, vc.indexB, outB);
    }
}

LLD void cs_apply2(const VConstraint& vc, V2 x, V2 a, V2 normal, V2& vB, float& wB, VCPoint& cp1, VCPoint& cp2) {
    V2 d = x - a;
    V2 P1 = d.x * normal, P2 = d.y * normal;
    vB = vB + vc.invMassB * (P1 + P2);
    wB += vc.invIB * (cross(cp1.rB, P1) + cross(cp2.rB, P2));
    cp1.normalImpulse = x.x; cp2.normalImpulse = x.y;
}

// b2ContactSolver::SolveVelocityConstraints for ONE constraint acting on the dynamic body whose velocity is velB
LLD void cs_friction_point(VCPoint& vcp, V2 tangent, float friction, float mB, float iB, V2& vB, float& wB) {
    V2 dv = vB + cross_sv(wB, vcp.rB);
    float vt = dot(dv, tangent) - 0.0f;
    float lambda = vcp.tangentMass * (-vt);
    float maxFriction = friction * vcp.normalImpulse;
    float newImpulse = fclamp(vcp.tangentImpulse + lambda, -maxFriction, maxFriction);
    lambda = newImpulse - vcp.tangentImpulse;
    vcp.tangentImpulse = newImpulse;
    V2 P = lambda * tangent;
    vB = vB + mB * P;
    wB += iB * cross(vcp.rB, P);
}
LLD void cs_solve_one(VConstraint& vc, Vel& velB) {
    const float mB = vc.invMassB, iB = vc.invIB;
    const int pointCount = vc.pointCount;
    V2 vB = velB.v; float wB = velB.w;
    V2 normal = vc.normal, tangent = cross_vs(normal, 1.0f);
    const float friction = vc.friction;
    // points addressed by constant index (a loop over `pointCount` would index the record dynamically and pin it
    // to scratch memory for the whole sweep loop)
    if (pointCount > 0) cs_friction_point(vc.points[0], tangent, friction, mB, iB, vB, wB);
    if (pointCount > 1) cs_friction_point(vc.points[1], tangent, friction, mB, iB, vB, wB);
    if (pointCount == 1) {
        VCPoint& vcp = vc.points[0];
        V2 dv = vB + cross_sv(wB, vcp.rB);
        float vn = dot(dv, normal);
        float lambda = -vcp.normalMass * (vn - vcp.velocityBias);
        float newImpulse = fmax2(vcp.normalImpulse + lambda, 0.0f);
        lambda = newImpulse - vcp.normalImpulse;
        vcp.normalImpulse = newImpulse;
        V2 P = lambda * normal;
        vB = vB + mB * P;
        wB += iB * cross(vcp.rB, P);
    } else {
        VCPoint& cp1 = vc.points[0];
        VCPoint& cp2 = vc.points[1];
        V2 a = mk(cp1.normalImpulse, cp2.normalImpulse);
        V2 dv1 = vB + cross_sv(wB, cp1.rB);
        V2 dv2 = vB + cross_sv(wB, cp2.rB);
        float vn1 = dot(dv1, normal), vn2 = dot(dv2, normal);
        V2 b = mk(vn1 - cp1.velocityBias, vn2 - cp2.velocityBias);
        b = b - mk(vc.k11 * a.x + vc.k12 * a.y, vc.k21 * a.x + vc.k22 * a.y);
        // The four cases of the block solver, evaluated side by side and selected (the reference tries them in this
        // order and takes the first that holds): lanes of a wave land in different cases, and the branchy form
        // would run the impulse application once per distinct case.
        const V2 x1 = -mk(vc.nm11 * b.x + vc.nm12 * b.y, vc.nm21 * b.x + vc.nm22 * b.y);
        const bool ok1 = x1.x >= 0.0f && x1.y >= 0.0f;
        const V2 x2 = mk(-cp1.normalMass * b.x, 0.0f);
        const float vn2_2 = vc.k21 * x2.x + b.y;
        const bool ok2 = x2.x >= 0.0f && vn2_2 >= 0.0f;
        const V2 x3 = mk(0.0f, -cp2.normalMass * b.y);
        const float vn1_3 = vc.k12 * x3.y + b.x;
        const bool ok3 = x3.y >= 0.0f && vn1_3 >= 0.0f;
        const bool ok4 = b.x >= 0.0f && b.y >= 0.0f;
        V2 x = mk(0.0f, 0.0f);
        x = ok3 ? x3 : x;
        x = ok2 ? x2 : x;
        x = ok1 ? x1 : x;
        if (ok1 || ok2 || ok3 || ok4) cs_apply2(vc, x, a, normal, vB, wB, cp1, cp2);
    }
    velB.v = vB; velB.w = wB;
}
// ... for constraints [i0, i1), all of which act on the SAME dynamic body (island contacts are grouped by body
// in DFS order), whose velocity lives in the caller's registers
LLD void cs_solve_velocity_range(CSolver& s, int i0, int i1, Vel& velB) {
    for (int i = i0; i < i1; ++i) cs_solve_one(cs_vc(s, i), velB);
}
LLD void cs_solve_one_on(VConstraint& vc, int body, Vel3& vel) {
    Vel vb = vel_pick(vel, body);
    cs_solve_one(vc, vb);
    vel_put(vel, body, vb);
}

LLD void cs_store_impulses(const CSolver& s, World& w) {  // b2ContactSolver::StoreImpulses
    for (int i = 0; i < s.count; ++i) {
        const VConstraint& vc = cs_vc(s, i);
        for (int j = 0; j < vc.pointCount; ++j) {
            ct_word(w.cs, vc.slot, 8 + 4 * j) = as_u32(vc.points[j].normalImpulse);
            ct_word(w.cs, vc.slot, 9 + 4 * j) = as_u32(vc.points[j].tangentImpulse);
        }
    }
}

// b2ContactSolver::SolvePositionConstraints (toi = false) / SolveTOIPositionConstraints (toi = true; the TOI
// body is always body B of every constraint in the TOI island, so its masses are kept)
LLD void pc_solve_point(const PConstraint& pc, V2 localPoint, bool toi, V2& cB, float& aB, float& minSeparation) {
    const float mB = pc.invMassB, iB = pc.invIB;
    Xf xfB;
    xfB.q = rot_set(aB);
    xfB.p = cB - rmul(xfB.q, pc.localCenterB);
    // b2PositionSolverManifold::Initialize with xfA = identity
    V2 normal, point; float separation;
    if (pc.type == FACE_A) {
        normal = pc.localNormal;
        V2 planePoint = pc.localPoint;
        V2 clipPoint = xmul(xfB, localPoint);
        separation = dot(clipPoint - planePoint, normal) - b2_polygonRadius - b2_polygonRadius;
        point = clipPoint;
    } else {
        normal = rmul(xfB.q, pc.localNormal);
        V2 planePoint = xmul(xfB, pc.localPoint);
        V2 clipPoint = localPoint;
        separation = dot(clipPoint - planePoint, normal) - b2_polygonRadius - b2_polygonRadius;
        point = clipPoint;
        normal = -normal;
    }
    V2 rB = point - cB;
    minSeparation = fmin2(minSeparation, separation);
    float C = fclamp((toi ? b2_toiBaumgarte : b2_baumgarte) * (separation + b2_linearSlop), -b2_maxLinearCorrection, 0.0f);
    float rnB = cross(rB, normal);
    float K = mB + iB * rnB * rnB;
    float impulse = K > 0.0f ? -C / K : 0.0f;
    V2 P = impulse * normal;
    cB = cB + mB * P;
    aB += iB * cross(rB, P);
}
LLD void pc_solve_one(const PConstraint& pc, bool toi, V2& cB, float& aB, float& minSeparation) {
    if (pc.pointCount > 0) pc_solve_point(pc, pc.localPoints[0], toi, cB, aB, minSeparation);  // constant indices: see cs_solve_one
    if (pc.pointCount > 1) pc_solve_point(pc, pc.localPoints[1], toi, cB, aB, minSeparation);
}
LLD bool cs_solve_position(const CSolver& s, Pos* pos, bool toi) {
    float minSeparation = 0.0f;
    for (int i = 0; i < s.count; ++i) {
        const PConstraint& pc = s.pc[i * s.ps];
        V2 cB = pos[pc.indexB].c; float aB = pos[pc.indexB].a;
        pc_solve_one(pc, toi, cB, aB, minSeparation);
        pos[pc.indexB].c = cB; pos[pc.indexB].a = aB;
    }
    return toi ? (minSeparation >= -1.5f * b2_linearSlop) : (minSeparation >= -3.0f * b2_linearSlop);
}

// ---- revolute joint ---------------------------------------------------------------------------------------
LLD void rj_init_velocity(Joint& j, int leg, const LLConst& k, float aA, float aB, Vel& velA, Vel& velB, float dtRatio) {
    const int ji = leg - 1;
    V2 vA = velA.v; float wA = velA.w;
    V2 vB = velB.v; float wB = velB.w;
    Rot qA = rot_set(aA), qB = rot_set(aB);
    j.rA = rmul(qA, mk(0.0f, 0.0f) - k.localCenter[0]);
    j.rB = rmul(qB, k.localAnchorB[ji] - k.localCenter[1]);
    const float mA = k.invMass[0], mB = k.invMass[1], iA = k.invI[0], iB = k.invI[1];
    j.K11 = mA + mB + j.rA.y * j.rA.y * iA + j.rB.y * j.rB.y * iB;
    j.K12 = -j.rA.y * j.rA.x * iA - j.rB.y * j.rB.x * iB;
    j.K21 = j.K12;
    j.K22 = mA + mB + j.rA.x * j.rA.x * iA + j.rB.x * j.rB.x * iB;
    {
        float det = j.K11 * j.K22 - j.K12 * j.K21;
        if (det != 0.0f) det = 1.0f / det;
        j.invDetK = det;
    }
    j.axialMass = iA + iB;
    j.axialMass = 1.0f / j.axialMass;  // iA + iB > 0 always
    j.angle = aB - aA - 0.0f;          // referenceAngle = 0
    j.impulse = dtRatio * j.impulse;
    j.motorImpulse *= dtRatio; j.lowerImpulse *= dtRatio; j.upperImpulse *= dtRatio;
    float axialImpulse = j.motorImpulse + j.lowerImpulse - j.upperImpulse;
    V2 P = j.impulse;
    vA = vA - mA * P;
    wA -= iA * (cross(j.rA, P) + axialImpulse);
    vB = vB + mB * P;
    wB += iB * (cross(j.rB, P) + axialImpulse);
    velA.v = vA; velA.w = wA; velB.v = vB; velB.w = wB;
}

LLD void rj_solve_velocity(Joint& j, int ji, const LLConst& k, V2& vA, float& wA, V2& vB, float& wB, float dt, float inv_dt) {
    const float mA = k.invMass[0], mB = k.invMass[1], iA = k.invI[0], iB = k.invI[1];
    {
        float Cdot = wB - wA - k.motorSpeed[ji];
        float impulse = -j.axialMass * Cdot;
        float oldImpulse = j.motorImpulse;
        float maxImpulse = dt * k.maxMotorTorque;
        j.motorImpulse = fclamp(oldImpulse + impulse, -maxImpulse, maxImpulse);
        impulse = j.motorImpulse - oldImpulse;
        wA -= iA * impulse;
        wB += iB * impulse;
    }
    {
        float C = j.angle - k.lowerAngle[ji];
        float Cdot = wB - wA;
        float impulse = -j.axialMass * (Cdot + fmax2(C, 0.0f) * inv_dt);
        float newImpulse = fmax2(j.lowerImpulse + impulse, 0.0f);
        impulse = newImpulse - j.lowerImpulse;
        j.lowerImpulse = newImpulse;
        wA -= iA * impulse;
        wB += iB * impulse;
    }
    {
        float C = k.upperAngle[ji] - j.angle;
        float Cdot = wA - wB;
        float impulse = -j.axialMass * (Cdot + fmax2(C, 0.0f) * inv_dt);
        float newImpulse = fmax2(j.upperImpulse + impulse, 0.0f);
        impulse = newImpulse - j.upperImpulse;
        j.upperImpulse = newImpulse;
        wA += iA * impulse;
        wB -= iB * impulse;
    }
    {
        V2 Cdot = vB + cross_sv(wB, j.rB) - vA - cross_sv(wA, j.rA);
        V2 nC = -Cdot;  // b2Mat22::Solve with the loop-invariant reciprocal determinant
        V2 impulse = mk(j.invDetK * (j.K22 * nC.x - j.K12 * nC.y), j.invDetK * (j.K11 * nC.y - j.K21 * nC.x));
        j.impulse.x += impulse.x;
        j.impulse.y += impulse.y;
        vA = vA - mA * impulse;
        wA -= iA * cross(j.rA, impulse);
        vB = vB + mB * impulse;
        wB += iB * cross(j.rB, impulse);
    }
}

LLD bool rj_solve_position(const Joint& j, int ji, const LLConst& k, Pos& pA, Pos& pB) {
    V2 cA = pA.c; float aA = pA.a; V2 cB = pB.c; float aB = pB.a;
    float angularError = 0.0f, positionError = 0.0f;
    const float mA = k.invMass[0], mB = k.invMass[1], iA = k.invI[0], iB = k.invI[1];
    {
        float angle = aB - aA - 0.0f;
        float C = 0.0f;
        if (fabs1(k.upperAngle[ji] - k.lowerAngle[ji]) < 2.0f * b2_angularSlop) {
            C = fclamp(angle - k.lowerAngle[ji], -b2_maxAngularCorrection, b2_maxAngularCorrection);
        } else if (angle <= k.lowerAngle[ji]) {
            C = fclamp(angle - k.lowerAngle[ji] + b2_angularSlop, -b2_maxAngularCorrection, 0.0f);
        } else if (angle >= k.upperAngle[ji]) {
            C = fclamp(angle - k.upperAngle[ji] - b2_angularSlop, 0.0f, b2_maxAngularCorrection);
        }
        float limitImpulse = -j.axialMass * C;
        aA -= iA * limitImpulse;
        aB += iB * limitImpulse;
        angularError = fabs1(C);
    }
    {
        Rot qA = rot_set(aA), qB = rot_set(aB);
        V2 rA = rmul(qA, mk(0.0f, 0.0f) - k.localCenter[0]);
        V2 rB = rmul(qB, k.localAnchorB[ji] - k.localCenter[1]);
        V2 C = cB + rB - cA - rA;
        positionError = len(C);
        float Kexx = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
        float Kexy = -iA * rA.x * rA.y - iB * rB.x * rB.y;
        float Keyx = Kexy;
        float Keyy = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
        V2 impulse = -mat22_solve(Kexx, Keyx, Kexy, Keyy, C);
        cA = cA - mA * impulse;
        aA -= iA * cross(rA, impulse);
        cB = cB + mB * impulse;
        aB += iB * cross(rB, impulse);
    }
    pA.c = cA; pA.a = aA; pB.c = cB; pB.a = aB;
    return positionError <= b2_linearSlop && angularError <= b2_angularSlop;
}

// b2Island::Solve pieces shared by the general island solver and the contact-free fast path (ll_free.h)
LLD void integrate_velocity(const Body& b, int t, const LLConst& k, float h, V2& v_out, float& w_out) {
    V2 v = b.v; float wv = b.w;
    v = v + (h * k.invMass[t]) * ((1.0f * k.mass[t]) * mk(0.0f, k.gravity) + b.force);
    wv += h * k.invI[t] * b.torque;
    v = (1.0f / (1.0f + h * 0.0f)) * v;
    wv *= 1.0f / (1.0f + h * 0.0f);
    v_out = v; w_out = wv;
}
LLD void integrate_position(V2& c, float& a, V2& v, float& wv, float h) {
    V2 translation = h * v;
    if (dot(translation, translation) > b2_maxTranslationSquared) {
        float ratio = b2_maxTranslation / len(translation);
        v = ratio * v;
    }
    float rotation = h * wv;
    if (rotation * rotation > b2_maxRotationSquared) {
        float ratio = b2_maxRotation / fabs1(rotation);
        wv *= ratio;
    }
    c = c + h * v;
    a += h * wv;
}
// sleep bookkeeping of one island body; returns the updated minSleepTime
LLD float sleep_update(Body& b, float h, float minSleepTime) {
    const float linTolSqr = b2_linearSleepTolerance * b2_linearSleepTolerance;
    const float angTolSqr = b2_angularSleepTolerance * b2_angularSleepTolerance;
    if (b.w * b.w > angTolSqr || dot(b.v, b.v) > linTolSqr) {
        b.sleepTime = 0.0f;
        return 0.0f;
    }
    b.sleepTime += h;
    return fmin2(minSleepTime, b.sleepTime);
}

// The 180 velocity sweeps of b2Island::Solve as an out-of-line function: the enclosing kernel is large and
// register-starved, so the hot loop gets its own register allocation (state is copied in, iterated in VGPRs,
// copied out).  The first two contact constraints ride in registers (their LDS copies could not be kept in
// registers by the compiler: every impulse store may alias them); the rest stay in LDS.
LL_NOINLINE void island_sweeps(Joint& J0_io, Joint& J1_io, Vel3& vel_io, CSolver& cs, const int* cstart, const int* ibody, int nb,
                               bool leg1_first, const LLConst& k_in, float dt, float inv_dt LL_WI_PARAM) {
    Joint J0 = J0_io, J1 = J1_io;
    Vel3 vel = vel_io;
    // `k_in` points at the kernel's constant block; read through a pointer inside this loop it would be re-loaded
    // every sweep (the impulse stores to the LDS constraints may alias it): keep the joint constants in registers
    LLConst k;
    k.invMass[0] = k_in.invMass[0]; k.invMass[1] = k_in.invMass[1]; k.invI[0] = k_in.invI[0]; k.invI[1] = k_in.invI[1];
    k.lowerAngle[0] = k_in.lowerAngle[0]; k.lowerAngle[1] = k_in.lowerAngle[1];
    k.upperAngle[0] = k_in.upperAngle[0]; k.upperAngle[1] = k_in.upperAngle[1];
    k.motorSpeed[0] = k_in.motorSpeed[0]; k.motorSpeed[1] = k_in.motorSpeed[1];
    k.maxMotorTorque = k_in.maxMotorTorque;
    const int n_cs = cs.count;
#ifndef LL_ISLAND_REGS
#define LL_ISLAND_REGS 4
#endif
    constexpr int kReg = LL_ISLAND_REGS;  // constraints held in registers (2..4; the kernel runs at one wave per SIMD: 512 VGPRs per lane)
    VConstraint r0, r1, r2, r3;
    int rb0 = -1, rb1 = -1, rb2 = -1, rb3 = -1;
    if (cs.count > 0) { r0 = cs_vc(cs, 0); rb0 = r0.indexB; }
    if (cs.count > 1) { r1 = cs_vc(cs, 1); rb1 = r1.indexB; }
    if (kReg > 2 && cs.count > 2) { r2 = cs_vc(cs, 2); rb2 = r2.indexB; }   // (the third and fourth may live in the far workspace)
    if (kReg > 3 && cs.count > 3) { r3 = cs_vc(cs, 3); rb3 = r3.indexB; }
    LL_DIAG_SWEEP_BEGIN(0);
    for (int it = 0; it < LL_WI_LIMIT(180); ++it) {
        if (leg1_first) {
            rj_solve_velocity(J1, 1, k, vel.b0.v, vel.b0.w, vel.b2.v, vel.b2.w, dt, inv_dt);
            rj_solve_velocity(J0, 0, k, vel.b0.v, vel.b0.w, vel.b1.v, vel.b1.w, dt, inv_dt);
        } else {
            rj_solve_velocity(J0, 0, k, vel.b0.v, vel.b0.w, vel.b1.v, vel.b1.w, dt, inv_dt);
            rj_solve_velocity(J1, 1, k, vel.b0.v, vel.b0.w, vel.b2.v, vel.b2.w, dt, inv_dt);
        }
        if (rb0 >= 0) cs_solve_one_on(r0, rb0, vel);
        if (rb1 >= 0) cs_solve_one_on(r1, rb1, vel);
        if (kReg > 2 && rb2 >= 0) cs_solve_one_on(r2, rb2, vel);
        if (kReg > 3 && rb3 >= 0) cs_solve_one_on(r3, rb3, vel);
        if (n_cs > kReg) {
            for (int q = 0; q < nb; ++q) {  // remaining contacts, grouped by body in DFS order
                const int c0 = cstart[q] > kReg ? cstart[q] : kReg, c1 = cstart[q + 1] < n_cs ? cstart[q + 1] : n_cs;
                if (c0 >= c1) continue;
                switch (ibody[q]) {
                case 0: cs_solve_velocity_range(cs, c0, c1, vel.b0); break;
                case 1: cs_solve_velocity_range(cs, c0, c1, vel.b1); break;
                default: cs_solve_velocity_range(cs, c0, c1, vel.b2); break;
                }
            }
        }
        LL_DIAG_SWEEP(0, it, n_cs <= 2 && rb2 < 0, vel.b0.v.x, vel.b0.v.y, vel.b0.w, vel.b1.v.x, vel.b1.v.y, vel.b1.w, vel.b2.v.x, vel.b2.v.y, vel.b2.w,
                      J0.impulse.x, J0.impulse.y, J0.motorImpulse, J0.lowerImpulse, J0.upperImpulse,
                      J1.impulse.x, J1.impulse.y, J1.motorImpulse, J1.lowerImpulse, J1.upperImpulse,
                      rb0 >= 0 ? r0.points[0].normalImpulse : 0.0f, rb0 >= 0 ? r0.points[0].tangentImpulse : 0.0f,
                      rb0 >= 0 && r0.pointCount > 1 ? r0.points[1].normalImpulse : 0.0f, rb0 >= 0 && r0.pointCount > 1 ? r0.points[1].tangentImpulse : 0.0f,
                      rb1 >= 0 ? r1.points[0].normalImpulse : 0.0f, rb1 >= 0 ? r1.points[0].tangentImpulse : 0.0f,
                      rb1 >= 0 && r1.pointCount > 1 ? r1.points[1].normalImpulse : 0.0f, rb1 >= 0 && r1.pointCount > 1 ? r1.points[1].tangentImpulse : 0.0f);
    }
    LL_DIAG_SWEEP_END(0);
    if (rb0 >= 0) cs_vc(cs, 0) = r0;
    if (rb1 >= 0) cs_vc(cs, 1) = r1;
    if (kReg > 2 && rb2 >= 0) cs_vc(cs, 2) = r2;
    if (kReg > 3 && rb3 >= 0) cs_vc(cs, 3) = r3;
    J0_io = J0; J1_io = J1; vel_io = vel;
}

// the 180 contact-only sweeps of b2Island::SolveTOI on the one dynamic body (impulses are not stored back).
// A sweep is a pure function F of the state S = (body velocity, accumulated impulses) — everything else in the
// constraints is constant — so the sequence S_0, S_1 = F(S_0), ... is eventually periodic, and in f32 it usually
// becomes so within a few dozen sweeps: one body against static ground settles on a fixed point (period 1) or on a
// short cycle of last-bit flips.  Once S_j == S_i bit for bit (i < j, p = j - i), S_180 = S_{j + (180 - j) mod p}:
// only (180 - j) mod p more sweeps are run, with the identical result.  S_j is compared with the previous state
// (fixed point) and with a snapshot renewed at doubling distances (Brent's cycle search: any period
// is found once the snapshot lies on the cycle and the window exceeds the period).  The joints of the full island
// never settle this way, which is why island_sweeps has no such exit.
// What is left (tools/ll_work_stats.sh: 4 % of one-constraint and 21 % of two-constraint sub-steps run all 180
// sweeps): slow geometric convergence of two coupled contacts, and states whose velocity has stopped moving while a
// tangent impulse still creeps by a few ulps per sweep.  Running only the friction rows in that second case (exact:
// the skipped rows see unchanged inputs) was built and measured — the bookkeeping cost every sweep more than the
// few lanes it helped, and a wave still waits for its slowest lane (1.79 -> 1.91 ms per step): not kept.
struct ToiSweepState { uint32_t w[11]; };
LLD bool toi_state_same(const ToiSweepState& a, const ToiSweepState& b) {
    bool same = true;
#pragma unroll
    for (int q = 0; q < 11; ++q) same = same && a.w[q] == b.w[q];
    return same;
}
// A third way out, for the states above whose VELOCITY has stopped while an impulse still creeps: the sweeps' only output is the body
// velocity (b2Island::SolveTOI does not store the impulses), and for one-point constraints it can be PROVEN constant from here on.
// With the velocity V fixed, every sweep computes the same increments lam_t, lam_n from V; the accumulated impulses then move by at most
//   |dn| <= lam_n + ulp(n)   (n' = max(n + lam_n, 0): a clamp at 0 moves it by less than lam_n; 0 when lam_n rounds away)
//   |dt| <= f |dn| + ulp while it slides along the friction bound (t' = clamp(t + lam_t, +-f n)), lam_t + ulp otherwise
// per sweep, and the velocity update of an impulse d along a unit direction u is v += mB (d u), w += iB cross(rB, d u).  If every one
// of those six terms, with d at its bound over all remaining sweeps (impulses at most 180 increments larger than now), stays below
// half the spacing of the floats just below the velocity component it is added to, each addition returns that component unchanged (round to nearest),
// so by induction over the updates of a sweep and over the sweeps V never changes again: stop.  Components that are exactly zero are
// excluded (an added zero can change the sign of a zero), as are two-point constraints (their block solver's increments depend on the
// accumulated impulses through a cancellation).  tests/native/ll_host_check.cpp re-runs every shortened solve in full and compares.
// A term of magnitude BELOW this, added to x, returns x (round to nearest): half the distance from x to its nearer neighbour — for |x| in
// (2^k, 2^(k+1)) that is 2^(k-24), for |x| = 2^k exactly 2^(k-25) (the floats below are spaced half as far).  0 when x is zero, denormal,
// tiny, infinite or NaN.
LLD float toi_half_spacing(float x) {
    const uint32_t u = as_u32(x), e = (u >> 23) & 0xffu;
    const uint32_t down = (u & 0x7fffffu) ? 24u : 25u;
    return (e > 26u && e < 255u) ? as_f32((e - down) << 23) : 0.0f;
}
LLD float toi_half_ulp(float x) {   // an upper bound of the rounding error of a sum that comes out near x: half the spacing of the floats at |x|
    const uint32_t e = (as_u32(x) >> 23) & 0xffu;
    return e > 24u ? as_f32((e - 24u) << 23) : 1.0e-37f;
}
LLD bool toi_velocity_settled_one(const VConstraint& vc, const Vel& V) {
    const VCPoint& p = vc.points[0];
    const V2 nrm = vc.normal, tan = cross_vs(nrm, 1.0f);
    const V2 dv = V.v + cross_sv(V.w, p.rB);
    const float lam_t_s = p.tangentMass * (-(dot(dv, tan) - 0.0f)), lam_t = fabs1(lam_t_s);           // as cs_friction_point forms them
    const float lam_n_s = -p.normalMass * (dot(dv, nrm) - p.velocityBias), lam_n = fabs1(lam_n_s);  // as cs_solve_one forms them
    const float kSlack = 1.0001f;                                         // the rounding of the bounds themselves
    const float f = vc.friction, n = p.normalImpulse, t = p.tangentImpulse, at = fabs1(t);
    // the normal impulse, n' = max(n + lam_n_s, 0).  An increment below half the spacing of the floats around n rounds away (n never moves
    // again); at n = 0 a non-positive increment is clamped away; otherwise it moves by the increment and the rounding of the sum
    float Bn, n_max;
    if ((n > 0.0f && lam_n < toi_half_spacing(n)) || (n == 0.0f && lam_n_s <= 0.0f)) { Bn = 0.0f; n_max = n; }
    else { n_max = (n + 180.0f * lam_n) * kSlack; Bn = (lam_n + toi_half_ulp(n_max)) * kSlack; }
    // the tangent impulse, t' = clamp(t + lam_t_s, -M, M), M = f n.  Sliding (pushed outward by more than M can move): t' = +-M every
    // sweep, so it moves by what separates it from the bound now and by f |dn| afterwards; otherwise by the increment
    const float M = f * n, D0 = fabs1(M - at);
    const bool outward = (lam_t_s < 0.0f && t <= 0.0f) || (lam_t_s > 0.0f && t >= 0.0f);
    const float m_err = Bn == 0.0f ? 0.0f : toi_half_ulp(f * n_max);   // the rounding of M = f n while n moves
    const float slide = (fmax2(D0, f * Bn) + m_err) * kSlack;
    float Bt;
    if (outward && lam_t > 2.0f * slide + 4.0f * toi_half_ulp(at + M)) Bt = slide;
    else { const float step = fmax2(lam_t, f * Bn); Bt = (step + toi_half_ulp((at + 180.0f * step) * kSlack) + m_err) * kSlack; }
    const float mB = vc.invMassB, iB = vc.invIB;
    const float ax = fabs1(nrm.x), ay = fabs1(nrm.y), rx = fabs1(p.rB.x), ry = fabs1(p.rB.y);
    // (0.999 of the half spacing: the rounding of the products on both sides of the comparison)
    const float vx = 0.999f * toi_half_spacing(V.v.x), vy = 0.999f * toi_half_spacing(V.v.y), vw = 0.999f * toi_half_spacing(V.w);
    bool ok = vx > 0.0f && vy > 0.0f && vw > 0.0f;                        // (false for zero, denormal, tiny, infinite and NaN velocities)
    ok = ok && Bn * (mB * ax) <= vx && Bn * (mB * ay) <= vy && Bn * (iB * (rx * ay + ry * ax)) <= vw;   // normal impulse: along n
    ok = ok && Bt * (mB * ay) <= vx && Bt * (mB * ax) <= vy && Bt * (iB * (rx * ax + ry * ay)) <= vw;   // tangent impulse: along (n.y, -n.x)
    return ok;
}

LL_NOINLINE void toi_sweeps(CSolver& cs, Vel& vd_io LL_WI_PARAM) {
    Vel vd = vd_io;
    VConstraint r0, r1;
    const bool h0 = cs.count > 0, h1 = cs.count > 1;
    if (h0) r0 = cs_vc(cs, 0);
    if (h1) r1 = cs_vc(cs, 1);
#ifdef LL_TOI_SWEEP_VERIFY
    const Vel v_in = vd; const VConstraint r0_in = r0, r1_in = r1;
#endif
    bool can_stop = cs.count <= 2;  // constraints beyond the two register-resident ones are not compared
#ifdef LL_TOI_NO_SETTLED_EXIT   // A/B builds (make EXTRA=-DLL_TOI_NO_SETTLED_EXIT): the settled-velocity proof is never tried, fixed points and cycles still end the loop
    const bool can_settle = false;
#else
    const bool can_settle = can_stop && (!h0 || r0.pointCount == 1) && (!h1 || r1.pointCount == 1);
#endif
    int settle_at = 2;              // the proof is tried when the velocity has not moved during a sweep, with exponential back-off
    auto state_now = [&]() {
        ToiSweepState st;
        st.w[0] = as_u32(vd.v.x); st.w[1] = as_u32(vd.v.y); st.w[2] = as_u32(vd.w);
        st.w[3] = h0 ? as_u32(r0.points[0].normalImpulse) : 0u; st.w[4] = h0 ? as_u32(r0.points[0].tangentImpulse) : 0u;
        st.w[5] = h0 ? as_u32(r0.points[1].normalImpulse) : 0u; st.w[6] = h0 ? as_u32(r0.points[1].tangentImpulse) : 0u;
        st.w[7] = h1 ? as_u32(r1.points[0].normalImpulse) : 0u; st.w[8] = h1 ? as_u32(r1.points[0].tangentImpulse) : 0u;
        st.w[9] = h1 ? as_u32(r1.points[1].normalImpulse) : 0u; st.w[10] = h1 ? as_u32(r1.points[1].tangentImpulse) : 0u;
        return st;
    };
    // (a one-word digest is compared first: two full 11-word comparisons per sweep were a sixth of a two-constraint sweep's instructions)
    auto digest = [](const ToiSweepState& st) {
        uint32_t f = st.w[0];
#pragma unroll
        for (int q = 1; q < 11; ++q) f = (f << 3) + (f >> 29) + st.w[q] * (uint32_t)(2 * q + 1);
        return f;
    };
    // The exits are looked for after every 16th sweep only.  Looking costs ~100 instructions (two state captures, two digests, the
    // comparisons) beside a ~350-instruction sweep, and a wave leaves the loop when its LAST lane does: measured at 262 144 envs, looking after
    // every sweep 1.082 ms per step, every 4th (after 32 dense ones) 1.048, every 8th 1.008, every 16th 0.995, every 32nd 1.015, every 64th
    // 1.09, never 1.23 (profiles/r03_lunarlander/ab_records.txt).  An exit seen late changes nothing: a fixed point stays one, a cycle is
    // met again at a later multiple of its period, a settled velocity stays settled.  (LL_TOI_CHECK_DENSE: look after every sweep up to
    // there; LL_TOI_CHECK_EVERY, a power of two, and LL_TOI_CHECK_PHASE: afterwards when (sweeps done) mod EVERY == PHASE.)
#ifndef LL_TOI_CHECK_DENSE
#define LL_TOI_CHECK_DENSE 0
#endif
#ifndef LL_TOI_CHECK_EVERY
#define LL_TOI_CHECK_EVERY 16
#endif
#ifndef LL_TOI_CHECK_PHASE
#define LL_TOI_CHECK_PHASE (LL_TOI_CHECK_EVERY - 1)
#endif
    constexpr int kEverySweep = LL_TOI_CHECK_DENSE, kLookMask = LL_TOI_CHECK_EVERY - 1, kLookPhase = LL_TOI_CHECK_PHASE;
    ToiSweepState prev = state_now(), snap = prev;
    uint32_t prev_f = digest(prev), snap_f = prev_f;
    int done = 0, left = LL_WI_LIMIT(180), snap_at = 0, next_snap = 1;
    LL_DIAG_SWEEP_BEGIN(1);
    while (left > 0) {
        const bool look = can_stop && (done < kEverySweep || (done & kLookMask) == kLookPhase);   // ... after the sweep that is about to run
        if (look && done >= kEverySweep) { prev = state_now(); prev_f = digest(prev); }   // the state before the sweep (below kEverySweep `prev` is the last sweep's `cur`)
        if (h0) cs_solve_one(r0, vd);
        if (h1) cs_solve_one(r1, vd);
        if (cs.count > 2) cs_solve_velocity_range(cs, 2, cs.count, vd);
        ++done; --left;
        if (look) {
            const ToiSweepState cur = state_now();
            LL_DIAG_SWEEP(1, done - 1, true, vd.v.x, vd.v.y, vd.w, h0 ? r0.points[0].normalImpulse : 0.0f, h0 ? r0.points[0].tangentImpulse : 0.0f,
                          h0 && r0.pointCount > 1 ? r0.points[1].normalImpulse : 0.0f, h0 && r0.pointCount > 1 ? r0.points[1].tangentImpulse : 0.0f,
                          h1 ? r1.points[0].normalImpulse : 0.0f, h1 ? r1.points[0].tangentImpulse : 0.0f,
                          h1 && r1.pointCount > 1 ? r1.points[1].normalImpulse : 0.0f, h1 && r1.pointCount > 1 ? r1.points[1].tangentImpulse : 0.0f);
#ifdef LL_TOI_SWEEP_TRACE
            LL_TOI_SWEEP_TRACE(done, cur, cs.count);
#endif
            const uint32_t cur_f = digest(cur);
            if (cur_f == prev_f && toi_state_same(cur, prev)) break;                                                // fixed point
            if (cur_f == snap_f && toi_state_same(cur, snap)) { left %= done - snap_at; can_stop = false; }          // on a cycle of period done - snap_at
            else if (done >= next_snap) { snap = cur; snap_f = cur_f; snap_at = done; next_snap = 2 * done; }
            if (can_settle && can_stop && done >= settle_at && cur.w[0] == prev.w[0] && cur.w[1] == prev.w[1] && cur.w[2] == prev.w[2]) {
                if ((!h0 || toi_velocity_settled_one(r0, vd)) && (!h1 || toi_velocity_settled_one(r1, vd))) break;   // the velocity is final
                settle_at = 2 * done;
            }
            prev = cur; prev_f = cur_f;
        }
    }
    LL_DIAG_SWEEP_END(1);
    LL_TOI_SWEEP_STAT(cs.count, done);
#ifdef LL_TOI_SWEEP_VERIFY
    if (cs.count <= 2) {
        Vel vp = v_in; VConstraint q0 = r0_in, q1 = r1_in;
        for (int it = 0; it < 180; ++it) { if (h0) cs_solve_one(q0, vp); if (h1) cs_solve_one(q1, vp); }
        LL_TOI_SWEEP_VERIFY(vp, vd, done);
    }
#endif
    vd_io = vd;
}

// The <=60 position iterations of b2Island::Solve (contacts, then the joints in island order) with the three
// body positions and the first two position constraints in registers, body selection branch-free.  An iteration
// is a pure function of the positions: when one leaves them bit-for-bit unchanged without meeting the
// tolerances, the remaining iterations would repeat it, so the loop ends there with the same result (not solved).
LLD void pc_solve_on(const PConstraint& pc, Pos3& pos, float& minSeparation) {
    Pos pb = pos_pick(pos, pc.indexB);
    pc_solve_one(pc, false, pb.c, pb.a, minSeparation);
    pos_put(pos, pc.indexB, pb);
}
LL_NOINLINE bool island_position(Pos3& pos_io, const Joint& J0, const Joint& J1, const CSolver& cs, bool leg1_first, const LLConst& k_in LL_WI_PARAM) {
    Pos3 pos = pos_io;
    LLConst k;
    k.invMass[0] = k_in.invMass[0]; k.invMass[1] = k_in.invMass[1]; k.invI[0] = k_in.invI[0]; k.invI[1] = k_in.invI[1];
    k.lowerAngle[0] = k_in.lowerAngle[0]; k.lowerAngle[1] = k_in.lowerAngle[1];
    k.upperAngle[0] = k_in.upperAngle[0]; k.upperAngle[1] = k_in.upperAngle[1];
    k.localCenter[0] = k_in.localCenter[0]; k.localCenter[1] = k_in.localCenter[1];
    k.localAnchorB[0] = k_in.localAnchorB[0]; k.localAnchorB[1] = k_in.localAnchorB[1];
    const int n_cs = cs.count, n_ps = cs.ps;
    PConstraint q0, q1;
    const bool h0 = n_cs > 0, h1 = n_cs > 1;
    if (h0) q0 = cs.pc[0];
    if (h1) q1 = cs.pc[n_ps];
    bool solved = false;
    for (int it = 0; it < LL_WI_LIMIT(60); ++it) {
        const bool look = (it & 7) == 7;   // the fixed-point exit is looked for after every 8th iteration (it has not fired once in the host statistics' 134 038 solves)
        Pos3 before;
        if (look) before = pos;
        float minSeparation = 0.0f;
        if (h0) pc_solve_on(q0, pos, minSeparation);
        if (h1) pc_solve_on(q1, pos, minSeparation);
        for (int i = 2; i < n_cs; ++i) pc_solve_on(cs.pc[i * n_ps], pos, minSeparation);
        const bool contactsOkay = minSeparation >= -3.0f * b2_linearSlop;
        bool okA, okB;
        if (leg1_first) {
            okA = rj_solve_position(J1, 1, k, pos.b0, pos.b2);
            okB = rj_solve_position(J0, 0, k, pos.b0, pos.b1);
        } else {
            okA = rj_solve_position(J0, 0, k, pos.b0, pos.b1);
            okB = rj_solve_position(J1, 1, k, pos.b0, pos.b2);
        }
#ifdef LL_POS_ITER_TRACE
        LL_POS_ITER_TRACE(it, pos);
#endif
        if (contactsOkay && okA && okB) { solved = true; LL_POS_ITER_STAT(0, n_cs, it + 1, 1); break; }
        if (look && pos_same(before.b0, pos.b0) && pos_same(before.b1, pos.b1) && pos_same(before.b2, pos.b2)) { LL_POS_ITER_STAT(0, n_cs, it + 1, 2); break; }
        if (it == 59) LL_POS_ITER_STAT(0, n_cs, 60, 0);
    }
    pos_io = pos;
    return solved;
}

// the <=20 position iterations of b2Island::SolveTOI: every constraint acts on the one dynamic body
LL_NOINLINE void toi_position(const CSolver& cs, Pos& p_io) {
    Pos p = p_io;
    const int n_cs = cs.count, n_ps = cs.ps;
    PConstraint q0, q1;
    const bool h0 = n_cs > 0, h1 = n_cs > 1;
    if (h0) q0 = cs.pc[0];
    if (h1) q1 = cs.pc[n_ps];
    for (int it = 0; it < 20; ++it) {
        const Pos before = p;
        float minSeparation = 0.0f;
        if (h0) pc_solve_one(q0, true, p.c, p.a, minSeparation);
        if (h1) pc_solve_one(q1, true, p.c, p.a, minSeparation);
        for (int i = 2; i < n_cs; ++i) pc_solve_one(cs.pc[i * n_ps], true, p.c, p.a, minSeparation);
        if (minSeparation >= -1.5f * b2_linearSlop) break;
        if (pos_same(before, p)) break;
    }
    p_io = p;
}

// ---- b2World::Solve + b2Island::Solve --------------------------------------------------------------------
LLD void solve_island(World& w, const PolyTab& tab, const LLConst& k, const CSolverMem& mem, float dt, float inv_dt, float dtRatio) {
    for (int i = 0; i < 3; ++i) w.b[i].islandFlag = false;
    for (int s = 0; s < kSlots; ++s) ct_set_key(w.cs, s, ct_key(w.cs, s) & ~CK_ISLAND);
    int seed = -1;
    for (int i = 2; i >= 0; --i)
        if (w.b[i].awake) { seed = i; break; }
    if (seed < 0) return;

    // depth-first search with Box2D's stack discipline (ground = 3 on the stack)
    int ibody[3], nb = 0, nc = 0, ijoint[2], nj = 0;
    uint8_t* const icontact = (uint8_t*)w.t->idx[1];
    int cstart[4] = {0, 0, 0, 0};  // constraints of the q-th visited body: [cstart[q], cstart[q + 1])
    bool jflag[2] = {false, false}, groundFlag = false;
    int stack[6], sc = 0;
    uint8_t* const order = (uint8_t*)w.t->idx[0];
    const int n_order = contact_order(w, order);
    stack[sc++] = seed; w.b[seed].islandFlag = true;
    while (sc > 0) {
        int bi = stack[--sc];
        if (bi == 3) continue;  // static body: in the island, not traversed, contributes nothing
        cstart[nb] = nc;
        ibody[nb++] = bi;
        w.b[bi].awake = true;
        for (int q = 0; q < n_order; ++q) {
            const uint32_t key = ct_key(w.cs, order[q]);
            if (ck_body(key) != bi || (key & CK_ISLAND)) continue;
            if (!(key & CK_ENABLED) || !(key & CK_TOUCHING)) continue;
            icontact[nc++] = order[q];
            ct_set_key(w.cs, order[q], key | CK_ISLAND);
            if (groundFlag) continue;
            stack[sc++] = 3; groundFlag = true;
        }
        for (int jj = 1; jj >= 0; --jj) {
            const int leg = 1 + jj;
            if (bi != 0 && bi != leg) continue;
            if (jflag[jj]) continue;
            int other = bi == 0 ? leg : 0;
            ijoint[nj++] = jj; jflag[jj] = true;
            if (w.b[other].islandFlag) continue;
            stack[sc++] = other; w.b[other].islandFlag = true;
        }
    }

    cstart[nb] = nc;
    Pos pos[3]; Vel3 vel;
    const float h = dt;
    for (int i = 0; i < 3; ++i) {  // all three bodies are always in the island (joints connect them)
        Body& b = w.b[i];
        const int t = poly_of(i);
        b.sw.c0 = b.sw.c; b.sw.a0 = b.sw.a;
        pos[i].c = b.sw.c; pos[i].a = b.sw.a;
        Vel vi;
        integrate_velocity(b, t, k, h, vi.v, vi.w);
        vel_set(vel, i, vi);
    }

    LL_STAMP(2);
    CSolver cs;
    cs_init(cs, mem, w, k, icontact, nc, true, dtRatio);
    cs_init_velocity(cs, w, pos, vel);
    cs_warm_start(cs, vel);
    // both joints are always in the island; DFS order is [joint(leg1), joint(leg0)] unless the seed was leg0
    const bool leg1_first = ijoint[0] == 1;
    Joint J0 = w.jt[0], J1 = w.jt[1];  // register copies for the 180 sweeps
    if (leg1_first) {
        rj_init_velocity(J1, 2, k, pos[0].a, pos[2].a, vel.b0, vel.b2, dtRatio);
        rj_init_velocity(J0, 1, k, pos[0].a, pos[1].a, vel.b0, vel.b1, dtRatio);
    } else {
        rj_init_velocity(J0, 1, k, pos[0].a, pos[1].a, vel.b0, vel.b1, dtRatio);
        rj_init_velocity(J1, 2, k, pos[0].a, pos[2].a, vel.b0, vel.b2, dtRatio);
    }
    LL_STAMP(3);
    island_sweeps(J0, J1, vel, cs, cstart, ibody, nb, leg1_first, k, dt, inv_dt LL_WI_ARG(w, WI_ISLAND_SWEEPS_10, 180, 10));
    LL_STAMP(4);
    w.jt[0] = J0; w.jt[1] = J1;
    cs_store_impulses(cs, w);

    integrate_position(pos[0].c, pos[0].a, vel.b0.v, vel.b0.w, h);
    integrate_position(pos[1].c, pos[1].a, vel.b1.v, vel.b1.w, h);
    integrate_position(pos[2].c, pos[2].a, vel.b2.v, vel.b2.w, h);

    bool positionSolved;
    {
        Pos3 p3; p3.b0 = pos[0]; p3.b1 = pos[1]; p3.b2 = pos[2];
        positionSolved = island_position(p3, J0, J1, cs, leg1_first, k LL_WI_ARG(w, WI_ONE_POSITION_ITER, 60, 1));
        pos[0] = p3.b0; pos[1] = p3.b1; pos[2] = p3.b2;
    }

    for (int i = 0; i < 3; ++i) {
        Body& b = w.b[i];
        const Vel vi = vel_get(vel, i);
        b.sw.c = pos[i].c; b.sw.a = pos[i].a; b.v = vi.v; b.w = vi.w;
        body_sync_transform(b);
    }

    LL_STAMP(5);
    float minSleepTime = FLT_MAX;
    for (int q = 0; q < nb; ++q) minSleepTime = sleep_update(w.b[ibody[q]], h, minSleepTime);
    if (minSleepTime >= b2_timeToSleep && positionSolved)
        for (int i = 0; i < 3; ++i) body_set_awake(w.b[i], false);

    int moved[3], nm = 0;
    for (int i = 2; i >= 0; --i)  // body list order: newest first
        if (body_sync_fixtures(w, tab, i)) moved[nm++] = i;
    if (!LL_WHATIF(w, WI_NO_FIND_CONTACTS)) find_new_contacts(w, moved, nm);
}

// ---- b2World::SolveTOI ----------------------------------------------------------------------------------
// The reference's loop "evaluate the invalidated times of impact, take the earliest, sub-step that body, repeat", cut
// into the pieces of one pass so that the kernels can run the evaluations of a whole wave side by side:
//   toi_begin     once per world.step: clear the per-step contact bits and the sweeps' alpha0
//   toi_list      which contacts need their time of impact computed, in list order; the bodies they belong to are put
//                 onto the common time interval (b2Sweep::Advance) — everything that ORDERS the evaluations
//   toi_evaluate  ONE evaluation: a pure function of the body's sweep and the edge, result into the contact's TOI word
//   toi_advance   take the minimum; advance that body, solve the TOI island, re-open that body's contacts
// Same operations on the same operands as b2World::SolveTOI's single walk.  Everything the loop carries from one pass to
// the next lives in World (sweeps incl. alpha0, per-contact toi / toiFlag / toiCount / enabled, gA), so it is also
// resumable: solve_toi_part runs `budget` sub-steps (< 0: to the end) and reports whether it finished; a later call
// with first = false continues exactly where this one stopped (follow-up launches over compacted lists, profiling only).
struct ToiLoop { int n_order; uint32_t order_seq; };

LLD bool toi_begin(World& w, ToiLoop& L, bool first) {  // false: no contacts, SolveTOI has nothing to do
    if (first) {
        for (int i = 0; i < 3; ++i) { w.b[i].islandFlag = false; w.b[i].sw.alpha0 = 0.0f; }
        bool any = false;
        for (int s = 0; s < kSlots; ++s) {
            const uint32_t key = ct_key(w.cs, s);
            if (!(key & CK_EXISTS)) continue;
            any = true;
            ct_set_key(w.cs, s, key & kCkRestMask);  // toiFlag = islandFlag = false, toiCount = 0
            ct_set_toi(w.cs, s, 1.0f);
        }
        w.gA = 0.0f;
        if (!any) return false;
    }
    // the contact list, newest first: SolveTOI only ever adds contacts (FindNewContacts after a sub-step), so the
    // order is rebuilt when the sequence counter has moved and not on every pass
    L.n_order = contact_order(w, (uint8_t*)w.t->idx[0]);
    L.order_seq = w.next_seq;
    return true;
}

// b2World::SolveTOI walks the contact list once, computing the time of impact of every contact whose cached value was
// invalidated and tracking the minimum.  Here the walk is three passes with the same per-contact order and
// arithmetic: (1) this one lists the contacts that need a computation (w.t->idx[1][0 .. n)), (2) toi_evaluate computes
// them, (3) toi_advance takes the minimum.
LLD int toi_list(World& w, ToiLoop& L) {
    uint8_t* const order = (uint8_t*)w.t->idx[0];
    if (w.next_seq != L.order_seq) { L.n_order = contact_order(w, order); L.order_seq = w.next_seq; }
    uint8_t* const need = (uint8_t*)w.t->idx[1];
    int n_need = 0;
    for (int q = 0; q < L.n_order; ++q) {
        const uint32_t key = ct_key(w.cs, order[q]);
        if (!(key & CK_ENABLED) || ck_toi_count(key) > b2_maxSubSteps || (key & CK_TOIFLAG)) continue;
        if (!w.b[ck_body(key)].awake) continue;
        need[n_need++] = order[q];
    }
    // Put the sweeps onto the same time interval, in evaluation order (the shared static body carries an alpha0 too: gA).
    // gA only ratchets up to the time of the last sub-step and every body's alpha0 is at most that, so a body is advanced
    // at most once per pass, before its first evaluation: its sweep is final for all of them, and an evaluation depends
    // on no other body — which is what lets the evaluations of a pass run in any order, or side by side.
    float gA = w.gA;
    for (int j = 0; j < n_need; ++j) {
        Body& bB = w.b[ck_body(ct_key(w.cs, need[j]))];
        if (gA < bB.sw.alpha0) gA = bB.sw.alpha0;
        else if (bB.sw.alpha0 < gA) sweep_advance(bB.sw, gA);
    }
    w.gA = gA;
    return n_need;
}

// (`wo` may be ANOTHER lane's World record, in LDS: the contact kernel deals the evaluations of a wave out over all its lanes)
LLD void toi_evaluate(World& wo, const PolyTab& tab, int slot) {
    const uint32_t key = ct_key(wo.cs, slot);
    const int body = ck_body(key);
    const Body& bB = wo.b[body];
    const float alpha0 = bB.sw.alpha0;  // == gA after toi_list
    V2 ev[2];
    edge_verts(wo, ck_edge(key), ev[0], ev[1]);
    float beta;
    LL_STAMP(8);
    int state = (LL_WHATIF(wo, WI_NO_TOI_EVAL) || (LL_WHATIF(wo, WI_SUB_NO_REEVAL) && wo.gA > 0.0f)) ? TOI_SEPARATED : time_of_impact(ev, tab, poly_of(body), bB.sw, beta);
    LL_STAMP(9);
    float alpha;
    if (state == TOI_TOUCHING) alpha = fmin2(alpha0 + (1.0f - alpha0) * beta, 1.0f);
    else alpha = 1.0f;
    ct_set_toi(wo.cs, slot, alpha);
    ct_set_key(wo.cs, slot, key | CK_TOIFLAG);
}

// The same evaluation as a TASK RECORD that any lane can run — for blocks whose World records are private to their lanes (64-lane blocks)
// and which still deal the evaluations of a pass out over the wave: the owner writes the operands (the body's sweep, the edge, the polygon:
// 14 words, column `t` of a [16][stride] table in LDS), some lane computes, the owner applies the result to its contact.  Same operations on the
// same operands as toi_evaluate.
constexpr int kToiTaskWords = 16;
LLD void toi_task_write(const World& wo, int slot, LL_LDS float* rec, int stride) {
    const uint32_t key = ct_key(wo.cs, slot);
    const int body = ck_body(key);
    const Sweep& sw = wo.b[body].sw;
    V2 v1, v2;
    edge_verts(wo, ck_edge(key), v1, v2);
    const bool skip = LL_WHATIF(wo, WI_NO_TOI_EVAL) || (LL_WHATIF(wo, WI_SUB_NO_REEVAL) && wo.gA > 0.0f);
    rec[0 * stride] = sw.localCenter.x; rec[1 * stride] = sw.localCenter.y; rec[2 * stride] = sw.c0.x; rec[3 * stride] = sw.c0.y;
    rec[4 * stride] = sw.c.x; rec[5 * stride] = sw.c.y; rec[6 * stride] = sw.a0; rec[7 * stride] = sw.a; rec[8 * stride] = sw.alpha0;
    rec[9 * stride] = v1.x; rec[10 * stride] = v1.y; rec[11 * stride] = v2.x; rec[12 * stride] = v2.y;
    rec[13 * stride] = as_f32(skip ? 0xffffffffu : (uint32_t)poly_of(body));
}
LLD void toi_task_run(LL_LDS float* rec, int stride, const PolyTab& tab) {
    Sweep sw;
    sw.localCenter = mk(rec[0 * stride], rec[1 * stride]); sw.c0 = mk(rec[2 * stride], rec[3 * stride]); sw.c = mk(rec[4 * stride], rec[5 * stride]);
    sw.a0 = rec[6 * stride]; sw.a = rec[7 * stride]; sw.alpha0 = rec[8 * stride];
    V2 ev[2];
    ev[0] = mk(rec[9 * stride], rec[10 * stride]); ev[1] = mk(rec[11 * stride], rec[12 * stride]);
    const uint32_t pi = as_u32(rec[13 * stride]);
    float beta;
    LL_STAMP(8);
    const int state = pi == 0xffffffffu ? TOI_SEPARATED : time_of_impact(ev, tab, (int)pi, sw, beta);
    LL_STAMP(9);
    rec[14 * stride] = state == TOI_TOUCHING ? fmin2(sw.alpha0 + (1.0f - sw.alpha0) * beta, 1.0f) : 1.0f;
}
LLD void toi_task_apply(World& wo, int slot, const LL_LDS float* rec, int stride) {
    ct_set_toi(wo.cs, slot, rec[14 * stride]);
    ct_set_key(wo.cs, slot, ct_key(wo.cs, slot) | CK_TOIFLAG);
}

enum { TOI_DONE = 0, TOI_AGAIN = 1, TOI_OUT_OF_BUDGET = 2 };
LLD int toi_advance(World& w, const PolyTab& tab, const LLConst& k, const CSolverMem& mem, float dt, ToiLoop& L, int& budget) {
    uint8_t* const order = (uint8_t*)w.t->idx[0];
    const int n_order = L.n_order;
    {
        int minSlot = -1;
        float minAlpha = 1.0f;
        for (int q = 0; q < n_order; ++q) {
            const uint32_t key = ct_key(w.cs, order[q]);
            if (!(key & CK_ENABLED)) continue;
            if (ck_toi_count(key) > b2_maxSubSteps) continue;
            if (!(key & CK_TOIFLAG)) continue;  // body asleep: skipped by the reference before any computation
            const float alpha = ct_toi(w.cs, order[q]);
            if (alpha < minAlpha) { minSlot = order[q]; minAlpha = alpha; }
        }
        LL_STAMP(16);
        if (minSlot < 0 || 1.0f - 10.0f * b2_epsilon < minAlpha || LL_WHATIF(w, WI_NO_SUBSTEPS)) return TOI_DONE;
        if (budget == 0) return TOI_OUT_OF_BUDGET;  // the evaluated times of impact are cached in the contacts: the next call picks the same minimum
        if (budget > 0) --budget;

        Contact& minContact = *(Contact*)&w.t->tmp;
        minContact = ct_get(w.cs, minSlot);
        const int dyn = minContact.body;
        Body& bB = w.b[dyn];
        Sweep backup = bB.sw;
        const float gA_backup = w.gA;
        w.gA = minAlpha;  // bA->Advance(minAlpha) on the ground
        body_advance(bB, minAlpha);
        contact_update(w, tab, minContact);
        minContact.toiFlag = false;
        ++minContact.toiCount;
        if (!minContact.enabled || !minContact.touching) {
            minContact.enabled = false;
            ct_put(w.cs, minSlot, minContact);
            bB.sw = backup;
            w.gA = gA_backup;
            body_sync_transform(bB);
            return TOI_AGAIN;
        }
        body_set_awake(bB, true);
        LL_STAMP(17);
#ifdef LL_DIAG_SUBSTEP_LANES
        LL_DIAG_SUBSTEP_LANES();   // tools/ll_phase_prof.hip: how many lanes of the wave take this sub-step
#endif

        uint8_t* const islandSlots = (uint8_t*)w.t->idx[2];
        int nc = 0;
        islandSlots[nc++] = minSlot;
        minContact.islandFlag = true;
        ct_put(w.cs, minSlot, minContact);
        // the body's other contacts, in list order; listed first so that the lanes of a wave run contact_update once per
        // list position of their own (an update changes no other contact's key)
        uint8_t* const upd = (uint8_t*)w.t->idx[1];   // (the list of evaluations has been consumed)
        int n_upd = 0;
        for (int q = 0; q < n_order; ++q) {
            const uint32_t key = ct_key(w.cs, order[q]);
            if (!(key & CK_EXISTS) || ck_body(key) != dyn || (key & CK_ISLAND)) continue;
            upd[n_upd++] = order[q];
        }
        if (LL_WHATIF(w, WI_NO_TOI_OTHER_UPDATES)) n_upd = 0;
        for (int j = 0; j < n_upd; ++j) {
            Contact& c = *(Contact*)&w.t->tmp;   // (the impacting contact has been put back by now)
            c = ct_get(w.cs, upd[j]);
            contact_update(w, tab, c);
            if (c.enabled && c.touching) { c.islandFlag = true; islandSlots[nc++] = upd[j]; }
            ct_put(w.cs, upd[j], c);
        }

        // b2Island::SolveTOI: positions/velocities of the one dynamic body (slot `dyn` of a 3-entry table)
        LL_STAMP(10);
        const float sub_dt = (1.0f - minAlpha) * dt;
        Pos pos[3]; Vel3 vel;
        pos[dyn].c = bB.sw.c; pos[dyn].a = bB.sw.a;
        Vel vd; vd.v = bB.v; vd.w = bB.w;
        vel.b0 = vd; vel.b1 = vd; vel.b2 = vd;  // cs_init_velocity reads slot `dyn` only
        CSolver cs;
        cs_init(cs, mem, w, k, islandSlots, nc, false, 1.0f);
        LL_STAMP(12);
        {
            Pos pd = pos[dyn];
            if (!LL_WHATIF(w, WI_SUB_NO_POSITION)) toi_position(cs, pd);
            pos[dyn] = pd;
        }
        LL_STAMP(13);
        bB.sw.c0 = pos[dyn].c; bB.sw.a0 = pos[dyn].a;
        cs_init_velocity(cs, w, pos, vel);
        LL_STAMP(14);
        toi_sweeps(cs, vd LL_WI_ARG(w, WI_TOI_SWEEPS_12, 180, 12));
        LL_STAMP(15);
        {
            const float h = sub_dt;
            V2 c = pos[dyn].c; float a = pos[dyn].a;
            integrate_position(c, a, vd.v, vd.w, h);
            bB.sw.c = c; bB.sw.a = a; bB.v = vd.v; bB.w = vd.w;
            body_sync_transform(bB);
        }
        LL_STAMP(11);
        int moved[1], nm = 0;
        if (!LL_WHATIF(w, WI_SUB_NO_SYNC) && body_sync_fixtures(w, tab, dyn)) moved[nm++] = dyn;
        for (int s = 0; s < kSlots; ++s) {
            const uint32_t key = ct_key(w.cs, s);
            if ((key & CK_EXISTS) && ck_body(key) == dyn) ct_set_key(w.cs, s, key & ~(CK_TOIFLAG | CK_ISLAND));
        }
        if (!LL_WHATIF(w, WI_SUB_NO_FIND)) find_new_contacts(w, moved, nm);
        LL_STAMP(18);
        if (LL_WHATIF(w, WI_ONE_SUBSTEP)) return TOI_DONE;
    }
    return TOI_AGAIN;
}

// one lane runs every piece itself (host check, the kernels that do not deal evaluations out)
LLD bool solve_toi_part(World& w, const PolyTab& tab, const LLConst& k, const CSolverMem& mem, float dt, bool first, int budget) {
    ToiLoop L;
    if (!toi_begin(w, L, first)) return true;
    for (;;) {
        const int n_need = toi_list(w, L);
        for (int j = 0; j < n_need; ++j) toi_evaluate(w, tab, ((const uint8_t*)w.t->idx[1])[j]);
        const int r = toi_advance(w, tab, k, mem, dt, L, budget);
        if (r != TOI_AGAIN) return r == TOI_DONE;
    }
}

// b2World::Step(1/50, 180, 60) in three pieces: everything before SolveTOI, SolveTOI (resumable, above), the tail.
constexpr float kStepDt = 1.0f / 50.0f;
LLD void world_step_begin(World& w, const PolyTab& tab, const LLConst& k, const CSolverMem& mem) {
    if (w.newContacts) {
        int order[3], n = 0;
        for (int i = 0; i < 3; ++i)
            if (w.pending & (1u << i)) order[n++] = i;
        find_new_contacts(w, order, n);
        w.pending = 0;
        w.newContacts = false;
    }
    const float dt = kStepDt;
    const float inv_dt = 1.0f / dt;
    const float dtRatio = (w.stepped_once ? inv_dt : 0.0f) * dt;
    LL_STAMP(0);
    collide(w, tab);
    LL_STAMP(1);
    solve_island(w, tab, k, mem, dt, inv_dt, dtRatio);
    LL_STAMP(6);
}
// the same in the pieces the contact kernel deals out over its wave: world_step_pre (-> number of manifold refreshes, listed),
// collide_refresh per listed contact (any lane), world_step_island
LLD int world_step_pre(World& w) {
    if (w.newContacts) {
        int order[3], n = 0;
        for (int i = 0; i < 3; ++i)
            if (w.pending & (1u << i)) order[n++] = i;
        find_new_contacts(w, order, n);
        w.pending = 0;
        w.newContacts = false;
    }
    LL_STAMP(0);
    return collide_list(w);
}
LLD void world_step_island(World& w, const PolyTab& tab, const LLConst& k, const CSolverMem& mem) {
    collide_apply(w);
    LL_STAMP(1);
    const float dt = kStepDt;
    const float inv_dt = 1.0f / dt;
    const float dtRatio = (w.stepped_once ? inv_dt : 0.0f) * dt;
    solve_island(w, tab, k, mem, dt, inv_dt, dtRatio);
    LL_STAMP(6);
}
LLD void world_step_end(World& w) {
    LL_STAMP(7);
    w.stepped_once = true;
    for (int i = 0; i < 3; ++i) { w.b[i].force = mk(0.0f, 0.0f); w.b[i].torque = 0.0f; }
}
LLD void world_step(World& w, const PolyTab& tab, const LLConst& k, const CSolverMem& mem) {
    world_step_begin(w, tab, k, mem);
    (void)solve_toi_part(w, tab, k, mem, kStepDt, true, -1);
    world_step_end(w);
}

}  // namespace ll
}  // namespace mgym
