// ll_b2.h — device-side rigid-body machinery for the LunarLander kernels.
//
// The reference's LunarLanderV3 (src/box_2d/lunar_lander.rs) drives box2d-rs 0.0.4 (a Rust port of
// Box2D v2.4.x; not vendored in the reference).  This header restates, for one environment per
// wavefront lane, the Box2D algorithms that `world.step(1/50, 180, 60)` (lunar_lander.rs:1066)
// executes for the lander scene: 1 static ground body with 11 two-sided edges, 3 dynamic polygon
// bodies (lander hexagon + 2 leg boxes), 2 revolute joints, edge-vs-polygon contacts.
//
// MI355X design notes
//  * The ground is static at the origin, so every "body A" term of Box2D's two-body formulas is an
//    exact zero (invMassA = invIA = 0, xfA = identity, vA = wA = 0); they are dropped, which is
//    bit-identical to evaluating them (x + 0 = x, x - 0*y = x).
//  * Per-lane working set: body positions/velocities and both joints live in VGPRs across the
//    180-iteration velocity loop; the contact cache (<= 8 slots) and the contact constraints live in
//    per-lane local arrays (rare path: only lanes near the ground touch them).
//  * Polygon tables (vertices/normals of the lander hexagon and the leg box) are indexed per lane
//    (support points, reference faces), so they are staged in LDS by the kernel prologue; uniform
//    scalars (masses, inertias, joint limits) travel in SGPRs.
//  * All arithmetic keeps Box2D's written operation order, compiled with -ffp-contract=off, IEEE
//    div/sqrt, sin/cos from mgym_math.h, so a step is bit-comparable with the CPU oracle.
#pragma once
#include <float.h>
#include <stdint.h>

#include "mgym_math.h"

namespace mgym {
namespace ll {

// ---- Box2D constants (b2_common.h) --------------------------------------------------------------
constexpr float b2_pi = 3.14159265359f;
constexpr float b2_linearSlop = 0.005f;
constexpr float b2_angularSlop = 2.0f / 180.0f * b2_pi;
constexpr float b2_polygonRadius = 2.0f * b2_linearSlop;
constexpr float b2_aabbExtension = 0.1f;
constexpr float b2_aabbMultiplier = 4.0f;
constexpr float b2_maxLinearCorrection = 0.2f;
constexpr float b2_maxAngularCorrection = 8.0f / 180.0f * b2_pi;
constexpr float b2_maxTranslation = 2.0f;
constexpr float b2_maxTranslationSquared = b2_maxTranslation * b2_maxTranslation;
constexpr float b2_maxRotation = 0.5f * b2_pi;
constexpr float b2_maxRotationSquared = b2_maxRotation * b2_maxRotation;
constexpr float b2_baumgarte = 0.2f;
constexpr float b2_toiBaumgarte = 0.75f;
constexpr float b2_timeToSleep = 0.5f;
constexpr float b2_linearSleepTolerance = 0.01f;
constexpr float b2_angularSleepTolerance = 2.0f / 180.0f * b2_pi;
constexpr int b2_maxSubSteps = 8;
constexpr float b2_epsilon = FLT_EPSILON;

constexpr int kEdges = 11;       // ground fixtures: base edge + 10 terrain edges
constexpr int kSlots = 12;       // contact cache capacity per environment (8 was hit ~5e-5 of env-steps in testing)
constexpr int kMaxPoly = 6;      // lander hexagon; legs use 4

// ---- math (b2_math.h) -----------------------------------------------------------------------------
struct V2 { float x, y; };
struct Rot { float s, c; };
struct Xf { V2 p; Rot q; };
struct AABB { V2 lo, hi; };

#define LLD __device__ __forceinline__
// out-of-line device functions (own register allocation; ONE copy of their local arrays in the caller's scratch frame)
#if defined(__HIPCC__)
#ifdef LL_CONTACT_NUM_VGPR   // register budget of the contact path (see lunar_lander.hip): the out-of-line functions count towards the kernel's
#define LL_NOINLINE __device__ __noinline__ __attribute__((amdgpu_num_vgpr(LL_CONTACT_NUM_VGPR)))
#else
#define LL_NOINLINE __device__ __noinline__
#endif
#else
#define LL_NOINLINE static
#endif
LLD V2 mk(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
LLD V2 operator+(V2 a, V2 b) { return mk(a.x + b.x, a.y + b.y); }
LLD V2 operator-(V2 a, V2 b) { return mk(a.x - b.x, a.y - b.y); }
LLD V2 operator-(V2 a) { return mk(-a.x, -a.y); }
LLD V2 operator*(float s, V2 a) { return mk(s * a.x, s * a.y); }
LLD float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
LLD float cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
LLD V2 cross_vs(V2 a, float s) { return mk(s * a.y, -s * a.x); }
LLD V2 cross_sv(float s, V2 a) { return mk(-s * a.y, s * a.x); }
LLD float len(V2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
LLD float len2(V2 a) { return a.x * a.x + a.y * a.y; }
LLD float normalize(V2& a) {
    float l = len(a);
    if (l < b2_epsilon) return 0.0f;
    float inv = 1.0f / l;
    a.x *= inv; a.y *= inv;
    return l;
}
LLD float fmin2(float a, float b) { return a < b ? a : b; }
LLD float fmax2(float a, float b) { return a > b ? a : b; }
LLD float fclamp(float a, float lo, float hi) { return fmax2(lo, fmin2(a, hi)); }
LLD float fabs1(float a) { return a > 0.0f ? a : -a; }
LLD Rot rot_set(float angle) { Rot q; mg_sincosf_u(angle, &q.s, &q.c); return q; }
LLD V2 rmul(Rot q, V2 v) { return mk(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
LLD V2 rmulT(Rot q, V2 v) { return mk(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
LLD V2 xmul(Xf T, V2 v) { return mk((T.q.c * v.x - T.q.s * v.y) + T.p.x, (T.q.s * v.x + T.q.c * v.y) + T.p.y); }
LLD V2 xmulT(Xf T, V2 v) {
    float px = v.x - T.p.x, py = v.y - T.p.y;
    return mk(T.q.c * px + T.q.s * py, -T.q.s * px + T.q.c * py);
}
LLD V2 mat22_solve(float a11, float a12, float a21, float a22, V2 b) {  // b2Mat22::Solve
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    return mk(det * (a22 * b.x - a12 * b.y), det * (a11 * b.y - a21 * b.x));
}

// ---- shapes: tables in LDS ---------------------------------------------------------------------------
// poly 0 = lander hexagon, poly 1 = leg box (both legs share it)
struct PolyTab {
    V2 v[2][kMaxPoly];
    V2 n[2][kMaxPoly];
    int count[2];
};

// uniform per-handle constants (SGPRs)
struct LLConst {
    float gravity, wind_power, turbulence_power;
    int enable_wind;
    float mass[2], invMass[2], invI[2];  // [0] lander, [1] leg
    V2 localCenter[2];
    float friction[2];                   // mixed with the ground's 0.1: sqrt(fA*fB)
    float lowerAngle[2], upperAngle[2], motorSpeed[2];  // joint of leg0, leg1
    V2 localAnchorB[2];
    float maxMotorTorque;
    V2 poly_v[2][kMaxPoly], poly_n[2][kMaxPoly];
    int poly_count[2];
};

struct Sweep { V2 localCenter, c0, c; float a0, a, alpha0; };

LLD Xf sweep_xf(const Sweep& s, float beta) {  // b2Sweep::GetTransform
    Xf xf;
    xf.p = (1.0f - beta) * s.c0 + beta * s.c;
    float angle = (1.0f - beta) * s.a0 + beta * s.a;
    xf.q = rot_set(angle);
    xf.p = xf.p - rmul(xf.q, s.localCenter);
    return xf;
}
LLD void sweep_advance(Sweep& s, float alpha) {  // b2Sweep::Advance
    float beta = (alpha - s.alpha0) / (1.0f - s.alpha0);
    s.c0 = s.c0 + beta * (s.c - s.c0);
    s.a0 += beta * (s.a - s.a0);
    s.alpha0 = alpha;
}
LLD void sweep_normalize(Sweep& s) {  // b2Sweep::Normalize
    float twoPi = 2.0f * b2_pi;
    float d = twoPi * floorf(s.a0 / twoPi);
    s.a0 -= d;
    s.a -= d;
}

// dynamic body (index 0 lander, 1 leg0, 2 leg1)
struct Body {
    Xf xf;        // xf.q == rot_set(sw.a) always
    Sweep sw;
    V2 v; float w;
    V2 force; float torque;
    float sleepTime;
    bool awake, islandFlag;
};

LLD void body_sync_transform(Body& b) {  // b2Body::SynchronizeTransform
    b.xf.q = rot_set(b.sw.a);
    b.xf.p = b.sw.c - rmul(b.xf.q, b.sw.localCenter);
}
LLD void body_set_awake(Body& b, bool flag) {  // b2Body::SetAwake
    if (flag) { b.awake = true; b.sleepTime = 0.0f; }
    else {
        b.awake = false; b.sleepTime = 0.0f;
        b.v = mk(0.0f, 0.0f); b.w = 0.0f; b.force = mk(0.0f, 0.0f); b.torque = 0.0f;
    }
}
LLD void body_advance(Body& b, float alpha) {  // b2Body::Advance
    sweep_advance(b.sw, alpha);
    b.sw.c = b.sw.c0;
    b.sw.a = b.sw.a0;
    b.xf.q = rot_set(b.sw.a);
    b.xf.p = b.sw.c - rmul(b.xf.q, b.sw.localCenter);
}

// ---- contact cache ---------------------------------------------------------------------------------------
struct CF { uint8_t indexA, indexB, typeA, typeB; };  // b2ContactFeature
LLD bool cf_equal(CF a, CF b) { return a.indexA == b.indexA && a.indexB == b.indexB && a.typeA == b.typeA && a.typeB == b.typeB; }
LLD uint32_t cf_pack(CF a) { return (uint32_t)a.indexA | ((uint32_t)a.indexB << 8) | ((uint32_t)a.typeA << 16) | ((uint32_t)a.typeB << 24); }
LLD CF cf_unpack(uint32_t u) { CF a; a.indexA = (uint8_t)u; a.indexB = (uint8_t)(u >> 8); a.typeA = (uint8_t)(u >> 16); a.typeB = (uint8_t)(u >> 24); return a; }
enum { CF_VERTEX = 0, CF_FACE = 1, FACE_A = 1, FACE_B = 2 };

struct MPoint { V2 localPoint; float normalImpulse, tangentImpulse; CF id; };
struct Manifold { MPoint points[2]; V2 localNormal, localPoint; int type, pointCount; };

struct Contact {
    bool exists, touching, enabled, islandFlag, toiFlag;
    uint32_t seq;   // creation order: Box2D pushes new contacts at the head of its lists
    int body, edge; // dynamic body 0..2, ground edge 0..10
    Manifold m;
    int toiCount; float toi;
};

// Contact cache STORAGE.  The 12 slots x 16 words of an environment live in the engine's HBM columns ([word][n_pad],
// lane = env index).  The manifold words (2..14) are worked on IN PLACE — they are only touched when a contact is
// updated or enters a solver, and a local Contact[12] would take 1.1 KB of scratch per lane (the runtime caps the
// concurrent waves of a kernel by its scratch size: profiles/r02_lunarlander/scratch_vs_concurrent_waves.txt).  The
// three words every list walk reads — KEY, SEQ, TOI — are staged in LDS for the duration of a step (CtStore::k, one
// column per lane): SolveTOI scans them again after every sub-step, and a chain of dependent loads costs ~100 cycles
// each from LDS against >= 500 from L2.  Only the contact being worked on is held in registers (struct Contact).
// Slot words: KEY SEQ LNX LNY LPX LPY P0X P0Y P0N P0T P1X P1Y P1N P1T IDS TOI.
// KEY bits: 0 exists, 1 touching, 2 enabled, 3-4 body, 5-8 edge, 9-10 manifold type, 11-12 pointCount, 13 toiFlag,
// 14-17 toiCount, 18 islandFlag (the last three and TOI only matter inside a step).  IDS = both contact-feature ids, 16 bits each.
#if defined(__HIP_DEVICE_COMPILE__)
#define LL_LDS __attribute__((address_space(3)))
#else
#define LL_LDS
#endif
struct CtHot { LL_LDS uint32_t* k; uint32_t ks; uint32_t toi_staged; };  // staged KEY / SEQ / TOI words: word j of slot s at k[(j * kSlots + s) * ks]; TOI only if toi_staged (else in place)
struct CtStore { uint32_t* p; uint64_t stride; CtHot hot; };
enum : uint32_t { CK_EXISTS = 1u, CK_TOUCHING = 2u, CK_ENABLED = 4u, CK_TOIFLAG = 1u << 13, CK_ISLAND = 1u << 18 };
constexpr uint32_t kCkRestMask = 0x1fffu;  // the key bits that persist between steps
LLD uint32_t& ct_word(const CtStore& c, int s, int k) { return c.p[(uint64_t)(16 * s + k) * c.stride]; }  // k in 2..14 (15: TOI when it is not staged)
LLD uint32_t ct_key(const CtStore& c, int s) { return c.hot.k[(uint32_t)s * c.hot.ks]; }
LLD void ct_set_key(const CtStore& c, int s, uint32_t v) { c.hot.k[(uint32_t)s * c.hot.ks] = v; }
LLD uint32_t ct_seq(const CtStore& c, int s) { return c.hot.k[(uint32_t)(kSlots + s) * c.hot.ks]; }
LLD void ct_set_seq(const CtStore& c, int s, uint32_t v) { c.hot.k[(uint32_t)(kSlots + s) * c.hot.ks] = v; }
LLD float ct_toi(const CtStore& c, int s) { return as_f32(c.hot.toi_staged ? c.hot.k[(uint32_t)(2 * kSlots + s) * c.hot.ks] : ct_word(c, s, 15)); }
LLD void ct_set_toi(const CtStore& c, int s, float v) {
    if (c.hot.toi_staged) c.hot.k[(uint32_t)(2 * kSlots + s) * c.hot.ks] = as_u32(v); else ct_word(c, s, 15) = as_u32(v);
}
LLD int ck_body(uint32_t key) { return (int)((key >> 3) & 3u); }
LLD int ck_edge(uint32_t key) { return (int)((key >> 5) & 15u); }
LLD int ck_toi_count(uint32_t key) { return (int)((key >> 14) & 15u); }
// both contact-feature ids of a manifold in one word (every field is < 16: vertex / face indices < 8, types 0 / 1)
LLD uint32_t cf16(CF a) { return (uint32_t)a.indexA | ((uint32_t)a.indexB << 4) | ((uint32_t)a.typeA << 8) | ((uint32_t)a.typeB << 12); }
LLD CF cf16_unpack(uint32_t u) {
    CF a; a.indexA = (uint8_t)(u & 15u); a.indexB = (uint8_t)((u >> 4) & 15u); a.typeA = (uint8_t)((u >> 8) & 15u); a.typeB = (uint8_t)((u >> 12) & 15u);
    return a;
}
LLD uint32_t ct_encode_key(const Contact& c) {
    return (c.exists ? CK_EXISTS : 0u) | (c.touching ? CK_TOUCHING : 0u) | (c.enabled ? CK_ENABLED : 0u) | ((uint32_t)c.body << 3) |
           ((uint32_t)c.edge << 5) | ((uint32_t)c.m.type << 9) | ((uint32_t)c.m.pointCount << 11) | (c.toiFlag ? CK_TOIFLAG : 0u) |
           (((uint32_t)c.toiCount & 15u) << 14) | (c.islandFlag ? CK_ISLAND : 0u);
}
LLD Contact ct_get(const CtStore& st, int s) {
    Contact c;
    const uint32_t key = ct_key(st, s);
    c.exists = key & CK_EXISTS; c.touching = key & CK_TOUCHING; c.enabled = key & CK_ENABLED;
    c.islandFlag = key & CK_ISLAND; c.toiFlag = key & CK_TOIFLAG;
    c.body = ck_body(key); c.edge = ck_edge(key);
    c.m.type = (int)((key >> 9) & 3u); c.m.pointCount = (int)((key >> 11) & 3u);
    c.toiCount = ck_toi_count(key);
    c.seq = ct_seq(st, s);
    c.m.localNormal = mk(as_f32(ct_word(st, s, 2)), as_f32(ct_word(st, s, 3)));
    c.m.localPoint = mk(as_f32(ct_word(st, s, 4)), as_f32(ct_word(st, s, 5)));
    for (int p = 0; p < 2; ++p) {
        c.m.points[p].localPoint = mk(as_f32(ct_word(st, s, 6 + 4 * p)), as_f32(ct_word(st, s, 7 + 4 * p)));
        c.m.points[p].normalImpulse = as_f32(ct_word(st, s, 8 + 4 * p));
        c.m.points[p].tangentImpulse = as_f32(ct_word(st, s, 9 + 4 * p));
    }
    const uint32_t ids = ct_word(st, s, 14);
    c.m.points[0].id = cf16_unpack(ids); c.m.points[1].id = cf16_unpack(ids >> 16);
    c.toi = ct_toi(st, s);
    return c;
}
LLD void ct_put(const CtStore& st, int s, const Contact& c) {
    ct_set_key(st, s, ct_encode_key(c));
    ct_set_seq(st, s, c.seq);
    ct_word(st, s, 2) = as_u32(c.m.localNormal.x); ct_word(st, s, 3) = as_u32(c.m.localNormal.y);
    ct_word(st, s, 4) = as_u32(c.m.localPoint.x); ct_word(st, s, 5) = as_u32(c.m.localPoint.y);
    for (int p = 0; p < 2; ++p) {
        ct_word(st, s, 6 + 4 * p) = as_u32(c.m.points[p].localPoint.x); ct_word(st, s, 7 + 4 * p) = as_u32(c.m.points[p].localPoint.y);
        ct_word(st, s, 8 + 4 * p) = as_u32(c.m.points[p].normalImpulse); ct_word(st, s, 9 + 4 * p) = as_u32(c.m.points[p].tangentImpulse);
    }
    ct_word(st, s, 14) = cf16(c.m.points[0].id) | (cf16(c.m.points[1].id) << 16);
    ct_set_toi(st, s, c.toi);
}

struct ClipV { V2 v; CF id; };

LLD int clip_segment(ClipV vOut[2], const ClipV vIn[2], V2 normal, float offset, int vertexIndexA) {  // b2ClipSegmentToLine
    int count = 0;
    float distance0 = dot(normal, vIn[0].v) - offset;
    float distance1 = dot(normal, vIn[1].v) - offset;
    if (distance0 <= 0.0f) vOut[count++] = vIn[0];
    if (distance1 <= 0.0f) vOut[count++] = vIn[1];
    if (distance0 * distance1 < 0.0f) {
        float interp = distance0 / (distance0 - distance1);
        vOut[count].v = vIn[0].v + interp * (vIn[1].v - vIn[0].v);
        vOut[count].id.indexA = (uint8_t)vertexIndexA;
        vOut[count].id.indexB = vIn[0].id.indexB;
        vOut[count].id.typeA = CF_VERTEX;
        vOut[count].id.typeB = CF_FACE;
        ++count;
    }
    return count;
}

// b2CollideEdgeAndPolygon (v2.4.1 form), two-sided edge in the ground frame (xfA = identity => xf = xfB)
// `tmp`: room for 2 * kMaxPoly vectors (the polygon's vertices and normals in the edge's frame, indexed at run time) supplied by
// the caller — World::poly_tmp, which is LDS when the World record is (a local array here would be scratch memory)
// On the device both `tab_in` (every kernel stages the polygon table in LDS) and `tmp` (WorldTmp::poly_tmp, or a helper lane's piece of the
// constraint columns) ARE LDS: an out-of-line function only sees generic pointers, and a generic access to LDS is a FLAT instruction —
// the address-space casts below turn the ~50 of them in here into ds_read / ds_write.
LL_NOINLINE void collide_edge_polygon(Manifold& manifold, V2 v1, V2 v2, const PolyTab& tab_in, int pi, Xf xfB, V2* tmp) {
    const LL_LDS PolyTab& tab = *(const LL_LDS PolyTab*)&tab_in;
    manifold.pointCount = 0;
    const Xf xf = xfB;
    V2 edge1 = v2 - v1;
    normalize(edge1);
    V2 normal1 = mk(edge1.y, -edge1.x);
    const int tcount = tab.count[pi];
    LL_LDS V2* const tv = (LL_LDS V2*)tmp;
    LL_LDS V2* const tn = tv + kMaxPoly;
    for (int i = 0; i < tcount; ++i) {
        tv[i] = xmul(xf, tab.v[pi][i]);
        tn[i] = rmul(xf.q, tab.n[pi][i]);
    }
    const float radius = b2_polygonRadius + b2_polygonRadius;

    int e_index = -1; float e_sep = -FLT_MAX; V2 e_normal = mk(0.0f, 0.0f);  // b2ComputeEdgeSeparation
    for (int j = 0; j < 2; ++j) {
        V2 axis = j == 0 ? normal1 : -normal1;
        float sj = FLT_MAX;
        for (int i = 0; i < tcount; ++i) {
            float si = dot(axis, tv[i] - v1);
            if (si < sj) sj = si;
        }
        if (sj > e_sep) { e_index = j; e_sep = sj; e_normal = axis; }
    }
    if (e_sep > radius) return;

    int p_index = -1; float p_sep = -FLT_MAX; V2 p_normal = mk(0.0f, 0.0f);  // b2ComputePolygonSeparation
    for (int i = 0; i < tcount; ++i) {
        V2 n = -tn[i];
        float s1 = dot(n, tv[i] - v1);
        float s2 = dot(n, tv[i] - v2);
        float s = fmin2(s1, s2);
        if (s > p_sep) { p_index = i; p_sep = s; p_normal = n; }
    }
    if (p_sep > radius) return;

    const float k_relativeTol = 0.98f, k_absoluteTol = 0.001f;
    bool primary_is_edge; V2 primary_normal; int primary_index;
    if (p_sep - radius > k_relativeTol * (e_sep - radius) + k_absoluteTol) {
        primary_is_edge = false; primary_normal = p_normal; primary_index = p_index;
    } else {
        primary_is_edge = true; primary_normal = e_normal; primary_index = e_index;
    }

    ClipV clipPoints[2];
    int ref_i1, ref_i2; V2 ref_v1, ref_v2, ref_normal, ref_side1, ref_side2;
    if (primary_is_edge) {
        manifold.type = FACE_A;
        int bestIndex = 0;
        float bestValue = dot(primary_normal, tn[0]);
        for (int i = 1; i < tcount; ++i) {
            float value = dot(primary_normal, tn[i]);
            if (value < bestValue) { bestValue = value; bestIndex = i; }
        }
        int i1 = bestIndex, i2 = i1 + 1 < tcount ? i1 + 1 : 0;
        clipPoints[0].v = tv[i1];
        clipPoints[0].id.indexA = 0; clipPoints[0].id.indexB = (uint8_t)i1; clipPoints[0].id.typeA = CF_FACE; clipPoints[0].id.typeB = CF_VERTEX;
        clipPoints[1].v = tv[i2];
        clipPoints[1].id.indexA = 0; clipPoints[1].id.indexB = (uint8_t)i2; clipPoints[1].id.typeA = CF_FACE; clipPoints[1].id.typeB = CF_VERTEX;
        ref_i1 = 0; ref_i2 = 1; ref_v1 = v1; ref_v2 = v2; ref_normal = primary_normal;
        ref_side1 = -edge1; ref_side2 = edge1;
    } else {
        manifold.type = FACE_B;
        clipPoints[0].v = v2;
        clipPoints[0].id.indexA = 1; clipPoints[0].id.indexB = (uint8_t)primary_index; clipPoints[0].id.typeA = CF_VERTEX; clipPoints[0].id.typeB = CF_FACE;
        clipPoints[1].v = v1;
        clipPoints[1].id.indexA = 0; clipPoints[1].id.indexB = (uint8_t)primary_index; clipPoints[1].id.typeA = CF_VERTEX; clipPoints[1].id.typeB = CF_FACE;
        ref_i1 = primary_index;
        ref_i2 = ref_i1 + 1 < tcount ? ref_i1 + 1 : 0;
        ref_v1 = tv[ref_i1]; ref_v2 = tv[ref_i2]; ref_normal = tn[ref_i1];
        ref_side1 = mk(ref_normal.y, -ref_normal.x);
        ref_side2 = -ref_side1;
    }
    float sideOffset1 = dot(ref_side1, ref_v1);
    float sideOffset2 = dot(ref_side2, ref_v2);
    ClipV clipPoints1[2], clipPoints2[2];
    int np = clip_segment(clipPoints1, clipPoints, ref_side1, sideOffset1, ref_i1);
    if (np < 2) return;
    np = clip_segment(clipPoints2, clipPoints1, ref_side2, sideOffset2, ref_i2);
    if (np < 2) return;
    if (primary_is_edge) {
        manifold.localNormal = ref_normal;
        manifold.localPoint = ref_v1;
    } else {
        manifold.localNormal = tab.n[pi][ref_i1];
        manifold.localPoint = tab.v[pi][ref_i1];
    }
    int pointCount = 0;
    for (int i = 0; i < 2; ++i) {
        float separation = dot(ref_normal, clipPoints2[i].v - ref_v1);
        if (separation <= radius) {
            MPoint& cp = manifold.points[pointCount];
            if (primary_is_edge) {
                cp.localPoint = xmulT(xf, clipPoints2[i].v);
                cp.id = clipPoints2[i].id;
            } else {
                cp.localPoint = clipPoints2[i].v;
                cp.id.typeA = clipPoints2[i].id.typeB;
                cp.id.typeB = clipPoints2[i].id.typeA;
                cp.id.indexA = clipPoints2[i].id.indexB;
                cp.id.indexB = clipPoints2[i].id.indexA;
            }
            ++pointCount;
        }
    }
    manifold.pointCount = pointCount;
}

// ---- contact solver (body A = static ground, dropped) ---------------------------------------------------
struct VCPoint { V2 rB; float normalImpulse, tangentImpulse, normalMass, tangentMass, velocityBias; };
struct VConstraint {
    VCPoint points[2];
    V2 normal;
    float nm11, nm12, nm21, nm22;
    float k11, k12, k21, k22;
    int indexB;       // dynamic body 0..2
    float invMassB, invIB, friction;
    int pointCount, slot;
    int pad_;     // 31 words: odd record stride => conflict-free per-lane columns in LDS
};
struct PConstraint {
    V2 localPoints[2], localNormal, localPoint, localCenterB;
    int indexB, type, pointCount;
    float invMassB, invIB;
};

// b2WorldManifold::Initialize with xfA = identity, radiusA = radiusB = polygonRadius
LLD void world_manifold(const Manifold& manifold, Xf xfB, V2& normal, V2 points[2]) {
    const float radiusA = b2_polygonRadius, radiusB = b2_polygonRadius;
    if (manifold.pointCount == 0) return;
    if (manifold.type == FACE_A) {
        normal = manifold.localNormal;
        V2 planePoint = manifold.localPoint;
        for (int i = 0; i < manifold.pointCount; ++i) {
            V2 clipPoint = xmul(xfB, manifold.points[i].localPoint);
            V2 cA = clipPoint + (radiusA - dot(clipPoint - planePoint, normal)) * normal;
            V2 cB = clipPoint - radiusB * normal;
            points[i] = 0.5f * (cA + cB);
        }
    } else {
        normal = rmul(xfB.q, manifold.localNormal);
        V2 planePoint = xmul(xfB, manifold.localPoint);
        for (int i = 0; i < manifold.pointCount; ++i) {
            V2 clipPoint = manifold.points[i].localPoint;
            V2 cB = clipPoint + (radiusB - dot(clipPoint - planePoint, normal)) * normal;
            V2 cA = clipPoint - radiusA * normal;
            points[i] = 0.5f * (cA + cB);
        }
        normal = -normal;
    }
}

struct Pos { V2 c; float a; };
struct Vel { V2 v; float w; };

// ---- GJK (b2Distance), proxy A = edge (2 vertices, identity transform), proxy B = polygon ---------------
struct SCache { float metric; int count; uint8_t indexA[3], indexB[3]; };
struct SVertex { V2 wA, wB, w; float a; int indexA, indexB; };
// The simplex vertices are three named records, never an indexed array: a dynamically indexed local array lives in
// scratch memory on the GPU, and b2Distance sits inside the time-of-impact root finder.
struct Simplex { SVertex v0, v1, v2; int count; };

LLD V2 edge_vertex(const V2 ev[2], int i) {  // ev[i], i in {0, 1}, without indexing
    return mk(i ? ev[1].x : ev[0].x, i ? ev[1].y : ev[0].y);
}
LLD int support2(const V2 ev[2], V2 d) {  // b2DistanceProxy::GetSupport on the edge
    float v0 = dot(ev[0], d), v1 = dot(ev[1], d);
    return v1 > v0 ? 1 : 0;
}
LLD int support_poly(const PolyTab& tab, int pi, V2 d) {
    int bestIndex = 0;
    float bestValue = dot(tab.v[pi][0], d);
    for (int i = 1; i < tab.count[pi]; ++i) {
        float value = dot(tab.v[pi][i], d);
        if (value > bestValue) { bestIndex = i; bestValue = value; }
    }
    return bestIndex;
}
LLD float simplex_metric(const Simplex& s) {
    if (s.count == 2) return len(s.v0.w - s.v1.w);
    if (s.count == 3) return cross(s.v1.w - s.v0.w, s.v2.w - s.v0.w);
    return 0.0f;
}
LLD void simplex_solve2(Simplex& s) {
    V2 w1 = s.v0.w, w2 = s.v1.w, e12 = w2 - w1;
    float d12_2 = -dot(w1, e12);
    if (d12_2 <= 0.0f) { s.v0.a = 1.0f; s.count = 1; return; }
    float d12_1 = dot(w2, e12);
    if (d12_1 <= 0.0f) { s.v1.a = 1.0f; s.count = 1; s.v0 = s.v1; return; }
    float inv_d12 = 1.0f / (d12_1 + d12_2);
    s.v0.a = d12_1 * inv_d12; s.v1.a = d12_2 * inv_d12; s.count = 2;
}
LLD void simplex_solve3(Simplex& s) {
    V2 w1 = s.v0.w, w2 = s.v1.w, w3 = s.v2.w;
    V2 e12 = w2 - w1;
    float w1e12 = dot(w1, e12), w2e12 = dot(w2, e12);
    float d12_1 = w2e12, d12_2 = -w1e12;
    V2 e13 = w3 - w1;
    float w1e13 = dot(w1, e13), w3e13 = dot(w3, e13);
    float d13_1 = w3e13, d13_2 = -w1e13;
    V2 e23 = w3 - w2;
    float w2e23 = dot(w2, e23), w3e23 = dot(w3, e23);
    float d23_1 = w3e23, d23_2 = -w2e23;
    float n123 = cross(e12, e13);
    float d123_1 = n123 * cross(w2, w3), d123_2 = n123 * cross(w3, w1), d123_3 = n123 * cross(w1, w2);
    if (d12_2 <= 0.0f && d13_2 <= 0.0f) { s.v0.a = 1.0f; s.count = 1; return; }
    if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) {
        float inv = 1.0f / (d12_1 + d12_2);
        s.v0.a = d12_1 * inv; s.v1.a = d12_2 * inv; s.count = 2; return;
    }
    if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) {
        float inv = 1.0f / (d13_1 + d13_2);
        s.v0.a = d13_1 * inv; s.v2.a = d13_2 * inv; s.count = 2; s.v1 = s.v2; return;
    }
    if (d12_1 <= 0.0f && d23_2 <= 0.0f) { s.v1.a = 1.0f; s.count = 1; s.v0 = s.v1; return; }
    if (d13_1 <= 0.0f && d23_1 <= 0.0f) { s.v2.a = 1.0f; s.count = 1; s.v0 = s.v2; return; }
    if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) {
        float inv = 1.0f / (d23_1 + d23_2);
        s.v1.a = d23_1 * inv; s.v2.a = d23_2 * inv; s.count = 2; s.v0 = s.v2; return;
    }
    float inv = 1.0f / (d123_1 + d123_2 + d123_3);
    s.v0.a = d123_1 * inv; s.v1.a = d123_2 * inv; s.v2.a = d123_3 * inv; s.count = 3;
}

LLD void simplex_read(SVertex& v, int indexA, int indexB, const V2 ev[2], const PolyTab& tab, int pi, Xf xfB) {
    v.indexA = indexA; v.indexB = indexB;
    v.wA = edge_vertex(ev, indexA);
    v.wB = xmul(xfB, tab.v[pi][indexB]);
    v.w = v.wB - v.wA;
    v.a = 0.0f;
}

// b2Distance, useRadii = false
LLD float gjk_distance(SCache& cache, const V2 ev[2], const PolyTab& tab, int pi, Xf xfB) {
    Simplex sx;
    sx.count = cache.count;  // ReadCache
    if (sx.count > 0) simplex_read(sx.v0, cache.indexA[0], cache.indexB[0], ev, tab, pi, xfB);
    if (sx.count > 1) simplex_read(sx.v1, cache.indexA[1], cache.indexB[1], ev, tab, pi, xfB);
    if (sx.count > 2) simplex_read(sx.v2, cache.indexA[2], cache.indexB[2], ev, tab, pi, xfB);
    if (sx.count > 1) {
        float metric1 = cache.metric, metric2 = simplex_metric(sx);
        if (metric2 < 0.5f * metric1 || 2.0f * metric1 < metric2 || metric2 < b2_epsilon) sx.count = 0;
    }
    if (sx.count == 0) {
        simplex_read(sx.v0, 0, 0, ev, tab, pi, xfB);
        sx.v0.a = 1.0f;
        sx.count = 1;
    }
    const int k_maxIters = 20;
    int saveA0 = 0, saveA1 = 0, saveA2 = 0, saveB0 = 0, saveB1 = 0, saveB2 = 0, saveCount = 0;
    int iter = 0;
    while (iter < k_maxIters) {
        saveCount = sx.count;
        if (saveCount > 0) { saveA0 = sx.v0.indexA; saveB0 = sx.v0.indexB; }
        if (saveCount > 1) { saveA1 = sx.v1.indexA; saveB1 = sx.v1.indexB; }
        if (saveCount > 2) { saveA2 = sx.v2.indexA; saveB2 = sx.v2.indexB; }
        if (sx.count == 2) simplex_solve2(sx);
        else if (sx.count == 3) simplex_solve3(sx);
        if (sx.count == 3) break;
        V2 d;  // GetSearchDirection
        if (sx.count == 1) d = -sx.v0.w;
        else {
            V2 e12 = sx.v1.w - sx.v0.w;
            float sgn = cross(e12, -sx.v0.w);
            d = sgn > 0.0f ? cross_sv(1.0f, e12) : cross_vs(e12, 1.0f);
        }
        if (len2(d) < b2_epsilon * b2_epsilon) break;
        SVertex vertex;  // the new vertex is slot `count` (1 or 2 here)
        vertex.indexA = support2(ev, -d);
        vertex.wA = edge_vertex(ev, vertex.indexA);
        vertex.indexB = support_poly(tab, pi, rmulT(xfB.q, d));
        vertex.wB = xmul(xfB, tab.v[pi][vertex.indexB]);
        vertex.w = vertex.wB - vertex.wA;
        vertex.a = sx.count == 1 ? sx.v1.a : sx.v2.a;  // untouched field of the slot it replaces
        if (sx.count == 1) sx.v1 = vertex; else sx.v2 = vertex;
        ++iter;
        bool duplicate = false;
        if (saveCount > 0 && vertex.indexA == saveA0 && vertex.indexB == saveB0) duplicate = true;
        if (saveCount > 1 && vertex.indexA == saveA1 && vertex.indexB == saveB1) duplicate = true;
        if (saveCount > 2 && vertex.indexA == saveA2 && vertex.indexB == saveB2) duplicate = true;
        if (duplicate) break;
        ++sx.count;
    }
    V2 pointA, pointB;  // GetWitnessPoints
    if (sx.count == 1) { pointA = sx.v0.wA; pointB = sx.v0.wB; }
    else if (sx.count == 2) {
        pointA = sx.v0.a * sx.v0.wA + sx.v1.a * sx.v1.wA;
        pointB = sx.v0.a * sx.v0.wB + sx.v1.a * sx.v1.wB;
    } else {
        pointA = sx.v0.a * sx.v0.wA + sx.v1.a * sx.v1.wA + sx.v2.a * sx.v2.wA;
        pointB = pointA;
    }
    float distance = len(pointA - pointB);
    cache.metric = simplex_metric(sx);  // WriteCache
    cache.count = sx.count;
    if (sx.count > 0) { cache.indexA[0] = (uint8_t)sx.v0.indexA; cache.indexB[0] = (uint8_t)sx.v0.indexB; }
    if (sx.count > 1) { cache.indexA[1] = (uint8_t)sx.v1.indexA; cache.indexB[1] = (uint8_t)sx.v1.indexB; }
    if (sx.count > 2) { cache.indexA[2] = (uint8_t)sx.v2.indexA; cache.indexB[2] = (uint8_t)sx.v2.indexB; }
    return distance;
}

// ---- b2TimeOfImpact (A = static edge: its sweep transform is the identity at every t) -------------------
enum { SEP_POINTS, SEP_FACE_A, SEP_FACE_B };
struct SepFn { int type; V2 localPoint, axis; };

LLD void sep_init(SepFn& f, const SCache& cache, const V2 ev[2], const PolyTab& tab, int pi, const Sweep& sweepB, float t1) {
    Xf xfB = sweep_xf(sweepB, t1);
    if (cache.count == 1) {
        f.type = SEP_POINTS;
        V2 pointA = edge_vertex(ev, cache.indexA[0]);
        V2 pointB = xmul(xfB, tab.v[pi][cache.indexB[0]]);
        f.axis = pointB - pointA;
        normalize(f.axis);
    } else if (cache.indexA[0] == cache.indexA[1]) {
        f.type = SEP_FACE_B;
        V2 localPointB1 = tab.v[pi][cache.indexB[0]], localPointB2 = tab.v[pi][cache.indexB[1]];
        f.axis = cross_vs(localPointB2 - localPointB1, 1.0f);
        normalize(f.axis);
        V2 normal = rmul(xfB.q, f.axis);
        f.localPoint = 0.5f * (localPointB1 + localPointB2);
        V2 pointB = xmul(xfB, f.localPoint);
        V2 pointA = edge_vertex(ev, cache.indexA[0]);
        float s = dot(pointA - pointB, normal);
        if (s < 0.0f) f.axis = -f.axis;
    } else {
        f.type = SEP_FACE_A;
        V2 localPointA1 = edge_vertex(ev, cache.indexA[0]), localPointA2 = edge_vertex(ev, cache.indexA[1]);
        f.axis = cross_vs(localPointA2 - localPointA1, 1.0f);
        normalize(f.axis);
        V2 normal = f.axis;
        f.localPoint = 0.5f * (localPointA1 + localPointA2);
        V2 pointA = f.localPoint;
        V2 pointB = xmul(xfB, tab.v[pi][cache.indexB[0]]);
        float s = dot(pointB - pointA, normal);
        if (s < 0.0f) f.axis = -f.axis;
    }
}
LLD float sep_find_min(const SepFn& f, const V2 ev[2], const PolyTab& tab, int pi, const Sweep& sweepB, int& indexA, int& indexB, float t) {
    Xf xfB = sweep_xf(sweepB, t);
    if (f.type == SEP_POINTS) {
        V2 axisA = f.axis, axisB = rmulT(xfB.q, -f.axis);
        indexA = support2(ev, axisA);
        indexB = support_poly(tab, pi, axisB);
        V2 pointA = edge_vertex(ev, indexA), pointB = xmul(xfB, tab.v[pi][indexB]);
        return dot(pointB - pointA, f.axis);
    } else if (f.type == SEP_FACE_A) {
        V2 normal = f.axis;
        V2 pointA = f.localPoint;
        V2 axisB = rmulT(xfB.q, -normal);
        indexA = -1;
        indexB = support_poly(tab, pi, axisB);
        V2 pointB = xmul(xfB, tab.v[pi][indexB]);
        return dot(pointB - pointA, normal);
    } else {
        V2 normal = rmul(xfB.q, f.axis);
        V2 pointB = xmul(xfB, f.localPoint);
        V2 axisA = -normal;
        indexB = -1;
        indexA = support2(ev, axisA);
        V2 pointA = edge_vertex(ev, indexA);
        return dot(pointA - pointB, normal);
    }
}
LLD float sep_evaluate(const SepFn& f, const V2 ev[2], const PolyTab& tab, int pi, const Sweep& sweepB, int indexA, int indexB, float t) {
    Xf xfB = sweep_xf(sweepB, t);
    if (f.type == SEP_POINTS) {
        V2 pointA = edge_vertex(ev, indexA), pointB = xmul(xfB, tab.v[pi][indexB]);
        return dot(pointB - pointA, f.axis);
    } else if (f.type == SEP_FACE_A) {
        V2 normal = f.axis;
        V2 pointA = f.localPoint;
        V2 pointB = xmul(xfB, tab.v[pi][indexB]);
        return dot(pointB - pointA, normal);
    } else {
        V2 normal = rmul(xfB.q, f.axis);
        V2 pointB = xmul(xfB, f.localPoint);
        V2 pointA = edge_vertex(ev, indexA);
        return dot(pointA - pointB, normal);
    }
}

enum { TOI_UNKNOWN, TOI_FAILED, TOI_OVERLAPPED, TOI_TOUCHING, TOI_SEPARATED };

LLD int time_of_impact(const V2 ev[2], const PolyTab& tab, int pi, Sweep sweepB, float& t_out) {
    int state = TOI_UNKNOWN;
    const float tMax = 1.0f;
    t_out = tMax;
    sweep_normalize(sweepB);
    const float totalRadius = b2_polygonRadius + b2_polygonRadius;
    const float target = fmax2(b2_linearSlop, totalRadius - 3.0f * b2_linearSlop);
    const float tolerance = 0.25f * b2_linearSlop;
    float t1 = 0.0f;
    const int k_maxIterations = 20;
    int iter = 0;
    SCache cache;
    cache.count = 0; cache.metric = 0.0f;
    for (;;) {
        Xf xfB = sweep_xf(sweepB, t1);
        float distance = gjk_distance(cache, ev, tab, pi, xfB);
        if (distance <= 0.0f) { state = TOI_OVERLAPPED; t_out = 0.0f; break; }
        if (distance < target + tolerance) { state = TOI_TOUCHING; t_out = t1; break; }
        SepFn fcn;
        sep_init(fcn, cache, ev, tab, pi, sweepB, t1);
        bool done = false;
        float t2 = tMax;
        int pushBackIter = 0;
        for (;;) {
            int indexA, indexB;
            float s2 = sep_find_min(fcn, ev, tab, pi, sweepB, indexA, indexB, t2);
            if (s2 > target + tolerance) { state = TOI_SEPARATED; t_out = tMax; done = true; break; }
            if (s2 > target - tolerance) { t1 = t2; break; }
            float s1 = sep_evaluate(fcn, ev, tab, pi, sweepB, indexA, indexB, t1);
            if (s1 < target - tolerance) { state = TOI_FAILED; t_out = t1; done = true; break; }
            if (s1 <= target + tolerance) { state = TOI_TOUCHING; t_out = t1; done = true; break; }
            int rootIterCount = 0;
            float a1 = t1, a2 = t2;
            for (;;) {
                float t;
                if (rootIterCount & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);
                else t = 0.5f * (a1 + a2);
                ++rootIterCount;
                float s = sep_evaluate(fcn, ev, tab, pi, sweepB, indexA, indexB, t);
                if (fabs1(s - target) < tolerance) { t2 = t; break; }
                if (s > target) { a1 = t; s1 = s; } else { a2 = t; s2 = s; }
                if (rootIterCount == 50) break;
            }
            ++pushBackIter;
            if (pushBackIter == 8) break;  // b2_maxPolygonVertices
        }
        ++iter;
        if (done) break;
        if (iter == k_maxIterations) { state = TOI_FAILED; t_out = t1; break; }
    }
    return state;
}

}  // namespace ll
}  // namespace mgym
