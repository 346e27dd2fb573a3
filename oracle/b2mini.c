/* b2mini.c — see b2mini.h.  Restatement of the published Box2D v2.4.1 algorithms (Erin Catto,
 * zlib licence) for the LunarLander scene; each function names the Box2D routine it follows.
 * TEST INFRASTRUCTURE ONLY. */
#include "b2mini.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ constants (b2_common.h) */
#define b2_pi 3.14159265359f
#define b2_linearSlop 0.005f
#define b2_angularSlop (2.0f / 180.0f * b2_pi)
#define b2_polygonRadius (2.0f * b2_linearSlop)
#define b2_aabbExtension 0.1f
#define b2_aabbMultiplier 4.0f
#define b2_maxLinearCorrection 0.2f
#define b2_maxAngularCorrection (8.0f / 180.0f * b2_pi)
#define b2_maxTranslation 2.0f
#define b2_maxTranslationSquared (b2_maxTranslation * b2_maxTranslation)
#define b2_maxRotation (0.5f * b2_pi)
#define b2_maxRotationSquared (b2_maxRotation * b2_maxRotation)
#define b2_baumgarte 0.2f
#define b2_toiBaumgarte 0.75f
#define b2_timeToSleep 0.5f
#define b2_linearSleepTolerance 0.01f
#define b2_angularSleepTolerance (2.0f / 180.0f * b2_pi)
#define b2_maxSubSteps 8
#define b2_maxTOIContacts 32
#define b2_epsilon FLT_EPSILON
#define b2_maxFloat FLT_MAX

/* ------------------------------------------------------------------------- b2_math.h */
/* Optional work statistics of the contact path (tools/ll_work_stats.c builds this file with -DORA_STATS);
 * never compiled into liboracle.so. */
#ifdef ORA_STATS
typedef struct {
    long steps, steps_with_contacts;
    long hist_contacts[16], hist_touching[16], hist_pos_iters[64];
    long toi_calls, toi_outer_iters, toi_root_iters, toi_pushback, gjk_calls, gjk_iters;
    long hist_toi_calls[64], hist_substeps[32], hist_rejected[32];
    long toi_island_contacts[16], toi_pos_iters;
    int cur_toi_calls, cur_substeps, cur_rejected;
} ora_b2_stats;
ora_b2_stats g_b2_stats;
#define STAT(x) do { x; } while (0)
#else
#define STAT(x) do { } while (0)
#endif

static b2v V(float x, float y) { b2v r = {x, y}; return r; }
static b2v vadd(b2v a, b2v b) { return V(a.x + b.x, a.y + b.y); }
static b2v vsub(b2v a, b2v b) { return V(a.x - b.x, a.y - b.y); }
static b2v vneg(b2v a) { return V(-a.x, -a.y); }
static b2v smul(float s, b2v a) { return V(s * a.x, s * a.y); }
static float dot(b2v a, b2v b) { return a.x * b.x + a.y * b.y; }
static float cross(b2v a, b2v b) { return a.x * b.y - a.y * b.x; }
static b2v cross_vs(b2v a, float s) { return V(s * a.y, -s * a.x); }
static b2v cross_sv(float s, b2v a) { return V(-s * a.y, s * a.x); }
static float vlen(b2v a) { return sqrtf(a.x * a.x + a.y * a.y); }
static float vlen2(b2v a) { return a.x * a.x + a.y * a.y; }
static float normalize(b2v *a) {
    float length = vlen(*a);
    if (length < b2_epsilon) return 0.0f;
    float inv = 1.0f / length;
    a->x *= inv; a->y *= inv;
    return length;
}
static float fmin2(float a, float b) { return a < b ? a : b; }
static float fmax2(float a, float b) { return a > b ? a : b; }
static float fclamp(float a, float lo, float hi) { return fmax2(lo, fmin2(a, hi)); }
static float fabs1(float a) { return a > 0.0f ? a : -a; }
static b2rot rot_set(float angle) { b2rot q; q.s = sinf(angle); q.c = cosf(angle); return q; }
static b2v rmul(b2rot q, b2v v) { return V(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
static b2v rmulT(b2rot q, b2v v) { return V(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
static b2v xmul(b2xf T, b2v v) {
    float x = (T.q.c * v.x - T.q.s * v.y) + T.p.x;
    float y = (T.q.s * v.x + T.q.c * v.y) + T.p.y;
    return V(x, y);
}
static b2xf xmulT_xf(b2xf A, b2xf B) { /* b2MulT(A, B) */
    b2xf C;
    C.q.s = A.q.c * B.q.s - A.q.s * B.q.c;
    C.q.c = A.q.c * B.q.c + A.q.s * B.q.s;
    C.p = rmulT(A.q, vsub(B.p, A.p));
    return C;
}
static b2v xmulT(b2xf T, b2v v) {
    float px = v.x - T.p.x, py = v.y - T.p.y;
    return V(T.q.c * px + T.q.s * py, -T.q.s * px + T.q.c * py);
}
/* b2Mat22::Solve */
static b2v mat22_solve(float a11, float a12, float a21, float a22, b2v b) {
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    return V(det * (a22 * b.x - a12 * b.y), det * (a11 * b.y - a21 * b.x));
}
/* b2Sweep::GetTransform / Advance / Normalize */
static b2xf sweep_xf(const b2sweep *s, float beta) {
    b2xf xf;
    xf.p = vadd(smul(1.0f - beta, s->c0), smul(beta, s->c));
    float angle = (1.0f - beta) * s->a0 + beta * s->a;
    xf.q = rot_set(angle);
    xf.p = vsub(xf.p, rmul(xf.q, s->localCenter));
    return xf;
}
static void sweep_advance(b2sweep *s, float alpha) {
    float beta = (alpha - s->alpha0) / (1.0f - s->alpha0);
    s->c0 = vadd(s->c0, smul(beta, vsub(s->c, s->c0)));
    s->a0 += beta * (s->a - s->a0);
    s->alpha0 = alpha;
}
static void sweep_normalize(b2sweep *s) {
    float twoPi = 2.0f * b2_pi;
    float d = twoPi * floorf(s->a0 / twoPi);
    s->a0 -= d;
    s->a -= d;
}

/* ------------------------------------------------------------------------- shapes */
/* b2PolygonShape::ComputeCentroid (v2.4.1: reference point = first vertex) */
static b2v poly_centroid(const b2v *vs, int count) {
    b2v c = V(0.0f, 0.0f);
    float area = 0.0f;
    b2v s = vs[0];
    const float inv3 = 1.0f / 3.0f;
    for (int i = 0; i < count; ++i) {
        b2v p1 = vsub(vs[0], s);
        b2v p2 = vsub(vs[i], s);
        b2v p3 = i + 1 < count ? vsub(vs[i + 1], s) : vsub(vs[0], s);
        b2v e1 = vsub(p2, p1), e2 = vsub(p3, p1);
        float D = cross(e1, e2);
        float triangleArea = 0.5f * D;
        area += triangleArea;
        c = vadd(c, smul(triangleArea * inv3, vadd(vadd(p1, p2), p3)));
    }
    c = vadd(smul(1.0f / area, c), s);
    return c;
}

/* b2PolygonShape::Set — weld, gift-wrap hull, normals, centroid */
static void poly_set(b2poly *p, const b2v *vertices, int count) {
    int n = count < B2_MAX_POLY ? count : B2_MAX_POLY;
    b2v ps[B2_MAX_POLY];
    int tempCount = 0;
    for (int i = 0; i < n; ++i) {
        b2v v = vertices[i];
        int unique = 1;
        for (int j = 0; j < tempCount; ++j) {
            b2v d = vsub(v, ps[j]);
            if (vlen2(d) < ((0.5f * b2_linearSlop) * (0.5f * b2_linearSlop))) { unique = 0; break; }
        }
        if (unique) ps[tempCount++] = v;
    }
    n = tempCount;
    int i0 = 0;
    float x0 = ps[0].x;
    for (int i = 1; i < n; ++i) {
        float x = ps[i].x;
        if (x > x0 || (x == x0 && ps[i].y < ps[i0].y)) { i0 = i; x0 = x; }
    }
    int hull[B2_MAX_POLY];
    int m = 0, ih = i0;
    for (;;) {
        hull[m] = ih;
        int ie = 0;
        for (int j = 1; j < n; ++j) {
            if (ie == ih) { ie = j; continue; }
            b2v r = vsub(ps[ie], ps[hull[m]]);
            b2v v = vsub(ps[j], ps[hull[m]]);
            float c = cross(r, v);
            if (c < 0.0f) ie = j;
            if (c == 0.0f && vlen2(v) > vlen2(r)) ie = j;
        }
        ++m;
        ih = ie;
        if (ie == i0) break;
    }
    p->count = m;
    for (int i = 0; i < m; ++i) p->v[i] = ps[hull[i]];
    for (int i = 0; i < m; ++i) {
        int i2 = i + 1 < m ? i + 1 : 0;
        b2v edge = vsub(p->v[i2], p->v[i]);
        p->n[i] = cross_vs(edge, 1.0f);
        normalize(&p->n[i]);
    }
    p->centroid = poly_centroid(p->v, m);
    p->radius = b2_polygonRadius;
}

/* b2PolygonShape::SetAsBox */
static void poly_set_box(b2poly *p, float hx, float hy) {
    p->count = 4;
    p->v[0] = V(-hx, -hy); p->v[1] = V(hx, -hy); p->v[2] = V(hx, hy); p->v[3] = V(-hx, hy);
    p->n[0] = V(0.0f, -1.0f); p->n[1] = V(1.0f, 0.0f); p->n[2] = V(0.0f, 1.0f); p->n[3] = V(-1.0f, 0.0f);
    p->centroid = V(0.0f, 0.0f);
    p->radius = b2_polygonRadius;
}

/* b2PolygonShape::ComputeMass */
static void poly_mass(const b2poly *p, float density, float *mass, b2v *center_out, float *I_out) {
    b2v center = V(0.0f, 0.0f);
    float area = 0.0f, I = 0.0f;
    b2v s = p->v[0];
    const float k_inv3 = 1.0f / 3.0f;
    for (int i = 0; i < p->count; ++i) {
        b2v e1 = vsub(p->v[i], s);
        b2v e2 = i + 1 < p->count ? vsub(p->v[i + 1], s) : vsub(p->v[0], s);
        float D = cross(e1, e2);
        float triangleArea = 0.5f * D;
        area += triangleArea;
        center = vadd(center, smul(triangleArea * k_inv3, vadd(e1, e2)));
        float ex1 = e1.x, ey1 = e1.y, ex2 = e2.x, ey2 = e2.y;
        float intx2 = ex1 * ex1 + ex2 * ex1 + ex2 * ex2;
        float inty2 = ey1 * ey1 + ey2 * ey1 + ey2 * ey2;
        I += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    *mass = density * area;
    center = smul(1.0f / area, center);
    *center_out = vadd(center, s);
    *I_out = density * I;
    *I_out += *mass * (dot(*center_out, *center_out) - dot(center, center));
}

/* b2PolygonShape::ComputeAABB / b2EdgeShape::ComputeAABB */
static b2aabb poly_aabb(const b2poly *p, b2xf xf) {
    b2v lower = xmul(xf, p->v[0]), upper = lower;
    for (int i = 1; i < p->count; ++i) {
        b2v v = xmul(xf, p->v[i]);
        lower = V(fmin2(lower.x, v.x), fmin2(lower.y, v.y));
        upper = V(fmax2(upper.x, v.x), fmax2(upper.y, v.y));
    }
    b2aabb a;
    a.lo = V(lower.x - p->radius, lower.y - p->radius);
    a.hi = V(upper.x + p->radius, upper.y + p->radius);
    return a;
}
static b2aabb edge_aabb(const b2edge *e) { /* ground transform is identity */
    b2v v1 = e->v1, v2 = e->v2;
    b2v lower = V(fmin2(v1.x, v2.x), fmin2(v1.y, v2.y)), upper = V(fmax2(v1.x, v2.x), fmax2(v1.y, v2.y));
    b2aabb a;
    a.lo = V(lower.x - e->radius, lower.y - e->radius);
    a.hi = V(upper.x + e->radius, upper.y + e->radius);
    return a;
}
static b2aabb aabb_combine(b2aabb a, b2aabb b) {
    b2aabb r;
    r.lo = V(fmin2(a.lo.x, b.lo.x), fmin2(a.lo.y, b.lo.y));
    r.hi = V(fmax2(a.hi.x, b.hi.x), fmax2(a.hi.y, b.hi.y));
    return r;
}
static int aabb_contains(b2aabb a, b2aabb b) { /* a.Contains(b) */
    int r = 1;
    r = r && a.lo.x <= b.lo.x; r = r && a.lo.y <= b.lo.y;
    r = r && b.hi.x <= a.hi.x; r = r && b.hi.y <= a.hi.y;
    return r;
}
static int aabb_overlap(b2aabb a, b2aabb b) { /* b2TestOverlap */
    b2v d1 = vsub(b.lo, a.hi), d2 = vsub(a.lo, b.hi);
    if (d1.x > 0.0f || d1.y > 0.0f) return 0;
    if (d2.x > 0.0f || d2.y > 0.0f) return 0;
    return 1;
}

/* ------------------------------------------------------------------------- bodies */
static void body_sync_transform(b2body *b) { /* b2Body::SynchronizeTransform */
    b->xf.q = rot_set(b->sw.a);
    b->xf.p = vsub(b->sw.c, rmul(b->xf.q, b->sw.localCenter));
}

void b2b_set_awake(b2world *w, int bi, int flag) { /* b2Body::SetAwake */
    b2body *b = &w->bodies[bi];
    if (b->type == B2_STATIC) return;
    if (flag) {
        b->awake = 1; b->sleepTime = 0.0f;
    } else {
        b->awake = 0; b->sleepTime = 0.0f;
        b->linvel = V(0.0f, 0.0f); b->angvel = 0.0f; b->force = V(0.0f, 0.0f); b->torque = 0.0f;
    }
}

/* ------------------------------------------------------------------------- broad phase */
static void buffer_move(b2world *w, int proxy) {
    if (w->move_count < 64) w->move_buffer[w->move_count++] = proxy;
}
static void create_proxy(b2world *w, int proxy, b2aabb aabb) { /* b2DynamicTree::CreateProxy + BufferMove */
    w->fat[proxy].lo = V(aabb.lo.x - b2_aabbExtension, aabb.lo.y - b2_aabbExtension);
    w->fat[proxy].hi = V(aabb.hi.x + b2_aabbExtension, aabb.hi.y + b2_aabbExtension);
    w->proxy_exists[proxy] = 1;
    w->moved[proxy] = 1;
    buffer_move(w, proxy);
}
/* b2DynamicTree::MoveProxy (v2.4.x) */
static void move_proxy(b2world *w, int proxy, b2aabb aabb, b2v displacement) {
    b2aabb fatAABB;
    fatAABB.lo = V(aabb.lo.x - b2_aabbExtension, aabb.lo.y - b2_aabbExtension);
    fatAABB.hi = V(aabb.hi.x + b2_aabbExtension, aabb.hi.y + b2_aabbExtension);
    b2v d = smul(b2_aabbMultiplier, displacement);
    if (d.x < 0.0f) fatAABB.lo.x += d.x; else fatAABB.hi.x += d.x;
    if (d.y < 0.0f) fatAABB.lo.y += d.y; else fatAABB.hi.y += d.y;
    b2aabb treeAABB = w->fat[proxy];
    if (aabb_contains(treeAABB, aabb)) {
        b2aabb huge;
        huge.lo = V(fatAABB.lo.x - 4.0f * b2_aabbExtension, fatAABB.lo.y - 4.0f * b2_aabbExtension);
        huge.hi = V(fatAABB.hi.x + 4.0f * b2_aabbExtension, fatAABB.hi.y + 4.0f * b2_aabbExtension);
        if (aabb_contains(huge, treeAABB)) return;
    }
    w->fat[proxy] = fatAABB;
    w->moved[proxy] = 1;
    buffer_move(w, proxy);
}
/* b2Fixture::Synchronize for the polygon fixture of dynamic body bi */
static void fixture_sync(b2world *w, int bi, b2xf xf1, b2xf xf2) {
    const b2poly *p = &w->polys[bi - 1];
    b2aabb a1 = poly_aabb(p, xf1), a2 = poly_aabb(p, xf2);
    b2aabb aabb = aabb_combine(a1, a2);
    b2v c1 = smul(0.5f, vadd(a1.lo, a1.hi)), c2 = smul(0.5f, vadd(a2.lo, a2.hi));
    move_proxy(w, 10 + bi, aabb, vsub(c2, c1));
}
static void body_sync_fixtures(b2world *w, int bi) { /* b2Body::SynchronizeFixtures */
    b2body *b = &w->bodies[bi];
    if (b->awake) {
        b2xf xf1;
        xf1.q = rot_set(b->sw.a0);
        xf1.p = vsub(b->sw.c0, rmul(xf1.q, b->sw.localCenter));
        fixture_sync(w, bi, xf1, b->xf);
    } else {
        fixture_sync(w, bi, b->xf, b->xf);
    }
}

static int contact_index(int body, int edge) { return (body - 1) * B2_N_EDGES + edge; }

/* b2ContactManager::AddPair for (ground edge, dynamic polygon) */
static void add_pair(b2world *w, int edge, int body) {
    b2contact *c = &w->contacts[contact_index(body, edge)];
    if (c->exists) return;
    memset(c, 0, sizeof(*c));
    c->exists = 1; c->enabled = 1;
    c->seq = w->next_seq++;
    c->body = body; c->edge = edge;
    c->friction = sqrtf(w->edge_friction * w->poly_friction[body - 1]); /* b2MixFriction */
    c->restitution = 0.0f;            /* b2MixRestitution = max(0, 0) */
    c->restitutionThreshold = 1.0f;   /* b2_velocityThreshold default */
    c->toi = 1.0f;
}

/* b2BroadPhase::UpdatePairs + b2ContactManager::FindNewContacts.  Brute-force partner search
 * (ascending proxy id) replaces the dynamic-tree query; filtering: ground (cat 1, mask 0xFFFF)
 * collides with lander (0x10/0x1) and legs (0x20/0x1); lander/legs never collide with each other. */
/* What-if switch for tools/ll_contact_order_probe.py ONLY (quantifies the one documented deviation): Box2D's order of
 * newly created contacts depends on dynamic-tree traversal; this restatement (and the kernels) use "moved proxies in body-list order,
 * partner edges by ascending id".  bit 0: partners by DESCENDING id; bit 1: moved proxies in reverse order.  0 = the product's order. */
int b2mini_order_variant = 0;
static void find_new_contacts(b2world *w) {
    for (int ii = 0; ii < w->move_count; ++ii) {
        int i = (b2mini_order_variant & 2) ? w->move_count - 1 - ii : ii;
        int q = w->move_buffer[i];
        for (int pp = 0; pp < B2_N_PROXIES; ++pp) {
            int p = (b2mini_order_variant & 1) ? B2_N_PROXIES - 1 - pp : pp;
            if (!w->proxy_exists[p] || p == q) continue;
            if (!aabb_overlap(w->fat[q], w->fat[p])) continue;
            if (w->moved[p] && p > q) continue; /* both moving: avoid duplicate pairs */
            int lo = p < q ? p : q, hi = p < q ? q : p;
            if (lo <= 10 && hi >= 11) add_pair(w, lo, hi - 10);
        }
    }
    for (int i = 0; i < w->move_count; ++i) w->moved[w->move_buffer[i]] = 0;
    w->move_count = 0;
}

/* ------------------------------------------------------------------------- collision */
typedef struct { b2v v; b2cf id; } clipv;
enum { CF_VERTEX = 0, CF_FACE = 1 };

/* b2ClipSegmentToLine */
static int clip_segment(clipv vOut[2], const clipv vIn[2], b2v normal, float offset, int vertexIndexA) {
    int count = 0;
    float distance0 = dot(normal, vIn[0].v) - offset;
    float distance1 = dot(normal, vIn[1].v) - offset;
    if (distance0 <= 0.0f) vOut[count++] = vIn[0];
    if (distance1 <= 0.0f) vOut[count++] = vIn[1];
    if (distance0 * distance1 < 0.0f) {
        float interp = distance0 / (distance0 - distance1);
        vOut[count].v = vadd(vIn[0].v, smul(interp, vsub(vIn[1].v, vIn[0].v)));
        vOut[count].id.indexA = (uint8_t)vertexIndexA;
        vOut[count].id.indexB = vIn[0].id.indexB;
        vOut[count].id.typeA = CF_VERTEX;
        vOut[count].id.typeB = CF_FACE;
        ++count;
    }
    return count;
}

/* b2CollideEdgeAndPolygon (v2.4.1), two-sided edge; xfA = identity (static ground at origin) */
static void collide_edge_polygon(b2manifold *manifold, const b2edge *edgeA, b2xf xfA, const b2poly *polygonB, b2xf xfB) {
    manifold->pointCount = 0;
    b2xf xf = xmulT_xf(xfA, xfB);
    b2v v1 = edgeA->v1, v2 = edgeA->v2;
    b2v edge1 = vsub(v2, v1);
    normalize(&edge1);
    b2v normal1 = V(edge1.y, -edge1.x);

    b2v tv[B2_MAX_POLY], tn[B2_MAX_POLY];
    int tcount = polygonB->count;
    for (int i = 0; i < tcount; ++i) {
        tv[i] = xmul(xf, polygonB->v[i]);
        tn[i] = rmul(xf.q, polygonB->n[i]);
    }
    float radius = polygonB->radius + edgeA->radius;

    /* b2ComputeEdgeSeparation */
    int e_index = -1; float e_sep = -FLT_MAX; b2v e_normal = V(0.0f, 0.0f);
    {
        b2v axes[2]; axes[0] = normal1; axes[1] = vneg(normal1);
        for (int j = 0; j < 2; ++j) {
            float sj = FLT_MAX;
            for (int i = 0; i < tcount; ++i) {
                float si = dot(axes[j], vsub(tv[i], v1));
                if (si < sj) sj = si;
            }
            if (sj > e_sep) { e_index = j; e_sep = sj; e_normal = axes[j]; }
        }
    }
    (void)e_index;
    if (e_sep > radius) return;

    /* b2ComputePolygonSeparation */
    int p_index = -1; float p_sep = -FLT_MAX; b2v p_normal = V(0.0f, 0.0f); int p_valid = 0;
    for (int i = 0; i < tcount; ++i) {
        b2v n = vneg(tn[i]);
        float s1 = dot(n, vsub(tv[i], v1));
        float s2 = dot(n, vsub(tv[i], v2));
        float s = fmin2(s1, s2);
        if (s > p_sep) { p_valid = 1; p_index = i; p_sep = s; p_normal = n; }
    }
    if (p_sep > radius) return;

    const float k_relativeTol = 0.98f, k_absoluteTol = 0.001f;
    int primary_is_edge;
    b2v primary_normal; int primary_index;
    if (p_sep - radius > k_relativeTol * (e_sep - radius) + k_absoluteTol) {
        primary_is_edge = 0; primary_normal = p_normal; primary_index = p_index;
    } else {
        primary_is_edge = 1; primary_normal = e_normal; primary_index = e_index;
    }
    (void)p_valid;

    clipv clipPoints[2];
    int ref_i1, ref_i2; b2v ref_v1, ref_v2, ref_normal, ref_side1, ref_side2;
    if (primary_is_edge) {
        manifold->type = B2_FACE_A;
        int bestIndex = 0;
        float bestValue = dot(primary_normal, tn[0]);
        for (int i = 1; i < tcount; ++i) {
            float value = dot(primary_normal, tn[i]);
            if (value < bestValue) { bestValue = value; bestIndex = i; }
        }
        int i1 = bestIndex, i2 = i1 + 1 < tcount ? i1 + 1 : 0;
        clipPoints[0].v = tv[i1];
        clipPoints[0].id.indexA = 0; clipPoints[0].id.indexB = (uint8_t)i1;
        clipPoints[0].id.typeA = CF_FACE; clipPoints[0].id.typeB = CF_VERTEX;
        clipPoints[1].v = tv[i2];
        clipPoints[1].id.indexA = 0; clipPoints[1].id.indexB = (uint8_t)i2;
        clipPoints[1].id.typeA = CF_FACE; clipPoints[1].id.typeB = CF_VERTEX;
        ref_i1 = 0; ref_i2 = 1; ref_v1 = v1; ref_v2 = v2; ref_normal = primary_normal;
        ref_side1 = vneg(edge1); ref_side2 = edge1;
    } else {
        manifold->type = B2_FACE_B;
        clipPoints[0].v = v2;
        clipPoints[0].id.indexA = 1; clipPoints[0].id.indexB = (uint8_t)primary_index;
        clipPoints[0].id.typeA = CF_VERTEX; clipPoints[0].id.typeB = CF_FACE;
        clipPoints[1].v = v1;
        clipPoints[1].id.indexA = 0; clipPoints[1].id.indexB = (uint8_t)primary_index;
        clipPoints[1].id.typeA = CF_VERTEX; clipPoints[1].id.typeB = CF_FACE;
        ref_i1 = primary_index;
        ref_i2 = ref_i1 + 1 < tcount ? ref_i1 + 1 : 0;
        ref_v1 = tv[ref_i1]; ref_v2 = tv[ref_i2]; ref_normal = tn[ref_i1];
        ref_side1 = V(ref_normal.y, -ref_normal.x);
        ref_side2 = vneg(ref_side1);
    }
    float sideOffset1 = dot(ref_side1, ref_v1);
    float sideOffset2 = dot(ref_side2, ref_v2);

    clipv clipPoints1[2], clipPoints2[2];
    int np = clip_segment(clipPoints1, clipPoints, ref_side1, sideOffset1, ref_i1);
    if (np < 2) return;
    np = clip_segment(clipPoints2, clipPoints1, ref_side2, sideOffset2, ref_i2);
    if (np < 2) return;

    if (primary_is_edge) {
        manifold->localNormal = ref_normal;
        manifold->localPoint = ref_v1;
    } else {
        manifold->localNormal = polygonB->n[ref_i1];
        manifold->localPoint = polygonB->v[ref_i1];
    }
    int pointCount = 0;
    for (int i = 0; i < 2; ++i) {
        float separation = dot(ref_normal, vsub(clipPoints2[i].v, ref_v1));
        if (separation <= radius) {
            b2mpoint *cp = &manifold->points[pointCount];
            if (primary_is_edge) {
                cp->localPoint = xmulT(xf, clipPoints2[i].v);
                cp->id = clipPoints2[i].id;
            } else {
                cp->localPoint = clipPoints2[i].v;
                cp->id.typeA = clipPoints2[i].id.typeB;
                cp->id.typeB = clipPoints2[i].id.typeA;
                cp->id.indexA = clipPoints2[i].id.indexB;
                cp->id.indexB = clipPoints2[i].id.indexA;
            }
            ++pointCount;
        }
    }
    manifold->pointCount = pointCount;
}

static int cf_equal(b2cf a, b2cf b) { return a.indexA == b.indexA && a.indexB == b.indexB && a.typeA == b.typeA && a.typeB == b.typeB; }

/* b2Contact::Update */
static void contact_update(b2world *w, b2contact *c) {
    b2manifold oldManifold = c->m;
    c->enabled = 1;
    int wasTouching = c->touching;
    b2body *bodyA = &w->bodies[0], *bodyB = &w->bodies[c->body];
    collide_edge_polygon(&c->m, &w->edges[c->edge], bodyA->xf, &w->polys[c->body - 1], bodyB->xf);
    int touching = c->m.pointCount > 0;
    for (int i = 0; i < c->m.pointCount; ++i) {
        b2mpoint *mp2 = &c->m.points[i];
        mp2->normalImpulse = 0.0f;
        mp2->tangentImpulse = 0.0f;
        for (int j = 0; j < oldManifold.pointCount; ++j) {
            const b2mpoint *mp1 = &oldManifold.points[j];
            if (cf_equal(mp1->id, mp2->id)) {
                mp2->normalImpulse = mp1->normalImpulse;
                mp2->tangentImpulse = mp1->tangentImpulse;
                break;
            }
        }
    }
    if (touching != wasTouching) {
        b2b_set_awake(w, 0, 1);
        b2b_set_awake(w, c->body, 1);
    }
    c->touching = touching;
    if (!wasTouching && touching && w->listener.begin) w->listener.begin(w->listener.ud, bodyA->userdata, bodyB->userdata);
    if (wasTouching && !touching && w->listener.end) w->listener.end(w->listener.ud, bodyA->userdata, bodyB->userdata);
}

/* world contact list order: newest first (b2ContactManager::AddPair inserts at the head) */
static int contact_order(const b2world *w, int *order) {
    int n = 0;
    for (int i = 0; i < B2_N_CONTACTS; ++i) if (w->contacts[i].exists) order[n++] = i;
    for (int i = 1; i < n; ++i) { /* insertion sort by seq descending */
        int k = order[i], j = i - 1;
        while (j >= 0 && w->contacts[order[j]].seq < w->contacts[k].seq) { order[j + 1] = order[j]; --j; }
        order[j + 1] = k;
    }
    return n;
}

static void contact_destroy(b2world *w, b2contact *c) { /* b2ContactManager::Destroy + b2Contact::Destroy */
    if (c->touching && w->listener.end) w->listener.end(w->listener.ud, w->bodies[0].userdata, w->bodies[c->body].userdata);
    if (c->m.pointCount > 0) { b2b_set_awake(w, 0, 1); b2b_set_awake(w, c->body, 1); }
    c->exists = 0;
}

/* b2ContactManager::Collide */
static void collide(b2world *w) {
    int order[B2_N_CONTACTS];
    int n = contact_order(w, order);
    for (int k = 0; k < n; ++k) {
        b2contact *c = &w->contacts[order[k]];
        b2body *bodyB = &w->bodies[c->body];
        int activeB = bodyB->awake && bodyB->type != B2_STATIC;
        if (!activeB) continue; /* activeA is false: ground is static */
        if (!aabb_overlap(w->fat[c->edge], w->fat[10 + c->body])) { contact_destroy(w, c); continue; }
        contact_update(w, c);
    }
}

/* ------------------------------------------------------------------------- contact solver */
typedef struct { b2v c; float a; } b2pos;
typedef struct { b2v v; float w; } b2vel;
typedef struct { b2v rA, rB; float normalImpulse, tangentImpulse, normalMass, tangentMass, velocityBias; } vcpoint;
typedef struct {
    vcpoint points[2];
    b2v normal;
    float nm11, nm12, nm21, nm22; /* normalMass (ex.x, ey.x, ex.y, ey.y) */
    float k11, k12, k21, k22;     /* K */
    int indexA, indexB;
    float invMassA, invMassB, invIA, invIB, friction, restitution, threshold, tangentSpeed;
    int pointCount, contactIndex;
} vconstraint;
typedef struct {
    b2v localPoints[2], localNormal, localPoint;
    int indexA, indexB;
    float invMassA, invMassB;
    b2v localCenterA, localCenterB;
    float invIA, invIB;
    int type;
    float radiusA, radiusB;
    int pointCount;
} pconstraint;

typedef struct {
    b2world *w;
    b2pos *positions; b2vel *velocities;
    b2contact **contacts; int count;
    vconstraint vc[B2_N_CONTACTS]; pconstraint pc[B2_N_CONTACTS];
    int warmStarting; float dtRatio;
    const int *islandIndex; /* body -> island index */
} csolver;

/* b2ContactSolver::b2ContactSolver */
static void cs_init(csolver *s, b2world *w, b2contact **contacts, int count, b2pos *positions, b2vel *velocities,
                    const int *islandIndex, int warmStarting, float dtRatio) {
    s->w = w; s->positions = positions; s->velocities = velocities; s->contacts = contacts; s->count = count;
    s->warmStarting = warmStarting; s->dtRatio = dtRatio; s->islandIndex = islandIndex;
    for (int i = 0; i < count; ++i) {
        b2contact *contact = contacts[i];
        b2body *bodyA = &w->bodies[0], *bodyB = &w->bodies[contact->body];
        b2manifold *manifold = &contact->m;
        int pointCount = manifold->pointCount;
        vconstraint *vc = &s->vc[i];
        vc->friction = contact->friction; vc->restitution = contact->restitution;
        vc->threshold = contact->restitutionThreshold; vc->tangentSpeed = 0.0f;
        vc->indexA = islandIndex[0]; vc->indexB = islandIndex[contact->body];
        vc->invMassA = bodyA->invMass; vc->invMassB = bodyB->invMass; vc->invIA = bodyA->invI; vc->invIB = bodyB->invI;
        vc->contactIndex = i; vc->pointCount = pointCount;
        vc->k11 = vc->k12 = vc->k21 = vc->k22 = 0.0f; vc->nm11 = vc->nm12 = vc->nm21 = vc->nm22 = 0.0f;
        pconstraint *pc = &s->pc[i];
        pc->indexA = vc->indexA; pc->indexB = vc->indexB;
        pc->invMassA = bodyA->invMass; pc->invMassB = bodyB->invMass;
        pc->localCenterA = bodyA->sw.localCenter; pc->localCenterB = bodyB->sw.localCenter;
        pc->invIA = bodyA->invI; pc->invIB = bodyB->invI;
        pc->localNormal = manifold->localNormal; pc->localPoint = manifold->localPoint;
        pc->pointCount = pointCount;
        pc->radiusA = w->edges[contact->edge].radius; pc->radiusB = w->polys[contact->body - 1].radius;
        pc->type = manifold->type;
        for (int j = 0; j < pointCount; ++j) {
            b2mpoint *cp = &manifold->points[j];
            vcpoint *vcp = &vc->points[j];
            if (warmStarting) {
                vcp->normalImpulse = dtRatio * cp->normalImpulse;
                vcp->tangentImpulse = dtRatio * cp->tangentImpulse;
            } else {
                vcp->normalImpulse = 0.0f; vcp->tangentImpulse = 0.0f;
            }
            vcp->rA = V(0.0f, 0.0f); vcp->rB = V(0.0f, 0.0f);
            vcp->normalMass = 0.0f; vcp->tangentMass = 0.0f; vcp->velocityBias = 0.0f;
            pc->localPoints[j] = cp->localPoint;
        }
    }
}

/* b2WorldManifold::Initialize */
static void world_manifold(const b2manifold *manifold, b2xf xfA, float radiusA, b2xf xfB, float radiusB, b2v *normal, b2v points[2]) {
    if (manifold->pointCount == 0) return;
    if (manifold->type == B2_FACE_A) {
        *normal = rmul(xfA.q, manifold->localNormal);
        b2v planePoint = xmul(xfA, manifold->localPoint);
        for (int i = 0; i < manifold->pointCount; ++i) {
            b2v clipPoint = xmul(xfB, manifold->points[i].localPoint);
            b2v cA = vadd(clipPoint, smul(radiusA - dot(vsub(clipPoint, planePoint), *normal), *normal));
            b2v cB = vsub(clipPoint, smul(radiusB, *normal));
            points[i] = smul(0.5f, vadd(cA, cB));
        }
    } else {
        *normal = rmul(xfB.q, manifold->localNormal);
        b2v planePoint = xmul(xfB, manifold->localPoint);
        for (int i = 0; i < manifold->pointCount; ++i) {
            b2v clipPoint = xmul(xfA, manifold->points[i].localPoint);
            b2v cB = vadd(clipPoint, smul(radiusB - dot(vsub(clipPoint, planePoint), *normal), *normal));
            b2v cA = vsub(clipPoint, smul(radiusA, *normal));
            points[i] = smul(0.5f, vadd(cA, cB));
        }
        *normal = vneg(*normal);
    }
}

/* b2ContactSolver::InitializeVelocityConstraints */
static void cs_init_velocity(csolver *s) {
    for (int i = 0; i < s->count; ++i) {
        vconstraint *vc = &s->vc[i];
        pconstraint *pc = &s->pc[i];
        float radiusA = pc->radiusA, radiusB = pc->radiusB;
        b2manifold *manifold = &s->contacts[vc->contactIndex]->m;
        int indexA = vc->indexA, indexB = vc->indexB;
        float mA = vc->invMassA, mB = vc->invMassB, iA = vc->invIA, iB = vc->invIB;
        b2v localCenterA = pc->localCenterA, localCenterB = pc->localCenterB;
        b2v cA = s->positions[indexA].c; float aA = s->positions[indexA].a;
        b2v vA = s->velocities[indexA].v; float wA = s->velocities[indexA].w;
        b2v cB = s->positions[indexB].c; float aB = s->positions[indexB].a;
        b2v vB = s->velocities[indexB].v; float wB = s->velocities[indexB].w;
        b2xf xfA, xfB;
        xfA.q = rot_set(aA); xfB.q = rot_set(aB);
        xfA.p = vsub(cA, rmul(xfA.q, localCenterA));
        xfB.p = vsub(cB, rmul(xfB.q, localCenterB));
        b2v wm_normal = V(0.0f, 0.0f), wm_points[2];
        world_manifold(manifold, xfA, radiusA, xfB, radiusB, &wm_normal, wm_points);
        vc->normal = wm_normal;
        int pointCount = vc->pointCount;
        for (int j = 0; j < pointCount; ++j) {
            vcpoint *vcp = &vc->points[j];
            vcp->rA = vsub(wm_points[j], cA);
            vcp->rB = vsub(wm_points[j], cB);
            float rnA = cross(vcp->rA, vc->normal), rnB = cross(vcp->rB, vc->normal);
            float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
            vcp->normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
            b2v tangent = cross_vs(vc->normal, 1.0f);
            float rtA = cross(vcp->rA, tangent), rtB = cross(vcp->rB, tangent);
            float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
            vcp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
            vcp->velocityBias = 0.0f;
            float vRel = dot(vc->normal, vsub(vsub(vadd(vB, cross_sv(wB, vcp->rB)), vA), cross_sv(wA, vcp->rA)));
            if (vRel < -vc->threshold) vcp->velocityBias = -vc->restitution * vRel;
        }
        if (vc->pointCount == 2) {
            vcpoint *vcp1 = &vc->points[0], *vcp2 = &vc->points[1];
            float rn1A = cross(vcp1->rA, vc->normal), rn1B = cross(vcp1->rB, vc->normal);
            float rn2A = cross(vcp2->rA, vc->normal), rn2B = cross(vcp2->rB, vc->normal);
            float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
            float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
            float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
            const float k_maxConditionNumber = 1000.0f;
            if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {
                vc->k11 = k11; vc->k21 = k12; /* ex = (k11, k12) */
                vc->k12 = k12; vc->k22 = k22; /* ey = (k12, k22) */
                /* b2Mat22::GetInverse */
                float a = k11, b = k12, c = k12, d = k22;
                float det = a * d - b * c;
                if (det != 0.0f) det = 1.0f / det;
                vc->nm11 = det * d; vc->nm12 = -det * b; vc->nm21 = -det * c; vc->nm22 = det * a;
            } else {
                vc->pointCount = 1;
            }
        }
    }
}

/* b2ContactSolver::WarmStart */
static void cs_warm_start(csolver *s) {
    for (int i = 0; i < s->count; ++i) {
        vconstraint *vc = &s->vc[i];
        int indexA = vc->indexA, indexB = vc->indexB;
        float mA = vc->invMassA, iA = vc->invIA, mB = vc->invMassB, iB = vc->invIB;
        b2v vA = s->velocities[indexA].v; float wA = s->velocities[indexA].w;
        b2v vB = s->velocities[indexB].v; float wB = s->velocities[indexB].w;
        b2v normal = vc->normal, tangent = cross_vs(normal, 1.0f);
        for (int j = 0; j < vc->pointCount; ++j) {
            vcpoint *vcp = &vc->points[j];
            b2v P = vadd(smul(vcp->normalImpulse, normal), smul(vcp->tangentImpulse, tangent));
            wA -= iA * cross(vcp->rA, P);
            vA = vsub(vA, smul(mA, P));
            wB += iB * cross(vcp->rB, P);
            vB = vadd(vB, smul(mB, P));
        }
        s->velocities[indexA].v = vA; s->velocities[indexA].w = wA;
        s->velocities[indexB].v = vB; s->velocities[indexB].w = wB;
    }
}

/* b2ContactSolver::SolveVelocityConstraints */
static void cs_solve_velocity(csolver *s) {
    for (int i = 0; i < s->count; ++i) {
        vconstraint *vc = &s->vc[i];
        int indexA = vc->indexA, indexB = vc->indexB;
        float mA = vc->invMassA, iA = vc->invIA, mB = vc->invMassB, iB = vc->invIB;
        int pointCount = vc->pointCount;
        b2v vA = s->velocities[indexA].v; float wA = s->velocities[indexA].w;
        b2v vB = s->velocities[indexB].v; float wB = s->velocities[indexB].w;
        b2v normal = vc->normal, tangent = cross_vs(normal, 1.0f);
        float friction = vc->friction;

        for (int j = 0; j < pointCount; ++j) {
            vcpoint *vcp = &vc->points[j];
            b2v dv = vsub(vsub(vadd(vB, cross_sv(wB, vcp->rB)), vA), cross_sv(wA, vcp->rA));
            float vt = dot(dv, tangent) - vc->tangentSpeed;
            float lambda = vcp->tangentMass * (-vt);
            float maxFriction = friction * vcp->normalImpulse;
            float newImpulse = fclamp(vcp->tangentImpulse + lambda, -maxFriction, maxFriction);
            lambda = newImpulse - vcp->tangentImpulse;
            vcp->tangentImpulse = newImpulse;
            b2v P = smul(lambda, tangent);
            vA = vsub(vA, smul(mA, P));
            wA -= iA * cross(vcp->rA, P);
            vB = vadd(vB, smul(mB, P));
            wB += iB * cross(vcp->rB, P);
        }

        if (pointCount == 1) {
            for (int j = 0; j < pointCount; ++j) {
                vcpoint *vcp = &vc->points[j];
                b2v dv = vsub(vsub(vadd(vB, cross_sv(wB, vcp->rB)), vA), cross_sv(wA, vcp->rA));
                float vn = dot(dv, normal);
                float lambda = -vcp->normalMass * (vn - vcp->velocityBias);
                float newImpulse = fmax2(vcp->normalImpulse + lambda, 0.0f);
                lambda = newImpulse - vcp->normalImpulse;
                vcp->normalImpulse = newImpulse;
                b2v P = smul(lambda, normal);
                vA = vsub(vA, smul(mA, P));
                wA -= iA * cross(vcp->rA, P);
                vB = vadd(vB, smul(mB, P));
                wB += iB * cross(vcp->rB, P);
            }
        } else {
            vcpoint *cp1 = &vc->points[0], *cp2 = &vc->points[1];
            b2v a = V(cp1->normalImpulse, cp2->normalImpulse);
            b2v dv1 = vsub(vsub(vadd(vB, cross_sv(wB, cp1->rB)), vA), cross_sv(wA, cp1->rA));
            b2v dv2 = vsub(vsub(vadd(vB, cross_sv(wB, cp2->rB)), vA), cross_sv(wA, cp2->rA));
            float vn1 = dot(dv1, normal), vn2 = dot(dv2, normal);
            b2v b = V(vn1 - cp1->velocityBias, vn2 - cp2->velocityBias);
            /* b -= K a  (b2Mul(Mat22, v) = (ex.x*v.x + ey.x*v.y, ex.y*v.x + ey.y*v.y)) */
            b = vsub(b, V(vc->k11 * a.x + vc->k12 * a.y, vc->k21 * a.x + vc->k22 * a.y));
            for (;;) {
                b2v x = vneg(V(vc->nm11 * b.x + vc->nm12 * b.y, vc->nm21 * b.x + vc->nm22 * b.y));
                if (x.x >= 0.0f && x.y >= 0.0f) {
                    b2v d = vsub(x, a);
                    b2v P1 = smul(d.x, normal), P2 = smul(d.y, normal);
                    vA = vsub(vA, smul(mA, vadd(P1, P2)));
                    wA -= iA * (cross(cp1->rA, P1) + cross(cp2->rA, P2));
                    vB = vadd(vB, smul(mB, vadd(P1, P2)));
                    wB += iB * (cross(cp1->rB, P1) + cross(cp2->rB, P2));
                    cp1->normalImpulse = x.x; cp2->normalImpulse = x.y;
                    break;
                }
                x.x = -cp1->normalMass * b.x; x.y = 0.0f;
                vn1 = 0.0f; vn2 = vc->k21 * x.x + b.y; /* K.ex.y */
                if (x.x >= 0.0f && vn2 >= 0.0f) {
                    b2v d = vsub(x, a);
                    b2v P1 = smul(d.x, normal), P2 = smul(d.y, normal);
                    vA = vsub(vA, smul(mA, vadd(P1, P2)));
                    wA -= iA * (cross(cp1->rA, P1) + cross(cp2->rA, P2));
                    vB = vadd(vB, smul(mB, vadd(P1, P2)));
                    wB += iB * (cross(cp1->rB, P1) + cross(cp2->rB, P2));
                    cp1->normalImpulse = x.x; cp2->normalImpulse = x.y;
                    break;
                }
                x.x = 0.0f; x.y = -cp2->normalMass * b.y;
                vn1 = vc->k12 * x.y + b.x; vn2 = 0.0f; /* K.ey.x */
                if (x.y >= 0.0f && vn1 >= 0.0f) {
                    b2v d = vsub(x, a);
                    b2v P1 = smul(d.x, normal), P2 = smul(d.y, normal);
                    vA = vsub(vA, smul(mA, vadd(P1, P2)));
                    wA -= iA * (cross(cp1->rA, P1) + cross(cp2->rA, P2));
                    vB = vadd(vB, smul(mB, vadd(P1, P2)));
                    wB += iB * (cross(cp1->rB, P1) + cross(cp2->rB, P2));
                    cp1->normalImpulse = x.x; cp2->normalImpulse = x.y;
                    break;
                }
                x.x = 0.0f; x.y = 0.0f;
                vn1 = b.x; vn2 = b.y;
                if (vn1 >= 0.0f && vn2 >= 0.0f) {
                    b2v d = vsub(x, a);
                    b2v P1 = smul(d.x, normal), P2 = smul(d.y, normal);
                    vA = vsub(vA, smul(mA, vadd(P1, P2)));
                    wA -= iA * (cross(cp1->rA, P1) + cross(cp2->rA, P2));
                    vB = vadd(vB, smul(mB, vadd(P1, P2)));
                    wB += iB * (cross(cp1->rB, P1) + cross(cp2->rB, P2));
                    cp1->normalImpulse = x.x; cp2->normalImpulse = x.y;
                    break;
                }
                break; /* no solution, give up */
            }
        }
        s->velocities[indexA].v = vA; s->velocities[indexA].w = wA;
        s->velocities[indexB].v = vB; s->velocities[indexB].w = wB;
    }
}

/* b2ContactSolver::StoreImpulses */
static void cs_store_impulses(csolver *s) {
    for (int i = 0; i < s->count; ++i) {
        vconstraint *vc = &s->vc[i];
        b2manifold *manifold = &s->contacts[vc->contactIndex]->m;
        for (int j = 0; j < vc->pointCount; ++j) {
            manifold->points[j].normalImpulse = vc->points[j].normalImpulse;
            manifold->points[j].tangentImpulse = vc->points[j].tangentImpulse;
        }
    }
}

/* b2PositionSolverManifold::Initialize */
static void psm_init(const pconstraint *pc, b2xf xfA, b2xf xfB, int index, b2v *normal, b2v *point, float *separation) {
    if (pc->type == B2_FACE_A) {
        *normal = rmul(xfA.q, pc->localNormal);
        b2v planePoint = xmul(xfA, pc->localPoint);
        b2v clipPoint = xmul(xfB, pc->localPoints[index]);
        *separation = dot(vsub(clipPoint, planePoint), *normal) - pc->radiusA - pc->radiusB;
        *point = clipPoint;
    } else {
        *normal = rmul(xfB.q, pc->localNormal);
        b2v planePoint = xmul(xfB, pc->localPoint);
        b2v clipPoint = xmul(xfA, pc->localPoints[index]);
        *separation = dot(vsub(clipPoint, planePoint), *normal) - pc->radiusA - pc->radiusB;
        *point = clipPoint;
        *normal = vneg(*normal);
    }
}

/* b2ContactSolver::SolvePositionConstraints / SolveTOIPositionConstraints */
static int cs_solve_position(csolver *s, int toi, int toiIndexA, int toiIndexB) {
    float minSeparation = 0.0f;
    for (int i = 0; i < s->count; ++i) {
        pconstraint *pc = &s->pc[i];
        int indexA = pc->indexA, indexB = pc->indexB;
        b2v localCenterA = pc->localCenterA, localCenterB = pc->localCenterB;
        int pointCount = pc->pointCount;
        float mA, iA, mB, iB;
        if (toi) {
            mA = 0.0f; iA = 0.0f;
            if (indexA == toiIndexA || indexA == toiIndexB) { mA = pc->invMassA; iA = pc->invIA; }
            mB = 0.0f; iB = 0.0f;
            if (indexB == toiIndexA || indexB == toiIndexB) { mB = pc->invMassB; iB = pc->invIB; }
        } else {
            mA = pc->invMassA; iA = pc->invIA; mB = pc->invMassB; iB = pc->invIB;
        }
        b2v cA = s->positions[indexA].c; float aA = s->positions[indexA].a;
        b2v cB = s->positions[indexB].c; float aB = s->positions[indexB].a;
        for (int j = 0; j < pointCount; ++j) {
            b2xf xfA, xfB;
            xfA.q = rot_set(aA); xfB.q = rot_set(aB);
            xfA.p = vsub(cA, rmul(xfA.q, localCenterA));
            xfB.p = vsub(cB, rmul(xfB.q, localCenterB));
            b2v normal, point; float separation;
            psm_init(pc, xfA, xfB, j, &normal, &point, &separation);
            b2v rA = vsub(point, cA), rB = vsub(point, cB);
            minSeparation = fmin2(minSeparation, separation);
            float C = fclamp((toi ? b2_toiBaumgarte : b2_baumgarte) * (separation + b2_linearSlop), -b2_maxLinearCorrection, 0.0f);
            float rnA = cross(rA, normal), rnB = cross(rB, normal);
            float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
            float impulse = K > 0.0f ? -C / K : 0.0f;
            b2v P = smul(impulse, normal);
            cA = vsub(cA, smul(mA, P));
            aA -= iA * cross(rA, P);
            cB = vadd(cB, smul(mB, P));
            aB += iB * cross(rB, P);
        }
        s->positions[indexA].c = cA; s->positions[indexA].a = aA;
        s->positions[indexB].c = cB; s->positions[indexB].a = aB;
    }
    return toi ? (minSeparation >= -1.5f * b2_linearSlop) : (minSeparation >= -3.0f * b2_linearSlop);
}

/* ------------------------------------------------------------------------- revolute joint */
/* b2RevoluteJoint::InitVelocityConstraints */
static void rj_init_velocity(b2world *w, b2revolute *j, b2pos *positions, b2vel *velocities, const int *islandIndex,
                             float dtRatio, int warmStarting) {
    b2body *bA = &w->bodies[j->bodyA], *bB = &w->bodies[j->bodyB];
    int indexA = islandIndex[j->bodyA], indexB = islandIndex[j->bodyB];
    j->localCenterA = bA->sw.localCenter; j->localCenterB = bB->sw.localCenter;
    j->invMassA = bA->invMass; j->invMassB = bB->invMass; j->invIA = bA->invI; j->invIB = bB->invI;
    float aA = positions[indexA].a; b2v vA = velocities[indexA].v; float wA = velocities[indexA].w;
    float aB = positions[indexB].a; b2v vB = velocities[indexB].v; float wB = velocities[indexB].w;
    b2rot qA = rot_set(aA), qB = rot_set(aB);
    j->rA = rmul(qA, vsub(j->localAnchorA, j->localCenterA));
    j->rB = rmul(qB, vsub(j->localAnchorB, j->localCenterB));
    float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
    j->K11 = mA + mB + j->rA.y * j->rA.y * iA + j->rB.y * j->rB.y * iB;  /* ex.x */
    j->K12 = -j->rA.y * j->rA.x * iA - j->rB.y * j->rB.x * iB;           /* ey.x */
    j->K21 = j->K12;                                                     /* ex.y */
    j->K22 = mA + mB + j->rA.x * j->rA.x * iA + j->rB.x * j->rB.x * iB;  /* ey.y */
    j->axialMass = iA + iB;
    int fixedRotation;
    if (j->axialMass > 0.0f) { j->axialMass = 1.0f / j->axialMass; fixedRotation = 0; } else fixedRotation = 1;
    j->angle = aB - aA - j->referenceAngle;
    if (j->enableLimit == 0 || fixedRotation) { j->lowerImpulse = 0.0f; j->upperImpulse = 0.0f; }
    if (j->enableMotor == 0 || fixedRotation) j->motorImpulse = 0.0f;
    if (warmStarting) {
        j->impulse = smul(dtRatio, j->impulse);
        j->motorImpulse *= dtRatio; j->lowerImpulse *= dtRatio; j->upperImpulse *= dtRatio;
        float axialImpulse = j->motorImpulse + j->lowerImpulse - j->upperImpulse;
        b2v P = V(j->impulse.x, j->impulse.y);
        vA = vsub(vA, smul(mA, P));
        wA -= iA * (cross(j->rA, P) + axialImpulse);
        vB = vadd(vB, smul(mB, P));
        wB += iB * (cross(j->rB, P) + axialImpulse);
    } else {
        j->impulse = V(0.0f, 0.0f); j->motorImpulse = 0.0f; j->lowerImpulse = 0.0f; j->upperImpulse = 0.0f;
    }
    velocities[indexA].v = vA; velocities[indexA].w = wA;
    velocities[indexB].v = vB; velocities[indexB].w = wB;
}

/* b2RevoluteJoint::SolveVelocityConstraints */
static void rj_solve_velocity(b2revolute *j, b2vel *velocities, const int *islandIndex, float dt, float inv_dt) {
    int indexA = islandIndex[j->bodyA], indexB = islandIndex[j->bodyB];
    b2v vA = velocities[indexA].v; float wA = velocities[indexA].w;
    b2v vB = velocities[indexB].v; float wB = velocities[indexB].w;
    float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
    int fixedRotation = (iA + iB == 0.0f);
    if (j->enableMotor && !fixedRotation) {
        float Cdot = wB - wA - j->motorSpeed;
        float impulse = -j->axialMass * Cdot;
        float oldImpulse = j->motorImpulse;
        float maxImpulse = dt * j->maxMotorTorque;
        j->motorImpulse = fclamp(oldImpulse + impulse, -maxImpulse, maxImpulse);
        impulse = j->motorImpulse - oldImpulse;
        wA -= iA * impulse;
        wB += iB * impulse;
    }
    if (j->enableLimit && !fixedRotation) {
        {
            float C = j->angle - j->lowerAngle;
            float Cdot = wB - wA;
            float impulse = -j->axialMass * (Cdot + fmax2(C, 0.0f) * inv_dt);
            float newImpulse = fmax2(j->lowerImpulse + impulse, 0.0f);
            impulse = newImpulse - j->lowerImpulse;
            j->lowerImpulse = newImpulse;
            wA -= iA * impulse;
            wB += iB * impulse;
        }
        {
            float C = j->upperAngle - j->angle;
            float Cdot = wA - wB;
            float impulse = -j->axialMass * (Cdot + fmax2(C, 0.0f) * inv_dt);
            float newImpulse = fmax2(j->upperImpulse + impulse, 0.0f);
            impulse = newImpulse - j->upperImpulse;
            j->upperImpulse = newImpulse;
            wA += iA * impulse;
            wB -= iB * impulse;
        }
    }
    {
        b2v Cdot = vsub(vsub(vadd(vB, cross_sv(wB, j->rB)), vA), cross_sv(wA, j->rA));
        b2v impulse = mat22_solve(j->K11, j->K12, j->K21, j->K22, vneg(Cdot));
        j->impulse.x += impulse.x;
        j->impulse.y += impulse.y;
        vA = vsub(vA, smul(mA, impulse));
        wA -= iA * cross(j->rA, impulse);
        vB = vadd(vB, smul(mB, impulse));
        wB += iB * cross(j->rB, impulse);
    }
    velocities[indexA].v = vA; velocities[indexA].w = wA;
    velocities[indexB].v = vB; velocities[indexB].w = wB;
}

/* b2RevoluteJoint::SolvePositionConstraints */
static int rj_solve_position(b2revolute *j, b2pos *positions, const int *islandIndex) {
    int indexA = islandIndex[j->bodyA], indexB = islandIndex[j->bodyB];
    b2v cA = positions[indexA].c; float aA = positions[indexA].a;
    b2v cB = positions[indexB].c; float aB = positions[indexB].a;
    b2rot qA, qB;
    float angularError = 0.0f, positionError = 0.0f;
    int fixedRotation = (j->invIA + j->invIB == 0.0f);
    if (j->enableLimit && !fixedRotation) {
        float angle = aB - aA - j->referenceAngle;
        float C = 0.0f;
        if (fabs1(j->upperAngle - j->lowerAngle) < 2.0f * b2_angularSlop) {
            C = fclamp(angle - j->lowerAngle, -b2_maxAngularCorrection, b2_maxAngularCorrection);
        } else if (angle <= j->lowerAngle) {
            C = fclamp(angle - j->lowerAngle + b2_angularSlop, -b2_maxAngularCorrection, 0.0f);
        } else if (angle >= j->upperAngle) {
            C = fclamp(angle - j->upperAngle - b2_angularSlop, 0.0f, b2_maxAngularCorrection);
        }
        float limitImpulse = -j->axialMass * C;
        aA -= j->invIA * limitImpulse;
        aB += j->invIB * limitImpulse;
        angularError = fabs1(C);
    }
    {
        qA = rot_set(aA); qB = rot_set(aB);
        b2v rA = rmul(qA, vsub(j->localAnchorA, j->localCenterA));
        b2v rB = rmul(qB, vsub(j->localAnchorB, j->localCenterB));
        b2v C = vsub(vsub(vadd(cB, rB), cA), rA);
        positionError = vlen(C);
        float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
        float Kexx = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
        float Kexy = -iA * rA.x * rA.y - iB * rB.x * rB.y;
        float Keyx = Kexy;
        float Keyy = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
        b2v impulse = vneg(mat22_solve(Kexx, Keyx, Kexy, Keyy, C));
        cA = vsub(cA, smul(mA, impulse));
        aA -= iA * cross(rA, impulse);
        cB = vadd(cB, smul(mB, impulse));
        aB += iB * cross(rB, impulse);
    }
    positions[indexA].c = cA; positions[indexA].a = aA;
    positions[indexB].c = cB; positions[indexB].a = aB;
    return positionError <= b2_linearSlop && angularError <= b2_angularSlop;
}

/* ------------------------------------------------------------------------- island */
/* b2World::Solve + b2Island::Solve for the single island {leg1, lander, leg0, ground}.
 * DFS from the first awake dynamic body of the body list (newest first: leg1, leg0, lander):
 * island bodies = [leg1, lander, leg0, ground*]; joints = [joint(leg1), joint(leg0)];
 * contacts = leg1's, lander's, leg0's touching contacts, each body's newest first. */
static void solve_island(b2world *w, float dt, float inv_dt, float dtRatio, int velIters, int posIters) {
    for (int b = 0; b < B2_N_BODIES; ++b) w->bodies[b].islandFlag = 0;
    for (int i = 0; i < B2_N_CONTACTS; ++i) w->contacts[i].islandFlag = 0;
    int seed = -1;
    for (int b = w->n_dyn; b >= 1; --b) {
        if (w->bodies[b].awake) { seed = b; break; }
    }
    if (seed < 0) return; /* everything asleep */

    /* DFS with Box2D's stack discipline */
    int islandBodies[B2_N_BODIES], nb = 0;
    b2contact *islandContacts[B2_N_CONTACTS]; int nc = 0;
    b2revolute *islandJoints[2]; int nj = 0; int jointFlag[2] = {0, 0};
    int stack[8], sc = 0;
    int order[B2_N_CONTACTS];
    int n_order = contact_order(w, order);
    stack[sc++] = seed; w->bodies[seed].islandFlag = 1;
    while (sc > 0) {
        int b = stack[--sc];
        islandBodies[nb++] = b;
        if (w->bodies[b].type == B2_STATIC) continue;
        w->bodies[b].awake = 1; /* without resetting the sleep timer */
        for (int k = 0; k < n_order; ++k) { /* this body's contact edges, newest first */
            b2contact *c = &w->contacts[order[k]];
            if (c->body != b) continue;
            if (c->islandFlag) continue;
            if (!c->enabled || !c->touching) continue;
            islandContacts[nc++] = c;
            c->islandFlag = 1;
            if (w->bodies[0].islandFlag) continue;
            stack[sc++] = 0; w->bodies[0].islandFlag = 1;
        }
        for (int jj = w->n_joints - 1; jj >= 0; --jj) { /* joint edges, newest first */
            b2revolute *j = &w->joints[jj];
            if (j->bodyA != b && j->bodyB != b) continue;
            if (jointFlag[jj]) continue;
            int other = j->bodyA == b ? j->bodyB : j->bodyA;
            islandJoints[nj++] = j; jointFlag[jj] = 1;
            if (w->bodies[other].islandFlag) continue;
            stack[sc++] = other; w->bodies[other].islandFlag = 1;
        }
    }

    /* ---- b2Island::Solve ---- */
    int islandIndex[B2_N_BODIES] = {-1, -1, -1, -1};
    b2pos positions[B2_N_BODIES]; b2vel velocities[B2_N_BODIES];
    float h = dt;
    for (int i = 0; i < nb; ++i) {
        b2body *b = &w->bodies[islandBodies[i]];
        islandIndex[islandBodies[i]] = i;
        b2v c = b->sw.c; float a = b->sw.a; b2v v = b->linvel; float wv = b->angvel;
        b->sw.c0 = b->sw.c; b->sw.a0 = b->sw.a;
        if (b->type == B2_DYNAMIC) {
            /* v += h * invMass * (gravityScale * mass * gravity + force);  gravityScale = 1 */
            v = vadd(v, smul(h * b->invMass, vadd(smul(1.0f * b->mass, w->gravity), b->force)));
            wv += h * b->invI * b->torque;
            v = smul(1.0f / (1.0f + h * 0.0f), v);   /* linearDamping = 0 */
            wv *= 1.0f / (1.0f + h * 0.0f);          /* angularDamping = 0 */
        }
        positions[i].c = c; positions[i].a = a; velocities[i].v = v; velocities[i].w = wv;
    }
    if (islandIndex[0] < 0) { /* ground not in the island: constraints never reference it, keep a valid slot */
        islandIndex[0] = nb;
        positions[nb].c = w->bodies[0].sw.c; positions[nb].a = w->bodies[0].sw.a;
        velocities[nb].v = V(0.0f, 0.0f); velocities[nb].w = 0.0f;
    }

    csolver cs;
    csolver *s = &cs;
    cs_init(s, w, islandContacts, nc, positions, velocities, islandIndex, 1, dtRatio);
    cs_init_velocity(s);
    cs_warm_start(s);
    for (int i = 0; i < nj; ++i) rj_init_velocity(w, islandJoints[i], positions, velocities, islandIndex, dtRatio, 1);
    for (int it = 0; it < velIters; ++it) {
        for (int j = 0; j < nj; ++j) rj_solve_velocity(islandJoints[j], velocities, islandIndex, dt, inv_dt);
        cs_solve_velocity(s);
    }
    cs_store_impulses(s);

    for (int i = 0; i < nb; ++i) {
        b2v c = positions[i].c; float a = positions[i].a; b2v v = velocities[i].v; float wv = velocities[i].w;
        b2v translation = smul(h, v);
        if (dot(translation, translation) > b2_maxTranslationSquared) {
            float ratio = b2_maxTranslation / vlen(translation);
            v = smul(ratio, v);
        }
        float rotation = h * wv;
        if (rotation * rotation > b2_maxRotationSquared) {
            float ratio = b2_maxRotation / fabs1(rotation);
            wv *= ratio;
        }
        c = vadd(c, smul(h, v));
        a += h * wv;
        positions[i].c = c; positions[i].a = a; velocities[i].v = v; velocities[i].w = wv;
    }

    int positionSolved = 0;
    STAT(g_b2_stats.hist_touching[nc < 15 ? nc : 15]++);
    int stat_pos_iters = 0; (void)stat_pos_iters;
    for (int it = 0; it < posIters; ++it) {
        STAT(stat_pos_iters++);
        int contactsOkay = cs_solve_position(s, 0, 0, 0);
        int jointsOkay = 1;
        for (int j = 0; j < nj; ++j) {
            int jointOkay = rj_solve_position(islandJoints[j], positions, islandIndex);
            jointsOkay = jointsOkay && jointOkay;
        }
        if (contactsOkay && jointsOkay) { positionSolved = 1; break; }
    }

    STAT(if (nc > 0) g_b2_stats.hist_pos_iters[stat_pos_iters < 63 ? stat_pos_iters : 63]++);
    for (int i = 0; i < nb; ++i) {
        b2body *b = &w->bodies[islandBodies[i]];
        b->sw.c = positions[i].c; b->sw.a = positions[i].a;
        b->linvel = velocities[i].v; b->angvel = velocities[i].w;
        body_sync_transform(b);
    }

    { /* allowSleep */
        float minSleepTime = b2_maxFloat;
        const float linTolSqr = b2_linearSleepTolerance * b2_linearSleepTolerance;
        const float angTolSqr = b2_angularSleepTolerance * b2_angularSleepTolerance;
        for (int i = 0; i < nb; ++i) {
            b2body *b = &w->bodies[islandBodies[i]];
            if (b->type == B2_STATIC) continue;
            if (b->angvel * b->angvel > angTolSqr || dot(b->linvel, b->linvel) > linTolSqr) {
                b->sleepTime = 0.0f;
                minSleepTime = 0.0f;
            } else {
                b->sleepTime += h;
                minSleepTime = fmin2(minSleepTime, b->sleepTime);
            }
        }
        if (minSleepTime >= b2_timeToSleep && positionSolved) {
            for (int i = 0; i < nb; ++i) b2b_set_awake(w, islandBodies[i], 0);
        }
    }

    /* b2World::Solve tail: synchronize fixtures of island bodies in body-list order, find new contacts */
    for (int b = w->n_dyn; b >= 1; --b) {
        if (!w->bodies[b].islandFlag) continue;
        body_sync_fixtures(w, b);
    }
    find_new_contacts(w);
}

/* ------------------------------------------------------------------------- GJK (b2Distance) */
typedef struct { const b2v *v; int count; float radius; } dproxy;
typedef struct { float metric; int count; uint8_t indexA[3], indexB[3]; } scache;
typedef struct { b2v wA, wB, w; float a; int indexA, indexB; } svertex;
typedef struct { svertex v[3]; int count; } simplex;

static int proxy_support(const dproxy *p, b2v d) {
    int bestIndex = 0;
    float bestValue = dot(p->v[0], d);
    for (int i = 1; i < p->count; ++i) {
        float value = dot(p->v[i], d);
        if (value > bestValue) { bestIndex = i; bestValue = value; }
    }
    return bestIndex;
}
static float simplex_metric(const simplex *s) {
    switch (s->count) {
    case 1: return 0.0f;
    case 2: return vlen(vsub(s->v[0].w, s->v[1].w));
    case 3: return cross(vsub(s->v[1].w, s->v[0].w), vsub(s->v[2].w, s->v[0].w));
    default: return 0.0f;
    }
}
static void simplex_read_cache(simplex *s, const scache *cache, const dproxy *pA, b2xf xfA, const dproxy *pB, b2xf xfB) {
    s->count = cache->count;
    for (int i = 0; i < s->count; ++i) {
        svertex *v = &s->v[i];
        v->indexA = cache->indexA[i]; v->indexB = cache->indexB[i];
        v->wA = xmul(xfA, pA->v[v->indexA]);
        v->wB = xmul(xfB, pB->v[v->indexB]);
        v->w = vsub(v->wB, v->wA);
        v->a = 0.0f;
    }
    if (s->count > 1) {
        float metric1 = cache->metric, metric2 = simplex_metric(s);
        if (metric2 < 0.5f * metric1 || 2.0f * metric1 < metric2 || metric2 < b2_epsilon) s->count = 0;
    }
    if (s->count == 0) {
        svertex *v = &s->v[0];
        v->indexA = 0; v->indexB = 0;
        v->wA = xmul(xfA, pA->v[0]);
        v->wB = xmul(xfB, pB->v[0]);
        v->w = vsub(v->wB, v->wA);
        v->a = 1.0f;
        s->count = 1;
    }
}
static void simplex_write_cache(const simplex *s, scache *cache) {
    cache->metric = simplex_metric(s);
    cache->count = s->count;
    for (int i = 0; i < s->count; ++i) { cache->indexA[i] = (uint8_t)s->v[i].indexA; cache->indexB[i] = (uint8_t)s->v[i].indexB; }
}
static b2v simplex_search_dir(const simplex *s) {
    if (s->count == 1) return vneg(s->v[0].w);
    if (s->count == 2) {
        b2v e12 = vsub(s->v[1].w, s->v[0].w);
        float sgn = cross(e12, vneg(s->v[0].w));
        if (sgn > 0.0f) return cross_sv(1.0f, e12);
        return cross_vs(e12, 1.0f);
    }
    return V(0.0f, 0.0f);
}
static void simplex_witness(const simplex *s, b2v *pA, b2v *pB) {
    if (s->count == 1) { *pA = s->v[0].wA; *pB = s->v[0].wB; }
    else if (s->count == 2) {
        *pA = vadd(smul(s->v[0].a, s->v[0].wA), smul(s->v[1].a, s->v[1].wA));
        *pB = vadd(smul(s->v[0].a, s->v[0].wB), smul(s->v[1].a, s->v[1].wB));
    } else {
        *pA = vadd(vadd(smul(s->v[0].a, s->v[0].wA), smul(s->v[1].a, s->v[1].wA)), smul(s->v[2].a, s->v[2].wA));
        *pB = *pA;
    }
}
static void simplex_solve2(simplex *s) {
    b2v w1 = s->v[0].w, w2 = s->v[1].w, e12 = vsub(w2, w1);
    float d12_2 = -dot(w1, e12);
    if (d12_2 <= 0.0f) { s->v[0].a = 1.0f; s->count = 1; return; }
    float d12_1 = dot(w2, e12);
    if (d12_1 <= 0.0f) { s->v[1].a = 1.0f; s->count = 1; s->v[0] = s->v[1]; return; }
    float inv_d12 = 1.0f / (d12_1 + d12_2);
    s->v[0].a = d12_1 * inv_d12; s->v[1].a = d12_2 * inv_d12; s->count = 2;
}
static void simplex_solve3(simplex *s) {
    b2v w1 = s->v[0].w, w2 = s->v[1].w, w3 = s->v[2].w;
    b2v e12 = vsub(w2, w1);
    float w1e12 = dot(w1, e12), w2e12 = dot(w2, e12);
    float d12_1 = w2e12, d12_2 = -w1e12;
    b2v e13 = vsub(w3, w1);
    float w1e13 = dot(w1, e13), w3e13 = dot(w3, e13);
    float d13_1 = w3e13, d13_2 = -w1e13;
    b2v e23 = vsub(w3, w2);
    float w2e23 = dot(w2, e23), w3e23 = dot(w3, e23);
    float d23_1 = w3e23, d23_2 = -w2e23;
    float n123 = cross(e12, e13);
    float d123_1 = n123 * cross(w2, w3), d123_2 = n123 * cross(w3, w1), d123_3 = n123 * cross(w1, w2);
    if (d12_2 <= 0.0f && d13_2 <= 0.0f) { s->v[0].a = 1.0f; s->count = 1; return; }
    if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) {
        float inv = 1.0f / (d12_1 + d12_2);
        s->v[0].a = d12_1 * inv; s->v[1].a = d12_2 * inv; s->count = 2; return;
    }
    if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) {
        float inv = 1.0f / (d13_1 + d13_2);
        s->v[0].a = d13_1 * inv; s->v[2].a = d13_2 * inv; s->count = 2; s->v[1] = s->v[2]; return;
    }
    if (d12_1 <= 0.0f && d23_2 <= 0.0f) { s->v[1].a = 1.0f; s->count = 1; s->v[0] = s->v[1]; return; }
    if (d13_1 <= 0.0f && d23_1 <= 0.0f) { s->v[2].a = 1.0f; s->count = 1; s->v[0] = s->v[2]; return; }
    if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) {
        float inv = 1.0f / (d23_1 + d23_2);
        s->v[1].a = d23_1 * inv; s->v[2].a = d23_2 * inv; s->count = 2; s->v[0] = s->v[2]; return;
    }
    float inv = 1.0f / (d123_1 + d123_2 + d123_3);
    s->v[0].a = d123_1 * inv; s->v[1].a = d123_2 * inv; s->v[2].a = d123_3 * inv; s->count = 3;
}

/* b2Distance with useRadii = false; returns distance */
static float gjk_distance(scache *cache, const dproxy *pA, b2xf xfA, const dproxy *pB, b2xf xfB) {
    simplex sx;
    simplex_read_cache(&sx, cache, pA, xfA, pB, xfB);
    const int k_maxIters = 20;
    int saveA[3], saveB[3], saveCount = 0;
    int iter = 0;
    while (iter < k_maxIters) {
        saveCount = sx.count;
        for (int i = 0; i < saveCount; ++i) { saveA[i] = sx.v[i].indexA; saveB[i] = sx.v[i].indexB; }
        if (sx.count == 2) simplex_solve2(&sx);
        else if (sx.count == 3) simplex_solve3(&sx);
        if (sx.count == 3) break;
        b2v d = simplex_search_dir(&sx);
        if (vlen2(d) < b2_epsilon * b2_epsilon) break;
        svertex *vertex = &sx.v[sx.count];
        vertex->indexA = proxy_support(pA, rmulT(xfA.q, vneg(d)));
        vertex->wA = xmul(xfA, pA->v[vertex->indexA]);
        vertex->indexB = proxy_support(pB, rmulT(xfB.q, d));
        vertex->wB = xmul(xfB, pB->v[vertex->indexB]);
        vertex->w = vsub(vertex->wB, vertex->wA);
        ++iter;
        int duplicate = 0;
        for (int i = 0; i < saveCount; ++i)
            if (vertex->indexA == saveA[i] && vertex->indexB == saveB[i]) { duplicate = 1; break; }
        if (duplicate) break;
        ++sx.count;
    }
    b2v pointA, pointB;
    simplex_witness(&sx, &pointA, &pointB);
    float distance = vlen(vsub(pointA, pointB));
    simplex_write_cache(&sx, cache);
    return distance;
}

/* ------------------------------------------------------------------------- b2TimeOfImpact */
enum { SEP_POINTS, SEP_FACE_A, SEP_FACE_B };
typedef struct { const dproxy *pA, *pB; b2sweep sweepA, sweepB; int type; b2v localPoint, axis; } sepfn;

static void sep_init(sepfn *f, const scache *cache, const dproxy *pA, const b2sweep *sweepA, const dproxy *pB,
                     const b2sweep *sweepB, float t1) {
    f->pA = pA; f->pB = pB;
    int count = cache->count;
    f->sweepA = *sweepA; f->sweepB = *sweepB;
    b2xf xfA = sweep_xf(&f->sweepA, t1), xfB = sweep_xf(&f->sweepB, t1);
    if (count == 1) {
        f->type = SEP_POINTS;
        b2v pointA = xmul(xfA, pA->v[cache->indexA[0]]);
        b2v pointB = xmul(xfB, pB->v[cache->indexB[0]]);
        f->axis = vsub(pointB, pointA);
        normalize(&f->axis);
    } else if (cache->indexA[0] == cache->indexA[1]) {
        f->type = SEP_FACE_B;
        b2v localPointB1 = pB->v[cache->indexB[0]], localPointB2 = pB->v[cache->indexB[1]];
        f->axis = cross_vs(vsub(localPointB2, localPointB1), 1.0f);
        normalize(&f->axis);
        b2v normal = rmul(xfB.q, f->axis);
        f->localPoint = smul(0.5f, vadd(localPointB1, localPointB2));
        b2v pointB = xmul(xfB, f->localPoint);
        b2v pointA = xmul(xfA, pA->v[cache->indexA[0]]);
        float s = dot(vsub(pointA, pointB), normal);
        if (s < 0.0f) f->axis = vneg(f->axis);
    } else {
        f->type = SEP_FACE_A;
        b2v localPointA1 = pA->v[cache->indexA[0]], localPointA2 = pA->v[cache->indexA[1]];
        f->axis = cross_vs(vsub(localPointA2, localPointA1), 1.0f);
        normalize(&f->axis);
        b2v normal = rmul(xfA.q, f->axis);
        f->localPoint = smul(0.5f, vadd(localPointA1, localPointA2));
        b2v pointA = xmul(xfA, f->localPoint);
        b2v pointB = xmul(xfB, pB->v[cache->indexB[0]]);
        float s = dot(vsub(pointB, pointA), normal);
        if (s < 0.0f) f->axis = vneg(f->axis);
    }
}
static float sep_find_min(const sepfn *f, int *indexA, int *indexB, float t) {
    b2xf xfA = sweep_xf(&f->sweepA, t), xfB = sweep_xf(&f->sweepB, t);
    if (f->type == SEP_POINTS) {
        b2v axisA = rmulT(xfA.q, f->axis), axisB = rmulT(xfB.q, vneg(f->axis));
        *indexA = proxy_support(f->pA, axisA);
        *indexB = proxy_support(f->pB, axisB);
        b2v pointA = xmul(xfA, f->pA->v[*indexA]), pointB = xmul(xfB, f->pB->v[*indexB]);
        return dot(vsub(pointB, pointA), f->axis);
    } else if (f->type == SEP_FACE_A) {
        b2v normal = rmul(xfA.q, f->axis);
        b2v pointA = xmul(xfA, f->localPoint);
        b2v axisB = rmulT(xfB.q, vneg(normal));
        *indexA = -1;
        *indexB = proxy_support(f->pB, axisB);
        b2v pointB = xmul(xfB, f->pB->v[*indexB]);
        return dot(vsub(pointB, pointA), normal);
    } else {
        b2v normal = rmul(xfB.q, f->axis);
        b2v pointB = xmul(xfB, f->localPoint);
        b2v axisA = rmulT(xfA.q, vneg(normal));
        *indexB = -1;
        *indexA = proxy_support(f->pA, axisA);
        b2v pointA = xmul(xfA, f->pA->v[*indexA]);
        return dot(vsub(pointA, pointB), normal);
    }
}
static float sep_evaluate(const sepfn *f, int indexA, int indexB, float t) {
    b2xf xfA = sweep_xf(&f->sweepA, t), xfB = sweep_xf(&f->sweepB, t);
    if (f->type == SEP_POINTS) {
        b2v pointA = xmul(xfA, f->pA->v[indexA]), pointB = xmul(xfB, f->pB->v[indexB]);
        return dot(vsub(pointB, pointA), f->axis);
    } else if (f->type == SEP_FACE_A) {
        b2v normal = rmul(xfA.q, f->axis);
        b2v pointA = xmul(xfA, f->localPoint);
        b2v pointB = xmul(xfB, f->pB->v[indexB]);
        return dot(vsub(pointB, pointA), normal);
    } else {
        b2v normal = rmul(xfB.q, f->axis);
        b2v pointB = xmul(xfB, f->localPoint);
        b2v pointA = xmul(xfA, f->pA->v[indexA]);
        return dot(vsub(pointA, pointB), normal);
    }
}

enum { TOI_UNKNOWN, TOI_FAILED, TOI_OVERLAPPED, TOI_TOUCHING, TOI_SEPARATED };

/* b2TimeOfImpact, tMax = 1 */
static int time_of_impact(const dproxy *proxyA, const dproxy *proxyB, b2sweep sweepA, b2sweep sweepB, float *t_out) {
    int state = TOI_UNKNOWN;
    const float tMax = 1.0f;
    *t_out = tMax;
    sweep_normalize(&sweepA);
    sweep_normalize(&sweepB);
    float totalRadius = proxyA->radius + proxyB->radius;
    float target = fmax2(b2_linearSlop, totalRadius - 3.0f * b2_linearSlop);
    float tolerance = 0.25f * b2_linearSlop;
    float t1 = 0.0f;
    const int k_maxIterations = 20;
    int iter = 0;
    scache cache;
    cache.count = 0; cache.metric = 0.0f;
    STAT(g_b2_stats.toi_calls++; g_b2_stats.cur_toi_calls++);
    for (;;) {
        STAT(g_b2_stats.toi_outer_iters++);
        b2xf xfA = sweep_xf(&sweepA, t1), xfB = sweep_xf(&sweepB, t1);
        float distance = gjk_distance(&cache, proxyA, xfA, proxyB, xfB);
        if (distance <= 0.0f) { state = TOI_OVERLAPPED; *t_out = 0.0f; break; }
        if (distance < target + tolerance) { state = TOI_TOUCHING; *t_out = t1; break; }
        sepfn fcn;
        sep_init(&fcn, &cache, proxyA, &sweepA, proxyB, &sweepB, t1);
        int done = 0;
        float t2 = tMax;
        int pushBackIter = 0;
        for (;;) {
            int indexA, indexB;
            float s2 = sep_find_min(&fcn, &indexA, &indexB, t2);
            if (s2 > target + tolerance) { state = TOI_SEPARATED; *t_out = tMax; done = 1; break; }
            if (s2 > target - tolerance) { t1 = t2; break; }
            float s1 = sep_evaluate(&fcn, indexA, indexB, t1);
            if (s1 < target - tolerance) { state = TOI_FAILED; *t_out = t1; done = 1; break; }
            if (s1 <= target + tolerance) { state = TOI_TOUCHING; *t_out = t1; done = 1; break; }
            int rootIterCount = 0;
            float a1 = t1, a2 = t2;
            for (;;) {
                float t;
                if (rootIterCount & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);
                else t = 0.5f * (a1 + a2);
                ++rootIterCount;
                STAT(g_b2_stats.toi_root_iters++);
                float s = sep_evaluate(&fcn, indexA, indexB, t);
                if (fabs1(s - target) < tolerance) { t2 = t; break; }
                if (s > target) { a1 = t; s1 = s; } else { a2 = t; s2 = s; }
                if (rootIterCount == 50) break;
            }
            ++pushBackIter;
            STAT(g_b2_stats.toi_pushback++);
            if (pushBackIter == B2_MAX_POLY) break;
        }
        ++iter;
        if (done) break;
        if (iter == k_maxIterations) { state = TOI_FAILED; *t_out = t1; break; }
    }
    return state;
}

/* ------------------------------------------------------------------------- b2World::SolveTOI */
static void body_advance(b2body *b, float alpha) { /* b2Body::Advance */
    sweep_advance(&b->sw, alpha);
    b->sw.c = b->sw.c0;
    b->sw.a = b->sw.a0;
    b->xf.q = rot_set(b->sw.a);
    b->xf.p = vsub(b->sw.c, rmul(b->xf.q, b->sw.localCenter));
}

static void solve_toi(b2world *w, float dt, int velIters) {
    for (int b = 0; b < B2_N_BODIES; ++b) { w->bodies[b].islandFlag = 0; w->bodies[b].sw.alpha0 = 0.0f; }
    for (int i = 0; i < B2_N_CONTACTS; ++i) {
        b2contact *c = &w->contacts[i];
        if (!c->exists) continue;
        c->toiFlag = 0; c->islandFlag = 0; c->toiCount = 0; c->toi = 1.0f;
    }
    for (;;) {
        b2contact *minContact = 0;
        float minAlpha = 1.0f;
        int order[B2_N_CONTACTS];
        int n_order = contact_order(w, order);
        for (int k = 0; k < n_order; ++k) {
            b2contact *c = &w->contacts[order[k]];
            if (!c->enabled) continue;
            if (c->toiCount > b2_maxSubSteps) continue;
            float alpha = 1.0f;
            if (c->toiFlag) {
                alpha = c->toi;
            } else {
                b2body *bA = &w->bodies[0], *bB = &w->bodies[c->body];
                int activeB = bB->awake; /* typeB dynamic; activeA false (static) */
                if (!activeB) continue;
                /* collideA = true (A is not dynamic) */
                float alpha0 = bA->sw.alpha0;
                if (bA->sw.alpha0 < bB->sw.alpha0) { alpha0 = bB->sw.alpha0; sweep_advance(&bA->sw, alpha0); }
                else if (bB->sw.alpha0 < bA->sw.alpha0) { alpha0 = bA->sw.alpha0; sweep_advance(&bB->sw, alpha0); }
                dproxy pA, pB;
                pA.v = &w->edges[c->edge].v1; pA.count = 2; pA.radius = w->edges[c->edge].radius;
                pB.v = w->polys[c->body - 1].v; pB.count = w->polys[c->body - 1].count; pB.radius = w->polys[c->body - 1].radius;
                float beta;
                int state = time_of_impact(&pA, &pB, bA->sw, bB->sw, &beta);
                if (state == TOI_TOUCHING) alpha = fmin2(alpha0 + (1.0f - alpha0) * beta, 1.0f);
                else alpha = 1.0f;
                c->toi = alpha;
                c->toiFlag = 1;
            }
            if (alpha < minAlpha) { minContact = c; minAlpha = alpha; }
        }
        if (minContact == 0 || 1.0f - 10.0f * b2_epsilon < minAlpha) break;

        b2body *bA = &w->bodies[0], *bB = &w->bodies[minContact->body];
        b2sweep backup1 = bA->sw, backup2 = bB->sw;
        body_advance(bA, minAlpha);
        body_advance(bB, minAlpha);
        contact_update(w, minContact);
        minContact->toiFlag = 0;
        ++minContact->toiCount;
        if (!minContact->enabled || !minContact->touching) {
            STAT(g_b2_stats.cur_rejected++);
            minContact->enabled = 0;
            bA->sw = backup1; bB->sw = backup2;
            body_sync_transform(bA); body_sync_transform(bB);
            continue;
        }
        b2b_set_awake(w, 0, 1);
        b2b_set_awake(w, minContact->body, 1);

        /* island: bodies [ground, dyn]; contacts: minContact + the dynamic body's other touching contacts */
        int dynBody = minContact->body;
        b2contact *islandContacts[B2_N_CONTACTS]; int nc = 0;
        islandContacts[nc++] = minContact;
        bA->islandFlag = 1; bB->islandFlag = 1; minContact->islandFlag = 1;
        for (int k = 0; k < n_order; ++k) { /* body B's contact edges, newest first (body A is static: skipped) */
            b2contact *c = &w->contacts[order[k]];
            if (!c->exists || c->body != dynBody) continue;
            if (nc == b2_maxTOIContacts) break;
            if (c->islandFlag) continue;
            /* other = ground (static): allowed; already in the island, so no tentative advance */
            contact_update(w, c);
            if (!c->enabled) continue;
            if (!c->touching) continue;
            c->islandFlag = 1;
            islandContacts[nc++] = c;
        }

        /* ---- b2Island::SolveTOI(subStep, toiIndexA = 0, toiIndexB = 1) ---- */
        float sub_dt = (1.0f - minAlpha) * dt;
        int islandIndex[B2_N_BODIES] = {0, 1, 1, 1};
        b2pos positions[2]; b2vel velocities[2];
        positions[0].c = bA->sw.c; positions[0].a = bA->sw.a; velocities[0].v = bA->linvel; velocities[0].w = bA->angvel;
        positions[1].c = bB->sw.c; positions[1].a = bB->sw.a; velocities[1].v = bB->linvel; velocities[1].w = bB->angvel;
        csolver cs2;
        csolver *s = &cs2;
        cs_init(s, w, islandContacts, nc, positions, velocities, islandIndex, 0, 1.0f);
        STAT(g_b2_stats.cur_substeps++; g_b2_stats.toi_island_contacts[nc < 15 ? nc : 15]++);
        for (int i = 0; i < 20; ++i) {
            STAT(g_b2_stats.toi_pos_iters++);
            int contactsOkay = cs_solve_position(s, 1, 0, 1);
            if (contactsOkay) break;
        }
        bA->sw.c0 = positions[0].c; bA->sw.a0 = positions[0].a;
        bB->sw.c0 = positions[1].c; bB->sw.a0 = positions[1].a;
        cs_init_velocity(s);
        for (int i = 0; i < velIters; ++i) cs_solve_velocity(s);
        float h = sub_dt;
        b2body *ib[2] = {bA, bB};
        for (int i = 0; i < 2; ++i) {
            b2v c = positions[i].c; float a = positions[i].a; b2v v = velocities[i].v; float wv = velocities[i].w;
            b2v translation = smul(h, v);
            if (dot(translation, translation) > b2_maxTranslationSquared) {
                float ratio = b2_maxTranslation / vlen(translation);
                v = smul(ratio, v);
            }
            float rotation = h * wv;
            if (rotation * rotation > b2_maxRotationSquared) {
                float ratio = b2_maxRotation / fabs1(rotation);
                wv *= ratio;
            }
            c = vadd(c, smul(h, v));
            a += h * wv;
            positions[i].c = c; positions[i].a = a; velocities[i].v = v; velocities[i].w = wv;
            ib[i]->sw.c = c; ib[i]->sw.a = a; ib[i]->linvel = v; ib[i]->angvel = wv;
            body_sync_transform(ib[i]);
        }

        /* reset island flags, synchronize the displaced dynamic body, invalidate its contact TOIs */
        bA->islandFlag = 0; bB->islandFlag = 0;
        body_sync_fixtures(w, dynBody);
        for (int i = 0; i < B2_N_CONTACTS; ++i) {
            b2contact *c = &w->contacts[i];
            if (c->exists && c->body == dynBody) { c->toiFlag = 0; c->islandFlag = 0; }
        }
        find_new_contacts(w);
    }
}

/* ------------------------------------------------------------------------- world API */
void b2w_init(b2world *w, b2v gravity) {
    memset(w, 0, sizeof(*w));
    w->gravity = gravity;
    b2body *g = &w->bodies[0]; /* static ground at the origin (lunar_lander.rs:777-781) */
    g->type = B2_STATIC;
    g->xf.q = rot_set(0.0f);
    g->awake = 1;
    g->userdata = 0;
    w->next_seq = 1;
}

void b2w_add_ground_edge(b2world *w, b2v v1, b2v v2, float friction) {
    int e = w->n_edges++;
    w->edges[e].v1 = v1; w->edges[e].v2 = v2; w->edges[e].radius = b2_polygonRadius; /* SetTwoSided */
    w->edge_friction = friction;
    create_proxy(w, e, edge_aabb(&w->edges[e]));
    w->newContacts = 1;
}

int b2w_add_body(b2world *w, b2v pos, float angle, int userdata) { /* b2World::CreateBody, dynamic */
    int bi = ++w->n_dyn;
    b2body *b = &w->bodies[bi];
    memset(b, 0, sizeof(*b));
    b->type = B2_DYNAMIC;
    b->xf.p = pos; b->xf.q = rot_set(angle);
    b->sw.localCenter = V(0.0f, 0.0f);
    b->sw.c0 = pos; b->sw.c = pos; b->sw.a0 = angle; b->sw.a = angle; b->sw.alpha0 = 0.0f;
    b->awake = 1;
    b->mass = 1.0f; b->invMass = 1.0f;
    b->userdata = userdata;
    return bi;
}

/* b2Body::CreateFixture + ResetMassData for the single polygon fixture of a dynamic body */
static void attach_polygon(b2world *w, int bi, float density, float friction) {
    b2body *b = &w->bodies[bi];
    w->poly_density[bi - 1] = density; w->poly_friction[bi - 1] = friction;
    create_proxy(w, 10 + bi, poly_aabb(&w->polys[bi - 1], b->xf));
    /* ResetMassData */
    float mass, I; b2v center;
    poly_mass(&w->polys[bi - 1], density, &mass, &center, &I);
    b->mass = mass; b->I = I;
    b2v localCenter = smul(mass, center);
    b->invMass = 1.0f / b->mass;
    localCenter = smul(b->invMass, localCenter);
    b->I -= b->mass * dot(localCenter, localCenter);
    b->invI = 1.0f / b->I;
    b2v oldCenter = b->sw.c;
    b->sw.localCenter = localCenter;
    b->sw.c0 = b->sw.c = xmul(b->xf, b->sw.localCenter);
    b->linvel = vadd(b->linvel, cross_sv(b->angvel, vsub(b->sw.c, oldCenter)));
    w->newContacts = 1;
}

void b2w_set_polygon(b2world *w, int bi, const b2v *pts, int n, float density, float friction) {
    poly_set(&w->polys[bi - 1], pts, n);
    attach_polygon(w, bi, density, friction);
}
void b2w_set_box(b2world *w, int bi, float hx, float hy, float density, float friction) {
    poly_set_box(&w->polys[bi - 1], hx, hy);
    attach_polygon(w, bi, density, friction);
}

void b2w_add_revolute(b2world *w, int bodyB, b2v anchorA, b2v anchorB, float lower, float upper, float maxMotorTorque,
                      float motorSpeed) {
    b2revolute *j = &w->joints[w->n_joints++];
    memset(j, 0, sizeof(*j));
    j->bodyA = 1; j->bodyB = bodyB;
    j->localAnchorA = anchorA; j->localAnchorB = anchorB;
    j->referenceAngle = 0.0f; /* b2RevoluteJointDef default: the reference fills the def by hand (lunar_lander.rs:887-902) */
    j->lowerAngle = lower; j->upperAngle = upper;
    j->maxMotorTorque = maxMotorTorque; j->motorSpeed = motorSpeed;
    j->enableLimit = 1; j->enableMotor = 1;
}

void b2b_set_transform(b2world *w, int bi, b2v pos, float angle) { /* b2Body::SetTransform */
    b2body *b = &w->bodies[bi];
    b->xf.q = rot_set(angle);
    b->xf.p = pos;
    b->sw.c = xmul(b->xf, b->sw.localCenter);
    b->sw.a = angle;
    b->sw.c0 = b->sw.c;
    b->sw.a0 = angle;
    fixture_sync(w, bi, b->xf, b->xf);
    w->newContacts = 1;
}
void b2b_set_linear_velocity(b2world *w, int bi, b2v v) {
    b2body *b = &w->bodies[bi];
    if (b->type == B2_STATIC) return;
    if (dot(v, v) > 0.0f) b2b_set_awake(w, bi, 1);
    b->linvel = v;
}
void b2b_set_angular_velocity(b2world *w, int bi, float wv) {
    b2body *b = &w->bodies[bi];
    if (b->type == B2_STATIC) return;
    if (wv * wv > 0.0f) b2b_set_awake(w, bi, 1);
    b->angvel = wv;
}
void b2b_apply_force_to_center(b2world *w, int bi, b2v f, int wake) {
    b2body *b = &w->bodies[bi];
    if (b->type != B2_DYNAMIC) return;
    if (wake && !b->awake) b2b_set_awake(w, bi, 1);
    if (b->awake) b->force = vadd(b->force, f);
}
void b2b_apply_torque(b2world *w, int bi, float t, int wake) {
    b2body *b = &w->bodies[bi];
    if (b->type != B2_DYNAMIC) return;
    if (wake && !b->awake) b2b_set_awake(w, bi, 1);
    if (b->awake) b->torque += t;
}
void b2b_apply_linear_impulse(b2world *w, int bi, b2v impulse, b2v point, int wake) {
    b2body *b = &w->bodies[bi];
    if (b->type != B2_DYNAMIC) return;
    if (wake && !b->awake) b2b_set_awake(w, bi, 1);
    if (b->awake) {
        b->linvel = vadd(b->linvel, smul(b->invMass, impulse));
        b->angvel += b->invI * cross(vsub(point, b->sw.c), impulse);
    }
}

/* b2World::Step */
void b2w_step(b2world *w, float dt, int velIters, int posIters) {
    if (w->newContacts) { find_new_contacts(w); w->newContacts = 0; }
    float inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
    float dtRatio = w->inv_dt0 * dt;
#ifdef ORA_STATS
    int stat_nc = 0;
    for (int i = 0; i < B2_N_CONTACTS; ++i) stat_nc += w->contacts[i].exists ? 1 : 0;
    g_b2_stats.steps++;
    g_b2_stats.cur_toi_calls = g_b2_stats.cur_substeps = g_b2_stats.cur_rejected = 0;
#endif
    collide(w);
    if (dt > 0.0f) solve_island(w, dt, inv_dt, dtRatio, velIters, posIters);
    if (dt > 0.0f) solve_toi(w, dt, velIters);
    if (dt > 0.0f) w->inv_dt0 = inv_dt;
#ifdef ORA_STATS
    if (stat_nc > 0) {
        g_b2_stats.steps_with_contacts++;
        g_b2_stats.hist_contacts[stat_nc < 15 ? stat_nc : 15]++;
        g_b2_stats.hist_toi_calls[g_b2_stats.cur_toi_calls < 63 ? g_b2_stats.cur_toi_calls : 63]++;
        g_b2_stats.hist_substeps[g_b2_stats.cur_substeps < 31 ? g_b2_stats.cur_substeps : 31]++;
        g_b2_stats.hist_rejected[g_b2_stats.cur_rejected < 31 ? g_b2_stats.cur_rejected : 31]++;
    }
#endif
    for (int b = 1; b <= w->n_dyn; ++b) { w->bodies[b].force = V(0.0f, 0.0f); w->bodies[b].torque = 0.0f; } /* ClearForces */
}
