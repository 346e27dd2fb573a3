/* b2mini.h — the subset of Box2D that LunarLanderV3 exercises, restated in plain C.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference's LunarLander arithmetic lives in the third-party crate box2d-rs 0.0.4
 * (Cargo.toml:11, Cargo.lock:84-90), a Rust port of Erin Catto's Box2D v2.4.x; its source is
 * NOT in /root/reference and cannot be fetched.  This file restates the PUBLISHED Box2D v2.4.1
 * algorithms for exactly the call sites the reference uses (lunar_lander.rs:735-905,949-957,
 * 999-1047,1066,1095-1098,1153 and the test seam :1371-1374,1434-1435,1481-1535):
 *   b2PolygonShape::Set / SetAsBox / ComputeMass, b2EdgeShape::SetTwoSided, ComputeAABB,
 *   b2Body (mass data, SetTransform, velocities, forces/impulses, sleep),
 *   broad-phase fat AABBs (b2DynamicTree::MoveProxy rules; pair search is brute force),
 *   b2ContactManager (AddPair / Collide / b2Contact::Update with Begin/EndContact),
 *   b2CollideEdgeAndPolygon (v2.4.1 form), b2ClipSegmentToLine, b2WorldManifold,
 *   b2ContactSolver (warm start, block solver, position correction, TOI variant),
 *   b2RevoluteJoint (motor + lower/upper limit impulses + point constraint),
 *   b2Island::Solve / SolveTOI, b2World::Step / Solve / SolveTOI,
 *   b2Distance (GJK) and b2TimeOfImpact (conservative advancement).
 * Parity pin for all of it: only the reference's LunarLander golden trajectory at its own
 * tolerances (obs 0.2, reward 5.0; lunar_lander.rs:1647-1655).  Where Box2D's result depends on
 * dynamic-tree traversal order (order of newly created contacts) this restatement uses a fixed
 * rule (moved proxies in body order, partners by ascending proxy id) — documented deviation.
 */
#ifndef MGYM_B2MINI_H
#define MGYM_B2MINI_H

#include <stdint.h>

typedef struct { float x, y; } b2v;
typedef struct { float s, c; } b2rot;
typedef struct { b2v p; b2rot q; } b2xf;
typedef struct { b2v localCenter, c0, c; float a0, a, alpha0; } b2sweep;
typedef struct { b2v lo, hi; } b2aabb;

#define B2_MAX_POLY 8
typedef struct { b2v v[B2_MAX_POLY], n[B2_MAX_POLY], centroid; int count; float radius; } b2poly;
typedef struct { b2v v1, v2; float radius; } b2edge;

enum { B2_STATIC = 0, B2_DYNAMIC = 2 };

typedef struct {
    int type;
    b2xf xf;
    b2sweep sw;
    b2v linvel; float angvel;
    b2v force; float torque;
    float mass, invMass, I, invI;
    float sleepTime;
    int awake, islandFlag, islandIndex;
    int userdata;
} b2body;

typedef struct { uint8_t indexA, indexB, typeA, typeB; } b2cf; /* b2ContactFeature; key = the 4 bytes */
typedef struct { b2v localPoint; float normalImpulse, tangentImpulse; b2cf id; } b2mpoint;
enum { B2_FACE_A = 1, B2_FACE_B = 2 };
typedef struct { b2mpoint points[2]; b2v localNormal, localPoint; int type, pointCount; } b2manifold;

typedef struct {
    int exists, touching, enabled, islandFlag, toiFlag;
    uint32_t seq;          /* creation order: Box2D pushes new contacts at the list head */
    int body, edge;        /* dynamic body 1..3, ground edge 0..10 */
    b2manifold m;
    float friction, restitution, restitutionThreshold;
    int toiCount; float toi;
} b2contact;

typedef struct {
    int bodyA, bodyB;      /* A = lander (1), B = leg */
    b2v localAnchorA, localAnchorB;
    float referenceAngle, lowerAngle, upperAngle, maxMotorTorque, motorSpeed;
    int enableLimit, enableMotor;
    b2v impulse; float motorImpulse, lowerImpulse, upperImpulse;
    /* solver temp */
    b2v rA, rB, localCenterA, localCenterB;
    float invMassA, invMassB, invIA, invIB;
    float K11, K12, K21, K22, angle, axialMass;
} b2revolute;

#define B2_N_EDGES 11
#define B2_N_BODIES 4   /* 0 ground, 1 lander, 2 leg0, 3 leg1 */
#define B2_N_PROXIES 14 /* 0..10 ground edges, 11 lander, 12 leg0, 13 leg1 */
#define B2_N_CONTACTS 33

typedef struct {
    void (*begin)(void *ud, int bodyA_userdata, int bodyB_userdata);
    void (*end)(void *ud, int bodyA_userdata, int bodyB_userdata);
    void *ud;
} b2listener;

typedef struct {
    b2v gravity;
    b2body bodies[B2_N_BODIES];
    b2edge edges[B2_N_EDGES];
    float edge_friction;
    b2poly polys[3];                 /* fixture of body 1..3 */
    float poly_friction[3], poly_density[3];
    int n_dyn;                       /* dynamic bodies created so far */
    int n_edges;
    b2aabb fat[B2_N_PROXIES];        /* broad-phase (tree) AABBs */
    int proxy_exists[B2_N_PROXIES];
    int moved[B2_N_PROXIES];
    int move_buffer[64]; int move_count;
    b2contact contacts[B2_N_CONTACTS];
    uint32_t next_seq;
    b2revolute joints[2]; int n_joints;
    int newContacts;
    float inv_dt0;
    b2listener listener;
} b2world;

void b2w_init(b2world *w, b2v gravity);
void b2w_add_ground_edge(b2world *w, b2v v1, b2v v2, float friction);
int b2w_add_body(b2world *w, b2v pos, float angle, int userdata);              /* dynamic; returns body index */
void b2w_set_polygon(b2world *w, int body, const b2v *pts, int n, float density, float friction);
void b2w_set_box(b2world *w, int body, float hx, float hy, float density, float friction);
void b2w_add_revolute(b2world *w, int bodyB, b2v anchorA, b2v anchorB, float lower, float upper,
                      float maxMotorTorque, float motorSpeed);
void b2w_step(b2world *w, float dt, int velIters, int posIters);

void b2b_set_transform(b2world *w, int body, b2v pos, float angle);
void b2b_set_linear_velocity(b2world *w, int body, b2v v);
void b2b_set_angular_velocity(b2world *w, int body, float wv);
void b2b_set_awake(b2world *w, int body, int flag);
void b2b_apply_force_to_center(b2world *w, int body, b2v f, int wake);
void b2b_apply_torque(b2world *w, int body, float t, int wake);
void b2b_apply_linear_impulse(b2world *w, int body, b2v impulse, b2v point, int wake);

#endif
