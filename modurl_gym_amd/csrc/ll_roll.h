// ll_roll.h — mgym_rollout for LunarLanderV3: K consecutive Gym::step calls of every environment in ONE persistent launch, each
// environment advancing as soon as IT is ready (reference loop: a trainer calling step() K times, src/box_2d/lunar_lander.rs:919-1167,
// with the reset of :727-917 between episodes under MGYM_FLAG_AUTO_RESET).
//
// Why: mgym_step is lock-step — a step ends when the slowest contact block ends (one environment's island solve + up to four
// time-of-impact sub-steps, ~0.88 ms), the ~700 environments per step that the free-flight role hands over travel through the contact
// path one lane per wave, and contact blocks run 25 lanes wide so that their chains stay short.  mgym_rollout's contract only fixes
// per-environment results ([K][n] actions in, [K][n] outputs out), so here nothing waits for anything but its own data:
//   * environments in free flight stay RESIDENT in a wave's registers from step to step (no state traffic at all, like the CartPole
//     rollout); a lane whose environment leaves (contact ahead, episode over, K reached) is refilled from the free-flight queue;
//   * environments that need the contact path are queued and taken 32 at a time — full blocks, whatever step index each lane is at —
//     through the same Collide / island / SolveTOI code as ll_step_kernel's contact role (hand-overs included: they are just queue
//     entries, resumed after their island solve);
//   * finished environments are queued for reset() (scene + implicit step(0)) and come back as free-flight entries.
// Every wave of the grid serves all three queues (one register allocation, one wave per SIMD, as ll_step_kernel).
//
// Queues: three multi-producer / multi-consumer rings in device memory (Vyukov sequence slots: {sequence, entry} in ONE 8-byte word, agent-scope
// atomics both sides), entry = step index << 24 | environment.  An environment is in at most one queue, so a ring of >= n slots never fills.
//
// Visibility between compute units inside the launch (per-XCD L2s are not coherent, a CU's vector L1 is never refreshed by another
// CU's stores — MI355X_MICROARCH.md "Workgroup dispatch"): placement-independent, never by co-location.
//   contact / reset batches : pop -> agent-scope ACQUIRE fence -> plain loads ... plain stores -> every lane's vmcnt(0) -> agent-scope
//                             RELEASE fence -> vmcnt(0) -> push   (one fence pair per ~0.5 ms batch)
//   free-flight residents   : refill = pop -> acquire fence -> 16-byte sc1 loads of the record's two lines; leaving = 16-byte sc1
//                             (write-through) stores -> vmcnt(0) -> push; terrain reads are agent-scope loads (ll_free.h, COH)
// Outputs ([K][n] reward / flags / observations) have exactly one writer each (the wave that finishes the step; the observation of a
// finished environment under auto-reset is written by the wave that resets it) and are only read after the launch.
// Every spin is bounded (slot waits, idle polls): on a timeout the wave raises the launch's abort word and a sticky internal error.
#pragma once
#include "ll_kernel_common.h"
#include "ll_roll_types.h"

namespace mgym {

// entry -> environment.  The value goes through an empty asm: hipcc (ROCm 7.2, clang 22) otherwise turns `(uint64_t)(ent & 0xffffff) * record
// bytes` into a 24-bit multiply, drops the mask as redundant for it and then re-forms a full v_mad_u64_u32 on the UNMASKED word — records of
// entries with a step index > 0 were addressed 16 Mi records too far (found with rocgdb on the faulting wave: ll_load's first load).
LLD uint32_t roll_env_of(uint32_t ent) { uint32_t i = ent & kRollEnvMask; asm volatile("" : "+v"(i)); return i; }
#define RQ_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RQ_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
LLD uint32_t* rq_ctl(const RollQ& q, int w) { return q.ctl + 32 * w; }
LLD unsigned long long* rq_slot(const RollQ& q, int which, uint32_t pos) { return q.ring + (uint64_t)which * ((uint64_t)q.mask + 1ull) + (pos & q.mask); }
LLD void rq_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
LLD bool rq_aborted(const RollQ& q) { return RQ_LOAD(rq_ctl(q, RC_ABORT)) != 0u; }
LLD void rq_abort(const LLDev& d, const RollQ& q) { atomicOr(rq_ctl(q, RC_ABORT), 1u); atomicOr(d.err, DEV_ERR_INTERNAL); }

// what waits in the four queues and how many environments are still to finish: ONE load instruction (lane w < 5 reads control word w)
struct RollCounters { int avail[RQ_COUNT]; uint32_t live; int headt[RQ_COUNT]; };
LLD RollCounters rq_counters(const RollQ& q) {
    const int lane = threadIdx.x & 63;
    uint32_t v = 0u;
    if (lane < RC_HEADT + RQ_COUNT) v = RQ_LOAD(rq_ctl(q, lane));
    RollCounters r;
    r.headt[0] = __builtin_amdgcn_readlane((int)v, RC_HEADT + 0); r.headt[1] = __builtin_amdgcn_readlane((int)v, RC_HEADT + 1);
    r.headt[2] = __builtin_amdgcn_readlane((int)v, RC_HEADT + 2); r.headt[3] = __builtin_amdgcn_readlane((int)v, RC_HEADT + 3);
    r.headt[4] = __builtin_amdgcn_readlane((int)v, RC_HEADT + 4); r.avail[4] = __builtin_amdgcn_readlane((int)v, 4);
    r.avail[0] = __builtin_amdgcn_readlane((int)v, 0); r.avail[1] = __builtin_amdgcn_readlane((int)v, 1); r.avail[2] = __builtin_amdgcn_readlane((int)v, 2);
    r.avail[3] = __builtin_amdgcn_readlane((int)v, 3);
    r.live = (uint32_t)__builtin_amdgcn_readlane((int)v, RC_LIVE);
    return r;
}
// what a subtraction of `want` from the semaphore (which held `old` before) has really obtained; the rest goes back
LLD int rq_sem_got(const RollQ& q, int which, int old, int want) {
    const int got = old <= 0 ? 0 : old < want ? old : want;
    if (got < want && (threadIdx.x & 63) == 0) atomicAdd(rq_ctl(q, RC_AVAIL + which), (uint32_t)(want - got));
    return got;
}
// the entries at ring positions [h, h + m): lane l < m takes entry h + l out of its slot (waiting, briefly, for a producer that has been
// handed the position but has not written yet)
LLD uint32_t rq_take(const LLDev& d, const RollQ& q, int which, uint32_t h, int m) {
    const int lane = threadIdx.x & 63;
    uint32_t ent = 0xffffffffu;
    if (lane < m) {
        unsigned long long* const s = rq_slot(q, which, h + (uint32_t)lane);
        const uint32_t want_seq = h + (uint32_t)lane + 1u;
        unsigned long long v = RQ_LOAD(s);
        if ((uint32_t)(v >> 32) != want_seq) {
            const long long t0 = wall_clock64();
            do { __builtin_amdgcn_s_sleep(2); v = RQ_LOAD(s); } while ((uint32_t)(v >> 32) != want_seq && wall_clock64() - t0 < kRollTimeoutTicks && !rq_aborted(q));
        }
        if ((uint32_t)(v >> 32) == want_seq) { ent = (uint32_t)v; RQ_STORE(s, (unsigned long long)(want_seq + q.mask) << 32); }   // free for position pos + cap
        else rq_abort(d, q);
        if (lane == 0 && ent != 0xffffffffu) RQ_STORE(rq_ctl(q, RC_HEADT + which), ent >> 24);   // how far this queue has got
    }
    return ent;
}
// Take up to `want` entries of queue `which`: wave-uniform result m, lane l < m gets its entry.  Two atomics with a result, one after the other.
LLD int rq_pop(const LLDev& d, const RollQ& q, int which, int want, uint32_t& ent) {
    int old = 0;
    if ((threadIdx.x & 63) == 0) old = (int)atomicSub(rq_ctl(q, RC_AVAIL + which), (uint32_t)want);
    old = __builtin_amdgcn_readfirstlane(old);
    const int m = rq_sem_got(q, which, old, want);
    ent = 0xffffffffu;
    if (m == 0) return 0;
    uint32_t h = 0u;
    if ((threadIdx.x & 63) == 0) h = atomicAdd(rq_ctl(q, RC_HEAD + which), (uint32_t)m);
    h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    ent = rq_take(d, q, which, h, m);
    return m;
}
// Append (wave-aggregated).  The caller has made the environment's state visible first (file header).
LLD void rq_push(const LLDev& d, const RollQ& q, int which, bool want, uint32_t ent) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(rq_ctl(q, RC_TAIL + which), (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (want) {
        const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        unsigned long long* const s = rq_slot(q, which, pos);
        if ((uint32_t)(RQ_LOAD(s) >> 32) != pos) {   // (the ring holds >= n slots and an environment is queued at most once: its last user is long gone)
            const long long t0 = wall_clock64();
            while ((uint32_t)(RQ_LOAD(s) >> 32) != pos && wall_clock64() - t0 < kRollTimeoutTicks && !rq_aborted(q)) __builtin_amdgcn_s_sleep(2);
        }
        if ((uint32_t)(RQ_LOAD(s) >> 32) == pos) RQ_STORE(s, ((unsigned long long)(pos + 1u) << 32) | ent);
        else rq_abort(d, q);
    }
    if (lane == leader) atomicAdd(rq_ctl(q, RC_AVAIL + which), (uint32_t)__popcll(mask));   // published (a taker that is faster than the slot store waits at the slot)
}
LLD void rq_retire(const RollQ& q, bool want) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;
    if ((int)(threadIdx.x & 63) == __ffsll((long long)mask) - 1) atomicSub(rq_ctl(q, RC_LIVE), (uint32_t)__popcll(mask));
}

// ---- the two leading lines of a record (words 0..55: everything a free-flight step touches) moved as 14 sixteen-byte accesses that
// bypass / write through the non-coherent caches (aux 16 = sc1)
struct RollRec { uint32_t w[56]; };
LLD void roll_rec_load(const __amdgpu_buffer_rsrc_t rs, uint32_t i, RollRec& r) {
    const uint32_t off = i * (uint32_t)(kRec * 4);
#pragma unroll
    for (int k = 0; k < 14; ++k) {
        const ll_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u * k, 0, 16);
        r.w[4 * k] = v.x; r.w[4 * k + 1] = v.y; r.w[4 * k + 2] = v.z; r.w[4 * k + 3] = v.w;
    }
}
LLD void roll_rec_store(const __amdgpu_buffer_rsrc_t rs, uint32_t i, const RollRec& r) {
    const uint32_t off = i * (uint32_t)(kRec * 4);
#pragma unroll
    for (int k = 0; k < 14; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(ll_u32x4{r.w[4 * k], r.w[4 * k + 1], r.w[4 * k + 2], r.w[4 * k + 3]}, rs, off + 16u * k, 0, 16);
}
// ll_free_load / ll_free_store / ll_free_store_resume (ll_free.h) on such a copy: the same words, the same meaning
LLD void roll_unpack(const LLDev& d, const RollRec& r, FreeRegs& f, EnvRegs& e) {
    const uint32_t flags = r.w[C_FLAGS];
    f.flags = flags;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        Body& bd = f.b[b];
        const int c = C_BODY + 9 * b;
        bd.xf.p = mk(as_f32(r.w[c + 0]), as_f32(r.w[c + 1]));
        bd.sw.c = mk(as_f32(r.w[c + 2]), as_f32(r.w[c + 3]));
        bd.sw.a = as_f32(r.w[c + 4]);
        bd.v = mk(as_f32(r.w[c + 5]), as_f32(r.w[c + 6]));
        bd.w = as_f32(r.w[c + 7]);
        bd.sleepTime = as_f32(r.w[c + 8]);
        bd.sw.localCenter = LLK(d).localCenter[b == 0 ? 0 : 1];
        bd.sw.c0 = bd.sw.c; bd.sw.a0 = bd.sw.a; bd.sw.alpha0 = 0.0f;
        bd.force = mk(0.0f, 0.0f); bd.torque = 0.0f;
        bd.awake = true; bd.islandFlag = false;
        f.fat[b].lo = mk(as_f32(r.w[C_FAT + 4 * b + 0]), as_f32(r.w[C_FAT + 4 * b + 1]));
        f.fat[b].hi = mk(as_f32(r.w[C_FAT + 4 * b + 2]), as_f32(r.w[C_FAT + 4 * b + 3]));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = C_JOINT + 5 * j;
        f.jt[j].impulse = mk(as_f32(r.w[c + 0]), as_f32(r.w[c + 1]));
        f.jt[j].motorImpulse = as_f32(r.w[c + 2]);
        f.jt[j].lowerImpulse = as_f32(r.w[c + 3]);
        f.jt[j].upperImpulse = as_f32(r.w[c + 4]);
    }
    e.prev_shaping = as_f32(r.w[C_PREV]);
    e.prev_some = flags & F_PREV_SOME;
    e.wind_idx = (int32_t)r.w[C_WIND]; e.torque_idx = (int32_t)r.w[C_TORQUE];
    e.step = r.w[C_STEP]; e.episode = r.w[C_EPISODE];
    e.has_world = true;
    e.deterministic = flags & F_DETERMINISTIC;
}
LLD void roll_pack_common(RollRec& r, const FreeRegs& f, uint32_t& awake_bits) {
    awake_bits = 0u;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const Body& bd = f.b[b];
        const int c = C_BODY + 9 * b;
        r.w[c + 0] = as_u32(bd.xf.p.x); r.w[c + 1] = as_u32(bd.xf.p.y);
        r.w[c + 2] = as_u32(bd.sw.c.x); r.w[c + 3] = as_u32(bd.sw.c.y);
        r.w[c + 4] = as_u32(bd.sw.a);
        r.w[c + 5] = as_u32(bd.v.x); r.w[c + 6] = as_u32(bd.v.y);
        r.w[c + 7] = as_u32(bd.w);
        r.w[c + 8] = as_u32(bd.sleepTime);
        if (bd.awake) awake_bits |= 1u << b;
        r.w[C_FAT + 4 * b + 0] = as_u32(f.fat[b].lo.x); r.w[C_FAT + 4 * b + 1] = as_u32(f.fat[b].lo.y);
        r.w[C_FAT + 4 * b + 2] = as_u32(f.fat[b].hi.x); r.w[C_FAT + 4 * b + 3] = as_u32(f.fat[b].hi.y);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = C_JOINT + 5 * j;
        r.w[c + 0] = as_u32(f.jt[j].impulse.x); r.w[c + 1] = as_u32(f.jt[j].impulse.y);
        r.w[c + 2] = as_u32(f.jt[j].motorImpulse); r.w[c + 3] = as_u32(f.jt[j].lowerImpulse); r.w[c + 4] = as_u32(f.jt[j].upperImpulse);
    }
}
// the flag word ll_free_store writes after a finished free-flight step
LLD uint32_t roll_flags_after_step(const FreeRegs& f, const EnvRegs& e) {
    uint32_t flags = f.flags & (F_GAME_OVER | F_LEG0 | F_LEG1 | F_HAS_WORLD | F_DETERMINISTIC);
#pragma unroll
    for (int b = 0; b < 3; ++b) if (f.b[b].awake) flags |= 1u << b;
    return flags | F_STEPPED | (e.prev_some ? F_PREV_SOME : 0u);
}
// ≙ ll_free_store (`seq_word`: C_SEQ shares a 16-byte piece with the counters and is carried along unchanged)
LLD void roll_pack_store(RollRec& r, const FreeRegs& f, const EnvRegs& e, uint32_t flags, uint32_t seq_word) {
    uint32_t awake_bits;
    roll_pack_common(r, f, awake_bits);
    r.w[C_FLAGS] = flags;
    r.w[C_PREV] = as_u32(e.prev_shaping);
    r.w[C_WIND] = (uint32_t)e.wind_idx; r.w[C_TORQUE] = (uint32_t)e.torque_idx;
    r.w[C_STEP] = e.step; r.w[C_EPISODE] = e.episode;
    r.w[C_SEQ] = seq_word;
}

// what a lane of a wave in free-flight mode carries from step to step
struct RollLane {
    bool have;        // an environment lives in this lane
    bool fresh;       // ... and has not been stepped since it was loaded (the record in HBM is its current state)
    uint32_t i, t;    // environment, index of its NEXT step
    uint32_t act;     // ... and that step's action, fetched a step ahead (before the previous step's outputs are stored: a load behind them waits for them)
    uint32_t seq_word;
    FreeRegs f; EnvRegs e; FreeSolve s;
};

LLD void roll_write_obs(const LLDev& d, const LLIo& io, uint32_t t, uint32_t i, const float state[8], bool last) {
    if (io.obs_out) {
#pragma unroll
        for (int q = 0; q < 8; ++q) io.obs_out[((uint64_t)t * 8u + (uint64_t)q) * d.n + i] = state[q];
    }
    if (last) {
#pragma unroll
        for (int q = 0; q < 8; ++q) d.obs[(uint64_t)q * d.n_pad + i] = state[q];
    }
}
LLD void roll_write_scalars(const LLDev& d, const LLIo& io, uint32_t t, uint32_t i, float reward, uint32_t done) {
    const uint64_t o = (uint64_t)t * d.n + i;
    if (io.rew) io.rew[o] = reward;
    if (io.done_out) io.done_out[o] = (uint8_t)done;
    if (io.trunc_out) io.trunc_out[o] = 0;  // :1165 truncated: false
}

// where an environment goes after step t has been finished with flag word `flags` (t1 = t + 1):
//   RESET  finished and auto-reset is on (also when t1 == K: the reset belongs to step t)
//   retire t1 == K
//   FREE / CONTACT by the flag word, as ll_epilogue_kernel classes the next step's population
enum { ROLL_NONE = 0, ROLL_TO_FREE, ROLL_TO_CONTACT, ROLL_TO_RESET, ROLL_TO_LIGHT, ROLL_TO_TOI, ROLL_RETIRE };
LLD int roll_route(const LLDev& d, const RollQ& q, uint32_t flags, uint32_t done, uint32_t t1) {
    if (done && d.auto_reset) return ROLL_TO_RESET;
    if (t1 >= q.K) return ROLL_RETIRE;
    return ll_free_eligible(flags) ? ROLL_TO_FREE : (flags & F_TOUCHING) ? ROLL_TO_CONTACT : ROLL_TO_LIGHT;
}
LLD void roll_dispatch(const LLDev& d, const RollQ& q, int route, uint32_t i, uint32_t t1) {
    const uint32_t ent = (t1 << 24) | i;
    rq_push(d, q, RQ_FREE, route == ROLL_TO_FREE, ent);
    rq_push(d, q, RQ_CONTACT, route == ROLL_TO_CONTACT, ent);
    rq_push(d, q, RQ_RESET, route == ROLL_TO_RESET, ent);
    rq_push(d, q, RQ_LIGHT, route == ROLL_TO_LIGHT, ent);
    rq_push(d, q, RQ_TOI, route == ROLL_TO_TOI, ent);
    rq_retire(q, route == ROLL_RETIRE);
}

// ---- contact batch: `m` entries in S.late, one per lane; the body of ll_contact_body<BLK> for one pass with a step index per lane.
// keep_which >= 0 (the queue the entries came from): an environment that ends its step in the SAME class stays in its lane for its next
// step — its record goes through memory as always (ll_store / ll_load by the same lane: nothing to publish, nothing to acquire) but not
// through the queue — and lanes that fall vacant are refilled from that queue, up to max_lanes on board.  K consecutive touching steps
// of one environment are the launch's longest chain: this takes the queue hand-over (a wave in free-flight mode looks at the queues once
// per step) out of every link of it.
template <int BLK>
__device__ __forceinline__ void roll_contact_batch(const LLDev& d, const LLIo& io, const RollQ& q, int m, ContactLds<BLK>& S, VConstraint* far_lane0, int far_stride,
                                                   uint32_t& overflow, uint32_t& finished, int keep_which, int max_lanes, unsigned long long& n_batches, unsigned long long& n_lanes,
                                                   bool resume_toi /* the entries stand at a sub-step (RQ_TOI) */, int toi_budget /* sub-steps this visit takes (< 0: all) */) {
    static_assert(BLK <= 32, "rollout: World records in LDS");
    constexpr int kThreads = ll_contact_threads(BLK);
    const PolyTab& tab = S.tab;
    const bool env_lane = threadIdx.x < BLK;
    const int own = env_lane ? (int)threadIdx.x : 0;
    const int lane = threadIdx.x & 63;
    const CtHot hot{(LL_LDS uint32_t*)S.hot + own, (uint32_t)BLK, 1u};
    PConstraint l_pc[kSolverCap];
    CSolverMem mem;
    mem.vc = &S.vc[own]; mem.vc_stride = BLK; mem.vc_near = d.vc_near < kVcNearLds ? d.vc_near : kVcNearLds; mem.pc = l_pc; mem.pc_stride = 1; mem.cap = kSolverCap;
    mem.vc_far = far_lane0 + own; mem.vc_far_stride = far_stride;
    bool have = env_lane && (int)threadIdx.x < m && S.late[threadIdx.x] != 0xffffffffu;   // (an entry is only missing after a timeout: the launch is being aborted)
    uint32_t i = 0u, t = 0u;
    if (have) { const uint32_t ent = S.late[threadIdx.x]; i = roll_env_of(ent); t = ent >> 24; }
    if (q.debug & 8u) { rq_retire(q, have); return; }   // (diagnosis: the batch framing alone)
    bool fresh = true;   // entries have come out of the queue since the last acquire
    for (;;) {
        if (fresh) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // their records were last written by other compute units, in this launch
        n_batches += 1ull; n_lanes += (unsigned long long)__popcll(__ballot(have));
        uint32_t action = 0u;
        bool stepping = false, islanding = false, unfinished = false;
        int n_refresh = 0;
        World& w = S.world[own];
        EnvRegs e;
        if (have) {
            w.t = (LL_LDS WorldTmp*)S.tmp + own;
            ll_load(d, i, w, e, hot, resume_toi);
        }
        if (q.debug & 16u) { if (have && e.step == 0xfffffff0u) atomicOr(d.err, DEV_ERR_INTERNAL); rq_retire(q, have); return; }   // (diagnosis: ... and the load)
        if (have) {
            action = io.act[(uint64_t)t * d.n + i];
            if (resume_toi) {   // world.step is past its island solve and its time-of-impact evaluations: on with SolveTOI
            } else if (w.resume) {   // the free-flight path has taken this step up to the end of the island solve (ll_free.h)
                ll_resume_after_island(w);
            } else {
                float d0, d1, m_power, s_power;
                ll_dispersion(d, i, e, d0, d1);
                ll_pre_step(w.b[0], w.legs[0], w.legs[1], e, LLK(d), action, d0, d1, m_power, s_power);   // wind / engines, :926-1048
                n_refresh = world_step_pre(w); islanding = true;                                          // world.step, :1066: Collide's list walk
            }
            stepping = true;
        }
        {   // b2ContactManager::Collide: the wave's manifold refreshes dealt out over all 64 lanes (see ll_contact_body)
            int incl = n_refresh;
            for (int dlt = 1; dlt < 64; dlt <<= 1) { const int v = __shfl_up(incl, dlt); if ((int)threadIdx.x >= dlt) incl += v; }
            const int n_tasks = __shfl(incl, 63);
            if (n_tasks > 0) {   // wave-uniform
                const int offs = incl - n_refresh;
                for (int j = 0; j < n_refresh; ++j) S.task[offs + j] = (uint16_t)(((uint32_t)own << 4) | ((LL_LDS WorldTmp*)S.tmp + own)->idx[1][j]);
                __syncthreads();
                constexpr int kHelpRoom = (int)(sizeof(S.vc) / (2 * kMaxPoly * sizeof(V2)));
                constexpr int kExec = BLK + ((kThreads - BLK) < kHelpRoom ? (kThreads - BLK) : kHelpRoom);
                V2* const poly_tmp = env_lane ? (V2*)((LL_LDS WorldTmp*)S.tmp + own)->poly_tmp : (V2*)S.vc + (size_t)(threadIdx.x - BLK) * (2 * kMaxPoly);
                if ((int)threadIdx.x < kExec)
                    for (int tk = (int)threadIdx.x; tk < n_tasks; tk += kExec) {
                        const uint32_t task = S.task[tk];
                        collide_refresh(S.world[task >> 4], tab, (int)(task & 15u), poly_tmp);
                    }
                __syncthreads();
            }
            if (islanding) world_step_island(w, tab, LLK(d), mem);   // the callbacks of Collide, then b2World::Solve
        }
        {   // b2World::SolveTOI, passes in lock step over the wave, time-of-impact evaluations dealt out over all 64 lanes
            ToiLoop L;
            bool running = stepping && toi_begin(w, L, !resume_toi);
            int budget = toi_budget;
            while (__any(running)) {   // wave-uniform
                const int n_need = running ? toi_list(w, L) : 0;
                int incl = n_need;
                for (int dlt = 1; dlt < 64; dlt <<= 1) { const int v = __shfl_up(incl, dlt); if ((int)threadIdx.x >= dlt) incl += v; }
                const int n_tasks = __shfl(incl, 63);
                const int offs = incl - n_need;
                for (int j = 0; j < n_need; ++j) S.task[offs + j] = (uint16_t)(((uint32_t)own << 4) | ((LL_LDS WorldTmp*)S.tmp + own)->idx[1][j]);
                __syncthreads();
                for (int tk = (int)threadIdx.x; tk < n_tasks; tk += kThreads) {
                    const uint32_t task = S.task[tk];
                    toi_evaluate(S.world[task >> 4], tab, (int)(task & 15u));
                }
                __syncthreads();
                if (running) {
                    const int r = toi_advance(w, tab, LLK(d), mem, kStepDt, L, budget);
                    running = r == TOI_AGAIN;
                    unfinished = r == TOI_OUT_OF_BUDGET;
                }
            }
        }
        int route = ROLL_NONE;
        bool is_done = false;
        if (stepping && unfinished) {   // stands at a sub-step: stored mid-step, continued by a batch of its own kind
            (void)ll_store(d, i, w, e, true);
            route = ROLL_TO_TOI;
            overflow |= w.overflow;
        } else if (stepping) {
            float state[8], reward; uint32_t done;
            ll_step_finish(w, e, action, state, reward, done);
            const uint32_t flags = (q.debug & (32u | 256u)) ? (F_HAS_WORLD | 7u) : ll_store(d, i, w, e);
            route = (q.debug & (32u | 128u)) ? ROLL_RETIRE : roll_route(d, q, flags, (q.debug & 256u) ? 0u : done, t + 1u);
            if (!(q.debug & 64u)) roll_write_scalars(d, io, t, i, reward, done);
            if (route != ROLL_TO_RESET && !(q.debug & 64u)) roll_write_obs(d, io, t, i, state, t + 1u == q.K);   // (a finished env's observation comes from its reset)
            is_done = done != 0u;
            overflow |= w.overflow;
        }
        finished += (uint32_t)__popcll(__ballot(is_done));
        const bool stay = keep_which >= 0 && route == ROLL_TO_FREE + keep_which;   // same class again: the next step right here
        if (stay) route = ROLL_NONE;
        if (__any(route != ROLL_NONE)) {   // make the stored records visible to whoever takes the environments next, then queue them
            rq_drain();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            rq_drain();
            roll_dispatch(d, q, route, i, route == ROLL_TO_TOI ? t : t + 1u);
        }
        if (keep_which < 0) return;
        have = stay;
        t += 1u;
        // vacant lanes: more of the same kind, if any wait
        fresh = false;
        const unsigned long long vac = __ballot(env_lane && !have);
        const int on_board = __popcll(__ballot(have));
        const int room = (max_lanes < BLK ? max_lanes : BLK) - on_board;
        if (room > 0) {
            int avail = 0;
            if (lane == 0) avail = (int)RQ_LOAD(rq_ctl(q, RC_AVAIL + keep_which));
            avail = __builtin_amdgcn_readfirstlane(avail);
            if (avail > 0) {
                uint32_t ent;
                const int mm = rq_pop(d, q, keep_which, room < avail ? room : avail, ent);
                if (mm > 0) {
                    __syncthreads();
                    if (lane < mm) S.late[lane] = ent;
                    __syncthreads();
                    const int rank = __popcll(vac & ((1ull << lane) - 1ull));
                    if (env_lane && !have && rank < mm && S.late[rank] != 0xffffffffu) {
                        const uint32_t e2 = S.late[rank];
                        i = roll_env_of(e2); t = e2 >> 24;
                        have = true;
                    }
                    __syncthreads();
                    fresh = true;
                }
            }
        }
        if (!__any(have)) return;
        // too few on board for a batch of its own: they go back to the queue, where they meet others (narrow batches cost the machine as much time
        // as wide ones: with every wave keeping its survivors the batches thinned out to 8 lanes, measured)
        if (__popcll(__ballot(have)) < (int)q.keep_min && !(__builtin_amdgcn_readfirstlane((int)RQ_LOAD(rq_ctl(q, RC_LIVE))) < (int)q.tail_live)) {
            rq_drain();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            rq_drain();
            roll_dispatch(d, q, have ? ROLL_TO_FREE + keep_which : ROLL_NONE, i, t);
            return;
        }
    }
}

// ---- reset batch: reset() of up to BLK finished environments — the scene (lunar_lander.rs:727-908), then the implicit step(0) on the
// free-flight path (:911-916; a fresh scene spawns far above the highest terrain vertex) — as ll_reset_pass does for mgym_step
template <int BLK>
__device__ __forceinline__ void roll_reset_batch(const LLDev& d, const LLIo& io, const RollQ& q, int m, ContactLds<BLK>& S) {
    const PolyTab& tab = S.tab;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const bool have = (int)threadIdx.x < BLK && (int)threadIdx.x < m && S.late[threadIdx.x] != 0xffffffffu;
    uint32_t i = 0u, t1 = 0u;
    int route = ROLL_NONE;
    if (have) {
        const CtHot hot{(LL_LDS uint32_t*)S.hot + threadIdx.x, (uint32_t)BLK, 1u};
        const uint32_t ent = S.late[threadIdx.x];
        i = roll_env_of(ent); t1 = ent >> 24;   // the step the new episode starts at; the reset itself belongs to step t1 - 1
        V2 force; float torque;
        {
            World w; EnvRegs e;
            w.t = nullptr;  // (a reset never reaches the contact path)
            ll_load(d, i, w, e, hot);
            ll_reset_scene(d, i, w, e, tab);
            force = w.b[0].force; torque = w.b[0].torque;  // the initial random push (:845-849) is not a state word
            ll_store(d, i, w, e);
        }
        FreeRegs f; EnvRegs e;
        ll_free_load(d, i, f, e);
        f.b[0].force = force; f.b[0].torque = torque;
        float state[8], reward, d0, d1; uint32_t done, moved;
        ll_dispersion(d, i, e, d0, d1);
        if (ll_free_env_step(d, i, f, e, tab, 0u, d0, d1, state, reward, done, moved) == FREE_DONE) {
            e.episode += 1u;
            ll_free_store(d, i, f, e);
            roll_write_obs(d, io, t1 - 1u, i, state, t1 == q.K);
            route = t1 >= q.K ? ROLL_RETIRE : ROLL_TO_FREE;
        } else {
            atomicOr(d.err, DEV_ERR_INTERNAL);   // (cannot happen: see above) — the environment is dropped from the launch, loudly
            route = ROLL_RETIRE;
        }
    }
    rq_drain();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    rq_drain();
    roll_dispatch(d, q, route, i, t1);
}

// ---- free-flight mode: the wave keeps its environments in registers from step to step; leaves when no lane holds one, or when it has
// taken a contact batch (then its residents are flushed back to the queue first) ----
// The queue traffic of a step is SOFTWARE-PIPELINED around the 180 sweeps (75 us without a single memory instruction): the atomics that
// reserve ring positions for the environments that left in the previous step, the claim (compare-and-swap) of free-flight entries for
// the vacant lanes and the next look at the queues' counters are issued right before the sweeps and their results are used right after
// them — issued and awaited on the spot they cost 170 us per wave-step with 1 024 waves at the same words (measured), a step 280 us.
// Returns 0 when no lane holds an environment any more, or m != 0: a contact batch of |m| entries has been taken and sits in S.late (m < 0: from the
// light queue; the residents have been flushed; the caller runs the batch — outside this function, so that the lanes' registers are dead by then)
struct RollPending { int route; uint32_t ent; };   // an environment that has left its slot; its record is on its way to memory (written through)
// A lane of a wave in free-flight mode can carry kRollPer environments side by side through the sweeps (two independent dependent chains in one
// instruction stream).  Built for LL_ROLL_PER = 2 and measured (profiles/r04_lunarlander/rollout_two_per_lane.txt): the double sweep is 230 VALU
// instructions, 137 of them packed (one environment: 124), and takes 85 us per 180 instead of 2 x 50 — but the parts around the sweeps (wind,
// sincos in f64, fixture synchronisation: 18 + 21 us against 9 + 11) do not overlap, and a wave of 128 slots runs 86 % full (refills arrive 64 at a
// time) instead of 98 %: 82 us per 64 environment-steps against 80.  One per lane is the default; the code below is written for any count.
#ifndef LL_ROLL_PER
#define LL_ROLL_PER 1
#endif
constexpr int kRollPer = LL_ROLL_PER;
// issue: ONE atomic instruction hands out the ring positions of the queues a free-flight environment can leave for (free flight, touching, reset,
// light: queue numbers 0..3) and takes the retiring environments off the live count (lane 4)
LLD uint32_t roll_dispatch_issue(const RollQ& q, const RollPending (&p)[kRollPer], unsigned long long (&mask)[kRollPer][5]) {
    const int lane = threadIdx.x & 63;
    uint32_t cnt = 0u;
#pragma unroll
    for (int z = 0; z < kRollPer; ++z) {
        mask[z][0] = __ballot(p[z].route == ROLL_TO_FREE); mask[z][1] = __ballot(p[z].route == ROLL_TO_CONTACT); mask[z][2] = __ballot(p[z].route == ROLL_TO_RESET);
        mask[z][3] = __ballot(p[z].route == ROLL_TO_LIGHT); mask[z][4] = __ballot(p[z].route == ROLL_RETIRE);
        cnt += (uint32_t)__popcll(lane == 0 ? mask[z][0] : lane == 1 ? mask[z][1] : lane == 2 ? mask[z][2] : lane == 3 ? mask[z][3] : mask[z][4]);
    }
    uint32_t ret = 0u;
    if (lane < 5 && cnt) ret = atomicAdd(lane < 4 ? rq_ctl(q, RC_TAIL + lane) : rq_ctl(q, RC_LIVE), lane < 4 ? cnt : 0u - cnt);
    return ret;
}
// complete: the entries go into the slots at their ring positions (the caller has waited for the records: rq_drain), then they count as published
LLD void roll_dispatch_complete(const LLDev& d, const RollQ& q, RollPending (&p)[kRollPer], const unsigned long long (&mask)[kRollPer][5], uint32_t ret) {
    const int lane = threadIdx.x & 63;
    uint32_t published = 0u;
#pragma unroll
    for (int which = 0; which < 4; ++which) {
        unsigned long long any = 0ull;
#pragma unroll
        for (int z = 0; z < kRollPer; ++z) any |= mask[z][which];
        if (any == 0ull) continue;   // wave-uniform
        uint32_t pos0 = __shfl(ret, which);
#pragma unroll
        for (int z = 0; z < kRollPer; ++z) {
            if (p[z].route == ROLL_TO_FREE + which) {
                const uint32_t pos = pos0 + (uint32_t)__popcll(mask[z][which] & ((1ull << lane) - 1ull));
                unsigned long long* const s = rq_slot(q, which, pos);
                if ((uint32_t)(RQ_LOAD(s) >> 32) != pos) {
                    const long long t0 = wall_clock64();
                    while ((uint32_t)(RQ_LOAD(s) >> 32) != pos && wall_clock64() - t0 < kRollTimeoutTicks && !rq_aborted(q)) __builtin_amdgcn_s_sleep(2);
                }
                if ((uint32_t)(RQ_LOAD(s) >> 32) == pos) RQ_STORE(s, ((unsigned long long)(pos + 1u) << 32) | p[z].ent);
                else rq_abort(d, q);
            }
            pos0 += (uint32_t)__popcll(mask[z][which]);
            if (lane == which) published += (uint32_t)__popcll(mask[z][which]);
        }
    }
    if (lane < 4 && published) atomicAdd(rq_ctl(q, RC_AVAIL + lane), published);   // one instruction, up to four words
#pragma unroll
    for (int z = 0; z < kRollPer; ++z) p[z].route = ROLL_NONE;
}
static_assert(ROLL_TO_CONTACT == ROLL_TO_FREE + RQ_CONTACT && ROLL_TO_RESET == ROLL_TO_FREE + RQ_RESET && ROLL_TO_LIGHT == ROLL_TO_FREE + RQ_LIGHT && ROLL_TO_TOI == ROLL_TO_FREE + RQ_TOI, "routes follow the queue numbers");

// Returns 0 when no slot holds an environment any more, or m != 0: a batch of the contact path has been taken and sits in S.late (m < 0: |m| entries
// from the light queue, m > 1000: m - 1000 from the sub-step queue, else m from the touching queue; the residents have been flushed; the caller runs
// the batch — outside this function, so that the slots' registers are dead by then)
// HELPER (ll_rollout_free_kernel): the wave never takes a batch of the contact path — the function returns 0 only.  LDS: anything with `tab` and `late`.
struct RollFreeLds { PolyTab tab; uint32_t late[64 * 4]; };
template <int BLK, bool HELPER = false, class LDS = ContactLds<BLK>>
__device__ __forceinline__ int roll_free_mode(const LLDev& d, const LLIo& io, const RollQ& q, const __amdgpu_buffer_rsrc_t rs, LDS& S, uint32_t& finished, RollStat& st) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const PolyTab& tab = S.tab;
    const LLConst& k = LLK(d);
    RollLane L[kRollPer] = {};
    RollPending P[kRollPer];
#pragma unroll
    for (int z = 0; z < kRollPer; ++z) { L[z].have = false; P[z].route = ROLL_NONE; P[z].ent = 0u; }
    int f_headt = 0, c_headt = 0, l_headt = 0;
    int t_avail = 0;
    int f_avail = 0, c_avail = 0, l_avail = 0;   // what waited in the free-flight / contact / light-contact queues when last looked at (one step ago)
    uint32_t on_board = 0u;         // steps since the wave last took a full load
    auto board = [&](RollLane& X, uint32_t e2) {   // an entry comes on board: its record's two leading lines, the action of its next step
        X.i = roll_env_of(e2); X.t = e2 >> 24;
        RollRec r;
        roll_rec_load(rs, X.i, r);
        roll_unpack(d, r, X.f, X.e);
        X.seq_word = r.w[C_SEQ];
        X.have = true; X.fresh = true;
        X.act = io.act[(uint64_t)X.t * d.n + X.i];
    };
    // take the entries at ring positions [h, h + m) into the vacant slots (slot order: all lanes' first, then all lanes' second): slots -> records
    // (the one exposed round trip of a refill)
    auto fill = [&](uint32_t h, int m, const unsigned long long (&vac)[kRollPer]) {
        for (int c0 = 0; c0 < m; c0 += 64) {   // (a wave takes 64 entries out of their slots at a time)
            const int mc = m - c0 < 64 ? m - c0 : 64;
            const uint32_t ent = rq_take(d, q, RQ_FREE, h + (uint32_t)c0, mc);
            __syncthreads();
            if (lane < mc) S.late[lane] = ent;
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            int before = 0;
#pragma unroll
            for (int z = 0; z < kRollPer; ++z) {
                const int rank = before + __popcll(vac[z] & below) - c0;
                if (((vac[z] >> lane) & 1ull) && rank >= 0 && rank < mc && S.late[rank] != 0xffffffffu) board(L[z], S.late[rank]);
                before += __popcll(vac[z]);
            }
            __syncthreads();
        }
    };
    auto flush_pending = [&]() {   // on the spot (leaving the mode, switching to a batch): what is pending goes into its queues now
        bool any = false;
#pragma unroll
        for (int z = 0; z < kRollPer; ++z) any = any || P[z].route != ROLL_NONE;
        if (__any(any)) {
            unsigned long long pm[kRollPer][5];
            const uint32_t ret = roll_dispatch_issue(q, P, pm);
            rq_drain();
            roll_dispatch_complete(d, q, P, pm, ret);
        }
    };
    auto leave_for_queue = [&](RollLane& X, RollPending& Pz) {   // a resident goes back to the free-flight queue (record written through if it was stepped here)
        if (X.have) {
            if (!X.fresh) {
                RollRec r;
                roll_pack_store(r, X.f, X.e, X.f.flags, X.seq_word);
                roll_rec_store(rs, X.i, r);
            }
            Pz.route = ROLL_TO_FREE; Pz.ent = (X.t << 24) | X.i;
            X.have = false;
        }
    };
    for (;;) {
        unsigned long long vac[kRollPer];
        int n_vac = 0;
#pragma unroll
        for (int z = 0; z < kRollPer; ++z) { vac[z] = __ballot(!L[z].have); n_vac += __popcll(vac[z]); }
        if (n_vac == 64 * kRollPer) {   // nobody on board (entering the mode, or everyone has left): finish what is pending, then a refill on the spot
            flush_pending();
            uint32_t ent;
            const long long tq0 = wall_clock64();
            const int m = rq_pop(d, q, RQ_FREE, 64, ent);
            st.v[RS_N_REFILLS] += 1ull;
            if (m == 0) return 0;
            __syncthreads();
            if (lane < m) S.late[lane] = ent;   // (rq_pop has taken the entries out of their slots already)
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (lane < m && S.late[lane] != 0xffffffffu) board(L[0], S.late[lane]);
            __syncthreads();
            st.v[RS_T_FREE_QUEUE] += (unsigned long long)(wall_clock64() - tq0);
            f_avail = 0; c_avail = 0; l_avail = 0; t_avail = 0; on_board = 0u;
            n_vac = 0;
#pragma unroll
            for (int z = 0; z < kRollPer; ++z) { vac[z] = __ballot(!L[z].have); n_vac += __popcll(vac[z]); }
            if (n_vac == 64 * kRollPer) return 0;
        }
        st.v[RS_N_FREE_STEPS] += 1ull; st.v[RS_N_FREE_LANE_STEPS] += (unsigned long long)(64 * kRollPer - n_vac);
        const long long ts0 = wall_clock64();
        // (1) one Gym::step of every resident environment: up to the sweeps
        bool run[kRollPer], declined[kRollPer];
#pragma unroll
        for (int z = 0; z < kRollPer; ++z) {
            run[z] = false; declined[z] = false;
            if (L[z].have) {
                float d0, d1;
                ll_dispersion(d, L[z].i, L[z].e, d0, d1);
                run[z] = ll_free_begin<true>(d, L[z].i, L[z].f, L[z].e, L[z].act, d0, d1, L[z].s);
                declined[z] = !run[z];   // a pending proxy overlaps the ground (only right after a reset / state import): the contact path, from the record in HBM
            }
        }
        // (2) queue traffic, issued now and used after the sweeps
        const long long ts1 = wall_clock64();
        unsigned long long pm[kRollPer][5];
        bool pend = false;
#pragma unroll
        for (int z = 0; z < kRollPer; ++z) pend = pend || P[z].route != ROLL_NONE;
        const bool pushing = __any(pend);
        uint32_t push_ret = 0u;
        if (pushing) push_ret = roll_dispatch_issue(q, P, pm);
        // ... the semaphore of the free-flight queue, for the vacant slots, if entries were seen waiting a step ago
        int claim_want = 0, claim_old = 0;
        const bool rotate = f_avail >= 64 * kRollPer && on_board + 1u >= q.residency;   // after this step everyone on board makes room for a load that waits
        if (rotate) {
            claim_want = 64 * kRollPer;
            if (lane == 0) claim_old = (int)atomicSub(rq_ctl(q, RC_AVAIL + RQ_FREE), (uint32_t)claim_want);
        } else if (f_avail > 0 && n_vac >= (int)q.refill_min) {
            claim_want = f_avail < n_vac ? f_avail : n_vac;
            if (lane == 0) claim_old = (int)atomicSub(rq_ctl(q, RC_AVAIL + RQ_FREE), (uint32_t)claim_want);
        }
        // ... and the semaphore of a contact-path queue, if a batch's worth was seen waiting (one attempt per step)
        int cclaim_want = 0, cclaim_old = 0;
        // (only a queue that is not AHEAD of the free-flight queue when q.fair: see RC_HEADT)
        const int cclaim_which = HELPER ? -1 : t_avail >= (int)q.toi_min ? RQ_TOI : (c_avail >= (int)q.heavy_min && (!q.fair || f_avail <= 0 || c_headt <= f_headt)) ? RQ_CONTACT
                               : (l_avail >= (int)q.contact_min && (!q.fair || f_avail <= 0 || l_headt <= f_headt)) ? RQ_LIGHT : -1;
        if (cclaim_which >= 0 && !(q.debug & 4u)) {
            cclaim_want = cclaim_which == RQ_CONTACT ? (c_headt < f_headt ? (int)q.heavy_narrow : (int)q.heavy_max) : BLK;
            if (lane == 0) cclaim_old = (int)atomicSub(rq_ctl(q, RC_AVAIL + cclaim_which), (uint32_t)cclaim_want);
        }
        uint32_t seen = 0u;   // next look at the semaphores
        if (lane < RC_HEADT + RQ_COUNT) seen = RQ_LOAD(rq_ctl(q, lane));
        // (3) the 180 velocity sweeps, the slots of a lane side by side
        const long long ts2 = wall_clock64();
        for (int it = 0; it < 180; ++it) {
#pragma unroll
            for (int z = 0; z < kRollPer; ++z) ll_free_sweep(L[z].f, k, L[z].s);
        }
        // (4) the environments that left in the previous step go into the slots reserved for them
        const long long tq0 = wall_clock64();
        st.v[RS_T_FREE_BEGIN] += (unsigned long long)(ts1 - ts0); st.v[RS_T_FREE_ISSUE] += (unsigned long long)(ts2 - ts1); st.v[RS_T_FREE_SWEEPS] += (unsigned long long)(tq0 - ts2);
        if (pushing) { rq_drain(); roll_dispatch_complete(d, q, P, pm, push_ret); }
        const long long ts3 = wall_clock64();
        // (5) the rest of the step; environments that leave their slot become pending
        uint32_t n_done = 0u;
#pragma unroll
        for (int z = 0; z < kRollPer; ++z) {
            RollLane& X = L[z];
            bool is_done = false;
            if (run[z]) {
                float state[8], reward; uint32_t done, moved;
                const int how = ll_free_finish<true>(d, X.i, X.f, X.e, tab, X.s, state, reward, done, moved);
                if (how == FREE_DONE) {
                    const uint32_t flags = roll_flags_after_step(X.f, X.e);
                    const int route = roll_route(d, q, flags, done, X.t + 1u);
                    if (route == ROLL_TO_FREE) X.act = io.act[(uint64_t)(X.t + 1u) * d.n + X.i];   // stays on board: next step's action, ahead of this step's stores
                    roll_write_scalars(d, io, X.t, X.i, reward, done);
                    if (route != ROLL_TO_RESET) roll_write_obs(d, io, X.t, X.i, state, X.t + 1u == q.K);
                    is_done = done != 0u;
                    // the slot's registers as a store + load of the record would leave them
                    X.f.flags = flags;
#pragma unroll
                    for (int b = 0; b < 3; ++b) { X.f.b[b].force = mk(0.0f, 0.0f); X.f.b[b].torque = 0.0f; }
                    X.t += 1u; X.fresh = false;
                    if (route != ROLL_TO_FREE) {   // the environment leaves the registers: its record goes back, written through
                        RollRec r;
                        roll_pack_store(r, X.f, X.e, flags, X.seq_word);
                        roll_rec_store(rs, X.i, r);
                        P[z].route = route; P[z].ent = (X.t << 24) | X.i;
                        X.have = false;
                    }
                } else {   // FREE_RESUME: a contact is being created at the end of the step — ≙ ll_free_store_resume, then the contact path finishes step t
                    RollRec r;
                    uint32_t awake_bits;
                    roll_pack_common(r, X.f, awake_bits);
                    const uint32_t flags = (X.f.flags & (F_GAME_OVER | F_LEG0 | F_LEG1 | F_HAS_WORLD | F_DETERMINISTIC | F_STEPPED | F_PREV_SOME)) | awake_bits;
                    r.w[C_FLAGS] = flags | F_RESUME | ((moved & 7u) << 9);
                    r.w[C_PREV] = as_u32(X.e.prev_shaping);                          // (unchanged by the unfinished step; the record may be older than the registers)
                    r.w[C_WIND] = (uint32_t)X.e.wind_idx; r.w[C_TORQUE] = (uint32_t)X.e.torque_idx;
                    r.w[C_STEP] = X.e.step; r.w[C_EPISODE] = X.e.episode; r.w[C_SEQ] = X.seq_word;
                    roll_rec_store(rs, X.i, r);
                    const uint32_t off = X.i * (uint32_t)(kRec * 4) + (uint32_t)(C_MID * 4);   // the sweeps' start: words C_MID .. C_MID + 12 (+ 3 unused) = four 16-byte pieces
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        __builtin_amdgcn_raw_buffer_store_b128(ll_u32x4{as_u32(X.f.b[b].sw.c0.x), as_u32(X.f.b[b].sw.c0.y), as_u32(X.f.b[b].sw.a0), as_u32(0.0f)}, rs, off + 16u * b, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(ll_u32x4{as_u32(0.0f), 0u, 0u, 0u}, rs, off + 48u, 0, 16);
                    P[z].route = ROLL_TO_LIGHT; P[z].ent = (X.t << 24) | X.i;   // same step index: the step is not finished
                    X.have = false;
                }
            } else if (declined[z]) {
                if (!X.fresh) atomicOr(d.err, DEV_ERR_INTERNAL);   // (a stepped environment has no pending proxies)
                P[z].route = ROLL_TO_LIGHT; P[z].ent = (X.t << 24) | X.i;
                X.have = false;
            }
            n_done += (uint32_t)__popcll(__ballot(is_done));
        }
        finished += n_done;
        const long long ts4 = wall_clock64();
        st.v[RS_T_FREE_FINISH] += (unsigned long long)(ts4 - ts3);
        // (6) what the free-flight semaphore has granted: ring positions for it, then fill the slots that were vacant before the sweeps
        if (claim_want > 0) {
            st.v[RS_N_REFILLS] += 1ull;
            const int got = rq_sem_got(q, RQ_FREE, __builtin_amdgcn_readfirstlane(claim_old), claim_want);
            if (got > 0) {
                uint32_t h = 0u;
                if (lane == 0) h = atomicAdd(rq_ctl(q, RC_HEAD + RQ_FREE), (uint32_t)got);
                h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
                if (rotate) {   // everyone still on board goes back to the queue ...
#pragma unroll
                    for (int z = 0; z < kRollPer; ++z) { leave_for_queue(L[z], P[z]); vac[z] = ~0ull; }
                    on_board = 0u;
                    st.v[RS_N_ROTATIONS] += 1ull;
                }
                fill(h, got, vac);   // ... and the waiting ones come on board
            }
        }
        on_board += 1u;
        f_avail = __builtin_amdgcn_readlane((int)seen, RQ_FREE); c_avail = __builtin_amdgcn_readlane((int)seen, RQ_CONTACT); l_avail = __builtin_amdgcn_readlane((int)seen, RQ_LIGHT); t_avail = __builtin_amdgcn_readlane((int)seen, RQ_TOI);
        f_headt = __builtin_amdgcn_readlane((int)seen, RC_HEADT + RQ_FREE); c_headt = __builtin_amdgcn_readlane((int)seen, RC_HEADT + RQ_CONTACT); l_headt = __builtin_amdgcn_readlane((int)seen, RC_HEADT + RQ_LIGHT);
        st.v[RS_T_FREE_QUEUE] += (unsigned long long)((wall_clock64() - ts4) + (ts3 - tq0));
        // (7) what the contact-path semaphore has granted: everyone leaves the registers, the batch goes to the caller
        if (cclaim_want > 0) {
            const int m = rq_sem_got(q, cclaim_which, __builtin_amdgcn_readfirstlane(cclaim_old), cclaim_want);
            if (m > 0) {
                uint32_t h = 0u;
                if (lane == 0) h = atomicAdd(rq_ctl(q, RC_HEAD + cclaim_which), (uint32_t)m);
                h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
                const uint32_t ent = rq_take(d, q, cclaim_which, h, m);
                flush_pending();   // the pending ones first, then the residents
#pragma unroll
                for (int z = 0; z < kRollPer; ++z) leave_for_queue(L[z], P[z]);
                flush_pending();
                __syncthreads();
                if (lane < m) S.late[lane] = ent;
                __syncthreads();
                st.v[RS_N_SWITCHES] += 1ull;
                return cclaim_which == RQ_LIGHT ? -m : cclaim_which == RQ_TOI ? m + 1000 : m;
            }
        }
    }
}

// ---- the launch ----
template <int BLK>
__global__ void __launch_bounds__(64) LL_CONTACT_ATTR
ll_rollout_kernel(LLDev d, LLIo io, RollQ q) {
    __shared__ ContactLds<BLK> S;
    const int lane = threadIdx.x & 63;
    uint32_t overflow = 0u, finished = 0u;
    bool not_reset = false;
    RollStat st = {};
    const long long t_begin = wall_clock64();
    int n_trace = 0, last_kind = 0;
    auto trace = [&](int kind) {   // (what the wave begins now; consecutive events of one kind are recorded once)
        if (q.trace && kind != last_kind && n_trace < kRollTraceLen && lane == 0)
            q.trace[(uint64_t)blockIdx.x * kRollTraceLen + n_trace] = ((unsigned long long)(wall_clock64() - t_begin) << 8) | (unsigned long long)kind;
        if (kind != last_kind) { ++n_trace; last_kind = kind; }
    };
    trace(RT_SEED);
    stage_tab(S.tab, LLK(d));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d.st, 0, 0xffffffff, 0x00020000);
    VConstraint* const far_lane0 = d.vc_far + (uint64_t)blockIdx.x * BLK;
    const int far_stride = (int)(gridDim.x * BLK);
    // seed: every environment enters a queue at step 0 (the records are the previous launches' — ordinary loads)
    for (;;) {
        uint32_t c = 0u;
        if (lane == 0) c = atomicAdd(rq_ctl(q, RC_CHUNK), 1u);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        if ((uint64_t)c * 64u >= d.n) break;
        const uint64_t i = (uint64_t)c * 64u + (uint64_t)lane;
        int route = ROLL_NONE;
        if (i < d.n) {
            const uint32_t flags = ST(C_FLAGS);
            if (!(flags & F_HAS_WORLD)) {  // assert!(self.lander.is_some(), "You forgot to call reset()") — :920, at every step
                not_reset = true;
                for (uint32_t t = 0; t < q.K; ++t) {
                    const uint64_t o = (uint64_t)t * d.n + i;
                    if (io.rew) io.rew[o] = 0.0f;
                    if (io.done_out) io.done_out[o] = 0;
                    if (io.trunc_out) io.trunc_out[o] = 0;
                }
                route = ROLL_RETIRE;
            } else {
                route = ll_free_eligible(flags) ? ROLL_TO_FREE : (flags & F_TOUCHING) ? ROLL_TO_CONTACT : ROLL_TO_LIGHT;
            }
        }
        roll_dispatch(d, q, route, (uint32_t)i, 0u);
    }
    st.v[RS_T_SEED] = (unsigned long long)(wall_clock64() - t_begin);
    if (q.debug & 1u) { ll_report(d, not_reset, overflow, finished); return; }   // (diagnosis: the seed phase alone)
    // serve the queues until every environment has done its K steps
    long long idle_since = -1;
    for (;;) {
        const long long t_look = wall_clock64();
        const RollCounters c = rq_counters(q);
        const int aF = c.avail[RQ_FREE], aC = c.avail[RQ_CONTACT], aR = c.avail[RQ_RESET], aL = c.avail[RQ_LIGHT];
        // the queue that is furthest behind among those with a batch's worth of entries (ties: touching contact, light contact, reset, free flight);
        // with nothing like that, any entries at all by the same rule
        int which = -1, best_t = 0x7fffffff;
        const int order[RQ_COUNT] = {RQ_TOI, RQ_CONTACT, RQ_LIGHT, RQ_RESET, RQ_FREE};
        const int need[RQ_COUNT] = {(int)q.free_min, c.live < q.tail_live ? 1 : (int)q.heavy_min, c.live < q.tail_live ? 1 : (int)q.reset_min, c.live < q.tail_live ? 1 : (int)q.contact_min,
                                    c.live < q.tail_live ? 1 : (int)q.toi_min};   // by queue number
#pragma unroll
        for (int o = 0; o < RQ_COUNT; ++o) { const int w = order[o]; const int ht = q.fair ? c.headt[w] : o; if (c.avail[w] >= need[w] && ht < best_t) { which = w; best_t = ht; } }
        if (which < 0) {
#pragma unroll
            for (int o = 0; o < RQ_COUNT; ++o) { const int w = order[o]; const int ht = q.fair ? c.headt[w] : o; if (c.avail[w] > 0 && ht < best_t) { which = w; best_t = ht; } }
        }
        const bool tail = c.live < q.tail_live;
        const int heavy_lanes = tail ? (int)q.tail_lanes : c.headt[RQ_CONTACT] < c.headt[RQ_FREE] ? (int)q.heavy_narrow : (int)q.heavy_max;
        (void)aF; (void)aC; (void)aR; (void)aL;
        int m = 0;
        if (which == RQ_FREE && !(q.debug & 2u)) {
            const long long tf0 = wall_clock64();
            trace(RT_FREE);
            m = roll_free_mode<BLK>(d, io, q, rs, S, finished, st);   // -> a contact batch it has taken (in S.late), or 0
            st.v[RS_T_FREE] += (unsigned long long)(wall_clock64() - tf0);
            idle_since = -1;
            if (m == 0) { st.v[RS_N_MAIN] += 1ull; continue; }
            which = m < 0 ? RQ_LIGHT : m > 1000 ? RQ_TOI : RQ_CONTACT;
            m = m < 0 ? -m : m > 1000 ? m - 1000 : m;
        } else if (which >= 0) {
            uint32_t ent;
            m = rq_pop(d, q, which, which == RQ_CONTACT ? heavy_lanes : (tail && (which == RQ_LIGHT || which == RQ_TOI)) ? (int)q.tail_lanes : BLK, ent);
            if (m == 0) { st.v[RS_N_MAIN] += 1ull; continue; }   // others were faster: look again
            __syncthreads();
            if (lane < m) S.late[lane] = ent;
            __syncthreads();
        }
        if (which >= 0) {
            const long long tb0 = wall_clock64();
            trace(which == RQ_RESET ? RT_RESET : which == RQ_LIGHT ? RT_LIGHT : which == RQ_TOI ? RT_TOI : RT_CONTACT);
            if (which == RQ_RESET) roll_reset_batch<BLK>(d, io, q, m, S);
            else {   // ONE call site for the contact path's three kinds of batch (the body is large)
                const int keep = which == RQ_LIGHT ? ((q.keep & 2u) ? RQ_LIGHT : -1) : (which == RQ_CONTACT && (q.keep & 1u)) ? RQ_CONTACT : -1;
                const int lanes_max = which == RQ_CONTACT ? heavy_lanes : tail ? (int)q.tail_lanes : BLK;
                const int budget = which == RQ_TOI ? ((q.toi_split & 4u) ? 1 : -1) : (which == RQ_LIGHT ? (q.toi_split & 1u) : which == RQ_CONTACT ? (q.toi_split & 2u) : 0u) ? 0 : -1;
                unsigned long long& nb = which == RQ_LIGHT ? st.v[RS_N_LIGHT_BATCHES] : which == RQ_TOI ? st.v[RS_N_TOI_BATCHES] : st.v[RS_N_CONTACT_BATCHES];
                unsigned long long& nl = which == RQ_LIGHT ? st.v[RS_N_LIGHT_LANES] : which == RQ_TOI ? st.v[RS_N_TOI_LANES] : st.v[RS_N_CONTACT_LANES];
                roll_contact_batch<BLK>(d, io, q, m, S, far_lane0, far_stride, overflow, finished, keep, lanes_max, nb, nl, which == RQ_TOI, budget);   // (diagnosis 2: free-flight entries come here too)
            }
            __syncthreads();
            const unsigned long long tb = (unsigned long long)(wall_clock64() - tb0);
            if (which == RQ_RESET) { st.v[RS_T_RESET] += tb; st.v[RS_N_RESET_LANES] += (unsigned long long)m; }
            else if (which == RQ_LIGHT) st.v[RS_T_LIGHT] += tb;
            else if (which == RQ_TOI) st.v[RS_T_TOI] += tb;
            else st.v[RS_T_CONTACT] += tb;
            idle_since = -1;
            continue;
        }
        if (c.live == 0u || rq_aborted(q)) break;
        trace(RT_IDLE);
        const long long now = wall_clock64();
        if (idle_since < 0) idle_since = now;
        else if (now - idle_since > kRollTimeoutTicks) { if (lane == 0) rq_abort(d, q); break; }
        __builtin_amdgcn_s_sleep(100);
        st.v[RS_T_IDLE] += (unsigned long long)(wall_clock64() - t_look);
    }
    trace(RT_END);
    st.v[RS_T_TOTAL] = (unsigned long long)(wall_clock64() - t_begin); st.v[RS_N_WAVES] = 1ull;
    if (lane < RS_COUNT && q.stat) { unsigned long long mine = 0ull;
#pragma unroll
        for (int z = 0; z < RS_COUNT; ++z) mine = lane == z ? st.v[z] : mine;
        if (mine) atomicAdd(q.stat + lane, mine); }
    ll_report(d, not_reset, overflow, finished);
}

// Free-flight helper waves of the same launch: the waves above are compiled for the contact path (one wave per SIMD, every register) and take their
// free-flight steps at that occupancy — the sweeps run at the SIMD's issue rate, but everything around them (records in and out, the step's begin and
// finish, the queues) waits with nothing to switch to.  These waves do nothing but free-flight steps, from the same queue, at several waves per SIMD
// on the SIMDs the main launch leaves free.  They are an accelerator, never a dependency: the main waves serve every queue, so the launch completes
// whether or not a single helper wave ever becomes resident; a helper exits when every environment has finished (or the launch was aborted).
#ifndef LL_ROLL_HELPER_OCC
#define LL_ROLL_HELPER_OCC 2
#endif
template <int BLK>
__global__ void __launch_bounds__(64, LL_ROLL_HELPER_OCC)
ll_rollout_free_kernel(LLDev d, LLIo io, RollQ q) {
    __shared__ RollFreeLds S;
    const int lane = threadIdx.x & 63;
    uint32_t finished = 0u;
    RollStat st = {};
    stage_tab(S.tab, LLK(d));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d.st, 0, 0xffffffff, 0x00020000);
    const long long t_begin = wall_clock64();
    for (;;) {
        uint32_t seen = 0u;
        if (lane < RC_HEADT) seen = RQ_LOAD(rq_ctl(q, lane));
        const int avail = __builtin_amdgcn_readlane((int)seen, RC_AVAIL + RQ_FREE);
        const uint32_t live = (uint32_t)__builtin_amdgcn_readlane((int)seen, RC_LIVE);
        if (live == 0u || rq_aborted(q)) break;
        if (wall_clock64() - t_begin > 4 * kRollTimeoutTicks) break;   // (never the one to abort: the main waves finish the launch without it)
        if (avail >= (int)q.helper_min) {
            roll_free_mode<BLK, true, RollFreeLds>(d, io, q, rs, S, finished, st);
            continue;
        }
        __builtin_amdgcn_s_sleep(64);
    }
    if (lane < RS_COUNT && q.stat) { unsigned long long mine = 0ull;   // (its free-flight steps are counted, its time is not: RS_N_WAVES stays the main launch's)
#pragma unroll
        for (int z = 0; z < RS_COUNT; ++z) mine = lane == z ? st.v[z] : mine;
        if (mine && (lane == RS_N_FREE_STEPS || lane == RS_N_FREE_LANE_STEPS)) atomicAdd(q.stat + RS_N_HELPER_STEPS + (lane == RS_N_FREE_LANE_STEPS ? 1 : 0), mine); }
    ll_flush_done(d, finished);
}

// slot k of every ring starts with sequence k; control words zero
__global__ void ll_rollout_ring_init_kernel(RollQ q) {
    const uint64_t cap = (uint64_t)q.mask + 1ull;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap * RQ_COUNT; s += (uint64_t)gridDim.x * blockDim.x)
        q.ring[s] = (unsigned long long)(uint32_t)(s & q.mask) << 32;
    if (blockIdx.x == 0 && threadIdx.x < RC_WORDS) q.ctl[32 * threadIdx.x] = 0u;
}
// before every launch: the chunk counter, the number of environments still to finish, the abort word (the rings carry on where the last launch left them)
__global__ void ll_rollout_begin_kernel(RollQ q, uint32_t n) {
    if (threadIdx.x == 0) { *rq_ctl(q, RC_CHUNK) = 0u; *rq_ctl(q, RC_LIVE) = n; *rq_ctl(q, RC_ABORT) = 0u; }
    if (threadIdx.x < RQ_COUNT) *rq_ctl(q, RC_HEADT + threadIdx.x) = 0u;
    if (threadIdx.x < RS_COUNT && q.stat) q.stat[threadIdx.x] = 0ull;
}

}  // namespace mgym
