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

namespace mgym {

typedef uint32_t ll_u32x4 __attribute__((ext_vector_type(4)));

enum { RQ_FREE = 0, RQ_CONTACT = 1, RQ_RESET = 2, RQ_COUNT = 3 };
// control words, each on a 128-byte line of its own
enum { RC_HEAD = 0 /* + queue */, RC_TAIL = 3 /* + queue */, RC_CHUNK = 6, RC_LIVE = 7, RC_ABORT = 8, RC_STAT = 9 /* .. 15: diagnostics */, RC_WORDS = 16 };
struct RollQ {
    unsigned long long* ring;   // [RQ_COUNT][cap] slots {sequence << 32 | entry}; slot k starts with sequence k
    uint32_t* ctl;              // [RC_WORDS][32]
    uint32_t mask;              // cap - 1 (cap: a power of two >= n)
    uint32_t K;                 // steps of this launch (<= kRollMaxK)
    uint32_t contact_min;       // a wave that has other work takes a contact batch only when at least this many entries wait
    uint32_t refill_min;        // a resident wave refills its vacant lanes only when at least this many are vacant (or none is left)
    uint32_t debug;             // diagnosis (MGYM_LL_ROLL_DEBUG): 1 stop after the seed phase, 2 no free-flight mode (every environment through the contact path), 4 resident waves never switch to a contact batch
};
constexpr uint32_t kRollMaxK = 240;          // step index in the top byte of an entry (0xff.. = empty is never a valid entry)
constexpr uint32_t kRollEnvMask = 0xffffffu;
constexpr long long kRollTimeoutTicks = 300000000ll;   // 3 s of the 100 MHz wall clock: a wave that waits this long gives up loudly

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

// Take up to `want` entries of queue `which` (only if at least `min_avail` wait): wave-uniform result m, lane l < m gets its entry.
LLD int rq_pop(const LLDev& d, const RollQ& q, int which, int want, int min_avail, uint32_t& ent) {
    const int lane = threadIdx.x & 63;
    uint32_t h = 0u; int m = 0;
    if (lane == 0) {
        for (int tries = 0; tries < 8; ++tries) {
            h = RQ_LOAD(rq_ctl(q, RC_HEAD + which));
            const uint32_t t = RQ_LOAD(rq_ctl(q, RC_TAIL + which));
            const int avail = (int)(t - h);
            if (avail <= 0 || avail < min_avail) { m = 0; break; }
            m = avail < want ? avail : want;
            if (atomicCAS(rq_ctl(q, RC_HEAD + which), h, h + (uint32_t)m) == h) break;
            m = 0;
        }
    }
    m = __builtin_amdgcn_readfirstlane(m); h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    ent = 0xffffffffu;
    if (lane < m) {
        unsigned long long* const s = rq_slot(q, which, h + (uint32_t)lane);
        const uint32_t want_seq = h + (uint32_t)lane + 1u;
        unsigned long long v = RQ_LOAD(s);
        if ((uint32_t)(v >> 32) != want_seq) {   // reserved by its producer, not written yet: a few hundred nanoseconds
            const long long t0 = wall_clock64();
            do { __builtin_amdgcn_s_sleep(2); v = RQ_LOAD(s); } while ((uint32_t)(v >> 32) != want_seq && wall_clock64() - t0 < kRollTimeoutTicks && !rq_aborted(q));
        }
        if ((uint32_t)(v >> 32) == want_seq) {
            ent = (uint32_t)v;
            RQ_STORE(s, (unsigned long long)(want_seq + q.mask) << 32);   // free for position pos + cap
        } else {
            rq_abort(d, q);
        }
    }
    return m;
}
// Append (wave-aggregated: one atomic per wave).  The caller has made the environment's state visible first (file header).
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
            if ((uint32_t)(RQ_LOAD(s) >> 32) != pos) { rq_abort(d, q); return; }
        }
        RQ_STORE(s, ((unsigned long long)(pos + 1u) << 32) | ent);
    }
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
enum { ROLL_NONE = 0, ROLL_TO_FREE, ROLL_TO_CONTACT, ROLL_TO_RESET, ROLL_RETIRE };
LLD int roll_route(const LLDev& d, const RollQ& q, uint32_t flags, uint32_t done, uint32_t t1) {
    if (done && d.auto_reset) return ROLL_TO_RESET;
    if (t1 >= q.K) return ROLL_RETIRE;
    return ll_free_eligible(flags) ? ROLL_TO_FREE : ROLL_TO_CONTACT;
}
LLD void roll_dispatch(const LLDev& d, const RollQ& q, int route, uint32_t i, uint32_t t1) {
    const uint32_t ent = (t1 << 24) | i;
    rq_push(d, q, RQ_FREE, route == ROLL_TO_FREE, ent);
    rq_push(d, q, RQ_CONTACT, route == ROLL_TO_CONTACT, ent);
    rq_push(d, q, RQ_RESET, route == ROLL_TO_RESET, ent);
    rq_retire(q, route == ROLL_RETIRE);
}

// ---- contact batch: `m` entries in S.late, one per lane; the body of ll_contact_body<BLK> for one pass with a step index per lane ----
template <int BLK>
__device__ __forceinline__ void roll_contact_batch(const LLDev& d, const LLIo& io, const RollQ& q, int m, ContactLds<BLK>& S, VConstraint* far_lane0, int far_stride,
                                                   uint32_t& overflow, uint32_t& finished) {
    static_assert(BLK <= 32, "rollout: World records in LDS");
    constexpr int kThreads = ll_contact_threads(BLK);
    const PolyTab& tab = S.tab;
    const bool env_lane = threadIdx.x < BLK;
    const int own = env_lane ? (int)threadIdx.x : 0;
    const CtHot hot{(LL_LDS uint32_t*)S.hot + own, (uint32_t)BLK, 1u};
    PConstraint l_pc[kSolverCap];
    CSolverMem mem;
    mem.vc = &S.vc[own]; mem.vc_stride = BLK; mem.vc_near = d.vc_near < kVcNearLds ? d.vc_near : kVcNearLds; mem.pc = l_pc; mem.pc_stride = 1; mem.cap = kSolverCap;
    mem.vc_far = far_lane0 + own; mem.vc_far_stride = far_stride;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the records were last written by other compute units, in this launch
    const bool have = env_lane && (int)threadIdx.x < m && S.late[threadIdx.x] != 0xffffffffu;   // (an entry is only missing after a timeout: the launch is being aborted)
    uint32_t i = 0u, t = 0u, action = 0u;
    bool stepping = false, islanding = false;
    int n_refresh = 0;
    World& w = S.world[own];
    EnvRegs e;
    if (q.debug & 8u) { rq_retire(q, have); return; }   // (diagnosis: the batch framing alone)
    if (have) {
        const uint32_t ent = S.late[threadIdx.x];
        i = roll_env_of(ent); t = ent >> 24;
        w.t = (LL_LDS WorldTmp*)S.tmp + own;
        ll_load(d, i, w, e, hot);
    }
    if (q.debug & 16u) { if (have && e.step == 0xfffffff0u) atomicOr(d.err, DEV_ERR_INTERNAL); rq_retire(q, have); return; }   // (diagnosis: ... and the load)
    if (have) {
        action = io.act[(uint64_t)t * d.n + i];
        if (w.resume) {   // the free-flight path has taken this step up to the end of the island solve (ll_free.h)
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
        bool running = stepping && toi_begin(w, L, true);
        int budget = -1;
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
            if (running) running = toi_advance(w, tab, LLK(d), mem, kStepDt, L, budget) == TOI_AGAIN;
        }
    }
    int route = ROLL_NONE;
    bool is_done = false;
    if (stepping) {
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
    // make the stored records visible to whoever takes the environments next, then queue them
    rq_drain();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    rq_drain();
    roll_dispatch(d, q, route, i, t + 1u);
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
// Returns 0 when no lane holds an environment any more, or m > 0: a contact batch of m entries has been claimed and sits in S.late (the residents
// have been flushed; the caller runs the batch — outside this function, so that the lanes' registers are dead by then)
template <int BLK>
__device__ __forceinline__ int roll_free_mode(const LLDev& d, const LLIo& io, const RollQ& q, const __amdgpu_buffer_rsrc_t rs, ContactLds<BLK>& S, uint32_t& finished) {
    const int lane = threadIdx.x & 63;
    const PolyTab& tab = S.tab;
    const LLConst& k = LLK(d);
    RollLane L = {};
    L.have = false;
    for (;;) {
        // (a) refill vacant lanes from the free-flight queue
        const unsigned long long vac = __ballot(!L.have);
        const int n_vac = __popcll(vac);
        if (n_vac == 64 || n_vac >= (int)q.refill_min) {
            uint32_t ent;
            const int m = rq_pop(d, q, RQ_FREE, n_vac, 1, ent);
            if (m > 0) {
                // lane l < m holds the entry for the l-th vacant lane: hand them over through LDS
                if (lane < m) S.late[lane] = ent;
                __syncthreads();
                const int rank = __popcll(vac & ((1ull << lane) - 1ull));
                const bool take = !L.have && rank < m && S.late[rank < m ? rank : 0] != 0xffffffffu;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (take) {
                    const uint32_t e2 = S.late[rank];
                    L.i = roll_env_of(e2); L.t = e2 >> 24;
                    RollRec r;
                    roll_rec_load(rs, L.i, r);
                    roll_unpack(d, r, L.f, L.e);
                    L.seq_word = r.w[C_SEQ];
                    L.have = true; L.fresh = true;
                }
                __syncthreads();
            }
        }
        if (!__any(L.have)) return 0;
        // (b) does a full contact block wait?  (looked at before the sweeps, acted upon after them)
        uint32_t c_head = 0u, c_tail = 0u;
        if (lane == 0) { c_head = RQ_LOAD(rq_ctl(q, RC_HEAD + RQ_CONTACT)); c_tail = RQ_LOAD(rq_ctl(q, RC_TAIL + RQ_CONTACT)); }
        // (c) one Gym::step of every resident environment
        bool run = false, declined = false;
        uint32_t action = 0u;
        if (L.have) {
            action = io.act[(uint64_t)L.t * d.n + L.i];
            float d0, d1;
            ll_dispersion(d, L.i, L.e, d0, d1);
            run = ll_free_begin<true>(d, L.i, L.f, L.e, action, d0, d1, L.s);
            declined = !run;   // a pending proxy overlaps the ground (only right after a reset / state import): the contact path, from the record in HBM
        }
        for (int it = 0; it < 180; ++it) ll_free_sweep(L.f, k, L.s);
        int route = ROLL_NONE;
        bool is_done = false, leave = false;
        uint32_t ent_t = L.t;
        if (run) {
            float state[8], reward; uint32_t done, moved;
            const int how = ll_free_finish<true>(d, L.i, L.f, L.e, tab, L.s, state, reward, done, moved);
            if (how == FREE_DONE) {
                const uint32_t flags = roll_flags_after_step(L.f, L.e);
                route = roll_route(d, q, flags, done, L.t + 1u);
                roll_write_scalars(d, io, L.t, L.i, reward, done);
                if (route != ROLL_TO_RESET) roll_write_obs(d, io, L.t, L.i, state, L.t + 1u == q.K);
                is_done = done != 0u;
                // the lane's registers as a store + load of the record would leave them
                L.f.flags = flags;
#pragma unroll
                for (int b = 0; b < 3; ++b) { L.f.b[b].force = mk(0.0f, 0.0f); L.f.b[b].torque = 0.0f; }
                L.t += 1u; L.fresh = false;
                ent_t = L.t;
                if (route != ROLL_TO_FREE) {   // the environment leaves the registers: its record goes back, written through
                    RollRec r;
                    roll_pack_store(r, L.f, L.e, flags, L.seq_word);
                    roll_rec_store(rs, L.i, r);
                    leave = true;
                } else {
                    route = ROLL_NONE;        // stays resident
                }
            } else {   // FREE_RESUME: a contact is being created at the end of the step — ≙ ll_free_store_resume, then the contact path finishes step t
                RollRec r;
                uint32_t awake_bits;
                roll_pack_common(r, L.f, awake_bits);
                const uint32_t flags = (L.f.flags & (F_GAME_OVER | F_LEG0 | F_LEG1 | F_HAS_WORLD | F_DETERMINISTIC | F_STEPPED | F_PREV_SOME)) | awake_bits;
                r.w[C_FLAGS] = flags | F_RESUME | ((moved & 7u) << 9);
                r.w[C_PREV] = as_u32(L.e.prev_shaping);                          // (unchanged by the unfinished step; the record may be older than the registers)
                r.w[C_WIND] = (uint32_t)L.e.wind_idx; r.w[C_TORQUE] = (uint32_t)L.e.torque_idx;
                r.w[C_STEP] = L.e.step; r.w[C_EPISODE] = L.e.episode; r.w[C_SEQ] = L.seq_word;
                roll_rec_store(rs, L.i, r);
                const uint32_t off = L.i * (uint32_t)(kRec * 4) + (uint32_t)(C_MID * 4);   // the sweeps' start: words C_MID .. C_MID + 12 (+ 3 unused) = four 16-byte pieces
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    __builtin_amdgcn_raw_buffer_store_b128(ll_u32x4{as_u32(L.f.b[b].sw.c0.x), as_u32(L.f.b[b].sw.c0.y), as_u32(L.f.b[b].sw.a0), as_u32(0.0f)}, rs, off + 16u * b, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(ll_u32x4{as_u32(0.0f), 0u, 0u, 0u}, rs, off + 48u, 0, 16);
                route = ROLL_TO_CONTACT; leave = true;   // same step index: the step is not finished
            }
        } else if (declined) {
            if (!L.fresh) atomicOr(d.err, DEV_ERR_INTERNAL);   // (a stepped environment has no pending proxies)
            route = ROLL_TO_CONTACT; leave = true;
        }
        finished += (uint32_t)__popcll(__ballot(is_done));
        if (__any(leave)) {
            rq_drain();   // the written-through records have arrived
            roll_dispatch(d, q, route, L.i, ent_t);
            if (leave) L.have = false;
        }
        // (d) a full contact block waits: take it (one wave wins the claim), flush the residents, run it, come back
        c_head = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_head); c_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_tail);
        if ((int)(c_tail - c_head) >= (int)q.contact_min && !(q.debug & 4u)) {
            uint32_t ent;
            const int m = rq_pop(d, q, RQ_CONTACT, BLK, (int)q.contact_min, ent);
            if (m > 0) {
                if (L.have) {
                    if (!L.fresh) {
                        RollRec r;
                        roll_pack_store(r, L.f, L.e, L.f.flags, L.seq_word);
                        roll_rec_store(rs, L.i, r);
                    }
                }
                rq_drain();
                rq_push(d, q, RQ_FREE, L.have, (L.t << 24) | L.i);
                __syncthreads();
                if (lane < m) S.late[lane] = ent;
                __syncthreads();
                return m;
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
                route = ll_free_eligible(flags) ? ROLL_TO_FREE : ROLL_TO_CONTACT;
            }
        }
        roll_dispatch(d, q, route, (uint32_t)i, 0u);
    }
    if (q.debug & 1u) { ll_report(d, not_reset, overflow, finished); return; }   // (diagnosis: the seed phase alone)
    // serve the queues until every environment has done its K steps
    long long idle_since = -1;
    for (;;) {
        uint32_t ent = 0xffffffffu;
        bool is_reset = false, in_late = false;
        int m = rq_pop(d, q, RQ_CONTACT, BLK, (int)q.contact_min, ent);
        if (m == 0) { m = rq_pop(d, q, RQ_RESET, BLK, 1, ent); is_reset = m > 0; }
        if (m == 0 && !(q.debug & 2u)) {
            const uint32_t fh = (uint32_t)__builtin_amdgcn_readfirstlane((int)RQ_LOAD(rq_ctl(q, RC_HEAD + RQ_FREE)));
            const uint32_t ft = (uint32_t)__builtin_amdgcn_readfirstlane((int)RQ_LOAD(rq_ctl(q, RC_TAIL + RQ_FREE)));
            if ((int)(ft - fh) > 0) {
                m = roll_free_mode<BLK>(d, io, q, rs, S, finished);   // -> a contact batch it has claimed (in S.late), or 0
                idle_since = -1;
                if (m == 0) continue;
                in_late = true;
            }
        }
        if (m == 0 && (q.debug & 2u)) m = rq_pop(d, q, RQ_FREE, BLK, 1, ent);   // (diagnosis: every environment through the contact path)
        if (m == 0) m = rq_pop(d, q, RQ_CONTACT, BLK, 1, ent);                    // nothing else to do: a partial block
        if (m > 0) {
            if (!in_late) {
                __syncthreads();
                if (lane < m) S.late[lane] = ent;
                __syncthreads();
            }
            if (is_reset) roll_reset_batch<BLK>(d, io, q, m, S);
            else roll_contact_batch<BLK>(d, io, q, m, S, far_lane0, far_stride, overflow, finished);
            __syncthreads();
            idle_since = -1;
            continue;
        }
        const uint32_t live = (uint32_t)__builtin_amdgcn_readfirstlane((int)RQ_LOAD(rq_ctl(q, RC_LIVE)));
        if (live == 0u || rq_aborted(q)) break;
        const long long now = wall_clock64();
        if (idle_since < 0) idle_since = now;
        else if (now - idle_since > kRollTimeoutTicks) { if (lane == 0) rq_abort(d, q); break; }
        __builtin_amdgcn_s_sleep(32);
    }
    ll_report(d, not_reset, overflow, finished);
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
}

}  // namespace mgym
