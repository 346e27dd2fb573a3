// lunar_lander.hip — LunarLanderV3 step()/reset() as gfx950 kernels, one wavefront lane per environment.
//
// Replaces the bodies of `impl Gym for LunarLanderV3` (reference src/box_2d/lunar_lander.rs: reset
// :727-917, step :919-1167) and the box2d-rs world it drives (ll_b2.h / ll_world.h restate Box2D).
//
// Data layout in HBM (engine-owned): one RECORD of 288 words (9 lines of 128 bytes) per environment — see enum Col, ll_env.h —
// plus 8 observation columns [8][n_pad].  Lines 0-1: what a free-flight step touches (3 bodies x {xf.p, sweep.c, angle, v, w,
// sleepTime}; 2 joints x {impulse.xy, motor, lower, upper}; 3 fat AABBs; prev_shaping; flag word; wind/torque indices; step/episode
// counters (Philox slots)); line 2: terrain smooth_y[11]; then 12 contact-cache slots x 16 words (worked on in place, KEY/SEQ/TOI staged
// in LDS: ll_b2.h) and the 13 words of an unfinished SolveTOI.  Every role moves whole lines of ITS environments: the contact role's
// list is a scattered ~9 % of the population, and with one column per word (rounds 1-3) each of its 4-byte accesses moved a 64-byte
// line — 568 MB counted per step of 262 144 envs against 274 MB with records (profiles/r03_lunarlander/ab_records.txt).
// Bound: f32 VALU issue / dependent-chain latency (180 Gauss-Seidel sweeps over 2 joints + contacts
// per step, ~3-5e4 flops per ~1 KB of state traffic) — NOT HBM; bench.py reports it that way.
//
// One mgym_step (LunarLanderEnv::step).  Below 376 832 envs per handle — TWO launches on the caller's stream:
//   ll_step_kernel<32>          the whole step, block roles by index: contact path over the list the previous call's epilogue built
//                               (Collide, island solve, SolveTOI with the wave's time-of-impact evaluations dealt out over all 64
//                               lanes), free-flight path over everyone else (register-only; an env that ends its step with a contact
//                               being created is resumed by the same wave on the contact path, after its island solve), preparation of
//                               the next resets into shadow records
//   ll_epilogue_kernel          fused auto-reset (prepared states copied in), the NEXT step's contact list and class bytes, counters
//   (+ ll_reset_kernel<32>, ll_general_kernel<64> for resets without a fitting prepared state: only while the host knows of an
//    event that can have produced one — set_state, dispersion override, unstaged resets)
// Larger populations (64-lane contact blocks, World records in registers) and MGYM_LL_SINGLE_LAUNCH=0: the multi-stream order of
// round 2 — ll_contact_kernel on the caller's stream beside ll_free_kernel + a short second contact launch on a helper stream (fork /
// join by events), then the same epilogue.  MGYM_LL_STAGED_RESET=0 computes a reset when the episode ends instead of preparing it ahead.
// mgym_rollout (K >= 8): ONE persistent launch, ll_roll.h.
// The product library carries these two orders and the knobs its tests use (mgym_get_info lists them).  The older orders and block sizes
// that only the measurements of rounds 2-3 used — counter memsets + ll_classify_kernel per step, select / copy launches, sequential kernels,
// follow-up time-of-impact launches, contact blocks of 8 / 16 / 40 / 48 lanes, other free-flight occupancies — are compiled with -DLL_DIAG
// only (tools/ build their own binaries that way).
#include <math.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#include <string.h>

#include "ll_kernel_common.h"
#include "ll_roll_types.h"

namespace mgym {

// Overlapped launch order, stage 0.  Which envs need the contact path is known from the flag word alone (a world exists,
// and a body sleeps or a contact is cached), so that list is built up front and the contact kernel runs BESIDE the
// free-flight kernel instead of behind it.  The class of every env is also written to LLDev::env_class: the two kernels
// run at the same time, and an env the contact kernel has already finished may look free-flight-eligible by its new flag
// word — the free-flight kernel goes by the class byte, which nothing changes during the step.
__global__ void __launch_bounds__(1024)
ll_classify_kernel(LLDev d) {
    __shared__ uint32_t s_cnt[17];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < d.n; base += stride) {  // block-uniform trip count
        const uint64_t i = base + threadIdx.x;
        bool to_general = false, touching = false;
        if (i < d.n) {
            const uint32_t flags = ST(C_FLAGS);
            to_general = (flags & F_HAS_WORLD) && !ll_free_eligible(flags);
            touching = d.bucket && (flags & F_TOUCHING);
            d.env_class[i] = to_general ? 1 : 0;
        }
        ll_push_block(d, L_GENERAL, to_general && !touching, (uint32_t)i, s_cnt);
        if (d.bucket) ll_push_back(d, L_GENERAL, L_GENERAL_T, to_general && touching, (uint32_t)i);
    }
}

// Every environment that is in free flight (no cached contact, all bodies awake) is stepped here with the
// register-only fast path (ll_free.h).  Sequential order: everything else — and every env the fast path declines —
// goes onto L_GENERAL for the contact kernel behind it.  Overlapped order (d.split): the contact class is being stepped
// by the contact kernel beside this one; only the declined envs are listed (L_LATE).  Finished envs go onto L_RESET.
// ll_free_pass: one pass of a wave over the 64 envs from `base`; returns whether this lane's env has to go down the contact path.
__device__ __forceinline__ bool ll_free_pass(const LLDev& d, const LLIo& io, const PolyTab& tab, uint64_t base, bool& not_reset, uint32_t& finished) {
    const uint64_t i = base + threadIdx.x;
    // overlapped order: envs of the contact class are being stepped (or were, already) by the contact kernel beside us
    const bool valid = i < d.n && !(d.split && d.env_class[i]);
    bool to_general = false, to_reset = false, is_done = false;
    if (valid) {
        const uint32_t flags = ST(C_FLAGS);
        if (!(flags & F_HAS_WORLD)) {  // assert!(self.lander.is_some(), "You forgot to call reset()") — :920
            not_reset = true;
            if (io.rew) io.rew[i] = 0.0f;
            if (io.done_out) io.done_out[i] = 0;
            if (io.trunc_out) io.trunc_out[i] = 0;
        } else if (!ll_free_eligible(flags)) {
            to_general = true;
        } else {
            FreeRegs f; EnvRegs e;
            ll_free_load(d, i, f, e);
            float state[8], reward, d0, d1; uint32_t done, moved;
            ll_dispersion(d, i, e, d0, d1);
            const int how = ll_free_env_step(d, i, f, e, tab, io.act[i], d0, d1, state, reward, done, moved);
            if (how == FREE_DONE) {
                ll_free_store(d, i, f, e);
                if (io.rew) io.rew[i] = reward;
                if (io.done_out) io.done_out[i] = (uint8_t)done;
                if (io.trunc_out) io.trunc_out[i] = 0;  // :1165 truncated: false
                ll_write_obs(d, io, i, state);
                to_reset = d.auto_reset && done;
                is_done = done != 0u;
                if (to_reset) d.env_class[i] = 2;   // finished: a fresh scene by the end of this call (ll_epilogue_kernel goes by this, not by the flag word)
            } else {
                // a contact is being created at the end of the step: the contact path finishes it, from the post-solve
                // state (FREE_RESUME) or, if a pending proxy already overlapped the ground at the start, from the old one
                if (how == FREE_RESUME && d.resume) ll_free_store_resume(d, i, f, e, moved);
                to_general = true;
            }
        }
    }
    ll_push(d, L_RESET, to_reset, (uint32_t)i);
    finished += (uint32_t)__popcll(__ballot(is_done));
    return to_general;
}

// The same for TWO environments per lane (base + lane and base + 64 + lane), taken through the 180 velocity sweeps side by side: the
// single-launch step kernel runs one wave per SIMD, and one environment per lane leaves that wave waiting on its own dependent chain
// (free-flight role alone: 0.59 ms against 0.30 ms for the stand-alone kernel at two waves per SIMD); two independent chains in one
// instruction stream fill those slots.  Same primitives per environment in the same order: identical results.
struct FreeLane {
    uint64_t i; bool run, to_general, to_reset, is_done;
    FreeRegs f; EnvRegs e; FreeSolve s;
};
__device__ __forceinline__ void ll_free_lane_begin(const LLDev& d, const LLIo& io, FreeLane& L, bool& not_reset) {
    const uint64_t i = L.i;
    L.run = false; L.to_general = false; L.to_reset = false; L.is_done = false;
    if (i < d.n) {
        // class byte, state columns and action are fetched together, before anything is decided: class byte -> flag word -> state would
        // be three dependent round trips to memory per environment (the envs of the contact class, one in ten, are fetched for nothing)
        const uint32_t cls = d.split ? (uint32_t)d.env_class[i] : 0u;
        const uint32_t action = io.act[i];
        ll_free_load(d, i, L.f, L.e);
        const uint32_t flags = L.f.flags;
        if (cls == 0u) {
            if (!(flags & F_HAS_WORLD)) {  // assert!(self.lander.is_some(), "You forgot to call reset()") — :920
                not_reset = true;
                if (io.rew) io.rew[i] = 0.0f;
                if (io.done_out) io.done_out[i] = 0;
                if (io.trunc_out) io.trunc_out[i] = 0;
            } else if (!ll_free_eligible(flags)) {
                L.to_general = true;
            } else {
                float d0, d1;
                ll_dispersion(d, i, L.e, d0, d1);
                L.run = ll_free_begin(d, i, L.f, L.e, action, d0, d1, L.s);
                L.to_general = !L.run;   // a pending proxy overlaps the ground: the contact path, from the old state
            }
        }
    }
}
__device__ __forceinline__ void ll_free_lane_finish(const LLDev& d, const LLIo& io, const PolyTab& tab, FreeLane& L) {
    if (!L.run) return;
    const uint64_t i = L.i;
    float state[8], reward; uint32_t done, moved;
    if (ll_free_finish(d, i, L.f, L.e, tab, L.s, state, reward, done, moved) == FREE_DONE) {
        ll_free_store(d, i, L.f, L.e);
        if (io.rew) io.rew[i] = reward;
        if (io.done_out) io.done_out[i] = (uint8_t)done;
        if (io.trunc_out) io.trunc_out[i] = 0;  // :1165 truncated: false
        ll_write_obs(d, io, i, state);
        L.to_reset = d.auto_reset && done;
        L.is_done = done != 0u;
        if (L.to_reset) d.env_class[i] = 2;
    } else {   // FREE_RESUME: a contact is being created at the end of the step
        ll_free_store_resume(d, i, L.f, L.e, moved);
        L.to_general = true;
    }
}
// Measured (profiles/r03_lunarlander/tune_free_role_envs_per_lane.txt): two per lane make the role alone 7 % faster (the compiler packs the
// pair into v_pk ops) but the step no faster, and double the step kernel's counted HBM traffic (1 156 vs 576 MB per launch: the second
// environment's registers spill around the sweep loop); three and four are slower.  One per lane is the default.
#ifndef LL_FREE_PER_LANE
#define LL_FREE_PER_LANE 1
#endif
constexpr int kFreePerLane = LL_FREE_PER_LANE;   // environments a lane of the single-launch step's free-flight role takes through the sweeps side by side
__device__ __forceinline__ void ll_free_passN(const LLDev& d, const LLIo& io, const PolyTab& tab, uint64_t base, bool& not_reset, uint32_t& finished,
                                              bool (&to_general)[kFreePerLane]) {
    FreeLane L[kFreePerLane] = {};   // (zeroed: a lane without an environment takes harmless numbers through the sweeps)
#pragma unroll
    for (int q = 0; q < kFreePerLane; ++q) { L[q].i = base + 64 * q + threadIdx.x; ll_free_lane_begin(d, io, L[q], not_reset); }
    const LLConst& k = LLK(d);
#ifndef LL_DIAG_FREE_SWEEPS   // diagnostic builds (tools/ll_role_time.hip) cut the sweep count to weigh the loop against the code around it
#define LL_DIAG_FREE_SWEEPS 180
#endif
    for (int it = 0; it < LL_DIAG_FREE_SWEEPS; ++it) {   // one basic block: the chains interleave
#pragma unroll
        for (int q = 0; q < kFreePerLane; ++q) ll_free_sweep(L[q].f, k, L[q].s);
    }
#pragma unroll
    for (int q = 0; q < kFreePerLane; ++q) {
        ll_free_lane_finish(d, io, tab, L[q]);
        ll_push(d, L_RESET, L[q].to_reset, (uint32_t)L[q].i);
        finished += (uint32_t)__popcll(__ballot(L[q].is_done));
        to_general[q] = L[q].to_general;
    }
}

template <int OCC>
__global__ void __launch_bounds__(kLLBlock, OCC)
ll_free_kernel(LLDev d, LLIo io) {
    __shared__ PolyTab tab;
    stage_tab(tab, LLK(d));
    bool not_reset = false;
    uint32_t finished = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < d.n; base += stride) {  // wave-uniform trip count
        const uint64_t i = base + threadIdx.x;
        const bool to_general = ll_free_pass(d, io, tab, base, not_reset, finished);
        if (d.split) {
            ll_push(d, L_LATE, to_general, (uint32_t)i);  // declined by the fast path: a short contact launch of its own
        } else {
            const bool touching = to_general && d.bucket && (ST(C_FLAGS) & F_TOUCHING);
            ll_push(d, L_GENERAL, to_general && !touching, (uint32_t)i);
            ll_push_back(d, L_GENERAL, L_GENERAL_T, to_general && touching, (uint32_t)i);
        }
    }
    ll_flush_done(d, finished);
    if (__any(not_reset) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_NOT_RESET);
}

// outputs of a finished step + its deferred reset (shared by the worklist kernels)
__device__ __forceinline__ void ll_emit(const LLDev& d, const LLIo& io, uint64_t i, const float state[8], float reward, uint32_t done) {
    if (io.rew) io.rew[i] = reward;
    if (io.done_out) io.done_out[i] = (uint8_t)done;
    if (io.trunc_out) io.trunc_out[i] = 0;  // :1165 truncated: false
    ll_write_obs(d, io, i, state);
}

// The contact path over a compacted list (`which` = L_GENERAL, or L_LATE for the second launch of the overlapped order):
// wind / engines, then b2World::Step — Collide, the 180-sweep island solve, position iterations, new contacts, SolveTOI
// with all its sub-steps (toi_budget < 0, the default).  With toi_budget >= 0 (MGYM_LL_TOI_ROUNDS, profiling only) SolveTOI
// stops after that many sub-steps and the env goes, with its unfinished state in the C_MID columns, onto L_TOI0 for
// ll_toi_kernel.
// where a wave of the contact path takes its environments from: entry q of a list that is filled from both ends
// (c0 entries from the front — rounded up to c0_up slots —, c1 from the back, `back` = index of the last slot)
// back_first (the product's bucketing, 64-lane blocks): the c1 touching envs from the back come FIRST (rounded up to c0_up slots, which then
// holds c1's round-up), the others behind them — blocks start in list order, so the long blocks start first and the short ones fill the end
struct ContactList { const uint32_t* list; uint64_t back, c0, c0_up, c1; bool spread, back_first; };
LLD uint64_t ll_list_total(const ContactList& L) { return L.back_first ? L.c0_up + L.c0 : L.c0_up + L.c1; }
LLD bool ll_list_has(const ContactList& L, uint64_t q, uint64_t total) {
    return L.back_first ? (q < L.c1 || (q >= L.c0_up && q < total)) : (q < L.c0 || (q >= L.c0_up && q < total));
}
LLD uint64_t ll_list_entry(const ContactList& L, uint64_t q, uint64_t total) {
    if (L.back_first) return q < L.c1 ? L.list[L.back - q] : L.list[q - L.c0_up];
    if (L.spread) {
        const uint64_t h0 = q * L.c1 / total, h1 = (q + 1) * L.c1 / total;
        return h1 > h0 ? L.list[L.back - h0] : L.list[q - h0];
    }
    return q < L.c0 ? L.list[q] : L.list[L.back - (q - L.c0_up)];
}

// The contact path over a compacted list: wind / engines, then b2World::Step — Collide, the 180-sweep island solve, position
// iterations, new contacts, SolveTOI with all its sub-steps (toi_budget < 0, the default).  With toi_budget >= 0
// (MGYM_LL_TOI_ROUNDS, profiling only) SolveTOI stops after that many sub-steps and the env goes, with its unfinished state in the
// C_MID columns, onto L_TOI0 for ll_toi_kernel.  The wave takes entries q_first + lane, q_first + q_stride + lane, ...
// (S.tab must have been staged.)
template <int BLK>
__device__ __forceinline__ void ll_contact_body(const LLDev& d, const LLIo& io, int toi_budget, const ContactList& CL, uint64_t q_first, uint64_t q_stride,
                                                VConstraint* far_lane0, int far_stride, ContactLds<BLK>& S, bool& not_reset, uint32_t& overflow, uint32_t& finished,
                                                int fill = BLK /* lanes of the block that take a list entry (the single-launch step deals a short list out over more blocks) */) {
    constexpr bool kWorldLds = BLK <= 32;
    constexpr int kThreads = ll_contact_threads(BLK);
    const PolyTab& tab = S.tab;
    const bool env_lane = threadIdx.x < BLK;
    const int own = env_lane ? (int)threadIdx.x : 0;             // per-lane records exist for the env lanes only
    const CtHot hot{(LL_LDS uint32_t*)S.hot + own, (uint32_t)BLK, BLK > 32 ? 0u : 1u};  // 64-lane blocks: KEY and SEQ only
    PConstraint l_pc[kSolverCap];
    CSolverMem mem;
    mem.vc = &S.vc[own]; mem.vc_stride = BLK; mem.vc_near = d.vc_near < kVcNearLds ? d.vc_near : kVcNearLds; mem.pc = l_pc; mem.pc_stride = 1; mem.cap = kSolverCap;
    mem.vc_far = far_lane0 + own; mem.vc_far_stride = far_stride;
    const uint64_t total = ll_list_total(CL);
    for (uint64_t q0 = q_first; q0 < total; q0 += q_stride) {  // block-uniform
        const uint64_t q = q0 + threadIdx.x;
        const bool have = env_lane && (int)threadIdx.x < fill && ll_list_has(CL, q, total);
        uint64_t i = 0;
        bool to_toi = false, to_reset = false, is_done = false;
        bool stepping = false;      // this lane is inside world.step
        bool islanding = false;     // ... between Collide's list walk and the island solve
        int n_refresh = 0;          // manifold refreshes this lane has listed
        bool step_complete = true;  // ... and has finished it (false only with toi_budget >= 0: profiling)
        World w_local;
        World& w = kWorldLds ? S.world[kWorldLds ? own : 0] : w_local;
        EnvRegs e;
        uint32_t action = 0u;
        if (have) {
            i = ll_list_entry(CL, q, total);
            w.t = (LL_LDS WorldTmp*)S.tmp + own;
            ll_load(d, i, w, e, hot);
            if (!e.has_world) {  // assert!(self.lander.is_some(), "You forgot to call reset()") — :920
                not_reset = true;
                if (io.rew) io.rew[i] = 0.0f;
                if (io.done_out) io.done_out[i] = 0;
                if (io.trunc_out) io.trunc_out[i] = 0;
            } else {
                action = io.act[i];
                if (w.resume) {   // the free-flight path has taken this step up to the end of the island solve (ll_free.h)
                    ll_resume_after_island(w);
                } else {
                    float d0, d1, m_power, s_power;
                    ll_dispersion(d, i, e, d0, d1);
                    ll_pre_step(w.b[0], w.legs[0], w.legs[1], e, LLK(d), action, d0, d1, m_power, s_power);   // wind / engines, :926-1048
                    if constexpr (kWorldLds) { n_refresh = world_step_pre(w); islanding = true; }              // world.step, :1066: Collide's list walk
                    else world_step_begin(w, tab, LLK(d), mem);                                                // ... or everything up to SolveTOI
                }
                stepping = true;
            }
        }
        // b2ContactManager::Collide: the manifold refreshes of the wave's cached contacts (b2CollideEdgeAndPolygon + warm-start matching,
        // ~10 k cycles each, up to ~7 per lane) are dealt out over all 64 lanes like the time-of-impact evaluations below; the callbacks
        // they imply are applied by the owner afterwards, in list order (ll_world.h collide_list / _refresh / _apply)
        if constexpr (kWorldLds) {
            int incl = n_refresh;
            for (int dlt = 1; dlt < 64; dlt <<= 1) { const int v = __shfl_up(incl, dlt); if ((int)threadIdx.x >= dlt) incl += v; }
            const int n_tasks = __shfl(incl, 63);
            if (n_tasks > 0) {   // wave-uniform
                const int offs = incl - n_refresh;
                for (int j = 0; j < n_refresh; ++j) S.task[offs + j] = (uint16_t)(((uint32_t)own << 4) | ((LL_LDS WorldTmp*)S.tmp + own)->idx[1][j]);
                __syncthreads();
                // the executing lane's polygon buffer: its own working record, or (helper lanes) a piece of the constraint columns,
                // which nothing uses before the island solver is set up
                constexpr int kHelpRoom = (int)(sizeof(S.vc) / (2 * kMaxPoly * sizeof(V2)));
                constexpr int kExec = BLK + ((kThreads - BLK) < kHelpRoom ? (kThreads - BLK) : kHelpRoom);   // lanes that have a polygon buffer (all 64 for 32-lane blocks)
                V2* const poly_tmp = env_lane ? (V2*)((LL_LDS WorldTmp*)S.tmp + own)->poly_tmp : (V2*)S.vc + (size_t)(threadIdx.x - BLK) * (2 * kMaxPoly);
                if ((int)threadIdx.x < kExec)
                    for (int t = (int)threadIdx.x; t < n_tasks; t += kExec) {
                        const uint32_t task = S.task[t];
                        collide_refresh(S.world[task >> 4], tab, (int)(task & 15u), poly_tmp);
                    }
                __syncthreads();
            }
            if (islanding) world_step_island(w, tab, LLK(d), mem);   // the callbacks of Collide, then b2World::Solve
        }
        // b2World::SolveTOI.  With the World records in LDS the wave runs its passes in lock step and DEALS THE
        // TIME-OF-IMPACT EVALUATIONS OF A PASS OUT OVER ALL 64 LANES: an evaluation (GJK + root finder: the longest
        // dependent chain of the step, ~25 k cycles) is a pure function of the body's sweep and the edge, both in the
        // owner's LDS record, and its result goes into the owner's staged TOI word — so any lane can do it.  A lane's own
        // list is 1-9 evaluations long while most lanes of the wave hold none: one lane per evaluation turns the wave's
        // ~18 sequential evaluation slots per step into ~6 (the arithmetic of each evaluation is unchanged).
        if constexpr (kWorldLds) {
            if (toi_budget < 0) {
                ToiLoop L;
                bool running = stepping && toi_begin(w, L, true);
                int budget = -1;
                while (__any(running)) {   // wave-uniform
                    const int n_need = running ? toi_list(w, L) : 0;
                    int incl = n_need;     // inclusive prefix sum over the wave: where this lane's evaluations go in the task table
                    for (int dlt = 1; dlt < 64; dlt <<= 1) { const int v = __shfl_up(incl, dlt); if ((int)threadIdx.x >= dlt) incl += v; }
                    const int n_tasks = __shfl(incl, 63);
                    const int offs = incl - n_need;
                    for (int j = 0; j < n_need; ++j) S.task[offs + j] = (uint16_t)(((uint32_t)own << 4) | ((LL_LDS WorldTmp*)S.tmp + own)->idx[1][j]);
                    __syncthreads();
                    for (int t = (int)threadIdx.x; t < n_tasks; t += kThreads) {
                        const uint32_t task = S.task[t];
                        toi_evaluate(S.world[task >> 4], tab, (int)(task & 15u));
                    }
                    __syncthreads();
                    if (running) running = toi_advance(w, tab, LLK(d), mem, kStepDt, L, budget) == TOI_AGAIN;
                }
            } else if (stepping) {
                step_complete = solve_toi_part(w, tab, LLK(d), mem, kStepDt, true, toi_budget);
            }
        } else if (toi_budget < 0) {
            // 64-lane blocks: every lane holds an env and the World records are private (registers / scratch), but a wave's evaluations are as
            // unevenly spread as in the small blocks (most lanes none, a few lanes up to nine per pass: the wave used to take ~12 evaluation slots
            // per step for ~2 evaluations per env).  They are dealt out as TASK RECORDS (ll_world.h toi_task_*): the owners write the operands into
            // the block's constraint columns — which nothing uses between two time-of-impact islands —, the wave's lanes take one record each,
            // the owners apply the results.
            constexpr int kTaskCap = (int)(sizeof(S.vc) / (kToiTaskWords * sizeof(float)));
            LL_LDS float* const trec = (LL_LDS float*)S.vc;
            ToiLoop L;
            bool running = stepping && toi_begin(w, L, true);
            int budget = -1;
            while (__any(running)) {   // wave-uniform (a 64-lane block is one wave)
                const int n_need = running ? toi_list(w, L) : 0;
                int incl = n_need;
                for (int dlt = 1; dlt < 64; dlt <<= 1) { const int v = __shfl_up(incl, dlt); if ((int)threadIdx.x >= dlt) incl += v; }
                const int n_tasks = __shfl(incl, 63);
                const int offs = incl - n_need;
                const uint8_t* const need = (const uint8_t*)((LL_LDS WorldTmp*)S.tmp + own)->idx[1];
                for (int c0 = 0; c0 < n_tasks; c0 += kTaskCap) {   // (one round unless the wave holds more than kTaskCap evaluations)
                    for (int j = 0; j < n_need; ++j) { const int t = offs + j - c0; if (t >= 0 && t < kTaskCap) toi_task_write(w, need[j], trec + t, kTaskCap); }
                    __syncthreads();
                    const int n_here = n_tasks - c0 < kTaskCap ? n_tasks - c0 : kTaskCap;
                    for (int t = (int)threadIdx.x; t < n_here; t += kThreads) toi_task_run(trec + t, kTaskCap, tab);
                    __syncthreads();
                    for (int j = 0; j < n_need; ++j) { const int t = offs + j - c0; if (t >= 0 && t < kTaskCap) toi_task_apply(w, need[j], trec + t, kTaskCap); }
                    __syncthreads();
                }
                if (running) running = toi_advance(w, tab, LLK(d), mem, kStepDt, L, budget) == TOI_AGAIN;
            }
        } else {
            if (stepping) step_complete = solve_toi_part(w, tab, LLK(d), mem, kStepDt, true, toi_budget);
        }
        if (stepping) {
            if (step_complete) {
                float state[8], reward; uint32_t done;
                ll_step_finish(w, e, action, state, reward, done);
                ll_store(d, i, w, e);
                ll_emit(d, io, i, state, reward, done);
                is_done = done != 0u;
                to_reset = d.auto_reset && done;
                if (to_reset) d.env_class[i] = 2;   // (see ll_free_pass)
            } else {
                ll_store(d, i, w, e, true);
                to_toi = true;
            }
            overflow |= w.overflow;
        }
        ll_push(d, L_TOI0, to_toi, (uint32_t)i);
        ll_push(d, L_RESET, to_reset, (uint32_t)i);
        finished += (uint32_t)__popcll(__ballot(is_done));
    }
}


// The contact path as a launch of its own (`which` = L_GENERAL, or L_LATE for the second launch of the overlapped order)
template <int BLK>
__global__ void __launch_bounds__(ll_contact_threads(BLK)) LL_CONTACT_ATTR
ll_contact_kernel(LLDev d, LLIo io, int toi_budget, int which) {
    __shared__ ContactLds<BLK> S;
    stage_tab(S.tab, LLK(d));
    bool not_reset = false;
    uint32_t overflow = 0u, finished = 0u;
    ContactList CL;
    CL.list = d.work_list + (uint64_t)which * d.n_pad;  // L_GENERAL (two-ended) or L_LATE
    CL.back = d.n_pad - 1u;
    CL.c0 = d.work_count[which]; CL.c1 = which == L_GENERAL ? d.work_count[L_GENERAL_T] : 0u;
    if (d.fused_tail && which == L_GENERAL && blockIdx.x == 0 && threadIdx.x == 0) {
        // fused order: the lists the LAST launches of the previous step consumed (its direct resets, behind the epilogue) are
        // cleared by the first kernel of this one; everything else is cleared by the epilogue's last block
        d.work_count[L_RESET_DIRECT] = 0u; d.work_count[L_RESET_SLOW] = 0u;
    }
    // bucket 1: the touching bucket starts at a block boundary (waves of one kind); bucket 2: the touching envs — the ones
    // that take time-of-impact sub-steps (59 % of them against 12 % of the others) — are dealt out evenly over the waves:
    // of the first q entries floor(q * c1 / total) come from the touching list
    CL.spread = d.bucket == 2;
    CL.back_first = d.bucket == 1;
    CL.c0_up = CL.spread ? CL.c0 : ((CL.back_first ? CL.c1 : CL.c0) + BLK - 1) / BLK * BLK;
    // the two contact launches of the overlapped order run at the same time: each has its own slice of the workspace
    ll_contact_body<BLK>(d, io, toi_budget, CL, (uint64_t)blockIdx.x * BLK, (uint64_t)gridDim.x * BLK,
                         (which == L_LATE ? d.vc_far_late : d.vc_far) + (uint64_t)blockIdx.x * BLK, (int)(gridDim.x * BLK), S, not_reset, overflow, finished);
    ll_report(d, not_reset, overflow, finished);
}

#ifdef LL_DIAG
// Stage 3 of mgym_step, rounds r = 0 .. kToiRounds-1 over ever shorter lists: continue b2World::SolveTOI of the envs on
// L_TOI0 + r by `budget` sub-steps (advance the body to its earliest impact, solve the TOI island, re-evaluate that body's
// times of impact); finished envs get the tail of the step (observation, reward, termination), the others move on to
// L_TOI0 + r + 1.  The last round runs with budget < 0 (to the end).  Per general env-step the reference does 0 sub-steps
// 74 % of the time, 1: 17 %, 2: 7 %, >= 3: 2 % (tools/ll_work_stats.c) — inside one launch every wave would wait for its
// slowest lane through all of them.
template <int BLK>
__global__ void __launch_bounds__(BLK)
ll_toi_kernel(LLDev d, LLIo io, int round, int budget) {
    __shared__ PolyTab tab;
    __shared__ VConstraint s_vc[kSolverCap * BLK];
    LL_HOT_DECL(BLK);
    LL_TMP_DECL(BLK);
    stage_tab(tab, LLK(d));
    PConstraint l_pc[kSolverCap];
    CSolverMem mem;
    mem.vc = &s_vc[threadIdx.x]; mem.vc_stride = BLK; mem.vc_near = kSolverCap; mem.vc_far = nullptr; mem.vc_far_stride = 0; mem.pc = l_pc; mem.pc_stride = 1; mem.cap = kSolverCap;
    uint32_t overflow = 0u;
    uint32_t finished = 0;
    const uint32_t* list = d.work_list + (uint64_t)(L_TOI0 + round) * d.n_pad;
    const uint64_t total = d.work_count[L_TOI0 + round];
    for (uint64_t q0 = (uint64_t)blockIdx.x * BLK; q0 < total; q0 += (uint64_t)gridDim.x * BLK) {  // block-uniform
        const uint64_t q = q0 + threadIdx.x;
        uint64_t i = 0;
        bool to_toi = false, to_reset = false, is_done = false;
        if (q < total) {
            i = list[q];
            World w; EnvRegs e;
            w.t = LL_TMP_PTR();
            ll_load(d, i, w, e, hot, true);
            if (ll_step_continue(w, tab, LLK(d), mem, budget)) {
                float state[8], reward; uint32_t done;
                ll_step_finish(w, e, io.act[i], state, reward, done);
                ll_store(d, i, w, e);
                ll_emit(d, io, i, state, reward, done);
                is_done = done != 0u;
                to_reset = d.auto_reset && done;
            } else {
                ll_store(d, i, w, e, true);
                to_toi = true;
            }
            overflow |= w.overflow;
        }
        if (round + 1 < kToiRounds) ll_push(d, L_TOI0 + round + 1, to_toi, (uint32_t)i);  // (never set when budget < 0)
        ll_push(d, L_RESET, to_reset, (uint32_t)i);
        finished += (uint32_t)__popcll(__ballot(is_done));
    }
    ll_flush_done(d, finished);
    if (__any(overflow & 1u) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_CONTACT_OVERFLOW);
    if (__any(overflow & 2u) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_SOLVER_OVERFLOW);
}

#endif   // LL_DIAG

// Whole general step in ONE launch (no time-of-impact rounds): the declined resets of L_RESET_SLOW (list != nullptr),
// Testable::reset_deterministic, and the MGYM_LL_GENERAL_ONLY debugging aid.  list == nullptr: every env with
// `forced` = 0 general step, 1 reset(), 2 Testable::reset_deterministic.
template <int BLK>
__global__ void __launch_bounds__(BLK)
ll_general_kernel(LLDev d, LLIo io, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, int forced) {
    __shared__ PolyTab tab;
    // velocity constraints of the contact solver: LDS, one column per lane (31-word records: conflict-free)
    __shared__ VConstraint s_vc[kSolverCap * BLK];
    LL_HOT_DECL(BLK);
    LL_TMP_DECL(BLK);
    stage_tab(tab, LLK(d));
    PConstraint l_pc[kSolverCap];
    CSolverMem mem;
    mem.vc = &s_vc[threadIdx.x]; mem.vc_stride = BLK; mem.vc_near = kSolverCap; mem.vc_far = nullptr; mem.vc_far_stride = 0; mem.pc = l_pc; mem.pc_stride = 1; mem.cap = kSolverCap;
    bool not_reset = false;
    uint32_t overflow = 0u;
    uint32_t finished = 0;
    const uint64_t total = list ? (uint64_t)*count : d.n;
    // block-uniform trip count (the deferred-reset push below is a wave-level collective)
    for (uint64_t q0 = (uint64_t)blockIdx.x * blockDim.x; q0 < total; q0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = q0 + threadIdx.x;
        uint64_t i = q;
        bool defer_reset = false, is_done = false;
        if (q < total) {
        int mode = forced;
        if (list) { uint32_t ent = list[q]; i = ent & ~kWorkReset; mode = (ent & kWorkReset) ? 1 : 0; }
        World w; EnvRegs e;
        w.t = LL_TMP_PTR();
        ll_load(d, i, w, e, hot);
        float state[8];
        if (mode == 0 && !e.has_world) {
            not_reset = true;
            if (io.rew) io.rew[i] = 0.0f;
            if (io.done_out) io.done_out[i] = 0;
            if (io.trunc_out) io.trunc_out[i] = 0;
        } else {
        // One inlined copy of the step body serves all three uses: a plain step (pass 0, mode 0), the implicit
        // step(0) that ends reset() (lunar_lander.rs:911), and the fused auto-reset after a finished step (pass 1).
        bool need_reset = mode == 1;
        for (int pass = 0; pass < 2; ++pass) {
            uint32_t action = 0u;
            if (mode == 2) {  // Testable::reset_deterministic: no step
                const float H = VIEWPORT_H / SCALE;
                float height[12];
                for (int t = 0; t < 12; ++t) height[t] = H / 8.0f;  // :1278-1280
                ll_build_scene(w, e, tab, LLK(d), height, VIEWPORT_H / SCALE * 0.8f, false, 0.0f, 0.0f, 0, 0, true);
                e.step = 1u;
                ll_observe(w.b[0], w.legs[0], w.legs[1], state);  // :1441
                break;
            }
            if (need_reset) {
                ll_reset_scene(d, i, w, e, tab);  // reset() up to the implicit step (lunar_lander.rs:727-908)
            } else {
                action = io.act[i];
            }
            float reward, d0, d1; uint32_t done;
            ll_dispersion(d, i, e, d0, d1);
            ll_env_step(w, e, tab, LLK(d), mem, action, d0, d1, state, reward, done);
            if (need_reset) { e.episode += 1u; break; }
            if (io.rew) io.rew[i] = reward;
            if (io.done_out) io.done_out[i] = (uint8_t)done;
            if (io.trunc_out) io.trunc_out[i] = 0;
            is_done = done != 0u;
            if (!(d.auto_reset && done)) break;
            if (list) { defer_reset = true; break; }  // compacted: ll_reset_kernel draws the new episode
            need_reset = true;
        }
        overflow |= w.overflow;
        ll_store(d, i, w, e);
        ll_write_obs(d, io, i, state);
        }
        }
        if (list) ll_push(d, L_RESET, defer_reset, (uint32_t)i);
        finished += (uint32_t)__popcll(__ballot(is_done));
    }
    ll_flush_done(d, finished);
    if (__any(not_reset) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_NOT_RESET);
    if (__any(overflow & 1u) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_CONTACT_OVERFLOW);
    if (__any(overflow & 2u) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_SOLVER_OVERFLOW);
}

// reset() on a compacted list (or on every env when list == nullptr): draw the new episode's scene
// (lunar_lander.rs:727-908), then the implicit step(0) that ends reset() (:911-916).  A fresh scene is in free
// flight, so that step runs on the register-only fast path; if the fast path declines (it never creates a
// contact) the env goes to the L_RESET_SLOW list and ll_general_kernel redoes the reset on the general path —
// the scene is a pure function of (seed, env id, episode), and the episode counter only advances on success.
// Keeping resets out of ll_general_kernel matters because nearly every one of its blocks holds a finishing env:
// an inline reset made every block pay a second full world step.
// (hot: this lane's staged contact words; one pass of the lanes of a wave over entries q of the list, or over env q itself)
__device__ __forceinline__ void ll_reset_pass(const LLDev& d, const LLIo& io, const PolyTab& tab, const CtHot& hot, const uint32_t* __restrict__ list, uint64_t q, uint64_t total, int slow_list) {
    uint64_t i = q;
    bool slow = false;
    if (q < total) {
        if (list) i = list[q];
        if (d.episode_src) ST(C_EPISODE) = ST_AT(d.episode_src, C_EPISODE);  // preparing in the shadow columns: the episode to draw is the live one
        V2 force; float torque;
        {
            World w; EnvRegs e;
            w.t = nullptr;  // (a reset never reaches the contact path)
            ll_load(d, i, w, e, hot);
            ll_reset_scene(d, i, w, e, tab);
            force = w.b[0].force; torque = w.b[0].torque;  // the initial random push (:845-849) is not a state column
            ll_store(d, i, w, e);
        }
        FreeRegs f; EnvRegs e;
        ll_free_load(d, i, f, e);
        f.b[0].force = force; f.b[0].torque = torque;
        float state[8], reward, d0, d1; uint32_t done;
        ll_dispersion(d, i, e, d0, d1);
        uint32_t moved;
        if (ll_free_env_step(d, i, f, e, tab, 0u, d0, d1, state, reward, done, moved) == FREE_DONE) {
            e.episode += 1u;
            ll_free_store(d, i, f, e);
            ll_write_obs(d, io, i, state);
        } else {
            slow = true;
        }
    }
    ll_push(d, slow_list, slow, (uint32_t)i | kWorkReset);
    if (d.prep && list) ll_push(d, L_PREP, q < total && !slow, (uint32_t)i);
}

template <int BLK>
__global__ void __launch_bounds__(BLK)
ll_reset_kernel(LLDev d, LLIo io, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, int slow_list) {
    __shared__ PolyTab tab;
    LL_HOT_DECL(BLK);
    stage_tab(tab, LLK(d));
    const uint64_t total = list ? (uint64_t)*count : d.n;
    for (uint64_t q0 = (uint64_t)blockIdx.x * blockDim.x; q0 < total; q0 += (uint64_t)gridDim.x * blockDim.x)  // block-uniform
        ll_reset_pass(d, io, tab, hot, list, q0 + threadIdx.x, total, slow_list);
}

// ---- the whole step in ONE launch ---------------------------------------------------------------------------------------------
// Blocks of one wave each, three roles by block index, all on the caller's stream — no helper streams, no events, nothing a
// hipGraph executor or the number of hardware queues could serialise:
//   [0, g_contact)   the contact path over L_GENERAL (ll_contact_body); dispatched first, so the step's longest dependent chains
//                    start at once on SIMDs of their own
//   [.., + g_free)   the free-flight path over everyone else, 128 envs per wave (two per lane, side by side through the sweeps); the few envs a wave has to hand to the contact path
//                    (a contact is created at the end of their step) it takes through ll_contact_body ITSELF, right away, from
//                    the post-solve state it has just stored (F_RESUME)
//   [.., + g_prep)   staged resets: the next episode of the envs the previous call's epilogue listed (L_PREP), into the shadow
//                    columns; these blocks come last and run when SIMDs fall free
// One register allocation serves all three (512 registers, one wave per SIMD — what the contact path needs; the free-flight
// path measures the same at that occupancy: profiles/r02_lunarlander/tune_free_kernel_occupancy_overlapped.txt).
template <int BLK>
__global__ void __launch_bounds__(64) LL_CONTACT_ATTR
ll_step_kernel(LLDev d, LLDev sh, LLIo io, unsigned g_contact, unsigned g_free) {
    static_assert(BLK <= 32, "single-launch step: World records in LDS");
    __shared__ ContactLds<BLK> S;
    bool not_reset = false;
    uint32_t overflow = 0u, finished = 0u;
#ifdef LL_ROLE_MASK   // diagnostic builds (tools/ll_role_time.sh): run only some of the roles, to time them apart (breaks the physics)
    if (!((LL_ROLE_MASK >> (blockIdx.x < g_contact ? 0 : blockIdx.x < g_contact + g_free ? 1 : 2)) & 1)) return;
#endif
    if (blockIdx.x < g_contact) {
        ContactList CL;
        CL.list = d.work_list + (uint64_t)L_GENERAL * d.n_pad; CL.back = 0; CL.c0 = d.work_count[L_GENERAL]; CL.c1 = 0; CL.spread = false; CL.back_first = false;
        CL.c0_up = (CL.c0 + BLK - 1) / BLK * BLK;
        if (blockIdx.x == 0 && threadIdx.x == 0) { d.work_count[L_RESET_DIRECT] = 0u; d.work_count[L_RESET_SLOW] = 0u; }   // (consumed by the last launches of the previous call)
        // LANES PER BLOCK by the length of the list: the step ends with the slowest contact block, and a block's chain grows with the lanes
        // it carries (sub-step passes = the maximum over its lanes, every pass as long as its slowest lane) — so a short list is dealt out
        // over MORE blocks of fewer lanes, as many as run at once beside the free-flight role (d.contact_blocks, from the population).
        // 65 536 envs: 0.95 -> 0.79 ms per step, 131 072: 0.99 -> 0.85, 196 608: 1.00 -> 0.92, 262 144: 1.00 -> 0.98
        // (profiles/r03_lunarlander/block_lanes_by_population.txt).  The lanes per block follow THIS step's list, so a policy that keeps more
        // envs on the ground simply gets fuller blocks.
        uint32_t fill = BLK;
        if (d.contact_blocks > 0) {
            fill = (uint32_t)((CL.c0 + (uint64_t)d.contact_blocks - 1) / (uint64_t)d.contact_blocks);
            fill = fill < 8u ? 8u : fill > (uint32_t)BLK ? (uint32_t)BLK : fill;
        }
        if ((uint64_t)blockIdx.x * fill >= CL.c0) return;   // most blocks of this role: the list is far shorter than the grid
        stage_tab(S.tab, LLK(d));
        ll_contact_body<BLK>(d, io, -1, CL, (uint64_t)blockIdx.x * fill, (uint64_t)g_contact * fill, d.vc_far + (uint64_t)blockIdx.x * BLK, (int)(g_contact * BLK),
                             S, not_reset, overflow, finished, (int)fill);
    } else if (blockIdx.x < g_contact + g_free) {
        const unsigned fb = blockIdx.x - g_contact;
        stage_tab(S.tab, LLK(d));
        constexpr int kPerWave = 64 * kFreePerLane;
        for (uint64_t base = (uint64_t)fb * kPerWave; base < d.n; base += (uint64_t)g_free * kPerWave) {  // wave-uniform trip count
            bool tg[kFreePerLane];
            ll_free_passN(d, io, S.tab, base, not_reset, finished, tg);
            unsigned long long hand[kFreePerLane], any = 0ull;
#pragma unroll
            for (int q = 0; q < kFreePerLane; ++q) { hand[q] = __ballot(tg[q]); any |= hand[q]; }
#ifdef LL_DIAG_DROP_HANDOVERS   // diagnostic builds (tools/ll_role_time.sh): what does the free-flight role cost WITHOUT taking its hand-overs through the contact path? (breaks the physics)
            any = 0ull;
#endif
            if (any != 0ull) {   // (about one wave in three)
                const unsigned long long below = (1ull << threadIdx.x) - 1ull;
                int at = 0;
#pragma unroll
                for (int q = 0; q < kFreePerLane; ++q) {
                    if (tg[q]) S.late[at + __popcll(hand[q] & below)] = (uint32_t)(base + 64 * q + threadIdx.x);
                    at += __popcll(hand[q]);
                }
                __syncthreads();
                ContactList CL;
                CL.list = S.late; CL.back = 0; CL.c0 = (uint64_t)at; CL.c1 = 0; CL.spread = false; CL.back_first = false; CL.c0_up = CL.c0;
                ll_contact_body<BLK>(d, io, -1, CL, 0, BLK, d.vc_far_late + (uint64_t)fb * BLK, (int)(g_free * BLK), S, not_reset, overflow, finished);
                __syncthreads();
            }
        }
    } else {
        const unsigned pb = blockIdx.x - g_contact - g_free, g_prep = gridDim.x - g_contact - g_free;
        stage_tab(S.tab, LLK(d));
        const uint32_t* list = d.work_list + (uint64_t)L_PREP * d.n_pad;
        const uint64_t total = d.work_count[L_PREP];
        const LLIo none{nullptr, nullptr, nullptr, nullptr, nullptr};
        if (threadIdx.x < BLK) {   // (the staged contact words of a reset: BLK columns)
            const CtHot hot{(LL_LDS uint32_t*)S.hot + threadIdx.x, (uint32_t)BLK, 1u};
            for (uint64_t q0 = (uint64_t)pb * BLK; q0 < total; q0 += (uint64_t)g_prep * BLK)
                ll_reset_pass(sh, none, S.tab, hot, list, q0 + threadIdx.x, total, (int)L_PREP_SLOW);
        }
        __syncthreads();
        if (threadIdx.x == 0) {   // the last block of this role clears the list it has consumed (the epilogue behind us fills it again)
            const uint32_t ticket = atomicAdd(d.work_count + C_TICKET2, 1u);
            if (ticket == g_prep - 1) { d.work_count[L_PREP] = 0u; d.work_count[L_PREP_SLOW] = 0u; d.work_count[C_TICKET2] = 0u; }
        }
    }
    ll_report(d, not_reset, overflow, finished);
}

// mgym_rollout: K steps in one persistent launch, environments advancing independently — ll_roll.h, compiled in ll_roll.hip
int ll_rollout_blocks_per_cu(int* per_cu);
void ll_rollout_ring_init(hipStream_t s, const RollQ& q);
void ll_rollout_begin(hipStream_t s, const RollQ& q, uint32_t n);
void ll_rollout_launch(hipStream_t s, unsigned grid, const LLDev& d, const LLIo& io, const RollQ& q);
void ll_rollout_helper_launch(hipStream_t s, unsigned grid, const LLDev& d, const LLIo& io, const RollQ& q);

// Staged resets.  The state an env has after reset() — scene, implicit step(0), observation — is a pure function of
// (seed, env id, episode counter), so it does not have to be computed when the episode ends, on the critical path of
// that step (reset kernel: 0.11 ms behind the contact kernel): it is PREPARED in advance, in shadow columns, by the same
// reset kernels on a third stream while later steps run, and the auto-reset of a finished env is a copy.
// The prepared state carries its episode counter: it fits iff shadow episode == live episode + 1; whatever changed the
// live counter in between (mgym_reset, mgym_set_state) just sends the env down the direct path (L_RESET_DIRECT) once.
#ifdef LL_DIAG   // (the unfused order's two launches; the fused order does both inside ll_epilogue_kernel)
// (1) which finished envs have a fitting prepared state: those go onto L_PREP (their reset is the copy below, and their
//     NEXT reset is prepared afterwards), the others onto L_RESET_DIRECT.  One thread per list entry, one atomic per block and list.
__global__ void __launch_bounds__(256)
ll_apply_select_kernel(LLDev d, const uint32_t* __restrict__ shadow) {
    __shared__ uint32_t s_cnt[5];
    const uint32_t* list = d.work_list + (uint64_t)L_RESET * d.n_pad;
    const uint64_t total = d.work_count[L_RESET];
    for (uint64_t q0 = (uint64_t)blockIdx.x * blockDim.x; q0 < total; q0 += (uint64_t)gridDim.x * blockDim.x) {  // block-uniform
        const uint64_t q = q0 + threadIdx.x;
        uint64_t i = 0;
        bool fits = false, direct = false;
        if (q < total) {
            i = list[q];
            fits = ST_AT(shadow, C_EPISODE) == ST(C_EPISODE) + 1u;
            direct = !fits;
        }
        ll_push_block(d, L_PREP, fits, (uint32_t)i, s_cnt);
        ll_push_block(d, L_RESET_DIRECT, direct, (uint32_t)i, s_cnt);
    }
}
// (2) the copy: one block per env of L_PREP, one thread per word (the columns of one env lie n_pad words apart)
__global__ void __launch_bounds__(128)
ll_apply_copy_kernel(LLDev d, const uint32_t* __restrict__ shadow, const float* __restrict__ shadow_obs, LLIo io) {
    const uint32_t* list = d.work_list + (uint64_t)L_PREP * d.n_pad;
    const uint64_t total = d.work_count[L_PREP];
    const int t = threadIdx.x;
    for (uint64_t q = blockIdx.x; q < total; q += gridDim.x) {
        const uint64_t i = list[q];
        if (t < C_CONTACT) ST(t) = ST_AT(shadow, t);
        else if (t < C_CONTACT + kSlots) ST(C_CONTACT + 16 * (t - C_CONTACT)) = 0u;  // a fresh world has no contacts
        else if (t < C_CONTACT + kSlots + 8) {
            const int k = t - C_CONTACT - kSlots;
            const float o = shadow_obs[(uint64_t)k * d.n_pad + i];
            d.obs[(uint64_t)k * d.n_pad + i] = o;
            if (io.obs_out) io.obs_out[(uint64_t)k * d.n + i] = o;
        }
    }
}

#endif   // LL_DIAG

// The END of a step in the fused order, one launch instead of counter memsets + classify (next step) + select + copy:
//  (1) staged resets: every finished env (L_RESET) whose prepared next episode fits gets it copied in (and goes onto L_PREP: its
//      NEXT reset is prepared afterwards), the others go onto L_RESET_DIRECT for the reset kernel behind this one;
//  (2) the contact list and class byte of the NEXT step, from the flag words the step kernels have just stored.  A finished
//      env (class byte 2, written by the kernel that finished it) is a fresh scene by the end of this call — copied in above by
//      another block, or reset by the launch behind — and is classed as free flight without looking at its flag word;
//  (3) the last block to finish publishes the new list's length and clears the counters this step has consumed.
__global__ void __launch_bounds__(1024)
ll_epilogue_kernel(LLDev d, const uint32_t* __restrict__ shadow, const float* __restrict__ shadow_obs, LLIo io, int staged, int direct_launch_follows) {
    __shared__ uint32_t s_cnt[17];
    __shared__ uint32_t s_ent[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    if (staged) {
        const uint32_t* list = d.work_list + (uint64_t)L_RESET * d.n_pad;
        const uint32_t total = d.work_count[L_RESET];
        const uint32_t per = (total + gridDim.x - 1) / gridDim.x;          // a contiguous share of the list per block
        const uint32_t lo = blockIdx.x * per, hi = lo + per < total ? lo + per : total;
        for (uint32_t base = lo; base < hi; base += blockDim.x) {          // block-uniform
            const uint32_t idx = base + tid;
            const bool valid = idx < hi;
            uint32_t i = 0u;
            bool fits = false;
            if (valid) {
                i = list[idx];
                fits = ST_AT(shadow, C_EPISODE) == ST(C_EPISODE) + 1u;
            }
            ll_push_block(d, L_PREP, fits, i, s_cnt);
            ll_push_block(d, L_RESET_DIRECT, valid && !fits, i, s_cnt);
            // the host only launches the direct resets while something it knows of may have invalidated prepared states (see
            // LunarLanderEnv::direct_possible); a misfit it did not expect would leave the env unreset: reported, never silent
            if (valid && !fits && !direct_launch_follows) atomicOr(d.err, DEV_ERR_INTERNAL);
            s_ent[tid] = fits ? i : 0xffffffffu;
            __syncthreads();
            const uint32_t n_here = hi - base < blockDim.x ? hi - base : blockDim.x;
            for (uint32_t k = wave; k < n_here; k += nw) {                 // one wave per env: its columns lie n_pad words apart
                const uint64_t i = s_ent[k];
                if (i == 0xffffffffu) continue;
                for (int t = lane; t < C_CONTACT + kSlots + 8; t += 64) {
                    if (t < C_CONTACT) ST(t) = ST_AT(shadow, t);
                    else if (t < C_CONTACT + kSlots) ST(C_CONTACT + 16 * (t - C_CONTACT)) = 0u;  // a fresh world has no contacts
                    else {
                        const int k8 = t - C_CONTACT - kSlots;
                        const float o = shadow_obs[(uint64_t)k8 * d.n_pad + i];
                        d.obs[(uint64_t)k8 * d.n_pad + i] = o;
                        if (io.obs_out) io.obs_out[(uint64_t)k8 * d.n + i] = o;
                    }
                }
            }
            __syncthreads();
        }
    }
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < d.n; base += stride) {  // block-uniform trip count
        const uint64_t i = base + tid;
        bool to_general = false, touching = false;
        if (i < d.n) {
            if (d.env_class[i] != 2) {
                const uint32_t flags = ST(C_FLAGS);
                to_general = (flags & F_HAS_WORLD) && !ll_free_eligible(flags);
                touching = d.bucket && (flags & F_TOUCHING);
            }
            d.env_class[i] = to_general ? 1 : 0;
        }
        ll_push_block(d, L_GENERAL, to_general && !touching, (uint32_t)i, s_cnt, C_NEXT);
        if (d.bucket) ll_push_block(d, L_GENERAL, to_general && touching, (uint32_t)i, s_cnt, C_NEXT_T, true);   // (block-uniform)
    }
    // (No device-scope fence: the list ENTRIES are only read by later kernels, and the counts travel by atomics that this thread has
    // already waited for.  A __threadfence() here costs 90 us per launch — every wave writes back its XCD's L2.)
    __syncthreads();
    if (tid == 0) {
        const uint32_t ticket = atomicAdd(d.work_count + C_TICKET, 1u);
        if (ticket == gridDim.x - 1) {   // every other block has added its count (its thread 0 waited for that atomic before taking a ticket)
            d.work_count[L_GENERAL] = atomicExch(d.work_count + C_NEXT, 0u);
            d.work_count[L_GENERAL_T] = atomicExch(d.work_count + C_NEXT_T, 0u); d.work_count[L_LATE] = 0u; d.work_count[L_RESET] = 0u;
            if (!staged) { d.work_count[L_RESET_SLOW] = 0u; d.work_count[L_RESET_DIRECT] = 0u; }
            d.work_count[C_TICKET] = 0u;
        }
    }
}

// masked reset: scan the masks (1-2 B/env) and compact the masked envs into the reset list
__global__ void __launch_bounds__(256)
ll_mask_scan_kernel(LLDev d, const uint8_t* __restrict__ m0, const uint8_t* __restrict__ m1) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < d.n; base += stride) {
        const uint64_t i = base + threadIdx.x;
        bool m = false;
        if (i < d.n) m = (m0 && m0[i]) || (m1 && m1[i]);
        ll_push(d, L_RESET, m, (uint32_t)i);
    }
}

// mgym_get_info: spins for `ticks` of the 100 MHz wall clock and records when it ran.  One launch per stream of a handle, issued
// back to back: launches that sit on different hardware queues overlap in time, launches that share a queue run one after the other.
__global__ void ll_queue_probe_kernel(unsigned long long* __restrict__ out, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long t1 = t0;
    while (t1 - t0 < ticks) t1 = wall_clock64();
    out[0] = t0; out[1] = t1;
}

// state blob (27 columns, identical to the CPU oracle's): raw {x,y,angle,vx,vy,w} of lander, leg0, leg1;
// leg0_contact, leg1_contact, game_over, lander awake, prev_shaping (NaN = None), wind_idx, torque_idx,
// step counter, episode counter.
__global__ void ll_export_kernel(LLDev d, uint32_t* __restrict__ blob) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    const uint32_t flags = ST(C_FLAGS);
    for (int b = 0; b < 3; ++b) {
        const int c = C_BODY + 9 * b;
        blob[(uint64_t)(6 * b + 0) * d.n + i] = ST(c + 0); blob[(uint64_t)(6 * b + 1) * d.n + i] = ST(c + 1);
        blob[(uint64_t)(6 * b + 2) * d.n + i] = ST(c + 4);
        blob[(uint64_t)(6 * b + 3) * d.n + i] = ST(c + 5); blob[(uint64_t)(6 * b + 4) * d.n + i] = ST(c + 6);
        blob[(uint64_t)(6 * b + 5) * d.n + i] = ST(c + 7);
    }
    blob[18 * d.n + i] = as_u32((flags & F_LEG0) ? 1.0f : 0.0f);
    blob[19 * d.n + i] = as_u32((flags & F_LEG1) ? 1.0f : 0.0f);
    blob[20 * d.n + i] = as_u32((flags & F_GAME_OVER) ? 1.0f : 0.0f);
    blob[21 * d.n + i] = as_u32((flags & 1u) ? 1.0f : 0.0f);
    blob[22 * d.n + i] = (flags & F_PREV_SOME) ? ST(C_PREV) : 0x7fc00000u;
    blob[23 * d.n + i] = ST(C_WIND); blob[24 * d.n + i] = ST(C_TORQUE);
    blob[25 * d.n + i] = ST(C_STEP); blob[26 * d.n + i] = ST(C_EPISODE);
}

// Testable::set_state (lunar_lander.rs:1444-1554) from blob words 0..19; counters from 25, 26
__global__ void __launch_bounds__(kLLBlock) ll_import_kernel(LLDev d, const uint32_t* __restrict__ blob) {
    __shared__ PolyTab tab;
    LL_HOT_DECL(kLLBlock);
    stage_tab(tab, LLK(d));
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    World w; EnvRegs e;
    w.t = nullptr;
    ll_load(d, i, w, e, hot);
    if (e.has_world) {
        float raw[18];
        for (int q = 0; q < 18; ++q) raw[q] = as_f32(blob[(uint64_t)q * d.n + i]);
        ll_apply_set_state(w, tab, raw, as_f32(blob[18 * d.n + i]) > 0.5f, as_f32(blob[19 * d.n + i]) > 0.5f);
    }
    e.step = blob[25 * d.n + i];
    e.episode = blob[26 * d.n + i];
    ll_store(d, i, w, e);
}

// tuning values of the persistent rollout launch: fixed in the product, read from the environment in -DLL_DIAG builds (tools/ll_roll_sweep.sh)
#ifdef LL_DIAG
#define LL_TUNE(name, dflt) (getenv(name) ? (uint32_t)atoi(getenv(name)) : (dflt))
#else
#define LL_TUNE(name, dflt) (dflt)
#endif
struct LunarLanderEnv final : Env {
    void* base = nullptr;
    void* obs_base = nullptr;
    void* work_base = nullptr;
    void* kdev = nullptr;
    void* vc_far_base = nullptr;
    hipStream_t aux = nullptr;          // helper stream of the overlapped launch order
    hipStream_t roll_aux = nullptr;     // the rollout's free-flight helper waves: a stream of their own at the default priority (created with the helpers)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void* shadow_base = nullptr;        // staged resets: shadow state columns [C_COUNT][n_pad] and observation [8][n_pad]
    void* shadow_obs = nullptr;
    hipStream_t aux2 = nullptr;         // ... prepared on this stream (its own: behind the free-flight kernel on `aux` it costs 0.1 ms per step)
    hipEvent_t ev_prep = nullptr, ev_prepared = nullptr, ev_free_done = nullptr;
    // Staged resets: every env has a fitting prepared episode (shadow episode == live + 1) from the moment a staged reset of ALL envs
    // has prepared them, and keeps one (each copy-in queues the next preparation; a fresh scene spawns far above the highest terrain, so
    // preparing it on the free-flight path cannot fail) UNLESS something else moves an episode counter or resets without preparing:
    // mgym_set_state, a dispersion override, a masked or unstaged reset, steps captured into a graph (they reset directly).  Only then
    // can a finished env need the direct path, and only then does step() launch its two (normally empty) kernels behind the epilogue.
    bool direct_possible = true;
    bool prep_due = false;              // a preparation has been marked (mark_prepare) and not launched yet
    bool ev_prep_valid = false;
    int staged_in_graph = getenv("MGYM_LL_STAGED_IN_GRAPH") ? atoi(getenv("MGYM_LL_STAGED_IN_GRAPH")) : 1;  // fused order: captured steps keep the staged resets (0: they reset directly)
    bool prep_pending = false;          // work was put on aux2 that `stream` has not waited for yet
    int staged = getenv("MGYM_LL_STAGED_RESET") ? atoi(getenv("MGYM_LL_STAGED_RESET")) : 1;  // 1 (default): auto-resets are prepared ahead (see ll_apply_select_kernel)
    uint8_t* env_class = nullptr;       // [n] class of each env for this step (ll_classify_kernel): 0 free flight, 1 contact path
#ifdef LL_DIAG
    int overlap = getenv("MGYM_LL_OVERLAP") ? atoi(getenv("MGYM_LL_OVERLAP")) : -1;  // 1 (default): contact kernel beside the free-flight kernel (see step()); 0: one after the other
#else
    int overlap = 1;
#endif
    LLDev dev{};
    bool general_only = getenv("MGYM_LL_GENERAL_ONLY") != nullptr;
    int gen_block = getenv("MGYM_LL_GENERAL_BLOCK") ? atoi(getenv("MGYM_LL_GENERAL_BLOCK")) : 0;   // lanes per block of the contact kernel; 0: by population (init)
    // the contact list by kind (multi-stream order): -1 by population (init) | 0 in index order | 1 the envs with a touching contact at the back of
    // the list and taken FIRST, in blocks of their own (the long blocks start first, the short ones fill the end) | 2 (-DLL_DIAG) dealt out evenly.
    // ms per step, 0 | 1, same box (profiles/r04_lunarlander/contact_list_by_kind.txt): 524 288 envs 1.37 | 1.57 (the step is still its slowest block, and a
    // block of 64 touching envs is slower than one with a few of them), 655 360: 1.58 | 1.60, 786 432: 2.11 | 1.70, 1 Mi: 2.45 | 2.20, 2 Mi: 5.44 | 4.67
    // (bound by SIMD occupancy: blocks of one kind keep more of their lanes busy).  The two ends as launches of their own — the touching end in 32-lane
    // blocks with their helper lanes, the rest in 64-lane blocks — were slower at every size (1 Mi: 2.97): built, measured, not kept.
    int bucket = getenv("MGYM_LL_BUCKET") ? atoi(getenv("MGYM_LL_BUCKET")) : -1;
#ifdef LL_DIAG
    int toi_block = getenv("MGYM_LL_TOI_BLOCK") ? atoi(getenv("MGYM_LL_TOI_BLOCK")) : 32;          // lanes per block of the time-of-impact round kernels
#endif
    // Launch structure of the contact path.  Measured on MI355X (profiles/r02_lunarlander/tune_launch_structure.txt), 262 144
    // envs, ms per step: whole world.step in the contact kernel, envs in index order 1.96 (default) | the same with
    // touching / non-touching envs bucketed into separate waves 2.12 | time-of-impact sub-steps in 1 / 4 follow-up
    // launches over compacted lists 2.33 / 3.02 | 16- or 8-lane blocks 2.6-3.7.  The kernel's duration is its SLOWEST
    // wave's dependent chain (island solve + up to 5 sub-steps of ~0.16 ms each), not a throughput limit: concentrating
    // heavy lanes (bucketing) lengthens that wave, and follow-up launches serialise the same chain behind launch
    // boundaries and a state round trip.  The alternatives stay selectable for profiling:
#ifdef LL_DIAG
    int toi_first = getenv("MGYM_LL_TOI_FIRST") ? atoi(getenv("MGYM_LL_TOI_FIRST")) : 0;  // with MGYM_LL_TOI_ROUNDS: sub-steps taken inside the contact kernel first
    int toi_rounds = getenv("MGYM_LL_TOI_ROUNDS") ? atoi(getenv("MGYM_LL_TOI_ROUNDS")) : 0;  // 0: none; 1 .. kToiRounds launches of ll_toi_kernel
    int fused_tail = getenv("MGYM_LL_FUSED_TAIL") ? atoi(getenv("MGYM_LL_FUSED_TAIL")) : 1;  // 1 (default): the fused order of step() (ll_epilogue_kernel); needs the overlapped order and none of the profiling knobs
#else
    static constexpr int toi_rounds = 0;
    int fused_tail = 1;   // (0 only with MGYM_LL_GENERAL_ONLY, the debugging aid that sends every env through ll_general_kernel)
#endif
    int single_launch = getenv("MGYM_LL_SINGLE_LAUNCH") ? atoi(getenv("MGYM_LL_SINGLE_LAUNCH")) : 1;  // 1 (default): contact path, free-flight path and reset preparation in ONE launch (ll_step_kernel); needs the fused order and 32-lane contact blocks
    int contact_blocks = getenv("MGYM_LL_CONTACT_BLOCKS") ? atoi(getenv("MGYM_LL_CONTACT_BLOCKS")) : -1;  // single-launch step: blocks the contact list is dealt out over (default 900: as many as run at once beside the free-flight role; 0: always 32 lanes per block)
    int resume = getenv("MGYM_LL_RESUME") ? atoi(getenv("MGYM_LL_RESUME")) : 1;  // 1 (default): envs the free-flight kernel stops at a new contact are resumed after their island solve (0: redone from the old state)
#ifdef LL_DIAG
    int free_occ = getenv("MGYM_LL_FREE_OCC") ? atoi(getenv("MGYM_LL_FREE_OCC")) : 2;  // waves/SIMD the free kernel is compiled for
#endif
    int vc_near_limit = kVcNearLds;     // velocity constraints per lane the contact kernel keeps in LDS (init(); MGYM_LL_VC_NEAR lowers it: test knob)
    // mgym_rollout as ONE persistent launch (ll_roll.h); MGYM_LL_ROLLOUT=0: K x step(), 1: always, unset: below 491 520 envs — above, K steps of the
    // multi-stream order are as fast or faster (524 288 envs, K = 64: 1.34 ms per step against 1.30; 1 Mi: 2.63 against 2.15: both bound by SIMD
    // occupancy there, and the step's free-flight kernel runs at two waves per SIMD; profiles/r04_lunarlander/rollout_by_population.txt)
    int roll_enabled = getenv("MGYM_LL_ROLLOUT") ? atoi(getenv("MGYM_LL_ROLLOUT")) : -1;
    // free-flight helper waves beside the main launch: 0 never (the default), 1 always, -1 by population (from 163 840 envs).  OFF by default: with them a launch hung —
    // waves of both kernels standing still — once in ~75 stress runs (tools/ll_roll_stress.sh: 4 hangs in ~350 runs, on either helper stream), never in 100 runs without them
    // or in the day's hundreds of runs before they existed; the cause was not found in the time left of round 4.  They are worth 4-6 %.
    int roll_helper = getenv("MGYM_LL_ROLL_HELPER") ? atoi(getenv("MGYM_LL_ROLL_HELPER")) : 0;
    unsigned roll_helper_grid = 0;
    int roll_min_k = getenv("MGYM_LL_ROLLOUT_MIN_K") ? atoi(getenv("MGYM_LL_ROLLOUT_MIN_K")) : -1;   // (-1: 8 from 163 840 envs, 12 below: init)   // shorter rollouts: K x step() (the launch's last environments take their last steps alone: ~2-3 ms, measured)
    void* roll_ring = nullptr;
    void* roll_ctl = nullptr;
    void* roll_vc_far = nullptr;
    RollQ rq{};
    unsigned roll_grid = 0;
    void* roll_trace = nullptr;

    ~LunarLanderEnv() override {
        if (base) (void)hipFree(base);
        if (obs_base) (void)hipFree(obs_base);
        if (work_base) (void)hipFree(work_base);
        if (kdev) (void)hipFree(kdev);
        if (vc_far_base) (void)hipFree(vc_far_base);
        if (roll_ring) (void)hipFree(roll_ring);
        if (roll_ctl) (void)hipFree(roll_ctl);
        if (roll_vc_far) (void)hipFree(roll_vc_far);
        if (roll_trace) (void)hipFree(roll_trace);
        if (aux) (void)hipStreamDestroy(aux);
        if (env_class) (void)hipFree(env_class);
        if (aux2) { (void)hipStreamSynchronize(aux2); (void)hipStreamDestroy(aux2); }
        if (roll_aux) { (void)hipStreamSynchronize(roll_aux); (void)hipStreamDestroy(roll_aux); }
        if (shadow_base) (void)hipFree(shadow_base);
        if (shadow_obs) (void)hipFree(shadow_obs);
        if (ev_prep) (void)hipEventDestroy(ev_prep);
        if (ev_prepared) (void)hipEventDestroy(ev_prepared);
        if (ev_free_done) (void)hipEventDestroy(ev_free_done);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }

    int init() override {
        // Contact-kernel block size and launch order (profiles/r02_lunarlander/population_block_overlap_matrix.txt, ms per step,
        // 32-lane / 64-lane blocks, sequential -> overlapped order):
        //     131 072 envs  1.47 -> 1.26 / 1.62 -> 1.39        524 288 envs  2.60 -> 2.09 / 2.35 -> 1.73
        //     262 144 envs  1.64 -> 1.30 / 1.89 -> 1.55      2 097 152 envs  7.59 -> 7.74 / 6.82 -> 6.16
        // 32-lane blocks (which also afford the World record in LDS) give the shortest waves: best while the whole
        // worklist is co-resident.  Beyond that the kernel is bound by wave slots — one 512-register wave per SIMD, four
        // blocks per CU by LDS — and 64-lane blocks (World record in scratch, working storage in LDS) carry twice
        // the environments per slot.  The overlapped order (contact kernel beside the free-flight kernel, see step()) pays
        // at every size once the blocks are chosen this way.
        // (round 3, single-launch step with 32-lane blocks against the multi-stream order with 64-lane blocks: 393 216 envs 1.54 / 1.57 ms,
        // 524 288: 2.04 / 1.68, 1 Mi: 3.81 / 2.86, 2 Mi: 7.38 / 6.11 — profiles/r03_lunarlander/population_block_matrix.txt)
        // (round 4, after the 64-lane blocks began to deal their time-of-impact evaluations out over the wave: 360 448 envs 1.26 / 1.29, 393 216: 1.39 / 1.30,
        // 425 984: 1.46 / 1.29 — profiles/r04_lunarlander/block_switch_by_population.txt)
        if (gen_block == 0) gen_block = n >= 376832 ? 64 : 32;
#ifndef LL_DIAG
        if (gen_block != 32 && gen_block != 64) { set_last_error("MGYM_LL_GENERAL_BLOCK: 32 or 64 (other contact-block sizes exist in -DLL_DIAG builds only)"); return MGYM_ERR_BAD_CONFIG; }
#endif
        if (overlap < 0) overlap = 1;
        obs_dim = 8;
        state_cols = 27;
        MGYM_HIP(hipMalloc(&base, (size_t)ll_state_words(n_pad) * sizeof(uint32_t)));
        MGYM_HIP(hipMemsetAsync(base, 0, (size_t)ll_state_words(n_pad) * sizeof(uint32_t), stream));  // has_world = false
        MGYM_HIP(hipMalloc(&obs_base, (size_t)8 * n_pad * sizeof(float)));
        MGYM_HIP(hipMemsetAsync(obs_base, 0, (size_t)8 * n_pad * sizeof(float), stream));
        dev.st = static_cast<uint32_t*>(base);
        dev.obs = static_cast<float*>(obs_base);
        MGYM_HIP(hipMalloc(&work_base, (size_t)(L_LISTS * n_pad + 64) * sizeof(uint32_t)));
        MGYM_HIP(hipMemsetAsync(work_base, 0, (size_t)(L_LISTS * n_pad + 64) * sizeof(uint32_t), stream));
        dev.work_count = static_cast<uint32_t*>(work_base);
        dev.work_list = static_cast<uint32_t*>(work_base) + 64;
        dev.disp = nullptr;
        dev.n = n; dev.n_pad = n_pad; dev.seed = cfg.seed; dev.env_id_base = cfg.env_id_base; dev.err = d_err; dev.done_count = d_done;
        dev.auto_reset = (cfg.flags & MGYM_FLAG_AUTO_RESET) ? 1 : 0;
        dev.resume = resume && !general_only;
        // contact blocks that run at once beside the free-flight role: 900 of the 1 024 one-wave-per-SIMD slots (measured 750 .. 1 000 at
        // 65 536 .. 327 680 envs: flat within 1-2 % from 850 up, profiles/r03_lunarlander/block_lanes_by_population.txt)
        if (contact_blocks < 0) contact_blocks = 900;
        dev.contact_blocks = contact_blocks;
        fused_tail = fused_tail && overlap && !general_only && toi_rounds == 0;
        dev.fused_tail = fused_tail;
        single_launch = single_launch && fused_tail && gen_block == 32 && resume;
        if (bucket < 0) bucket = (!single_launch && gen_block == 64 && n >= 688128) ? 1 : 0;
        if (single_launch) bucket = 0;   // (its contact role deals ONE list out over its blocks)
#ifndef LL_DIAG
        if (bucket > 1) bucket = 1;
#endif
        dev.bucket = bucket;
        ll_make_const(dev.k, cfg.gravity, cfg.enable_wind, cfg.wind_power, cfg.turbulence_power);
        MGYM_HIP(hipMalloc(&kdev, sizeof(LLConst)));
        MGYM_HIP(hipMemcpyAsync(kdev, &dev.k, sizeof(LLConst), hipMemcpyHostToDevice, stream));
        dev.kd = static_cast<const LLConst*>(kdev);
        // far velocity constraints of the contact kernel: one column per lane of its (bounded) grid, [constraint][lane].  The main
        // launch and the late launch of the overlapped order run concurrently, so each gets a slice of its own, sized from the
        // grids step() really launches (MGYM_LL_VC_NEAR=0, a test knob, sends every constraint there: kSolverCap per lane).
        vc_near_limit = getenv("MGYM_LL_VC_NEAR") ? atoi(getenv("MGYM_LL_VC_NEAR")) : kVcNearLds;
        if (vc_near_limit < 0) vc_near_limit = 0;
        if (vc_near_limit > kVcNearLds) vc_near_limit = kVcNearLds;
        {
            const size_t per_lane = (size_t)(kSolverCap - vc_near_limit);
            // (single-launch step: every wave of the free-flight role may take envs through the contact path itself: a slice per such block)
            const size_t lanes_main = (size_t)main_contact_grid() * gen_block, lanes_late = (size_t)(single_launch ? free2_grid() : late_contact_grid()) * 32;
            MGYM_HIP(hipMalloc(&vc_far_base, per_lane * (lanes_main + lanes_late) * sizeof(VConstraint)));
            dev.vc_far = static_cast<VConstraint*>(vc_far_base);
            dev.vc_far_late = dev.vc_far + per_lane * lanes_main;
            dev.vc_near = vc_near_limit;
        }
        {   // the helper stream carries the work that is NOT on the critical path: lowest priority (1 Mi envs: 3.65 -> 3.52 ms per step)
            int lo = 0, hi = 0;
            MGYM_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            MGYM_HIP(hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, getenv("MGYM_LL_AUX_PRIO") ? atoi(getenv("MGYM_LL_AUX_PRIO")) : lo));
        }
        MGYM_HIP(hipMalloc((void**)&env_class, n_pad));
        MGYM_HIP(hipMemsetAsync(env_class, 0, n_pad, stream));
        dev.env_class = env_class;
        MGYM_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        MGYM_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        dev.split = 0;
        dev.prep = 0; dev.episode_src = nullptr;
        // (pays while the step is bound by the contact kernel's chain: 131 072 envs 1.43 -> 1.36 ms per step, 262 144 1.50 -> 1.46,
        // 393 216 1.67 -> 1.60; a population that fills every wave slot gains nothing from moving the reset work and loses to the
        // extra launches: 524 288 envs 1.86 -> 1.83, 1 Mi envs 3.53 -> 3.84 — with one column per word.  With records the copy of a prepared
        // state moves nine whole lines and staging pays at every size: 524 288 envs 1.55 -> 1.46, 786 432 2.27 -> 2.18, 1 Mi 2.57 -> 2.52,
        // 2 Mi 5.67 -> 5.56 — profiles/r03_lunarlander/population_block_matrix.txt)
        staged = staged >= 1 && (cfg.flags & MGYM_FLAG_AUTO_RESET) && !general_only;
        if (staged) {
            MGYM_HIP(hipMalloc(&shadow_base, (size_t)ll_state_words(n_pad) * sizeof(uint32_t)));
            MGYM_HIP(hipMemsetAsync(shadow_base, 0, (size_t)ll_state_words(n_pad) * sizeof(uint32_t), stream));
            MGYM_HIP(hipMalloc(&shadow_obs, (size_t)8 * n_pad * sizeof(float)));
            MGYM_HIP(hipMemsetAsync(shadow_obs, 0, (size_t)8 * n_pad * sizeof(float), stream));
            int lo = 0, hi = 0;
            MGYM_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            MGYM_HIP(hipStreamCreateWithPriority(&aux2, hipStreamNonBlocking, lo));
            MGYM_HIP(hipEventCreateWithFlags(&ev_prep, hipEventDisableTiming));
            MGYM_HIP(hipEventCreateWithFlags(&ev_prepared, hipEventDisableTiming));
            MGYM_HIP(hipEventCreateWithFlags(&ev_free_done, hipEventDisableTiming));
        }
        // mgym_rollout's persistent launch: rings of >= n slots per queue, one wave per SIMD as many as are resident at once, a slice of
        // far-constraint workspace per wave.  (Records are addressed by 32-bit byte offsets there: n_pad x 1152 B < 4 GiB.)
        if (roll_enabled < 0) roll_enabled = n < 491520 ? 1 : 0;
        roll_enabled = roll_enabled && !general_only && n > 0 && n <= kRollEnvMask && (uint64_t)n_pad * kRec * 4ull < 0xffffffffull;
        // 8-step rollouts of small populations are no faster than 8 steps (131 072 envs: 0.88 ms per step-equivalent against 0.85, 65 536: 0.80 / 0.80; K = 16: 0.70 / 0.85
        // and 0.62 / 0.79): the persistent launch from 12 steps there, from 8 at 163 840 envs and more
        if (roll_min_k < 0) roll_min_k = n >= 163840 ? 8 : 12;
        if (roll_enabled) {
            uint64_t cap = 64;
            while (cap < n) cap <<= 1;
            rq.mask = (uint32_t)(cap - 1);
            MGYM_HIP(hipMalloc(&roll_ring, (size_t)cap * RQ_COUNT * sizeof(unsigned long long)));
            MGYM_HIP(hipMalloc(&roll_ctl, (size_t)RC_WORDS * 32 * sizeof(uint32_t) + RS_COUNT * sizeof(unsigned long long)));
            rq.ring = static_cast<unsigned long long*>(roll_ring);
            rq.ctl = static_cast<uint32_t*>(roll_ctl);
            rq.stat = reinterpret_cast<unsigned long long*>(rq.ctl + RC_WORDS * 32);
            rq.contact_min = LL_TUNE("MGYM_LL_ROLL_CONTACT_MIN", 32u);
            rq.heavy_min = LL_TUNE("MGYM_LL_ROLL_HEAVY_MIN", 16u);
            rq.heavy_max = LL_TUNE("MGYM_LL_ROLL_HEAVY_MAX", 32u);
            if (rq.heavy_max > 32u) rq.heavy_max = 32u;
            if (rq.heavy_max < 1u) rq.heavy_max = 1u;
            rq.tail_live = LL_TUNE("MGYM_LL_ROLL_TAIL_LIVE", 32768u);
            rq.tail_lanes = LL_TUNE("MGYM_LL_ROLL_TAIL_LANES", 4u);
            if (rq.tail_lanes > 32u) rq.tail_lanes = 32u;
            if (rq.tail_lanes < 1u) rq.tail_lanes = 1u;
            rq.toi_split = LL_TUNE("MGYM_LL_ROLL_TOI_SPLIT", 7u);
            rq.toi_min = LL_TUNE("MGYM_LL_ROLL_TOI_MIN", 24u);
            rq.keep_min = LL_TUNE("MGYM_LL_ROLL_KEEP_MIN", 24u);
            // (with the helper waves on — the main waves live on the contact path — light batches that keep their residents too, and a rotation after 3 steps: 262 144 envs
            // K = 64 0.68 -> 0.62 ms per step, K = 16 0.77 -> 0.76, K = 8 unchanged; profiles/r04_lunarlander/rollout_tuning_with_helper_waves.txt)
            rq.keep = LL_TUNE("MGYM_LL_ROLL_KEEP", 3u);
            rq.fair = LL_TUNE("MGYM_LL_ROLL_FAIR", 0u);
            rq.heavy_narrow = LL_TUNE("MGYM_LL_ROLL_HEAVY_NARROW", 12u);
            if (rq.heavy_narrow > rq.heavy_max) rq.heavy_narrow = rq.heavy_max;
            if (rq.heavy_narrow < 1u) rq.heavy_narrow = 1u;
            rq.reset_min = LL_TUNE("MGYM_LL_ROLL_RESET_MIN", 16u);
            rq.free_min = LL_TUNE("MGYM_LL_ROLL_FREE_MIN", 32u);
            rq.residency = LL_TUNE("MGYM_LL_ROLL_RESIDENCY", 3u);
            rq.refill_min = LL_TUNE("MGYM_LL_ROLL_REFILL_MIN", 8u);
            rq.debug = LL_TUNE("MGYM_LL_ROLL_DEBUG", 0u);
            ll_rollout_ring_init(stream, rq);
            int per_cu = 0, dev_id = 0;
            hipDeviceProp_t prop;
            MGYM_HIP(hipGetDevice(&dev_id));
            MGYM_HIP(hipGetDeviceProperties(&prop, dev_id));
            if (ll_rollout_blocks_per_cu(&per_cu) != MGYM_OK) return MGYM_ERR_HIP;
            if (per_cu < 1) per_cu = 1;
            uint64_t g = (uint64_t)per_cu * (uint64_t)prop.multiProcessorCount;
            // Free-flight helper waves (ll_rollout_free_kernel) beside the main launch: the main waves fill a SIMD each (256 + 256 registers), so they
            // get three SIMDs of every CU and the helpers — two waves per SIMD — the fourth.  262 144 envs, ms per step-equivalent without | with
            // (profiles/r04_lunarlander/rollout_helper_waves.txt): K = 8 0.91 | 0.87, K = 16 0.81 | 0.77, K = 64 0.71 | 0.68; the helpers then take 86 % of the
            // free-flight steps at 46 us of SIMD time per 64 env-steps against 79 in a main wave, and the 768 main waves are busy with the contact
            // path throughout (35 of 45 ms).  Two SIMDs per CU for the helpers: 1.05 (the contact path is the larger share); four main waves per CU:
            // no helper ever becomes resident and the launch runs as before (0.71) — the helpers are an accelerator, never a dependency.
            // Small populations end a launch with a longer tail than busy phase, and there a fourth main wave per CU is worth more than the helpers:
            // 65 536 envs K = 16 0.71 | 0.81, 98 304: 0.66 | 0.70, 131 072: 0.72 | 0.64 (K = 8: 0.78 | 0.82), 196 608: 0.77 | 0.73 (K = 8: 0.90 | 0.85) — on from 163 840 envs.
            if (roll_helper < 0) roll_helper = n >= 163840 ? 1 : 0;
            roll_helper_grid = 0;
            if (roll_helper && per_cu >= 4) {
                const unsigned main_per_cu = LL_TUNE("MGYM_LL_ROLL_MAIN_PER_CU", 3u), help_per_cu = LL_TUNE("MGYM_LL_ROLL_HELPER_PER_CU", 2u);
                if ((int)main_per_cu < per_cu && main_per_cu >= 1) g = (uint64_t)main_per_cu * (uint64_t)prop.multiProcessorCount;
                roll_helper_grid = help_per_cu * (unsigned)prop.multiProcessorCount;
                if ((uint64_t)roll_helper_grid * 64 > n) roll_helper_grid = (unsigned)(n / 64);
                // (MGYM_LL_ROLL_HELPER_LOWPRIO=1: on the step's helper stream, which has the lowest priority)
                if (roll_helper_grid && !(getenv("MGYM_LL_ROLL_HELPER_LOWPRIO") && atoi(getenv("MGYM_LL_ROLL_HELPER_LOWPRIO")))) MGYM_HIP(hipStreamCreateWithFlags(&roll_aux, hipStreamNonBlocking));
            }
            rq.helper_min = LL_TUNE("MGYM_LL_ROLL_HELPER_MIN", 256u);
            if (LL_TUNE("MGYM_LL_ROLL_GRID", 0u)) g = (uint64_t)LL_TUNE("MGYM_LL_ROLL_GRID", 0u);
            const uint64_t need = (n + 31) / 32;   // more waves than 32-lane batches can never be busy
            if (g > need) g = need;
            if (g < 1) g = 1;
            roll_grid = (unsigned)g;
            const size_t per_lane = (size_t)(kSolverCap - vc_near_limit);
            MGYM_HIP(hipMalloc(&roll_vc_far, per_lane * (size_t)roll_grid * 32 * sizeof(VConstraint)));
            if (getenv("MGYM_LL_ROLL_TRACE")) {   // diagnosis: what every wave did when (written to that file after each launch; synchronises)
                MGYM_HIP(hipMalloc(&roll_trace, (size_t)roll_grid * kRollTraceLen * sizeof(unsigned long long)));
                rq.trace = static_cast<unsigned long long*>(roll_trace);
            }
        }
        return MGYM_OK;
    }
    // K steps of every environment in ONE persistent launch (ll_roll.h): identical per-environment results to K calls of step()
    int rollout(const void* actions, int K, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (!roll_enabled || n == 0 || K < roll_min_k) return Env::rollout(actions, K, obs_out, reward, done, trunc);
        int st = join_helpers();
        if (st != MGYM_OK) return st;
        LLDev rd = dev; rd.split = 0;
        rd.vc_far = static_cast<VConstraint*>(roll_vc_far); rd.vc_far_late = rd.vc_far;
        for (int k0 = 0; k0 < K; k0 += (int)kRollMaxK) {
            const int kc = K - k0 < (int)kRollMaxK ? K - k0 : (int)kRollMaxK;
            LLIo io{static_cast<const uint32_t*>(actions) + (size_t)k0 * n, obs_out ? obs_out + (size_t)k0 * 8 * n : nullptr, reward ? reward + (size_t)k0 * n : nullptr,
                    done ? done + (size_t)k0 * n : nullptr, trunc ? trunc + (size_t)k0 * n : nullptr};
            RollQ q = rq; q.K = (uint32_t)kc;
            if (roll_trace) MGYM_HIP(hipMemsetAsync(roll_trace, 0, (size_t)roll_grid * kRollTraceLen * sizeof(unsigned long long), stream));
            ll_rollout_begin(stream, q, (uint32_t)n);
            if (roll_helper_grid) MGYM_HIP(hipEventRecord(ev_fork, stream));
            ll_rollout_launch(stream, roll_grid, rd, io, q);
            if (roll_helper_grid) {   // the helper waves beside it, on the helper stream (fork / join by events: stream-ordered for the caller, capturable)
                hipStream_t hs = roll_aux ? roll_aux : aux;
                MGYM_HIP(hipStreamWaitEvent(hs, ev_fork, 0));
                ll_rollout_helper_launch(hs, roll_helper_grid, rd, io, q);
                MGYM_HIP(hipEventRecord(ev_join, hs));
                MGYM_HIP(hipStreamWaitEvent(stream, ev_join, 0));
            }
        }
        if (roll_trace) {
            std::vector<unsigned long long> h((size_t)roll_grid * kRollTraceLen);
            MGYM_HIP(hipStreamSynchronize(stream));
            MGYM_HIP(hipMemcpy(h.data(), roll_trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE* f = fopen(getenv("MGYM_LL_ROLL_TRACE"), "wb")) { const unsigned hdr[2] = {roll_grid, (unsigned)kRollTraceLen}; fwrite(hdr, sizeof hdr, 1, f); fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
        }
        if (getenv("MGYM_LL_ROLL_STATS")) {   // diagnosis: where the waves of the last launch spent their time (synchronises)
            unsigned long long h[RS_COUNT];
            MGYM_HIP(hipStreamSynchronize(stream));
            MGYM_HIP(hipMemcpy(h, rq.stat, sizeof h, hipMemcpyDeviceToHost));
            const double w = h[RS_N_WAVES] ? (double)h[RS_N_WAVES] : 1.0, us = 0.01;
            auto per = [&](int t, int nn) { return h[nn] ? h[t] * us / h[nn] : 0.0; };
            fprintf(stderr, "ll_rollout K=%d waves=%.0f: per wave us: total %.0f seed %.0f | touching-contact %.0f (%.1f batches of %.1f lanes, %.0f us each) | light-contact %.0f (%.1f batches of %.1f lanes, %.0f us each) | "
                            "sub-steps %.0f (%.1f batches of %.1f lanes, %.0f us each) | reset %.0f (%.1f lanes) | free %.0f (%.1f steps of %.1f lanes, %.1f us each: begin %.1f issue %.1f sweeps %.1f finish %.1f queue %.1f; %.1f refills %.1f switches %.1f rotations) | idle %.0f, %.1f looks without work\n",
                    K, w, h[RS_T_TOTAL] * us / w, h[RS_T_SEED] * us / w,
                    h[RS_T_CONTACT] * us / w, h[RS_N_CONTACT_BATCHES] / w, h[RS_N_CONTACT_BATCHES] ? (double)h[RS_N_CONTACT_LANES] / h[RS_N_CONTACT_BATCHES] : 0.0, per(RS_T_CONTACT, RS_N_CONTACT_BATCHES),
                    h[RS_T_LIGHT] * us / w, h[RS_N_LIGHT_BATCHES] / w, h[RS_N_LIGHT_BATCHES] ? (double)h[RS_N_LIGHT_LANES] / h[RS_N_LIGHT_BATCHES] : 0.0, per(RS_T_LIGHT, RS_N_LIGHT_BATCHES),
                    h[RS_T_TOI] * us / w, h[RS_N_TOI_BATCHES] / w, h[RS_N_TOI_BATCHES] ? (double)h[RS_N_TOI_LANES] / h[RS_N_TOI_BATCHES] : 0.0, per(RS_T_TOI, RS_N_TOI_BATCHES),
                    h[RS_T_RESET] * us / w, h[RS_N_RESET_LANES] / w,
                    h[RS_T_FREE] * us / w, h[RS_N_FREE_STEPS] / w, h[RS_N_FREE_STEPS] ? (double)h[RS_N_FREE_LANE_STEPS] / h[RS_N_FREE_STEPS] : 0.0, per(RS_T_FREE, RS_N_FREE_STEPS),
                    per(RS_T_FREE_BEGIN, RS_N_FREE_STEPS), per(RS_T_FREE_ISSUE, RS_N_FREE_STEPS), per(RS_T_FREE_SWEEPS, RS_N_FREE_STEPS), per(RS_T_FREE_FINISH, RS_N_FREE_STEPS), per(RS_T_FREE_QUEUE, RS_N_FREE_STEPS),
                    h[RS_N_REFILLS] / w, h[RS_N_SWITCHES] / w, h[RS_N_ROTATIONS] / w, h[RS_T_IDLE] * us / w, h[RS_N_MAIN] / w);
            if (roll_helper_grid) fprintf(stderr, "   free-flight helper waves (%u blocks): %.0f wave-steps of %.1f lanes (main waves: %.0f wave-steps)\n", roll_helper_grid, (double)h[RS_N_HELPER_STEPS],
                                          h[RS_N_HELPER_STEPS] ? (double)h[RS_N_HELPER_LANE_STEPS] / h[RS_N_HELPER_STEPS] : 0.0, (double)h[RS_N_FREE_STEPS]);
        }
        if (staged) direct_possible = true;   // episodes that ended inside the launch were reset there: their prepared successors no longer fit
        if (fused_tail) { st = rebuild_list(); if (st != MGYM_OK) return st; }
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }
    bool staging() const { return staged && dev.disp == nullptr; }  // (a dispersion override would be baked into prepared states)
    bool capturing() const {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) return false;
        return cs != hipStreamCaptureStatusNone;
    }
    // Staged resets while the caller's stream is being captured: the fused order takes its preparation branch into the graph (every
    // event it needs is recorded inside the capture: see step()); the unfused order resets directly in captured steps — replayed as a
    // hipGraph its extra branch cost more than it saved (profiles/r02_lunarlander/launch_modes.txt) — and their envs' prepared states
    // are then simply stale (episode check) the next time a step runs eagerly.
    bool staging_now() const {
        if (!staging()) return false;
        return (fused_tail && staged_in_graph) || !capturing();
    }
    // Preparation state.  prep_due: L_PREP may hold envs (put there by an epilogue) that no preparation launch has been issued for;
    // ev_prep_valid: ev_prep is a real record on `stream` behind that epilogue (a record made while capturing is not);
    // prep_pending: a launch is on aux2 that `stream` has not waited for yet (never true between calls).
    // prepared states no longer fit: episode 0 in every shadow record never equals a live counter + 1
    hipError_t clear_shadow_episodes() {
#if LL_RECORD_LAYOUT
        return hipMemset2DAsync(static_cast<uint32_t*>(shadow_base) + C_EPISODE, (size_t)kRec * sizeof(uint32_t), 0, sizeof(uint32_t), n_pad, stream);
#else
        return hipMemsetAsync(static_cast<uint32_t*>(shadow_base) + (size_t)C_EPISODE * n_pad, 0, n_pad * sizeof(uint32_t), stream);
#endif
    }
    // `stream` waits for the preparation work put on aux2 so far: before anything appends to L_PREP or reads the shadow columns
    int join_helpers() override {
        if (prep_due && !capturing()) {   // marked by the last step, no step followed
            int st = launch_prepare(false, true, false);
            if (st != MGYM_OK) return st;
        }
        if (prep_pending) { MGYM_HIP(hipStreamWaitEvent(stream, ev_prepared, 0)); prep_pending = false; }
        return MGYM_OK;
    }
    // prepare the next reset of the envs on L_PREP (or of every env), after what `stream` holds so far.
    // Fused order: the preparation that follows a step is only MARKED (mark_prepare) and launched by the next step() behind its
    // free-flight kernel (after_free) — started at once it competes with the contact kernel for the SIMDs at the very moment that
    // kernel's one-wave-per-SIMD blocks are being placed (contact kernel 1.04 -> 1.15 ms, itself 0.11 -> 0.7 ms).
    int mark_prepare(bool in_capture) {
        if (!in_capture) MGYM_HIP(hipEventRecord(ev_prep, stream));
        ev_prep_valid = !in_capture;
        prep_due = true;
        return MGYM_OK;
    }
    // deferred: the launch belongs to an earlier mark_prepare (no new record of ev_prep, unless that one was made while capturing);
    // in_capture: everything this launch waits for is recorded inside the capture (ev_free_done), and what it prepares is whatever
    // L_PREP holds when the graph runs
    int launch_prepare(bool all, bool deferred = false, bool after_free = false, bool in_capture = false) {
        LLDev sh = dev;
        sh.st = static_cast<uint32_t*>(shadow_base); sh.obs = static_cast<float*>(shadow_obs);
        sh.episode_src = dev.st; sh.prep = 0; sh.split = 0;
        const LLIo none{nullptr, nullptr, nullptr, nullptr, nullptr};
        if (!in_capture) {
            if (!deferred || !ev_prep_valid) MGYM_HIP(hipEventRecord(ev_prep, stream));
            ev_prep_valid = true;
            prep_due = false;
            MGYM_HIP(hipStreamWaitEvent(aux2, ev_prep, 0));
        }
        if (after_free) MGYM_HIP(hipStreamWaitEvent(aux2, ev_free_done, 0));
        if (all) hipLaunchKernelGGL(ll_reset_kernel<64>, grid(), dim3(64), 0, aux2, sh, none, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (int)L_PREP_SLOW);
        else hipLaunchKernelGGL(ll_reset_kernel<32>, dim3(work_grid().x * 2), dim3(32), 0, aux2, sh, none, list_ptr(L_PREP), (const uint32_t*)dev.work_count + L_PREP, (int)L_PREP_SLOW);
        // (an env the fast path declines — it never does for a fresh scene — keeps a shadow episode that does not fit and
        // takes the direct path at its next reset; L_PREP_SLOW is only a sink)
        MGYM_HIP(hipMemsetAsync(dev.work_count + L_PREP, 0, 2 * sizeof(uint32_t), aux2));
        MGYM_HIP(hipEventRecord(ev_prepared, aux2));
        prep_pending = true;
        return MGYM_OK;
    }

    dim3 grid() const {
        uint64_t b = (n + kLLBlock - 1) / kLLBlock;
        if (b > 65536) b = 65536;
        return dim3((unsigned)(b ? b : 1));
    }
    unsigned main_contact_grid() const { return (unsigned)(((uint64_t)work_grid().x * 64 + gen_block - 1) / gen_block); }  // blocks of gen_block lanes
    unsigned late_contact_grid() const { return work_grid().x < 256 ? work_grid().x : 256; }                               // blocks of 32 lanes
    dim3 work_grid() const {  // the worklist length is only known on the device: fixed grid, grid-stride inside
        uint64_t b = (n + kLLBlock - 1) / kLLBlock;
        if (b > 4096) b = 4096;
        return dim3((unsigned)(b ? b : 1));
    }

    const uint32_t* list_ptr(int which) const { return dev.work_list + (size_t)which * n_pad; }
    unsigned free2_grid() const { const uint64_t per = 64 * kFreePerLane; return (unsigned)((n + per - 1) / per ? (n + per - 1) / per : 1); }   // free-flight role of the single-launch kernel: 64 x kFreePerLane envs per wave
    unsigned classify_grid() const { return (unsigned)((n + 1023) / 1024 < 256 ? (n + 1023) / 1024 : 256); }
    // The fused order keeps the contact list and class bytes of the NEXT step valid at all times: ll_epilogue_kernel builds them
    // at the end of every step, and whatever else changes the state (reset, set_state, reset_deterministic) rebuilds them here.
    int rebuild_list() {
        LLDev sd = dev; sd.split = 1;
        MGYM_HIP(hipMemsetAsync(dev.work_count, 0, L_COUNT * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(ll_classify_kernel, dim3(classify_grid()), dim3(1024), 0, stream, sd);
        return MGYM_OK;
    }
    // resets of the envs on the L_RESET list (or of every env): fast path, then the (normally empty) declined list
    void launch_resets(const LLIo& io, bool all, int from_list = L_RESET, bool prep = false) {
        LLDev d = dev;
        d.prep = prep ? 1 : 0;   // the envs reset here are listed so that their next reset gets prepared
        if (all) hipLaunchKernelGGL(ll_reset_kernel<64>, grid(), dim3(64), 0, stream, d, io, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (int)L_RESET_SLOW);
        else hipLaunchKernelGGL(ll_reset_kernel<32>, dim3(work_grid().x * 2), dim3(32), 0, stream, d, io, list_ptr(from_list), (const uint32_t*)dev.work_count + from_list, (int)L_RESET_SLOW);
        // declined resets are rare to non-existent: a small grid (grid-stride inside) keeps the usual empty launch cheap
        hipLaunchKernelGGL(ll_general_kernel<64>, dim3(work_grid().x < 64 ? work_grid().x : 64), dim3(64), 0, stream, d, io, list_ptr(L_RESET_SLOW), (const uint32_t*)dev.work_count + L_RESET_SLOW, 0);
    }
    int reset(const uint8_t* m0, const uint8_t* m1, bool all, float* obs_out) override {
        if (n == 0) return MGYM_OK;
        LLIo io{nullptr, obs_out, nullptr, nullptr, nullptr};
        MGYM_HIP(hipMemsetAsync(dev.work_count, 0, L_COUNT * sizeof(uint32_t), stream));
        if (general_only && all) {  // debugging aid: reset + implicit step on the general path
            hipLaunchKernelGGL(ll_general_kernel<64>, grid(), dim3(kLLBlock), 0, stream, dev, io, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 1);
        } else {
            const bool stage = staging_now() && !capturing();   // (a captured reset computes directly and prepares nothing)
            if (stage) { int st = join_helpers(); if (st != MGYM_OK) return st; }
            if (!all) hipLaunchKernelGGL(ll_mask_scan_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dev, m0, m1);
            launch_resets(io, all, L_RESET, stage);
            if (stage) {   // (waited for at once: between calls nothing is left pending on the helper streams)
                int st = launch_prepare(all);
                if (st == MGYM_OK) st = join_helpers();
                if (st != MGYM_OK) return st;
            }
            if (stage && all) direct_possible = false;   // every env has a fitting prepared successor from here on
            else if (!stage) direct_possible = true;     // (a masked staged reset prepares the envs it resets and says nothing about the others: unchanged)
        }
        if (fused_tail) { int st = rebuild_list(); if (st != MGYM_OK) return st; }
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }
    int reset_deterministic(float* obs_out) override {
        if (n == 0) return MGYM_OK;
        LLIo io{nullptr, obs_out, nullptr, nullptr, nullptr};
        direct_possible = true;
        hipLaunchKernelGGL(ll_general_kernel<64>, grid(), dim3(kLLBlock), 0, stream, dev, io, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 2);
        if (fused_tail) { int st = rebuild_list(); if (st != MGYM_OK) return st; }
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }
    void launch_free(hipStream_t s, const LLDev& d, const LLIo& io) {
#ifdef LL_DIAG
        if (free_occ == 1) { hipLaunchKernelGGL(ll_free_kernel<1>, grid(), dim3(kLLBlock), 0, s, d, io); return; }
        if (free_occ == 3) { hipLaunchKernelGGL(ll_free_kernel<3>, grid(), dim3(kLLBlock), 0, s, d, io); return; }
#endif
        hipLaunchKernelGGL(ll_free_kernel<2>, grid(), dim3(kLLBlock), 0, s, d, io);   // two waves per SIMD (profiles/r02_lunarlander/tune_free_kernel_occupancy_overlapped.txt)
    }
    void launch_contact(hipStream_t s, int block, unsigned gb, const LLDev& d, const LLIo& io, int toi_budget, int which) {
        switch (block) {
#ifdef LL_DIAG
        case 8: hipLaunchKernelGGL(ll_contact_kernel<8>, dim3(gb), dim3(ll_contact_threads(8)), 0, s, d, io, toi_budget, which); break;
        case 16: hipLaunchKernelGGL(ll_contact_kernel<16>, dim3(gb), dim3(ll_contact_threads(16)), 0, s, d, io, toi_budget, which); break;
        case 40: hipLaunchKernelGGL(ll_contact_kernel<40>, dim3(gb), dim3(ll_contact_threads(40)), 0, s, d, io, toi_budget, which); break;
        case 48: hipLaunchKernelGGL(ll_contact_kernel<48>, dim3(gb), dim3(ll_contact_threads(48)), 0, s, d, io, toi_budget, which); break;
#endif
        case 64: hipLaunchKernelGGL(ll_contact_kernel<64>, dim3(gb), dim3(ll_contact_threads(64)), 0, s, d, io, toi_budget, which); break;
        default: hipLaunchKernelGGL(ll_contact_kernel<32>, dim3(gb), dim3(ll_contact_threads(32)), 0, s, d, io, toi_budget, which); break;
        }
    }
    int step(const void* actions, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0) return MGYM_OK;
        LLIo io{static_cast<const uint32_t*>(actions), obs_out, reward, done, trunc};
        if (general_only) {  // debugging aid (MGYM_LL_GENERAL_ONLY=1): every env through the general kernel
            hipLaunchKernelGGL(ll_general_kernel<64>, grid(), dim3(kLLBlock), 0, stream, dev, io, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0);
            MGYM_HIP(hipGetLastError());
            return MGYM_OK;
        }
        const unsigned gb = main_contact_grid();
        if (fused_tail) {
            // Fused order (the default): this step's contact list and class bytes were built by the previous call's last kernel, so
            // the contact kernel starts at once; beside it, on the helper stream, the free-flight kernel and the short contact launch
            // for the envs it hands over; after the join ONE epilogue launch: prepared resets copied in, next step's list, counters.
            LLDev sd = dev; sd.split = 1;
            if (single_launch) {
                // ONE launch for the step (ll_step_kernel) and one for its end (ll_epilogue_kernel), both on the caller's stream: no
                // helper streams, no events — nothing depends on hardware queues or on how a graph executor schedules branches.
                const bool stage1 = dev.auto_reset && staging();
                LLDev sh = sd;
                if (stage1) {
                    sh.st = static_cast<uint32_t*>(shadow_base); sh.obs = static_cast<float*>(shadow_obs);
                    sh.episode_src = dev.st; sh.prep = 0; sh.split = 0;
                }
                const unsigned g_contact = gb, g_free = free2_grid(), g_prep = stage1 ? 128u : 0u;
                hipLaunchKernelGGL(ll_step_kernel<32>, dim3(g_contact + g_free + g_prep), dim3(64), 0, stream, sd, sh, io, g_contact, g_free);
                LLIo rio1{nullptr, obs_out, nullptr, nullptr, nullptr};
                // A step that is being CAPTURED always carries the two (normally empty) direct-reset launches: whether a finished env will
                // find a fitting prepared episode is decided when the graph RUNS — after a later mgym_set_state, say — not now.
                const bool direct_now = direct_possible || capturing();
                if (stage1) {
                    hipLaunchKernelGGL(ll_epilogue_kernel, dim3(classify_grid()), dim3(1024), 0, stream, dev, (const uint32_t*)shadow_base, (const float*)shadow_obs, rio1, 1, direct_now ? 1 : 0);
                    if (direct_now) launch_resets(rio1, false, L_RESET_DIRECT, true);   // (two launches, normally empty)
                } else {
                    if (dev.auto_reset) launch_resets(rio1, false);
                    if (dev.auto_reset && staged) direct_possible = true;   // resets computed here leave no prepared successor behind
                    hipLaunchKernelGGL(ll_epilogue_kernel, dim3(classify_grid()), dim3(1024), 0, stream, dev, (const uint32_t*)nullptr, (const float*)nullptr, rio1, 0, 1);
                }
                MGYM_HIP(hipGetLastError());
                return MGYM_OK;
            }
            const bool stage = dev.auto_reset && staging_now();
            const bool cap = stage && capturing();
            if (stage && !cap && prep_due && !ev_prep_valid) {   // marked by a captured step: the record has to be made now, BEFORE this step's contact kernel
                MGYM_HIP(hipEventRecord(ev_prep, stream));
                ev_prep_valid = true;
            }
            MGYM_HIP(hipEventRecord(ev_fork, stream));
            MGYM_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
            launch_contact(stream, gen_block, gb, sd, io, -1, L_GENERAL);
            launch_free(aux, sd, io);
            if (stage && (cap || prep_due)) {   // the preparation the previous step marked: behind the free-flight kernel, beside the late contact launch
                MGYM_HIP(hipEventRecord(ev_free_done, aux));
                int st = launch_prepare(false, true, true, cap);
                if (st != MGYM_OK) return st;
            }
            launch_contact(aux, 32, late_contact_grid(), sd, io, -1, L_LATE);
            MGYM_HIP(hipEventRecord(ev_join, aux));
            MGYM_HIP(hipStreamWaitEvent(stream, ev_join, 0));
            LLIo rio{nullptr, obs_out, nullptr, nullptr, nullptr};
            if (stage) {
                int st = join_helpers();
                if (st != MGYM_OK) return st;
                const bool direct_now = direct_possible || capturing();   // (see the single-launch order above)
                hipLaunchKernelGGL(ll_epilogue_kernel, dim3(classify_grid()), dim3(1024), 0, stream, dev, (const uint32_t*)shadow_base, (const float*)shadow_obs, rio, 1, direct_now ? 1 : 0);
                if (direct_now) launch_resets(rio, false, L_RESET_DIRECT, true);   // (two launches, normally empty)
                st = mark_prepare(cap);
                if (st != MGYM_OK) return st;
            } else {
                if (dev.auto_reset) launch_resets(rio, false);
                if (dev.auto_reset && staged) direct_possible = true;   // resets computed here leave no prepared successor behind
                hipLaunchKernelGGL(ll_epilogue_kernel, dim3(classify_grid()), dim3(1024), 0, stream, dev, (const uint32_t*)nullptr, (const float*)nullptr, rio, 0, 1);
            }
            MGYM_HIP(hipGetLastError());
            return MGYM_OK;
        }
#ifdef LL_DIAG   // the unfused orders of rounds 1-2 (MGYM_LL_FUSED_TAIL=0, MGYM_LL_OVERLAP=0, MGYM_LL_TOI_ROUNDS, MGYM_LL_BUCKET): profiling builds only
        MGYM_HIP(hipMemsetAsync(dev.work_count, 0, L_COUNT * sizeof(uint32_t), stream));
        const int first_budget = toi_rounds > 0 ? toi_first : -1;  // sub-steps the contact kernel takes itself before handing an env to ll_toi_kernel
        if (overlap) {
            // Overlapped order (see ll_classify_kernel): the contact kernel starts at once on the caller's stream; beside it,
            // on the helper stream, the free-flight kernel over everyone else, followed by a short contact launch for the
            // envs it declined (a contact was created at the end of their step).  Stream-ordered for the caller and
            // capturable (fork / join by events).  262 144 envs: 1.82 -> 1.52 ms per step (table in init()).
            LLDev sd = dev; sd.split = 1;
            hipLaunchKernelGGL(ll_classify_kernel, dim3(classify_grid()), dim3(1024), 0, stream, sd);
            MGYM_HIP(hipEventRecord(ev_fork, stream));
            launch_contact(stream, gen_block, gb, sd, io, first_budget, L_GENERAL);
            MGYM_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
            launch_free(aux, sd, io);
            launch_contact(aux, 32, late_contact_grid(), sd, io, first_budget, L_LATE);
            MGYM_HIP(hipEventRecord(ev_join, aux));
            MGYM_HIP(hipStreamWaitEvent(stream, ev_join, 0));
        } else {
            launch_free(stream, dev, io);
            // contact path on the compacted list, then the time-of-impact rounds over ever shorter lists (fixed grids,
            // grid-stride inside: the list lengths only exist on the device)
            launch_contact(stream, gen_block, gb, dev, io, first_budget, L_GENERAL);
        }
        for (int r = 0; r < toi_rounds; ++r) {
            unsigned g = (work_grid().x * (64 / toi_block)) >> (r + 1);
            if (g < 64) g = 64;
            const int budget = r + 1 < toi_rounds ? 1 : -1;   // the last round runs every remaining sub-step
            switch (toi_block) {
            case 8: hipLaunchKernelGGL(ll_toi_kernel<8>, dim3(g), dim3(8), 0, stream, dev, io, r, budget); break;
            case 16: hipLaunchKernelGGL(ll_toi_kernel<16>, dim3(g), dim3(16), 0, stream, dev, io, r, budget); break;
            case 64: hipLaunchKernelGGL(ll_toi_kernel<64>, dim3(g), dim3(64), 0, stream, dev, io, r, budget); break;
            default: hipLaunchKernelGGL(ll_toi_kernel<32>, dim3(g), dim3(32), 0, stream, dev, io, r, budget); break;
            }
        }
        if (dev.auto_reset) {  // finished envs were compacted onto L_RESET by both kernels; outputs of the step stay as written
            LLIo rio{nullptr, obs_out, nullptr, nullptr, nullptr};
            if (staging_now()) {
                int st = join_helpers();
                if (st != MGYM_OK) return st;
                hipLaunchKernelGGL(ll_apply_select_kernel, dim3(work_grid().x < 64 ? work_grid().x : 64), dim3(256), 0, stream, dev, (const uint32_t*)shadow_base);
                hipLaunchKernelGGL(ll_apply_copy_kernel, dim3(work_grid().x < 2048 ? work_grid().x : 2048), dim3(128), 0, stream, dev,
                                   (const uint32_t*)shadow_base, (const float*)shadow_obs, rio);
                launch_resets(rio, false, L_RESET_DIRECT, true);   // normally empty
                st = launch_prepare(false);
                if (st != MGYM_OK) return st;
            } else {
                launch_resets(rio, false);
            }
        }
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
#else
        set_last_error("LunarLander step: no launch order selected");   // (unreachable: general_only returned above, the fused order is the only other one)
        return MGYM_ERR_BAD_CONFIG;
#endif
    }
    int observation(const float** obs, uint64_t* col_stride) override {
        *obs = dev.obs; *col_stride = n_pad;
        return MGYM_OK;
    }
    // mgym_get_info: the launch structure of this handle and whether its streams really run side by side
    int info(std::string& out) override {
        out += "contact_block=" + std::to_string(gen_block) + "\nlaunch_order=" + (single_launch ? "single_launch" : overlap ? "overlapped" : "sequential") +
               "\nstaged_resets=" + (staged ? "1" : "0") + "\ncontact_list_by_kind=" + std::to_string(dev.bucket) + "\ncontact_blocks_target=" + std::to_string(single_launch ? dev.contact_blocks : 0) +
               "\nknobs=MGYM_LL_GENERAL_BLOCK(32|64) MGYM_LL_SINGLE_LAUNCH MGYM_LL_STAGED_RESET MGYM_LL_STAGED_IN_GRAPH MGYM_LL_CONTACT_BLOCKS MGYM_LL_VC_NEAR MGYM_LL_RESUME MGYM_LL_GENERAL_ONLY "
               "MGYM_LL_AUX_PRIO MGYM_LL_BUCKET MGYM_LL_ROLLOUT MGYM_LL_ROLLOUT_MIN_K MGYM_LL_ROLL_HELPER MGYM_LL_ROLL_STATS MGYM_LL_ROLL_TRACE" +
               "\nrollout=" + (roll_enabled ? "persistent_launch" : "k_steps") + "\nrollout_min_k=" + std::to_string(roll_min_k) + "\nrollout_waves=" + std::to_string(roll_grid) + "\nrollout_helper_blocks=" + std::to_string(roll_helper_grid) + "\n";
        if (capturing()) { set_last_error("mgym_get_info: the stream is being captured (the call launches probe kernels and synchronises)"); return MGYM_ERR_BAD_ARG; }
        hipStream_t ss[3] = {stream, aux, aux2};
        const int ns = aux2 ? 3 : 2;
        unsigned long long* d_t = nullptr;
        unsigned long long h_t[6] = {0, 0, 0, 0, 0, 0};
        MGYM_HIP(hipStreamSynchronize(stream));
        MGYM_HIP(hipStreamSynchronize(aux));
        if (aux2) MGYM_HIP(hipStreamSynchronize(aux2));
        MGYM_HIP(hipMalloc((void**)&d_t, sizeof h_t));
        hipError_t perr = hipSuccess;
        for (int q = 0; q < ns; ++q) hipLaunchKernelGGL(ll_queue_probe_kernel, dim3(1), dim3(1), 0, ss[q], d_t + 2 * q, 5000ull);  // 50 us each
        for (int q = 0; q < ns && perr == hipSuccess; ++q) perr = hipStreamSynchronize(ss[q]);
        if (perr == hipSuccess) perr = hipMemcpy(h_t, d_t, sizeof h_t, hipMemcpyDeviceToHost);
        (void)hipFree(d_t);   // (also on the error paths)
        MGYM_HIP(perr);
        // the largest set of probes whose intervals share an instant: count the intervals covering each start time
        int best = 1;
        for (int a = 0; a < ns; ++a) {
            int c = 0;
            for (int b = 0; b < ns; ++b) c += (h_t[2 * b] <= h_t[2 * a] && h_t[2 * a] < h_t[2 * b + 1]) ? 1 : 0;
            if (c > best) best = c;
        }
        out += "streams=" + std::to_string(ns) + "\nconcurrent_streams=" + std::to_string(best) + "\n";
        return MGYM_OK;
    }
    int get_state(void* blob) override {
        if (n == 0) return MGYM_OK;
        hipLaunchKernelGGL(ll_export_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dev, static_cast<uint32_t*>(blob));
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }
    int set_state(const void* blob) override {
        if (n == 0) return MGYM_OK;
        direct_possible = true;   // imported episode counters need not match the prepared ones
        hipLaunchKernelGGL(ll_import_kernel, dim3((unsigned)((n + kLLBlock - 1) / kLLBlock)), dim3(kLLBlock), 0, stream, dev, static_cast<const uint32_t*>(blob));
        if (fused_tail) { int st = rebuild_list(); if (st != MGYM_OK) return st; }
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }
    int set_dispersion(const float* disp) override {
        direct_possible = true;
        if (disp != dev.disp) ++config_epoch;   // the dispersion source is a kernel argument: graphs captured under the other one are refused (abi.hip)
        if (staged) {  // states prepared under the other dispersion source no longer fit: episode 0 never equals a live counter + 1
            int st = join_helpers();
            if (st != MGYM_OK) return st;
            MGYM_HIP(clear_shadow_episodes());
        }
        dev.disp = disp;
        return MGYM_OK;
    }
};

Env* make_lunarlander() { return new LunarLanderEnv(); }

}  // namespace mgym
