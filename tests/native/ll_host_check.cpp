// CPU check of the LunarLander KERNEL SOURCE (modurl_gym_amd/csrc/ll_*.h compiled for the host through a
// small shim) against the CPU oracle: same seeds, actions and dispersion, step by step.  Built and run by
// tests/test_ll_host.py; no GPU involved.  This is a test of the kernel logic, not a product CPU path.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define __device__
#define __host__
#define __forceinline__ inline
// every time-of-impact velocity solve that stopped before its 180th sweep (fixed point, cycle, settled-velocity proof) is re-run in
// full and must give the same velocity bit for bit
static unsigned long g_verify_total, g_verify_short, g_verify_bad;
#define LL_TOI_SWEEP_VERIFY(v_full, v_got, done) do { g_verify_total++; if ((done) < 180) g_verify_short++; \
    if (memcmp(&(v_full).v.x, &(v_got).v.x, 4) || memcmp(&(v_full).v.y, &(v_got).v.y, 4) || memcmp(&(v_full).w, &(v_got).w, 4)) { if (g_verify_bad++ < 5) printf("TOI SWEEP SHORTCUT MISMATCH after %d sweeps: %.9g %.9g %.9g vs %.9g %.9g %.9g\n", (int)(done), (v_got).v.x, (v_got).v.y, (v_got).w, (v_full).v.x, (v_full).v.y, (v_full).w); } } while (0)
#ifndef LL_TRACE_N
#define LL_TRACE_N 0
#define LL_TRACE_COUNT 1
#endif
#ifdef LL_HOST_STATS  // -DLL_HOST_STATS: histogram of the sub-step velocity sweeps actually run, by constraint count (tools/ll_work_stats.sh)
static unsigned long g_toi_sweeps[13][181], g_rounds_by_touch[2][6], g_rounds_by_ncont[9][6];
#define LL_TOI_SWEEP_STAT(count, done) do { g_toi_sweeps[(count) < 12 ? (count) : 12][(done)]++; if ((done) == 180 && g_trace_left > 0 && (count) == LL_TRACE_COUNT) { --g_trace_left; printf("-- a sub-step that ran all 180 sweeps (%d constraints); state words after sweeps 1..180 (v.x v.y w n00 t00 n01 t01 n10 t10 n11 t11):\n", (int)(count)); for (int q_ = 0; q_ < 180; ++q_) if (q_ < 6 || q_ % 20 == 0 || q_ >= 174) { printf("   %3d:", q_ + 1); for (int z_ = 0; z_ < 11; ++z_) printf(" %08x", g_trace[q_][z_]); printf("\n"); } } } while (0)
static unsigned g_trace[181][11]; static int g_trace_left = LL_TRACE_N;
static unsigned long g_pos_iters[2][13][61], g_pos_how[2][3];
static unsigned g_ptrace[60][9]; static unsigned long g_pcycle_at[62], g_pcycle_period[62]; static int g_ptrace_print = 3;
#define LL_POS_ITER_TRACE(it, pos) do { const float z_[9] = {(pos).b0.c.x, (pos).b0.c.y, (pos).b0.a, (pos).b1.c.x, (pos).b1.c.y, (pos).b1.a, (pos).b2.c.x, (pos).b2.c.y, (pos).b2.a}; memcpy(g_ptrace[(it)], z_, sizeof z_); } while (0)
#define LL_POS_ITER_STAT(kind, count, iters, solved) do { if ((kind) == 0 && (solved) == 0) { int at_ = 61, per_ = 0; for (int j_ = 1; j_ < 60 && at_ == 61; ++j_) for (int i_ = j_ - 1; i_ >= 0; --i_) if (!memcmp(g_ptrace[j_], g_ptrace[i_], 36)) { at_ = j_; per_ = j_ - i_; break; } g_pcycle_at[at_]++; g_pcycle_period[per_]++; \
    if (at_ == 61 && g_ptrace_print > 0) { --g_ptrace_print; printf("-- island position solve that ran out without repeating (%d contacts):\n", (int)(count)); for (int q_ = 0; q_ < 60; q_ += (q_ < 4 || q_ >= 54) ? 1 : 10) { printf("   %2d:", q_); for (int z_ = 0; z_ < 9; ++z_) printf(" %08x", g_ptrace[q_][z_]); printf("\n"); } } } \
    g_pos_iters[kind][(count) < 12 ? (count) : 12][(iters)]++; g_pos_how[kind][(solved)]++; } while (0)
#define LL_POS_ITER_STAT_OLD(kind, count, iters, solved) do { g_pos_iters[kind][(count) < 12 ? (count) : 12][(iters)]++; g_pos_how[kind][(solved)]++; } while (0)
static unsigned long g_free_pos[62]; static int g_free_pos_wave_max; static unsigned long g_free_pos_wavehist[62];
#define LL_FREE_POS_STAT(iters) do { g_free_pos[(iters)]++; if ((iters) > g_free_pos_wave_max) g_free_pos_wave_max = (iters); } while (0)
#define LL_TOI_SWEEP_TRACE(done, cur, count) do { for (int z_ = 0; z_ < 11; ++z_) g_trace[(done) - 1][z_] = (cur).w[z_]; } while (0)
#endif
#include "../../modurl_gym_amd/csrc/ll_free.h"
extern "C" {
#include "../../oracle/oracle.h"
}
using namespace mgym;

static bool closef(float a, float b) { return fabsf(a - b) <= 1e-5f * fmaxf(fabsf(a), fabsf(b)) + 1e-6f || (a != a && b != b); }

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 0) : 64;
    const int steps = argc > 2 ? atoi(argv[2]) : 300;
    const int wind = argc > 3 ? atoi(argv[3]) : 0;
    const int deterministic = argc > 4 ? atoi(argv[4]) : 0;
    const unsigned toi_staged = argc > 5 ? (unsigned)atoi(argv[5]) : 1u;  // 0: the TOI word of the contact cache stays in place (64-lane blocks)
    // other physics settings than the defaults (the early exits of the sweep loops rest on bounds derived from the constraint masses, not on
    // these values, and must hold for every setting the builder accepts: lunar_lander.rs:278-296)
    const float gravity = argc > 6 ? (float)atof(argv[6]) : -10.0f, wind_power = argc > 7 ? (float)atof(argv[7]) : 15.0f, turbulence = argc > 8 ? (float)atof(argv[8]) : 1.5f;
    const uint64_t seed = 77;
    // product side: SoA state exactly as the kernels keep it
    LLDev d;
    memset(&d, 0, sizeof d);
    d.n = n; d.n_pad = (n + 1023) / 1024 * 1024; d.seed = seed; d.env_id_base = 0;
    std::vector<uint32_t> st((size_t)ll_state_words(d.n_pad), 0u);
    std::vector<float> obsbuf((size_t)8 * d.n_pad, 0.0f);
    d.st = st.data(); d.obs = obsbuf.data();
    ll_make_const(d.k, gravity, wind, wind_power, turbulence);
    d.kd = &d.k;
    PolyTab tab;
    for (int p = 0; p < 2; ++p) { tab.count[p] = d.k.poly_count[p]; for (int q = 0; q < kMaxPoly; ++q) { tab.v[p][q] = d.k.poly_v[p][q]; tab.n[p][q] = d.k.poly_n[p][q]; } }
    // oracle side
    ora_vec_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.kind = 3; cfg.n_envs = n; cfg.seed = seed; cfg.gravity = gravity; cfg.enable_wind = wind; cfg.wind_power = wind_power; cfg.turbulence_power = turbulence;
    int status = 0;
    ora_vec* ov = ora_vec_new(&cfg, &status);
    std::vector<float> oobs(8 * n), orew(n); std::vector<uint8_t> odone(n), otr(n), mask(n);
    std::vector<uint32_t> act(n);
    if (deterministic) ora_vec_reset_deterministic(ov, oobs.data()); else ora_vec_reset(ov, NULL, oobs.data(), 1);
    WorldTmp h_tmp;  // the per-lane working storage (LDS on the GPU)
    uint32_t h_hot[3 * kSlots]; const CtHot hot{h_hot, 1u, toi_staged};  // the staged KEY / SEQ / TOI words (LDS on the GPU)
    VConstraint h_vc0[kSlots]; PConstraint h_pc0[kSlots];
    CSolverMem mem0; mem0.vc = h_vc0; mem0.vc_stride = 1; mem0.vc_near = kSlots; mem0.vc_far = nullptr; mem0.vc_far_stride = 0; mem0.pc = h_pc0; mem0.pc_stride = 1; mem0.cap = kSlots;
    unsigned long fast_steps = 0, general_steps = 0, resumed_steps = 0;
    int max_slots = 0, max_rounds = 0; unsigned long hist[kSlots + 1] = {0}, round_hist[8] = {0};
    unsigned long mism = 0, exact = 0, total = 0, done_total = 0, overflow = 0;
    for (uint64_t i = 0; i < n; ++i) {
        World w; EnvRegs e; float state[8];
        w.t = &h_tmp;
        ll_load(d, i, w, e, hot);
        if (deterministic) {
            float height[12]; for (int q = 0; q < 12; ++q) height[q] = (400.0f / 30.0f) / 8.0f;
            ll_build_scene(w, e, tab, d.k, height, 400.0f / 30.0f * 0.8f, false, 0.0f, 0.0f, 0, 0, true);
            e.step = 1u; ll_observe(w.b[0], w.legs[0], w.legs[1], state);
        } else ll_env_reset(d, i, w, e, tab, mem0, state);
        ll_store(d, i, w, e);
        for (int q = 0; q < 8; ++q) { total++; if (state[q] == oobs[q * n + i]) exact++; if (!closef(state[q], oobs[q * n + i])) { if (mism < 10) printf("reset env %lu obs[%d] %.9g vs %.9g\n", (unsigned long)i, q, state[q], oobs[q * n + i]); mism++; } }
    }
    VConstraint h_vc[kSlots]; PConstraint h_pc[kSlots];
    VConstraint h_vc_far[kSlots];  // constraints 2.. go through the far (global workspace) path here, as in the 32-lane blocks on the GPU (64-lane blocks: 4..)
    CSolverMem mem; mem.vc = h_vc; mem.vc_stride = 1; mem.vc_near = 2; mem.vc_far = h_vc_far; mem.vc_far_stride = 1; mem.pc = h_pc; mem.pc_stride = 1; mem.cap = kSlots;
    uint32_t rs = 12345;
    for (int t = 0; t < steps; ++t) {
        for (uint64_t i = 0; i < n; ++i) { rs = rs * 1664525u + 1013904223u; act[i] = (rs >> 16) & 3u; }
        ora_vec_step(ov, act.data(), oobs.data(), orew.data(), odone.data(), otr.data(), 1);
        for (uint64_t i = 0; i < n; ++i) {
            float state[8], reward, d0, d1; uint32_t done;
#ifdef LL_HOST_STATS
            if ((i & 63u) == 0u) { if (i) g_free_pos_wavehist[g_free_pos_wave_max]++; g_free_pos_wave_max = 0; }
#endif
            // same dispatch as ll_free_kernel / ll_general_kernel: fast path when eligible and it accepts
            bool fast = false;
            if (ll_free_eligible(ST(C_FLAGS))) {
                FreeRegs f; EnvRegs e;
                ll_free_load(d, i, f, e);
                ll_dispersion(d, i, e, d0, d1);
                uint32_t moved;
                const int how = ll_free_env_step(d, i, f, e, tab, act[i], d0, d1, state, reward, done, moved);
                if (how == FREE_DONE) { ll_free_store(d, i, f, e); fast = true; fast_steps++; }
                else if (how == FREE_RESUME && (i & 1u) == 0u) { ll_free_store_resume(d, i, f, e, moved); resumed_steps++; }   // as ll_free_kernel does (every other env: both hand-over forms are checked)
            }
            if (!fast) {
                // same structure as ll_general_kernel + the ll_toi_kernel rounds: world.step up to its first
                // time-of-impact evaluation, then one sub-step per "launch", the environment going through its
                // mid-step columns (ll_store / ll_load with mid = true) in between, as on the GPU
                World w; EnvRegs e;
                w.t = &h_tmp;
#ifdef LL_HOST_STATS
                const uint32_t flags0 = ST(C_FLAGS);
                const int touching0 = (flags0 & F_TOUCHING) ? 1 : 0, ncont0 = (int)((flags0 >> F_NCONTACT_SHIFT) & 15u);
#endif
                ll_load(d, i, w, e, hot);
                ll_dispersion(d, i, e, d0, d1);
                bool fin;
                if (w.resume) { ll_resume_after_island(w); fin = solve_toi_part(w, tab, d.k, mem, kStepDt, true, 0); }   // as ll_contact_kernel does
                else fin = ll_step_begin(w, e, tab, d.k, mem, act[i], d0, d1, 0);
                int rounds = 0;
                while (!fin) {
                    ll_store(d, i, w, e, true);
                    World w2; EnvRegs e2;
                    w2.t = &h_tmp;
                    ll_load(d, i, w2, e2, hot, true);
                    w2.overflow |= w.overflow;
                    w = w2; e = e2;
                    fin = ll_step_continue(w, tab, d.k, mem, rounds < 3 ? 1 : -1);
                    ++rounds;
                }
                if (rounds > max_rounds) max_rounds = rounds;
                round_hist[rounds < 7 ? rounds : 7]++;
#ifdef LL_HOST_STATS
                g_rounds_by_touch[touching0][rounds < 5 ? rounds : 5]++; g_rounds_by_ncont[ncont0 < 8 ? ncont0 : 8][rounds < 5 ? rounds : 5]++;
#endif
                ll_step_finish(w, e, act[i], state, reward, done);
                if (w.overflow) overflow++;
                { int nc = 0; for (int q = 0; q < kSlots; ++q) nc += (ct_key(w.cs, q) & CK_EXISTS) ? 1 : 0; if (nc > max_slots) max_slots = nc; hist[nc]++; }
                ll_store(d, i, w, e);
                general_steps++;
            }
            bool bad = done != odone[i] || !closef(reward, orew[i]);
            for (int q = 0; q < 8; ++q) { total++; if (state[q] == oobs[q * n + i]) exact++; bad |= !closef(state[q], oobs[q * n + i]); }
            if (bad) {
                if (mism < 10) {
                    printf("step %d env %lu act %u: done %u vs %u reward %.9g vs %.9g\n", t, (unsigned long)i, act[i], done, odone[i], reward, orew[i]);
                    for (int q = 0; q < 8; ++q) printf("   obs[%d] %.9g vs %.9g\n", q, state[q], oobs[q * n + i]);
                }
                mism++;
            }
            mask[i] = odone[i]; done_total += odone[i];
        }
        if (t % 2 == 0 && !deterministic) {  // masked reset of finished envs on both sides
            ora_vec_reset(ov, mask.data(), NULL, 1);
            for (uint64_t i = 0; i < n; ++i) if (mask[i]) { World w; EnvRegs e; float state[8]; w.t = &h_tmp; ll_load(d, i, w, e, hot); ll_env_reset(d, i, w, e, tab, mem, state); ll_store(d, i, w, e); }
        }
    }
    printf("envs=%lu steps=%d wind=%d det=%d mismatches=%lu exact_words=%lu/%lu episodes_done=%lu overflow=%lu\n", (unsigned long)n, steps, wind, deterministic, mism, exact, total, done_total, overflow);
    printf("time-of-impact velocity solves %lu, of which shortened %lu, all re-run in full: %lu differ\n", g_verify_total, g_verify_short, g_verify_bad);
    if (g_verify_bad) mism += g_verify_bad;
    printf("fast-path steps %lu, general-path steps %lu (of which resumed after the free path's island solve: %lu)\n", fast_steps, general_steps, resumed_steps);
    printf("max simultaneous cached contacts %d; histogram:", max_slots);
    for (int q = 0; q <= kSlots; ++q) printf(" %lu", hist[q]);
    printf("\n");
    printf("time-of-impact rounds per general step (max %d):", max_rounds);
    for (int q = 0; q < 8; ++q) printf(" %lu", round_hist[q]);
    printf("\n");
#ifdef LL_HOST_STATS
    { unsigned long tot = 0, sum = 0; for (int q = 0; q < 62; ++q) { tot += g_free_pos[q]; sum += g_free_pos[q] * (q > 60 ? 60 : q); }
      printf("free-flight steps %lu: joint position iterations mean %.2f; histogram 1..10:", tot, (double)sum / (tot ? tot : 1)); for (int q = 1; q <= 10; ++q) printf(" %.2f%%", 100.0 * g_free_pos[q] / (tot ? tot : 1));
      unsigned long m20 = 0, m60 = 0; for (int q = 11; q < 62; ++q) { if (q <= 20) m20 += g_free_pos[q]; else m60 += g_free_pos[q]; } printf("  11-20: %.3f%%  21-60+: %.3f%% (ran out: %.3f%%)\n", 100.0 * m20 / (tot ? tot : 1), 100.0 * m60 / (tot ? tot : 1), 100.0 * g_free_pos[61] / (tot ? tot : 1));
      unsigned long wt = 0, ws = 0; for (int q = 0; q < 62; ++q) { wt += g_free_pos_wavehist[q]; ws += g_free_pos_wavehist[q] * (q > 60 ? 60 : q); }
      printf("   maximum over each group of 64 consecutive envs (what a wave runs): mean %.1f; histogram 0..12:", (double)ws / (wt ? wt : 1)); for (int q = 0; q <= 12; ++q) printf(" %.1f%%", 100.0 * g_free_pos_wavehist[q] / (wt ? wt : 1));
      unsigned long w60 = 0; for (int q = 13; q < 62; ++q) w60 += g_free_pos_wavehist[q]; printf("  13+: %.1f%% (60: %.1f%%)\n", 100.0 * w60 / (wt ? wt : 1), 100.0 * (g_free_pos_wavehist[60] + g_free_pos_wavehist[61]) / (wt ? wt : 1)); }
    for (int tq = 0; tq < 2; ++tq) { unsigned long tot = 0, sum = 0; for (int r = 0; r < 6; ++r) { tot += g_rounds_by_touch[tq][r]; sum += g_rounds_by_touch[tq][r] * r; }
        printf("general steps that start %s a touching contact: %lu; sub-steps 0/1/2/3/4/5+:", tq ? "with" : "without", tot); for (int r = 0; r < 6; ++r) printf(" %.1f%%", 100.0 * g_rounds_by_touch[tq][r] / (tot ? tot : 1)); printf("  mean %.2f\n", (double)sum / (tot ? tot : 1)); }
    for (int c = 0; c < 9; ++c) { unsigned long tot = 0, sum = 0; for (int r = 0; r < 6; ++r) { tot += g_rounds_by_ncont[c][r]; sum += g_rounds_by_ncont[c][r] * r; }
        if (tot) { printf("general steps that start with %d cached contacts: %lu; sub-steps 0/1/2/3/4/5+:", c, tot); for (int r = 0; r < 6; ++r) printf(" %.1f%%", 100.0 * g_rounds_by_ncont[c][r] / tot); printf("  mean %.2f\n", (double)sum / tot); } }
    for (int c = 0; c < 13; ++c) {
        unsigned long tot = 0, full = 0, le16 = 0, le48 = 0, sum = 0;
        for (int q = 0; q <= 180; ++q) { tot += g_toi_sweeps[c][q]; sum += g_toi_sweeps[c][q] * q; if (q == 180) full += g_toi_sweeps[c][q]; if (q <= 16) le16 += g_toi_sweeps[c][q]; if (q <= 48) le48 += g_toi_sweeps[c][q]; }
        if (tot) printf("sub-step velocity solves with %d constraints: %lu; sweeps run: mean %.1f, <=16: %.1f%%, <=48: %.1f%%, all 180: %.1f%%\n", c, tot, (double)sum / tot, 100.0 * le16 / tot, 100.0 * le48 / tot, 100.0 * full / tot);
    }
#endif
#ifdef LL_HOST_STATS
    for (int c = 0; c < 13; ++c) {
        unsigned long tot = 0, sum = 0, full = 0, le4 = 0, le16 = 0;
        for (int q = 0; q <= 60; ++q) { tot += g_pos_iters[0][c][q]; sum += g_pos_iters[0][c][q] * q; if (q == 60) full += g_pos_iters[0][c][q]; if (q <= 4) le4 += g_pos_iters[0][c][q]; if (q <= 16) le16 += g_pos_iters[0][c][q]; }
        if (tot) printf("island position solves with %d contact constraints: %lu; iterations: mean %.1f, <=4: %.1f%%, <=16: %.1f%%, all 60: %.1f%%\n", c, tot, (double)sum / tot, 100.0 * le4 / tot, 100.0 * le16 / tot, 100.0 * full / tot);
    }
    printf("island position solves that ran out: first repeated state at iteration (61 = never):"); for (int q = 0; q < 62; ++q) if (g_pcycle_at[q]) printf(" %d:%lu", q, g_pcycle_at[q]); printf("\n   period:"); for (int q = 0; q < 62; ++q) if (g_pcycle_period[q]) printf(" %d:%lu", q, g_pcycle_period[q]); printf("\n");
    printf("island position solves ended: solved %lu, fixed point (unsolved) %lu, ran out %lu\n", g_pos_how[0][1], g_pos_how[0][2], g_pos_how[0][0]);
#endif
    ora_vec_free(ov);
    return mism ? 1 : 0;
}
