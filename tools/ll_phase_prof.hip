// tools/ll_phase_prof.hip — diagnostic build of the LunarLander kernels with per-phase wave-cycle stamps
// (NOT product code; never quote its run time).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <initializer_list>
#include <stdint.h>
__device__ unsigned long long g_prof[24];
__device__ unsigned long long g_cnt[24];
__device__ unsigned long long g_blk[8192 * 24];   // per block (wave) of the general kernel: cycles per phase, this launch
__device__ unsigned long long g_blkcnt[8192 * 24];
#define LL_STAMP(id) do { if ((int)(threadIdx.x & 63) == __ffsll((unsigned long long)__ballot(1)) - 1) { unsigned long long t_ = __builtin_readcyclecounter(); \
    if (id != 0) { atomicAdd(&g_prof[id], t_ - s_last); atomicAdd(&g_cnt[id], 1ull); if (blockIdx.x < 8192) { g_blk[blockIdx.x * 24 + id] += t_ - s_last; g_blkcnt[blockIdx.x * 24 + id] += 1; } } s_last = t_; } } while (0)
static __device__ __shared__ unsigned long long s_last;
// lanes of the wave that take a sub-step, summed over the wave's passes of one launch (slot 19 of the per-block COUNT table)
#define LL_DIAG_SUBSTEP_LANES() do { const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll(m_) - 1 && blockIdx.x < 8192) g_blkcnt[blockIdx.x * 24 + 19] += (unsigned long long)__popcll(m_); } while (0)
// sweep-state cycle diagnostics: per lane, hash the state after every sweep; record the first sweep at which the
// hash equals the one 1..4 sweeps earlier (period p), or 180 if never.  g_cyc[kind][period 0..4][bucket of 10 sweeps]
__device__ unsigned long long g_cyc[2][5][19];
__device__ inline unsigned long long diag_hash(std::initializer_list<float> w) { unsigned long long h = 1469598103934665603ull; for (float f : w) { h ^= __float_as_uint(f); h *= 1099511628211ull; } return h; }
#ifdef LL_CYCLE_DIAG  // opt-in: the hashing inflates the sweep phases
#define LL_DIAG_SWEEP 1
#define LL_DIAG_SWEEP_BEGIN(kind) unsigned long long dh_[4] = {1, 2, 3, 4}; int dper_ = 0, dit_ = 180; bool dok_ = true
#define LL_DIAG_SWEEP(kind, it, ok, ...) do { dok_ = dok_ && (ok); unsigned long long h_ = diag_hash({__VA_ARGS__}); if (dper_ == 0) { for (int p_ = 0; p_ < 4; ++p_) if (h_ == dh_[p_]) { dper_ = p_ + 1; dit_ = it; break; } } dh_[3] = dh_[2]; dh_[2] = dh_[1]; dh_[1] = dh_[0]; dh_[0] = h_; } while (0)
#define LL_DIAG_SWEEP_END(kind) do { if (dok_) atomicAdd(&g_cyc[kind][dper_][dit_ / 10], 1ull); } while (0)
#endif
#define LL_DIAG 1   // the measurement-only launch orders and block sizes (lunar_lander.hip)
#include "../modurl_gym_amd/csrc/lunar_lander.hip"
// (the persistent rollout launch lives in ll_roll.hip; these stand-alone measurement binaries never call it)
namespace mgym { int ll_rollout_blocks_per_cu(int* per_cu) { *per_cu = 1; return 0; } void ll_rollout_ring_init(hipStream_t, const RollQ&) {}
void ll_rollout_begin(hipStream_t, const RollQ&, uint32_t) {} void ll_rollout_launch(hipStream_t, unsigned, const LLDev&, const LLIo&, const RollQ&) {}
void ll_rollout_helper_launch(hipStream_t, unsigned, const LLDev&, const LLIo&, const RollQ&) {} }

#include <stdio.h>
#include <vector>
#include <algorithm>
namespace mgym {
void set_last_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); }
int hip_fail(hipError_t e, const char* what, const char*, int) { fprintf(stderr, "HIP %d %s\n", (int)e, what); return 4; }
Env* make_cartpole() { return nullptr; }
Env* make_mountaincar(bool) { return nullptr; }
}
using namespace mgym;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], 0, 0) : (1u << 18);
    int steps = argc > 2 ? atoi(argv[2]) : 150;
    int warm = argc > 3 ? atoi(argv[3]) : 100;
    LunarLanderEnv env; env.cfg.struct_size = sizeof(mgym_config); env.cfg.flags = MGYM_FLAG_AUTO_RESET; env.cfg.seed = 5; env.cfg.gravity = -10.f;
    env.cfg.enable_wind = 1; env.cfg.wind_power = 15.f; env.cfg.turbulence_power = 1.5f;
    env.n = n; env.n_pad = round_up(n, 1024);
    CK(hipStreamCreate(&env.stream)); CK(hipMalloc((void**)&env.d_err, 4)); CK(hipMemset(env.d_err, 0, 4));
    CK(hipMalloc((void**)&env.d_done, kDoneShards * 8)); CK(hipMemset(env.d_done, 0, kDoneShards * 8));
    env.init();
    uint32_t* act; float* rew; uint8_t *dn, *tr;
    CK(hipMalloc((void**)&act, 16 * n * 4)); CK(hipMalloc((void**)&rew, n * 4)); CK(hipMalloc((void**)&dn, n)); CK(hipMalloc((void**)&tr, n));
    { std::vector<uint32_t> h(16 * n); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 16) & 3; } CK(hipMemcpy(act, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    env.reset(nullptr, nullptr, true, nullptr);
    for (int t = 0; t < warm; ++t) env.step(act + (uint64_t)(t % 16) * n, nullptr, rew, dn, tr);  // reach the steady mix of flight / contact
    CK(hipStreamSynchronize(env.stream));
    unsigned long long z[24] = {0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), z, sizeof z));
    static unsigned long long hb[8192 * 24], hc[8192 * 24], zz[8192 * 24];
    std::vector<double> all_tot;
    static double tab_t[6][7], tab_n[6][7];
    double worst_phase[24] = {0}, worst_cnt[24] = {0}, worst_tot = 0, mean_tot = 0; long nwaves = 0;
    // contact list by kind (MGYM_LL_BUCKET=1, 64-lane blocks): the blocks of the touching end (the first ceil(c1 / 64)) and the others, each by phase
    static double kind_phase[2][24], kind_cnt[2][24]; double kind_blocks[2] = {0, 0}, kind_tot[2] = {0, 0}, kind_max[2] = {0, 0}, kind_envs[2] = {0, 0};
    for (int t = 0; t < steps; ++t) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_blk), zz, sizeof zz)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_blkcnt), zz, sizeof zz));
        uint32_t wc[2] = {0, 0};
        CK(hipMemcpy(wc, env.dev.work_count, sizeof wc, hipMemcpyDeviceToHost));   // L_GENERAL (front), L_GENERAL_T (back) of the step about to run
        env.step(act + (uint64_t)(t % 16) * n, nullptr, rew, dn, tr);
        CK(hipStreamSynchronize(env.stream));
        CK(hipMemcpyFromSymbol(hb, HIP_SYMBOL(g_blk), sizeof hb)); CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_blkcnt), sizeof hc));
        if (env.dev.bucket == 1) {
            const int heavy_blocks = (int)((wc[1] + 63) / 64);
            kind_envs[0] += wc[1]; kind_envs[1] += wc[0];
            for (int b = 0; b < 8192; ++b) {
                if (!hc[b * 24 + 1]) continue;
                const int k = b < heavy_blocks ? 0 : 1;
                double tt = 0;
                for (int q = 0; q < 24; ++q) { kind_phase[k][q] += (double)hb[b * 24 + q]; kind_cnt[k][q] += (double)hc[b * 24 + q]; tt += (double)hb[b * 24 + q]; }
                kind_blocks[k] += 1; kind_tot[k] += tt; if (tt > kind_max[k]) kind_max[k] = tt;
            }
        }
        int wb = -1; double wt = 0;
        for (int b = 0; b < 8192; ++b) { double tt = 0; for (int q = 0; q < 24; ++q) tt += (double)hb[b * 24 + q]; if (tt > 0) { mean_tot += tt; nwaves++; all_tot.push_back(tt); } if (tt > wt) { wt = tt; wb = b; } }
        for (int b = 0; b < 8192; ++b) {   // contact-role blocks (they pass Collide): time by sub-step passes and by lanes x sub-steps
            if (!hc[b * 24 + 1]) continue;
            double tt = 0; for (int q = 0; q < 24; ++q) tt += (double)hb[b * 24 + q];
            const int passes = (int)std::min<unsigned long long>(hc[b * 24 + 17], 5), ls = (int)hc[b * 24 + 19];
            const int col = ls == 0 ? 0 : ls <= 2 ? 1 : ls <= 4 ? 2 : ls <= 6 ? 3 : ls <= 9 ? 4 : ls <= 13 ? 5 : 6;
            tab_t[passes][col] += tt; tab_n[passes][col] += 1;
        }
        if (wb >= 0) { worst_tot += wt; for (int q = 0; q < 24; ++q) { worst_phase[q] += (double)hb[wb * 24 + q]; worst_cnt[q] += (double)hc[wb * 24 + q]; } }
    }
    CK(hipStreamSynchronize(env.stream));
    unsigned long long p[24], c[24];
    CK(hipMemcpyFromSymbol(p, HIP_SYMBOL(g_prof), sizeof p)); CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cnt), sizeof c));
    const char* names[24] = {"", "collide", "island: DFS+integrate", "island: constraint init + joint init", "island: 180 sweeps", "island: integrate pos + position iters + sync",
                             "island: sleep + fixture sync + find contacts", "solve_toi total tail", "toi: contact order + list of evaluations (and gaps between evaluations)", "toi: time_of_impact", "toi: other contacts of the body: update + island build", "toi: integrate + sync", "toi: cs_init", "toi: position iterations (<=20)", "toi: cs_init_velocity", "toi: velocity sweeps (<=180)",
                             "toi: list contacts, pick the earliest", "toi: advance body + update the impacting contact", "toi: fixture sync + flags + find contacts", "", "", "", "", ""};
    double tot = 0; for (int i = 0; i < 24; ++i) tot += (double)p[i];
    printf("wave-cycles by phase over %d steps of %llu envs (general kernel only; shares, not run time):\n", steps, (unsigned long long)n);
    for (int i = 1; i < 24; ++i) if (c[i]) printf("  %-52s %6.2f%%  (%llu stamps, %.0f cyc each)\n", names[i], 100.0 * p[i] / tot, c[i], (double)p[i] / c[i]);
    for (int k = 0; k < 2; ++k) if (kind_blocks[k] > 0) {
        printf("%s blocks: %.0f per launch, %.0f envs per launch, mean %.0f cycles per block (slowest %.0f); by phase (share, stamps per block, cycles per stamp):\n", k ? "OTHER (no touching contact)" : "TOUCHING-end",
               kind_blocks[k] / steps, kind_envs[k] / steps, kind_tot[k] / kind_blocks[k], kind_max[k]);
        for (int i = 1; i < 24; ++i) if (kind_cnt[k][i] > 0) printf("  %-52s %6.2f%%  %6.2f  %8.0f\n", names[i], 100.0 * kind_phase[k][i] / kind_tot[k], kind_cnt[k][i] / kind_blocks[k], kind_phase[k][i] / kind_cnt[k][i]);
    }
    printf("slowest wave per launch: %.0f cycles on average (mean wave %.0f); its phases:\n", worst_tot / steps, mean_tot / (nwaves ? nwaves : 1));
    for (int i = 1; i < 24; ++i) if (worst_phase[i] > 0) printf("  %-52s %6.2f%%  (%.1f stamps per launch)\n", names[i], 100.0 * worst_phase[i] / worst_tot, worst_cnt[i] / steps);
    { unsigned long long cyc[2][5][19]; CK(hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof cyc));
      for (int k = 0; k < 2; ++k) for (int p = 0; p < 5; ++p) { unsigned long long t = 0; for (int b = 0; b < 19; ++b) t += cyc[k][p][b]; if (!t) continue;
        printf("%s sweeps, <=2 contacts: period %d (0 = none found) n=%llu; first-detected-at histogram by 10 sweeps:", k ? "toi" : "island", p, t); for (int b = 0; b < 19; ++b) printf(" %llu", cyc[k][p][b]); printf("\n"); } }
    printf("contact-role blocks: mean cycles (count) by passes with a sub-step (rows 0..5+) and by lane-sub-steps of the block (columns 0, 1-2, 3-4, 5-6, 7-9, 10-13, 14+):\n");
    for (int a = 0; a < 6; ++a) { printf("  %d:", a); for (int b = 0; b < 7; ++b) if (tab_n[a][b] > 0) printf("  %8.0f (%6.0f)", tab_t[a][b] / tab_n[a][b], tab_n[a][b]); else printf("  %8s (%6d)", "-", 0); printf("\n"); }
    std::sort(all_tot.begin(), all_tot.end());
    if (!all_tot.empty()) {
        printf("per-block cycles (all launches): ");
        for (double q : {0.1, 0.5, 0.9, 0.99, 0.999, 1.0}) printf("p%g=%.0f ", q * 100, all_tot[(size_t)((all_tot.size() - 1) * q)]);
        printf(" blocks/launch=%.0f\n", (double)all_tot.size() / steps);
    }
    return 0;
}
