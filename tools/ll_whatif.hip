// tools/ll_whatif.hip — "what would the step cost without X?" for the LunarLander contact path (NOT product code).
// The same kernels, compiled with run-time switches that cut one piece of the solver out (which breaks the physics):
// the population is first brought to its steady flight / contact mix with the real code, its state is saved, and
// every variant then runs the same few steps from that same state, timed with HIP events — an upper bound on what
// any optimisation of that piece can gain, free of the distortion per-phase cycle stamps add to short phases.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math tools/ll_whatif.hip -o tools/_ll_whatif
//   tools/_ll_whatif [envs=262144] [timed steps=1] [warm-up steps=400]
#include <hip/hip_runtime.h>
#include <stdint.h>
#define LL_WHATIF_BUILD 1
#define LL_WHATIF(w, bit) (((w).whatif & (bit)) != 0u)
#define LL_DIAG 1   // the measurement-only launch orders and block sizes (lunar_lander.hip)
#include "../modurl_gym_amd/csrc/lunar_lander.hip"
// (the persistent rollout launch lives in ll_roll.hip; these stand-alone measurement binaries never call it)
namespace mgym { int ll_rollout_blocks_per_cu(int* per_cu) { *per_cu = 1; return 0; } void ll_rollout_ring_init(hipStream_t, const RollQ&) {}
void ll_rollout_begin(hipStream_t, const RollQ&, uint32_t) {} void ll_rollout_launch(hipStream_t, unsigned, const LLDev&, const LLIo&, const RollQ&) {}
void ll_rollout_helper_launch(hipStream_t, unsigned, const LLDev&, const LLIo&, const RollQ&) {} }

#include <stdio.h>
#include <vector>
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
    int steps = argc > 2 ? atoi(argv[2]) : 1;  // 1: every variant does the first step's work on identical input
    int warm = argc > 3 ? atoi(argv[3]) : 400;
    LunarLanderEnv env; env.cfg.struct_size = sizeof(mgym_config); env.cfg.flags = MGYM_FLAG_AUTO_RESET; env.cfg.seed = 5; env.cfg.gravity = -10.f;
    env.cfg.enable_wind = 1; env.cfg.wind_power = 15.f; env.cfg.turbulence_power = 1.5f;
    env.n = n; env.n_pad = round_up(n, 1024);
    CK(hipStreamCreate(&env.stream)); CK(hipMalloc((void**)&env.d_err, 4)); CK(hipMemset(env.d_err, 0, 4));
    CK(hipMalloc((void**)&env.d_done, kDoneShards * 8)); CK(hipMemset(env.d_done, 0, kDoneShards * 8));
    env.init();
    env.dev.whatif = 0u;
    uint32_t* act; float* rew; uint8_t *dn, *tr;
    CK(hipMalloc((void**)&act, 16 * n * 4)); CK(hipMalloc((void**)&rew, n * 4)); CK(hipMalloc((void**)&dn, n)); CK(hipMalloc((void**)&tr, n));
    { std::vector<uint32_t> h(16 * n); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 16) & 3; } CK(hipMemcpy(act, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    env.reset(nullptr, nullptr, true, nullptr);
    for (int t = 0; t < warm; ++t) env.step(act + (uint64_t)(t % 16) * n, nullptr, rew, dn, tr);
    CK(hipStreamSynchronize(env.stream));
    const size_t st_bytes = (size_t)ll_state_words(env.n_pad) * 4;
    void* backup; CK(hipMalloc(&backup, st_bytes));
    CK(hipMemcpy(backup, env.dev.st, st_bytes, hipMemcpyDeviceToDevice));
    struct V { uint32_t mask; const char* what; };
    const V variants[] = {
        {0u, "the real step"},
        {WI_NO_COLLIDE_UPDATE, "Collide: cached contacts are not updated (no edge-polygon manifolds before the island solve)"},
        {WI_ISLAND_SWEEPS_10, "island solve: 10 velocity sweeps instead of 180"},
        {WI_ONE_POSITION_ITER, "island solve: 1 position iteration instead of <= 60"},
        {WI_NO_FIND_CONTACTS, "island solve: no FindNewContacts after the fixture sync"},
        {WI_NO_TOI_EVAL, "SolveTOI: no time_of_impact evaluation at all (hence no sub-steps)"},
        {WI_NO_SUBSTEPS, "SolveTOI: times of impact evaluated once, no sub-step taken"},
        {WI_TOI_SWEEPS_12, "sub-steps: at most 12 velocity sweeps"},
        {WI_NO_TOI_OTHER_UPDATES, "sub-steps: the body's other contacts are not updated"},
        {WI_ONE_SUBSTEP, "sub-steps: at most one per env"},
        {WI_SUB_NO_REEVAL, "sub-steps: no time_of_impact re-evaluation after the first sub-step (so at most one sub-step per body)"},
        {WI_SUB_NO_POSITION, "sub-steps: no position iterations"},
        {WI_SUB_NO_SYNC | WI_SUB_NO_FIND, "sub-steps: no fixture sync, no FindNewContacts"},
        {WI_TOI_SWEEPS_12 | WI_SUB_NO_POSITION | WI_SUB_NO_SYNC | WI_SUB_NO_FIND | WI_NO_TOI_OTHER_UPDATES, "sub-steps: all of the sub-step cuts above except the count (what remains: evaluate, advance, update one contact, solver set-up, 12 sweeps)"},
        {WI_NO_COLLIDE_UPDATE | WI_ISLAND_SWEEPS_10 | WI_ONE_POSITION_ITER | WI_NO_FIND_CONTACTS | WI_NO_TOI_EVAL, "all of the above: load, integrate, store"},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("LunarLander, %llu envs, wind on; %d timed steps from the same saved state (after %d real steps), ms per step:\n", (unsigned long long)n, steps, warm);
    const int reps = 6;  // the first repetition warms caches and is dropped
    for (const V& v : variants) {
        double sum = 0, best = 1e30;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipMemcpy(env.dev.st, backup, st_bytes, hipMemcpyDeviceToDevice));
            env.dev.whatif = v.mask;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, env.stream));
            for (int t = 0; t < steps; ++t) env.step(act + (uint64_t)((warm + t) % 16) * n, nullptr, rew, dn, tr);
            CK(hipEventRecord(e1, env.stream));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) { sum += ms / steps; if (ms / steps < best) best = ms / steps; }
        }
        printf("  mean %7.3f  best %7.3f  %s\n", sum / (reps - 1), best, v.what);
    }
    return 0;
}
