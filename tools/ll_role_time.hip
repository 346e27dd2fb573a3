// tools/ll_role_time.hip — times ONE single-launch LunarLander step from a saved steady-state population, with only some block roles
// of ll_step_kernel doing their work (-DLL_ROLE_MASK: bit 0 contact path, bit 1 free-flight path, bit 2 reset preparation).
// NOT product code: a step with a role cut out is wrong physics; every variant restores the same saved state before its timed step.
#include <hip/hip_runtime.h>
#include <stdint.h>
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
    // the steady population is produced by the FULL kernel of this same binary?  No: a masked build cannot step correctly, so the
    // warm-up state is loaded from a file written by the mask-7 build (first run) — /tmp/ll_role_state.bin
    LunarLanderEnv env; env.cfg.struct_size = sizeof(mgym_config); env.cfg.flags = MGYM_FLAG_AUTO_RESET; env.cfg.seed = 5; env.cfg.gravity = -10.f;
    env.cfg.enable_wind = 1; env.cfg.wind_power = 15.f; env.cfg.turbulence_power = 1.5f;
    env.n = n; env.n_pad = round_up(n, 1024);
    CK(hipStreamCreate(&env.stream)); CK(hipMalloc((void**)&env.d_err, 4)); CK(hipMemset(env.d_err, 0, 4));
    CK(hipMalloc((void**)&env.d_done, kDoneShards * 8)); CK(hipMemset(env.d_done, 0, kDoneShards * 8));
    env.init();
    uint32_t* act; float* rew; uint8_t *dn, *tr;
    CK(hipMalloc((void**)&act, 16 * n * 4)); CK(hipMalloc((void**)&rew, n * 4)); CK(hipMalloc((void**)&dn, n)); CK(hipMalloc((void**)&tr, n));
    { std::vector<uint32_t> h(16 * n); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 16) & 3; } CK(hipMemcpy(act, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    const size_t st_bytes = (size_t)ll_state_words(env.n_pad) * 4, wc_bytes = (size_t)(L_LISTS * env.n_pad + 64) * 4;
    std::vector<char> st(st_bytes), sh(st_bytes), wc(wc_bytes), cls(env.n_pad);
    const char* path = "/tmp/ll_role_state.bin";
    FILE* f = fopen(path, "rb");
    if (LL_ROLE_MASK == 7 && !f) {
        env.reset(nullptr, nullptr, true, nullptr);
        for (int t = 0; t < 640; ++t) env.step(act + (uint64_t)(t % 16) * n, nullptr, rew, dn, tr);
        CK(hipStreamSynchronize(env.stream));
        CK(hipMemcpy(st.data(), env.base, st_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(sh.data(), env.shadow_base, st_bytes, hipMemcpyDeviceToHost));
        CK(hipMemcpy(wc.data(), env.work_base, wc_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(cls.data(), env.env_class, env.n_pad, hipMemcpyDeviceToHost));
        f = fopen(path, "wb"); fwrite(st.data(), 1, st_bytes, f); fwrite(sh.data(), 1, st_bytes, f); fwrite(wc.data(), 1, wc_bytes, f); fwrite(cls.data(), 1, env.n_pad, f); fclose(f);
    } else if (f) {
        if (fread(st.data(), 1, st_bytes, f) != st_bytes || fread(sh.data(), 1, st_bytes, f) != st_bytes || fread(wc.data(), 1, wc_bytes, f) != wc_bytes || fread(cls.data(), 1, env.n_pad, f) != env.n_pad) { printf("bad state file\n"); return 1; }
        fclose(f);
    } else { printf("run the mask-7 build first\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int reps = 12;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemcpy(env.base, st.data(), st_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(env.shadow_base, sh.data(), st_bytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(env.work_base, wc.data(), wc_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(env.env_class, cls.data(), env.n_pad, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, env.stream));
        env.step(act + (uint64_t)(640 % 16) * n, nullptr, rew, dn, tr);
        CK(hipEventRecord(e1, env.stream));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) { sum += ms; if (ms < best) best = ms; }
    }
    printf("one step from the saved state: mean %.3f ms  best %.3f ms\n", sum / (reps - 1), best);
    return 0;
}
