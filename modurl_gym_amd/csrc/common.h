// common.h — engine-internal declarations shared by the kernels' host wrappers and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/mgym.h"

namespace mgym {

void set_last_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define MGYM_HIP(expr)                                                        \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return ::mgym::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kDoneShards = 256;     // per-handle finished-episode counter: partial sums, one fire-and-forget atomic per wave
constexpr int kMaxBlocks = 8192;     // grid-stride beyond this (measured best of 2048/4096/8192/uncapped at 32 Mi envs)

inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }
inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
inline int grid_for(uint64_t work_items) {
    uint64_t b = (work_items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > (uint64_t)kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// device-side sticky status bits (OR-ed into Env::d_err)
enum : uint32_t { DEV_ERR_INVALID_ACTION = 1u, DEV_ERR_NOT_RESET = 2u, DEV_ERR_CONTACT_OVERFLOW = 4u, DEV_ERR_SOLVER_OVERFLOW = 8u, DEV_ERR_INTERNAL = 16u };

// Base of every environment family: one device, one stream, engine-owned SoA state.
struct Env {
    mgym_config cfg{};
    int kind = 0;
    int obs_dim = 0;
    int state_cols = 0;
    uint64_t n = 0;      // environments
    uint64_t n_pad = 0;  // column stride (multiple of 1024 words: every column 4 KiB aligned)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t* d_err = nullptr;   // sticky device status word
    uint32_t* h_err = nullptr;   // pinned mirror
    unsigned long long* d_done = nullptr;  // kDoneShards partial counts of finished env-steps (done | truncated)
    unsigned long long* h_done = nullptr;  // pinned mirror
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    virtual ~Env() {}
    virtual int init() = 0;
    virtual int reset(const uint8_t* m0, const uint8_t* m1, bool all, float* obs_out) = 0;
    virtual int step(const void* actions, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) = 0;
    // K fused steps; default: K launches of step() with strided per-step buffers
    virtual int rollout(const void* actions, int K, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) {
        for (int k = 0; k < K; ++k) {
            const char* a = static_cast<const char*>(actions) + (size_t)k * n * 4;
            int st = step(a, obs_out ? obs_out + (size_t)k * obs_dim * n : nullptr, reward ? reward + (size_t)k * n : nullptr,
                          done ? done + (size_t)k * n : nullptr, trunc ? trunc + (size_t)k * n : nullptr);
            if (st != MGYM_OK) return st;
        }
        return MGYM_OK;
    }
    // K fused steps under the on-device uniform random policy (SURVEY §8f-1 policy hook)
    virtual int rollout_uniform(uint64_t, int, void*, float*, float*, uint8_t*, uint8_t*) {
        set_last_error("mgym_rollout_uniform: not available for this environment family");
        return MGYM_ERR_BAD_ARG;
    }
    // K fused steps under an on-device linear policy; policy = obs_dim weights + bias (host)
    virtual int rollout_linear(const float*, int, void*, float*, float*, uint8_t*, uint8_t*) {
        set_last_error("mgym_rollout_linear: not available for this environment family");
        return MGYM_ERR_BAD_ARG;
    }
    virtual int observation(const float** obs, uint64_t* col_stride) = 0;
    virtual int get_state(void* blob) = 0;
    virtual int set_state(const void* blob) = 0;
    // Testable::reset_deterministic of the reference's test modules (cartpole.rs:437-442, mountain_car.rs:403-408,
    // lunar_lander.rs:1249-1442)
    virtual int reset_deterministic(float* obs_out) = 0;
    // helper streams forked from `stream` must have rejoined it before a stream capture ends (mgym_graph_end)
    virtual int join_helpers() { return MGYM_OK; }
    // mgym_get_info: family-specific "key=value" lines appended to `out` (may launch probe kernels and synchronise)
    virtual int info(std::string&) { return MGYM_OK; }
    uint64_t config_epoch = 0;   // moves whenever something changes that captured launches have frozen into their arguments (abi.hip GraphBox)
    std::string hwq_env;   // GPU_MAX_HW_QUEUES as the environment held it at mgym_create ("unset" if absent)
    virtual int set_dispersion(const float*) { set_last_error("dispersion override: LunarLander only"); return MGYM_ERR_BAD_ARG; }
};

Env* make_cartpole();
Env* make_mountaincar(bool continuous);
Env* make_lunarlander();

}  // namespace mgym
