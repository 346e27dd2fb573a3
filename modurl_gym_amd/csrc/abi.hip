// abi.hip — the extern "C" entry points declared in include/mgym.h.
// Thin: argument checks, device selection, dispatch to the per-family Env object.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "common.h"

namespace mgym {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_last_error = buf;
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) ? MGYM_ERR_NO_DEVICE
                                                                                                      : MGYM_ERR_HIP;
}

static int bad_arg(const char* msg) {
    g_last_error = msg;
    return MGYM_ERR_BAD_ARG;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace mgym

using namespace mgym;

// SoA [obs_dim][stride] -> AoS [n][obs_dim]; one lane per env, obs_dim (<= 8) consecutive words per lane
__global__ void obs_to_aos_kernel(const float* __restrict__ soa, uint64_t stride, int obs_dim, uint64_t n, float* __restrict__ aos) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < obs_dim; ++k) aos[i * obs_dim + k] = soa[(uint64_t)k * stride + i];
}

struct mgym_env {
    Env* impl;
};

extern "C" {

int mgym_abi_version(void) { return MGYM_ABI_VERSION; }

int mgym_default_config(int kind, mgym_config* cfg) {
    if (!cfg) return bad_arg("mgym_default_config: cfg is NULL");
    if (kind < MGYM_CARTPOLE || kind > MGYM_LUNARLANDER) return bad_arg("mgym_default_config: unknown kind");
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = (uint32_t)sizeof *cfg;
    cfg->kind = kind;
    cfg->n_envs = 1;
    cfg->sutton_barto_reward = 0;  // cartpole.rs:39
    cfg->is_euler = 1;             // cartpole.rs:40
    cfg->goal_velocity = 0.0f;     // mountain_car.rs:33
    cfg->gravity = -10.0f;         // lunar_lander.rs:282
    cfg->enable_wind = 0;          // :283
    cfg->wind_power = 15.0f;       // :284
    cfg->turbulence_power = 1.5f;  // :285
    return MGYM_OK;
}

int mgym_get_spec(int kind, mgym_spec* s) {
    if (!s) return bad_arg("mgym_get_spec: spec is NULL");
    memset(s, 0, sizeof *s);
    const float inf = __builtin_inff();
    switch (kind) {
    case MGYM_CARTPOLE: {  // cartpole.rs:58-69
        s->obs_dim = 4; s->n_actions = 2; s->state_cols = 7;
        const float thr = 12.0f * 2.0f * 3.14159265358979323846f / 360.0f;
        const float hi[4] = {2.4f * 2.0f, inf, thr * 2.0f, inf};
        for (int k = 0; k < 4; ++k) { s->obs_high[k] = hi[k]; s->obs_low[k] = -hi[k]; }
        break;
    }
    case MGYM_MOUNTAINCAR:
    case MGYM_MOUNTAINCAR_CONT:  // mountain_car.rs:42-48
        s->obs_dim = 2; s->state_cols = 3;
        s->obs_low[0] = -1.2f; s->obs_low[1] = -0.07f; s->obs_high[0] = 0.6f; s->obs_high[1] = 0.07f;
        if (kind == MGYM_MOUNTAINCAR) { s->n_actions = 3; }
        else { s->n_actions = 0; s->action_is_float = 1; s->action_low = -1.0f; s->action_high = 1.0f; }
        break;
    case MGYM_LUNARLANDER: {  // lunar_lander.rs:1169-1200
        s->obs_dim = 8; s->n_actions = 4; s->state_cols = 27;
        const float twopi = 2.0f * 3.14159265358979323846f;
        const float hi[8] = {2.5f, 2.5f, 10.0f, 10.0f, twopi, 10.0f, 1.0f, 1.0f};
        for (int k = 0; k < 8; ++k) { s->obs_high[k] = hi[k]; s->obs_low[k] = k < 6 ? -hi[k] : 0.0f; }
        break;
    }
    default:
        return bad_arg("mgym_get_spec: unknown kind");
    }
    return MGYM_OK;
}

int mgym_device_count(int* count) {
    if (!count) return bad_arg("mgym_device_count: NULL");
    MGYM_HIP(hipGetDeviceCount(count));
    return MGYM_OK;
}

int mgym_create(const mgym_config* cfg, mgym_env** out) {
    if (!cfg || !out) return bad_arg("mgym_create: NULL argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(mgym_config)) return bad_arg("mgym_create: struct_size mismatch (ABI version?)");
    if (cfg->kind == MGYM_LUNARLANDER && !(-12.0f < cfg->gravity && cfg->gravity < 0.0f)) {
        // lunar_lander.rs:292-296
        char buf[128];
        snprintf(buf, sizeof buf, "gravity (current value: %g) must be between -12 and 0", (double)cfg->gravity);
        set_last_error(buf);
        return MGYM_ERR_BAD_CONFIG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_last_error("no HIP device visible: libmgym has no CPU fallback (build/run on an MI355X)");
        return MGYM_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) return bad_arg("mgym_create: device ordinal out of range");
    DeviceGuard guard(cfg->device);
    if (!guard.ok) return hip_fail(hipErrorInvalidDevice, "hipSetDevice", __FILE__, __LINE__);

    Env* impl = nullptr;
    switch (cfg->kind) {
    case MGYM_CARTPOLE: impl = make_cartpole(); break;
    case MGYM_MOUNTAINCAR: impl = make_mountaincar(false); break;
    case MGYM_MOUNTAINCAR_CONT: impl = make_mountaincar(true); break;
    case MGYM_LUNARLANDER: impl = make_lunarlander(); break;
    default: return bad_arg("mgym_create: unknown kind");
    }
    if (!impl) { set_last_error("mgym_create: environment family not available in this build"); return MGYM_ERR_BAD_CONFIG; }
    impl->cfg = *cfg;
    { const char* q = getenv("GPU_MAX_HW_QUEUES"); impl->hwq_env = q ? q : "unset"; }
    impl->kind = cfg->kind;
    impl->n = cfg->n_envs;
    impl->n_pad = round_up(cfg->n_envs ? cfg->n_envs : 1, 1024);
    int st = MGYM_OK;
    do {
        if (hipStreamCreateWithFlags(&impl->stream, hipStreamNonBlocking) != hipSuccess) { st = MGYM_ERR_HIP; set_last_error("hipStreamCreate failed"); break; }
        impl->own_stream = true;
        if (hipMalloc((void**)&impl->d_err, sizeof(uint32_t)) != hipSuccess) { st = MGYM_ERR_HIP; set_last_error("hipMalloc failed"); break; }
        if (hipHostMalloc((void**)&impl->h_err, sizeof(uint32_t), 0) != hipSuccess) { st = MGYM_ERR_HIP; set_last_error("hipHostMalloc failed"); break; }
        if (hipMemsetAsync(impl->d_err, 0, sizeof(uint32_t), impl->stream) != hipSuccess) { st = MGYM_ERR_HIP; break; }
        if (hipMalloc((void**)&impl->d_done, kDoneShards * sizeof(unsigned long long)) != hipSuccess) { st = MGYM_ERR_HIP; set_last_error("hipMalloc failed"); break; }
        if (hipHostMalloc((void**)&impl->h_done, kDoneShards * sizeof(unsigned long long), 0) != hipSuccess) { st = MGYM_ERR_HIP; set_last_error("hipHostMalloc failed"); break; }
        if (hipMemsetAsync(impl->d_done, 0, kDoneShards * sizeof(unsigned long long), impl->stream) != hipSuccess) { st = MGYM_ERR_HIP; break; }
        if (hipEventCreate(&impl->ev0) != hipSuccess || hipEventCreate(&impl->ev1) != hipSuccess) { st = MGYM_ERR_HIP; break; }
        st = impl->init();
        if (st != MGYM_OK) break;
        if (hipStreamSynchronize(impl->stream) != hipSuccess) { st = MGYM_ERR_HIP; set_last_error("init sync failed"); break; }
    } while (0);
    if (st != MGYM_OK) {
        mgym_env tmp{impl};
        mgym_env* p = new mgym_env(tmp);
        mgym_destroy(p);
        return st;
    }
    *out = new mgym_env{impl};
    return MGYM_OK;
}

int mgym_destroy(mgym_env* env) {
    if (!env) return MGYM_OK;
    Env* e = env->impl;
    if (e) {
        DeviceGuard guard(e->cfg.device);
        if (e->stream) (void)hipStreamSynchronize(e->stream);
        if (e->ev0) (void)hipEventDestroy(e->ev0);
        if (e->ev1) (void)hipEventDestroy(e->ev1);
        if (e->d_err) (void)hipFree(e->d_err);
        if (e->h_err) (void)hipHostFree(e->h_err);
        if (e->d_done) (void)hipFree(e->d_done);
        if (e->h_done) (void)hipHostFree(e->h_done);
        hipStream_t s = e->own_stream ? e->stream : nullptr;
        delete e;
        if (s) (void)hipStreamDestroy(s);
    }
    delete env;
    return MGYM_OK;
}

#define ENV_OR_FAIL(env)                                        \
    if (!(env) || !(env)->impl) return bad_arg("NULL env handle"); \
    Env* e = (env)->impl;                                       \
    DeviceGuard guard(e->cfg.device);

int mgym_set_stream(mgym_env* env, void* hip_stream) {
    ENV_OR_FAIL(env);
    if (e->own_stream && e->stream) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamDestroy(e->stream);
    }
    e->stream = static_cast<hipStream_t>(hip_stream);
    e->own_stream = false;
    return MGYM_OK;
}

void* mgym_get_stream(mgym_env* env) { return (env && env->impl) ? (void*)env->impl->stream : nullptr; }

int mgym_reset(mgym_env* env, const uint8_t* mask, float* obs_out) {
    ENV_OR_FAIL(env);
    return e->reset(mask, nullptr, mask == nullptr, obs_out);
}

int mgym_reset_done(mgym_env* env, const uint8_t* done, const uint8_t* truncated, float* obs_out) {
    ENV_OR_FAIL(env);
    if (!done && !truncated) return bad_arg("mgym_reset_done: both masks NULL (use mgym_reset(env, NULL, ..) to reset all)");
    return e->reset(done, truncated, false, obs_out);
}

int mgym_reset_deterministic(mgym_env* env, float* obs_out) {
    ENV_OR_FAIL(env);
    return e->reset_deterministic(obs_out);
}

int mgym_step(mgym_env* env, const void* actions, float* obs_out, float* reward_out, uint8_t* done_out,
              uint8_t* trunc_out) {
    ENV_OR_FAIL(env);
    if (!actions && e->n) return bad_arg("mgym_step: actions is NULL");
    return e->step(actions, obs_out, reward_out, done_out, trunc_out);
}

int mgym_rollout(mgym_env* env, const void* actions, int32_t K, float* obs_out, float* reward_out, uint8_t* done_out,
                 uint8_t* trunc_out) {
    ENV_OR_FAIL(env);
    if (K < 0) return bad_arg("mgym_rollout: K < 0");
    if (!actions && e->n && K) return bad_arg("mgym_rollout: actions is NULL");
    if (e->cfg.kind == MGYM_LUNARLANDER) {
        // While the stream is being captured a LunarLander rollout records K steps, not its persistent launch: replays of a captured persistent
        // launch ended in an abort inside the HIP runtime (35 of 40 fresh processes; eager launches of the same kernel and captured steps do not),
        // unexplained at the end of round 4 — the per-environment results are the same words either way.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(e->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return e->Env::rollout(actions, K, obs_out, reward_out, done_out, trunc_out);
    }
    return e->rollout(actions, K, obs_out, reward_out, done_out, trunc_out);
}

int mgym_rollout_uniform(mgym_env* env, uint64_t policy_seed, int32_t K, void* actions_out, float* obs_out, float* reward_out,
                         uint8_t* done_out, uint8_t* trunc_out) {
    ENV_OR_FAIL(env);
    if (K < 0) return bad_arg("mgym_rollout_uniform: K < 0");
    if (e->cfg.env_id_base % 4 != 0) return bad_arg("mgym_rollout_uniform: env_id_base must be a multiple of 4 (policy bits are keyed per 4 consecutive global env ids)");
    return e->rollout_uniform(policy_seed, K, actions_out, obs_out, reward_out, done_out, trunc_out);
}

int mgym_rollout_linear(mgym_env* env, const float* policy, int32_t K, void* actions_out, float* obs_out, float* reward_out,
                        uint8_t* done_out, uint8_t* trunc_out) {
    ENV_OR_FAIL(env);
    if (K < 0) return bad_arg("mgym_rollout_linear: K < 0");
    if (!policy) return bad_arg("mgym_rollout_linear: policy is NULL");
    return e->rollout_linear(policy, K, actions_out, obs_out, reward_out, done_out, trunc_out);
}

int mgym_episode_count(mgym_env* env, uint64_t* finished) {
    ENV_OR_FAIL(env);
    if (!finished) return bad_arg("mgym_episode_count: NULL");
    MGYM_HIP(hipMemcpyAsync(e->h_done, e->d_done, kDoneShards * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    MGYM_HIP(hipStreamSynchronize(e->stream));
    uint64_t tot = 0;
    for (int i = 0; i < kDoneShards; ++i) tot += e->h_done[i];
    *finished = tot;
    return MGYM_OK;
}

int mgym_observation(mgym_env* env, const float** obs, uint64_t* col_stride) {
    ENV_OR_FAIL(env);
    if (!obs || !col_stride) return bad_arg("mgym_observation: NULL out pointer");
    return e->observation(obs, col_stride);
}

int mgym_observation_aos(mgym_env* env, float* out_aos) {
    ENV_OR_FAIL(env);
    if (!out_aos && e->n) return bad_arg("mgym_observation_aos: out is NULL");
    if (e->n == 0) return MGYM_OK;
    const float* obs = nullptr;
    uint64_t stride = 0;
    int st = e->observation(&obs, &stride);
    if (st != MGYM_OK) return st;
    hipLaunchKernelGGL(obs_to_aos_kernel, dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, e->stream, obs, stride, e->obs_dim, e->n, out_aos);
    MGYM_HIP(hipGetLastError());
    return MGYM_OK;
}

int mgym_get_state(mgym_env* env, void* blob) {
    ENV_OR_FAIL(env);
    if (!blob && e->n) return bad_arg("mgym_get_state: blob is NULL");
    return e->get_state(blob);
}

int mgym_set_state(mgym_env* env, const void* blob) {
    ENV_OR_FAIL(env);
    if (!blob && e->n) return bad_arg("mgym_set_state: blob is NULL");
    return e->set_state(blob);
}

int mgym_set_dispersion_override(mgym_env* env, const float* disp) {
    ENV_OR_FAIL(env);
    return e->set_dispersion(disp);
}

int mgym_sync(mgym_env* env) {
    ENV_OR_FAIL(env);
    MGYM_HIP(hipMemcpyAsync(e->h_err, e->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    MGYM_HIP(hipMemsetAsync(e->d_err, 0, sizeof(uint32_t), e->stream));
    MGYM_HIP(hipStreamSynchronize(e->stream));
    uint32_t bits = *e->h_err;
    if (bits & DEV_ERR_INVALID_ACTION) {
        set_last_error("invalid action: assertion `action_space.contains(&action)` failed for at least one environment");
        return MGYM_ERR_INVALID_ACTION;
    }
    if (bits & DEV_ERR_CONTACT_OVERFLOW) {
        set_last_error("LunarLander: contact cache overflow (more than 12 body/ground-edge pairs with overlapping broad-phase boxes in one "
                       "environment: a pair was dropped; that environment's results are undefined until it is reset)");
        return MGYM_ERR_CAPACITY;
    }
    if (bits & DEV_ERR_SOLVER_OVERFLOW) {
        set_last_error("LunarLander: island solver overflow (more than 9 touching contacts in one environment: the extra constraints were "
                       "not solved; that environment's results are undefined until it is reset)");
        return MGYM_ERR_CAPACITY;
    }
    if (bits & DEV_ERR_INTERNAL) {
        set_last_error("LunarLander: internal error — a finished environment had no fitting prepared reset while the engine believed every one had "
                       "(staged resets, lunar_lander.hip direct_possible); that environment was not reset.  Please report; MGYM_LL_STAGED_RESET=0 avoids the path");
        return MGYM_ERR_HIP;
    }
    if (bits & DEV_ERR_NOT_RESET) {
        set_last_error("You forgot to call reset()");
        return MGYM_ERR_NOT_RESET;
    }
    return MGYM_OK;
}

const char* mgym_last_error(void) { return g_last_error.c_str(); }

int mgym_get_info(mgym_env* env, char* buf, size_t cap) {
    ENV_OR_FAIL(env);
    if (!buf || cap == 0) return bad_arg("mgym_get_info: buf is NULL or cap is 0");
    static const char* const kinds[] = {"cartpole", "mountain_car", "mountain_car_continuous", "lunar_lander"};
    std::string out = std::string("kind=") + kinds[e->kind] + "\nn_envs=" + std::to_string(e->n) + "\ndevice=" + std::to_string(e->cfg.device) +
                      "\nauto_reset=" + ((e->cfg.flags & MGYM_FLAG_AUTO_RESET) ? "1" : "0") + "\nGPU_MAX_HW_QUEUES=" + e->hwq_env + "\n";
    int st = e->info(out);
    if (st != MGYM_OK) return st;
    if (out.size() + 1 > cap) { buf[0] = 0; return bad_arg("mgym_get_info: buffer too small"); }
    memcpy(buf, out.c_str(), out.size() + 1);
    return MGYM_OK;
}

int mgym_malloc(int device, size_t bytes, void** out) {
    if (!out) return bad_arg("mgym_malloc: NULL");
    DeviceGuard guard(device);
    if (!guard.ok) { set_last_error("mgym_malloc: no such HIP device"); return MGYM_ERR_NO_DEVICE; }
    MGYM_HIP(hipMalloc(out, bytes ? bytes : 16));
    return MGYM_OK;
}

int mgym_free(int device, void* ptr) {
    DeviceGuard guard(device);
    MGYM_HIP(hipFree(ptr));
    return MGYM_OK;
}

// The engine's kernels run on a hipStreamNonBlocking stream, which the null stream does not order against: these
// helpers therefore return only when the copy has fully completed (hipMemcpy blocks the host; the explicit null-stream
// synchronise covers staged pageable-memory copies), so a launch issued afterwards on any stream sees the data.
int mgym_memcpy_h2d(int device, void* dst, const void* src, size_t bytes) {
    DeviceGuard guard(device);
    if (bytes) {
        MGYM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        MGYM_HIP(hipStreamSynchronize(nullptr));
    }
    return MGYM_OK;
}

int mgym_memcpy_d2h(int device, void* dst, const void* src, size_t bytes) {
    DeviceGuard guard(device);
    if (bytes) MGYM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return MGYM_OK;
}

int mgym_timer_start(mgym_env* env) {
    ENV_OR_FAIL(env);
    MGYM_HIP(hipEventRecord(e->ev0, e->stream));
    return MGYM_OK;
}

int mgym_timer_stop(mgym_env* env, float* elapsed_ms) {
    ENV_OR_FAIL(env);
    if (!elapsed_ms) return bad_arg("mgym_timer_stop: NULL");
    MGYM_HIP(hipEventRecord(e->ev1, e->stream));
    MGYM_HIP(hipEventSynchronize(e->ev1));
    MGYM_HIP(hipEventElapsedTime(elapsed_ms, e->ev0, e->ev1));
    return MGYM_OK;
}

int mgym_graph_begin(mgym_env* env) {
    ENV_OR_FAIL(env);
    { int st = e->join_helpers(); if (st != MGYM_OK) return st; }  // helper-stream work issued so far is waited for OUTSIDE the capture
    MGYM_HIP(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    return MGYM_OK;
}

// What mgym_graph_end hands out: the executable graph and the handle configuration it was captured under.  Kernel arguments are
// frozen at capture — among them the dispersion source of mgym_set_dispersion_override — so a graph captured under another
// configuration is refused at launch (loudly) instead of silently stepping with the old one.
struct GraphBox { hipGraphExec_t exec; uint64_t config_epoch; };

int mgym_graph_end(mgym_env* env, void** graph_exec_out) {
    ENV_OR_FAIL(env);
    if (!graph_exec_out) return bad_arg("mgym_graph_end: NULL");
    hipGraph_t g = nullptr;
    { int st = e->join_helpers(); if (st != MGYM_OK) return st; }
    MGYM_HIP(hipStreamEndCapture(e->stream, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t err = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (err != hipSuccess) return hip_fail(err, "hipGraphInstantiate", __FILE__, __LINE__);
    *graph_exec_out = new GraphBox{ge, e->config_epoch};
    return MGYM_OK;
}

int mgym_graph_launch(mgym_env* env, void* graph_exec) {
    ENV_OR_FAIL(env);
    if (!graph_exec) return bad_arg("mgym_graph_launch: NULL");
    GraphBox* b = static_cast<GraphBox*>(graph_exec);
    if (b->config_epoch != e->config_epoch)
        return bad_arg("mgym_graph_launch: the handle's configuration has changed since this graph was captured (mgym_set_dispersion_override): capture it again");
    MGYM_HIP(hipGraphLaunch(b->exec, e->stream));
    return MGYM_OK;
}

int mgym_graph_destroy(void* graph_exec) {
    if (graph_exec) {
        GraphBox* b = static_cast<GraphBox*>(graph_exec);
        hipError_t err = hipGraphExecDestroy(b->exec);
        delete b;
        MGYM_HIP(err);
    }
    return MGYM_OK;
}

}  // extern "C"
