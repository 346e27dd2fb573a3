// mgym.hpp — C++ host-side mirror of the reference's environment interface, over the C ABI (mgym.h).
//
// The reference (ModuRL/ModuRL_Gym) is a Rust crate; its toolchain is absent from the build image, so
// the host side above the C ABI is C++ (the reference is compiled code).  Names, argument meaning and
// error behaviour follow the reference's `impl Gym for ...` blocks:
//     CartPoleV1::builder().sutton_barto_reward(b).is_euler(b).build()     cartpole.rs:34-44
//     MountainCarV0::builder().goal_velocity(v).build()                   mountain_car.rs:25-34
//     LunarLanderV3::builder().gravity(g).enable_wind(b).wind_power(p)
//                             .turbulence_power(p).seed(s).build()        lunar_lander.rs:278-291
//     reset() -> state;  step(action) -> StepInfo{state, reward, done, truncated}
// Where the reference panics (assert!), these throw: std::invalid_argument for an invalid action
// (cartpole.rs:252, mountain_car.rs:294) and bad gravity (lunar_lander.rs:292-296), std::logic_error
// for step-before-reset (lunar_lander.rs:920).  `VecGym` is the batched superset (device pointers).
// Header-only; link with libmgym.so.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "mgym.h"

namespace mgym_host {

struct StepInfo {  // modurl::gym::StepInfo
    std::vector<float> state;
    float reward;
    bool done;
    bool truncated;
};

inline void check(int st) {
    if (st == MGYM_OK) return;
    std::string msg = mgym_last_error();
    switch (st) {
    case MGYM_ERR_INVALID_ACTION: throw std::invalid_argument("invalid action: " + msg);
    case MGYM_ERR_BAD_CONFIG: throw std::invalid_argument(msg);
    case MGYM_ERR_NOT_RESET: throw std::logic_error(msg);
    default: throw std::runtime_error("mgym status " + std::to_string(st) + ": " + msg);
    }
}

// n_envs environments of one family on one GPU; buffers are device pointers owned by the caller.
class VecGym {
  public:
    explicit VecGym(const mgym_config& cfg) : cfg_(cfg) {
        check(mgym_create(&cfg_, &env_));
        mgym_spec s;
        check(mgym_get_spec(cfg_.kind, &s));
        obs_dim_ = s.obs_dim;
    }
    ~VecGym() { mgym_destroy(env_); }
    VecGym(const VecGym&) = delete;
    VecGym& operator=(const VecGym&) = delete;

    void reset(const uint8_t* mask = nullptr, float* obs_out = nullptr) { check(mgym_reset(env_, mask, obs_out)); }
    void reset_done(const uint8_t* done, const uint8_t* trunc, float* obs_out = nullptr) { check(mgym_reset_done(env_, done, trunc, obs_out)); }
    void step(const void* actions, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) {
        check(mgym_step(env_, actions, obs_out, reward, done, trunc));
    }
    void sync() { check(mgym_sync(env_)); }
    // K fused steps under the on-device uniform random policy (CartPole); actions_out may be null
    void rollout_linear(const float* policy, int K, void* actions_out, float* obs, float* reward, uint8_t* done, uint8_t* trunc) {
        check(mgym_rollout_linear(env_, policy, K, actions_out, obs, reward, done, trunc));
    }
    void rollout_uniform(uint64_t policy_seed, int K, void* actions_out, float* obs, float* reward, uint8_t* done, uint8_t* trunc) {
        check(mgym_rollout_uniform(env_, policy_seed, K, actions_out, obs, reward, done, trunc));
    }
    // env-steps that returned done or truncated since creation (synchronises)
    uint64_t episode_count() {
        uint64_t c = 0;
        check(mgym_episode_count(env_, &c));
        return c;
    }
    const float* observation(uint64_t* col_stride) {
        const float* p = nullptr;
        check(mgym_observation(env_, &p, col_stride));
        return p;
    }
    mgym_env* handle() { return env_; }
    int obs_dim() const { return obs_dim_; }
    uint64_t n_envs() const { return cfg_.n_envs; }
    int device() const { return cfg_.device; }

  private:
    mgym_config cfg_;
    mgym_env* env_ = nullptr;
    int obs_dim_ = 0;
};

// single environment with the reference's scalar signature (n_envs = 1)
class SingleGym {
  public:
    // owns five device allocations and a handle: not copyable (a copy would free them twice), movable
    SingleGym(const SingleGym&) = delete;
    SingleGym& operator=(const SingleGym&) = delete;
    SingleGym(SingleGym&& o) noexcept
        : vec_(o.vec_), d_act_(o.d_act_), d_obs_(o.d_obs_), d_rew_(o.d_rew_), d_done_(o.d_done_), d_trunc_(o.d_trunc_) {
        o.vec_ = nullptr; o.d_act_ = nullptr; o.d_obs_ = nullptr; o.d_rew_ = nullptr; o.d_done_ = nullptr; o.d_trunc_ = nullptr;
    }
    SingleGym& operator=(SingleGym&&) = delete;
    std::vector<float> reset() {
        vec_->reset(nullptr, d_obs_);
        vec_->sync();
        return fetch_obs();
    }
    StepInfo step(uint32_t action) { return step_bits(action); }

  protected:
    explicit SingleGym(const mgym_config& cfg) {
        vec_ = new VecGym(cfg);
        const int dev = cfg.device;
        void* p = nullptr;
        check(mgym_malloc(dev, 64, &p)); d_act_ = p;
        check(mgym_malloc(dev, sizeof(float) * 8, &p)); d_obs_ = static_cast<float*>(p);
        check(mgym_malloc(dev, sizeof(float) * 4, &p)); d_rew_ = static_cast<float*>(p);
        check(mgym_malloc(dev, 16, &p)); d_done_ = static_cast<uint8_t*>(p);
        check(mgym_malloc(dev, 16, &p)); d_trunc_ = static_cast<uint8_t*>(p);
    }
    ~SingleGym() {
        if (!vec_) return;  // moved from
        const int dev = vec_->device();
        mgym_free(dev, d_act_); mgym_free(dev, d_obs_); mgym_free(dev, d_rew_); mgym_free(dev, d_done_); mgym_free(dev, d_trunc_);
        delete vec_;
    }
    StepInfo step_bits(uint32_t bits) {
        const int dev = vec_->device();
        check(mgym_memcpy_h2d(dev, d_act_, &bits, 4));
        vec_->step(d_act_, d_obs_, d_rew_, d_done_, d_trunc_);
        vec_->sync();
        StepInfo si;
        si.state = fetch_obs();
        uint8_t d = 0, t = 0;
        check(mgym_memcpy_d2h(dev, &si.reward, d_rew_, 4));
        check(mgym_memcpy_d2h(dev, &d, d_done_, 1));
        check(mgym_memcpy_d2h(dev, &t, d_trunc_, 1));
        si.done = d != 0; si.truncated = t != 0;
        return si;
    }
    std::vector<float> fetch_obs() {
        std::vector<float> o(vec_->obs_dim());
        check(mgym_memcpy_d2h(vec_->device(), o.data(), d_obs_, sizeof(float) * o.size()));
        return o;
    }
    static mgym_config defaults(int kind, int device, uint64_t seed) {
        mgym_config c;
        check(mgym_default_config(kind, &c));
        c.device = device; c.seed = seed; c.n_envs = 1;
        return c;
    }
    VecGym* vec_ = nullptr;
    void* d_act_ = nullptr;
    float *d_obs_ = nullptr, *d_rew_ = nullptr;
    uint8_t *d_done_ = nullptr, *d_trunc_ = nullptr;
};

class CartPoleV1 : public SingleGym {  // cartpole.rs:13-357
  public:
    struct Builder {
        int device_ = 0; bool sutton_barto_reward_ = false; bool is_euler_ = true; uint64_t seed_ = 0;
        Builder& device(int d) { device_ = d; return *this; }
        Builder& sutton_barto_reward(bool b) { sutton_barto_reward_ = b; return *this; }
        Builder& is_euler(bool b) { is_euler_ = b; return *this; }
        Builder& seed(uint64_t s) { seed_ = s; return *this; }
        CartPoleV1* build() { return new CartPoleV1(*this); }
    };
    static Builder builder() { return Builder(); }

  private:
    static mgym_config cfg(const Builder& b) {
        mgym_config c = defaults(MGYM_CARTPOLE, b.device_, b.seed_);
        c.sutton_barto_reward = b.sutton_barto_reward_; c.is_euler = b.is_euler_;
        return c;
    }
    explicit CartPoleV1(const Builder& b) : SingleGym(cfg(b)) {}
};

class MountainCarV0 : public SingleGym {  // mountain_car.rs:10-339
  public:
    struct Builder {
        int device_ = 0; float goal_velocity_ = 0.0f; uint64_t seed_ = 0;
        Builder& device(int d) { device_ = d; return *this; }
        Builder& goal_velocity(float v) { goal_velocity_ = v; return *this; }
        Builder& seed(uint64_t s) { seed_ = s; return *this; }
        MountainCarV0* build() { return new MountainCarV0(*this); }
    };
    static Builder builder() { return Builder(); }

  private:
    static mgym_config cfg(const Builder& b) {
        mgym_config c = defaults(MGYM_MOUNTAINCAR, b.device_, b.seed_);
        c.goal_velocity = b.goal_velocity_;
        return c;
    }
    explicit MountainCarV0(const Builder& b) : SingleGym(cfg(b)) {}
};

class LunarLanderV3 : public SingleGym {  // lunar_lander.rs:232-1201
  public:
    struct Builder {
        int device_ = 0; float gravity_ = -10.0f; bool enable_wind_ = false; float wind_power_ = 15.0f, turbulence_power_ = 1.5f;
        uint64_t seed_ = 0;
        Builder& device(int d) { device_ = d; return *this; }
        Builder& gravity(float g) { gravity_ = g; return *this; }
        Builder& enable_wind(bool b) { enable_wind_ = b; return *this; }
        Builder& wind_power(float p) { wind_power_ = p; return *this; }
        Builder& turbulence_power(float p) { turbulence_power_ = p; return *this; }
        Builder& seed(uint64_t s) { seed_ = s; return *this; }
        LunarLanderV3* build() { return new LunarLanderV3(*this); }
    };
    static Builder builder() { return Builder(); }

  private:
    static mgym_config cfg(const Builder& b) {
        mgym_config c = defaults(MGYM_LUNARLANDER, b.device_, b.seed_);
        c.gravity = b.gravity_; c.enable_wind = b.enable_wind_; c.wind_power = b.wind_power_; c.turbulence_power = b.turbulence_power_;
        return c;
    }
    explicit LunarLanderV3(const Builder& b) : SingleGym(cfg(b)) {}
};

}  // namespace mgym_host
