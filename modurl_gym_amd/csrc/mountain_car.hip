// mountain_car.hip — MountainCarV0 (discrete) and MountainCarContinuous-v0 step()/reset()
// as struct-of-arrays gfx950 kernels.
//
// MountainCarV0 replaces `impl Gym for MountainCarV0` (reference
// src/classic_control/mountain_car.rs: reset :279-291, step :293-330).
// MountainCarContinuous-v0 is NOT in the reference (BASELINE config 3 asks for it):
// gymnasium semantics, parity unpinned; it shares everything but the action decoding,
// goal position and reward.
//
// Data layout in HBM (engine-owned): position[n] velocity[n] f32 (the observation),
// episode[n] u32 (touched only by reset).  One lane owns 4 consecutive environments
// (16-byte accesses).  Algorithmic bytes per env-step: 26 (state 8 R + 8 W, action 4 R,
// reward 4 W, done 1 W, truncated 1 W).
#include "common.h"
#include "mgym_math.h"
#include "philox.h"

namespace mgym {

struct MountainCarParams {
    float min_position, max_position, max_speed, goal_position, goal_velocity;
    float force, gravity;  // discrete: 0.001, 0.0025; continuous: power 0.0015 in `force`
};

struct MountainCarDev {
    float *pos, *vel;
    uint32_t* episode;
    uint64_t n, seed, env_id_base;
    uint32_t* err;
    unsigned long long* done_count;  // kDoneShards partial counts of finished env-steps (mgym_episode_count)
    MountainCarParams p;
    int auto_reset;
};

// done-mask reduction: per-lane counts of finished envs summed over the wave, one fire-and-forget atomic per wave
__device__ __forceinline__ void mountaincar_flush_counts(const MountainCarDev& d, uint32_t finished, bool bad) {
    for (int o = 32; o > 0; o >>= 1) finished += __shfl_down(finished, o);
    if ((threadIdx.x & 63) == 0 && finished)
        atomicAdd(d.done_count + ((blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) & (kDoneShards - 1)), (unsigned long long)finished);
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_INVALID_ACTION);
    }
}

typedef float mc_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float* p, float4 v) {
    mc_f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<mc_f32x4*>(p));
}

// f32::clamp: if self < min {min} else if self > max {max} else {self}
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// reset(): mountain_car.rs:279-291 — position ~ U[-0.6,-0.4) sampled f64 -> f32, velocity 0
__device__ __forceinline__ void mountaincar_reset_one(const MountainCarDev& d, uint64_t i, float& pos, float& vel) {
    uint32_t ep = d.episode[i];
    Philox4 a = env_draw(d.seed, d.env_id_base + i, ep, SLOT_RESET0);
    pos = uniform_f64_to_f32(-0.6, -0.4, a.w[0], a.w[1]);
    vel = 0.0f;
    d.episode[i] = ep + 1u;
}

template <bool CONT>
__device__ __forceinline__ void mountaincar_step_one(const MountainCarParams& p, float& position, float& velocity,
                                                     uint32_t action_bits, float& reward, uint32_t& done, bool& bad) {
    float push;  // the action-dependent velocity increment
    float a_f = 0.0f;
    if (CONT) {
        a_f = as_f32(action_bits);
        if (a_f != a_f) { bad = true; reward = 0.0f; done = 0u; return; }
        float force = clampf(a_f, -1.0f, 1.0f);
        push = force * p.force;
    } else {
        if (action_bits >= 3u) { bad = true; reward = 0.0f; done = 0u; return; }  // :294
        push = ((float)action_bits - 1.0f) * p.force;                               // :301
    }
    velocity += push + mg_cosf_u(3.0f * position) * (-p.gravity);                     // :301-302
    velocity = clampf(velocity, -p.max_speed, p.max_speed);                         // :304
    position += velocity;                                                           // :306
    position = clampf(position, p.min_position, p.max_position);                    // :308
    if (position == p.min_position && velocity < 0.0f) velocity = 0.0f;             // :311-313
    bool terminated = position >= p.goal_position && velocity >= p.goal_velocity;   // :318
    done = terminated ? 1u : 0u;
    if (CONT) {
        float r = 0.0f;
        if (terminated) r = 100.0f;
        r -= a_f * a_f * 0.1f;
        reward = r;
    } else {
        reward = -1.0f;  // :319
    }
}

template <int VEC, bool CONT>
__global__ void __launch_bounds__(kBlock)
mountaincar_step_kernel(MountainCarDev d, const uint32_t* __restrict__ act, float* __restrict__ obs_out,
                        float* __restrict__ rew, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    const uint64_t groups = (d.n + VEC - 1) / VEC;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    uint32_t finished = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const uint64_t i0 = g * VEC;
        float ps[VEC], vs[VEC], r[VEC];
        uint32_t a[VEC], dn[VEC];
        const bool full = (VEC == 1) || (i0 + VEC <= d.n);
        if (VEC == 4 && full) {
            float4 vp = *reinterpret_cast<const float4*>(d.pos + i0);
            float4 vv = *reinterpret_cast<const float4*>(d.vel + i0);
            uint4 va = *reinterpret_cast<const uint4*>(act + i0);
            ps[0] = vp.x; ps[1 % VEC] = vp.y; ps[2 % VEC] = vp.z; ps[3 % VEC] = vp.w;
            vs[0] = vv.x; vs[1 % VEC] = vv.y; vs[2 % VEC] = vv.z; vs[3 % VEC] = vv.w;
            a[0] = va.x; a[1 % VEC] = va.y; a[2 % VEC] = va.z; a[3 % VEC] = va.w;
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                uint64_t i = i0 + k;
                bool in = i < d.n;
                ps[k] = in ? d.pos[i] : 0.0f;
                vs[k] = in ? d.vel[i] : 0.0f;
                a[k] = in ? act[i] : (CONT ? 0u : 1u);
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            mountaincar_step_one<CONT>(d.p, ps[k], vs[k], a[k], r[k], dn[k], bad);
            finished += (dn[k] && (i0 + k < d.n)) ? 1u : 0u;
            if (d.auto_reset && dn[k] && (i0 + k < d.n)) mountaincar_reset_one(d, i0 + k, ps[k], vs[k]);
        }
        if (VEC == 4 && full) {
            float4 op = make_float4(ps[0], ps[1 % VEC], ps[2 % VEC], ps[3 % VEC]);
            float4 ov = make_float4(vs[0], vs[1 % VEC], vs[2 % VEC], vs[3 % VEC]);
            // non-temporal output stores (measured on the CartPole kernel: ~5 % at 1 Mi envs)
            nt_store4(d.pos + i0, op);
            nt_store4(d.vel + i0, ov);
            if (obs_out) {
                nt_store4(obs_out + i0, op);
                nt_store4(obs_out + d.n + i0, ov);
            }
            if (rew) nt_store4(rew + i0, make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]));
            if (done_out)
                __builtin_nontemporal_store(dn[0] | (dn[1 % VEC] << 8) | (dn[2 % VEC] << 16) | (dn[3 % VEC] << 24),
                                            reinterpret_cast<uint32_t*>(done_out + i0));
            if (trunc_out) __builtin_nontemporal_store(0u, reinterpret_cast<uint32_t*>(trunc_out + i0));  // :328 truncated: false, always
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                uint64_t i = i0 + k;
                if (i < d.n) {
                    d.pos[i] = ps[k]; d.vel[i] = vs[k];
                    if (obs_out) { obs_out[i] = ps[k]; obs_out[d.n + i] = vs[k]; }
                    if (rew) rew[i] = r[k];
                    if (done_out) done_out[i] = (uint8_t)dn[k];
                    if (trunc_out) trunc_out[i] = 0;
                }
            }
        }
    }
    mountaincar_flush_counts(d, finished, bad);
}

// The 16-byte path (caller buffers aligned, n % 4 == 0): one lane owns 4 consecutive environments; block-uniform passes;
// engine columns through one buffer resource (descriptor and column / pass offsets in SGPRs, one constant VGPR lane
// offset: no per-lane address arithmetic; n <= 2^27 so every byte offset fits 32 bits); the columns are padded to n_pad
// (a multiple of the block's 1024 envs), so every lane may load them — only in-range lanes store.  The done mask is
// reduced with __ballot + popcount for mgym_episode_count; resets are rare here (episodes last >= 100 steps) and stay
// in place.
typedef uint32_t mc_u32x4v __attribute__((ext_vector_type(4)));
template <bool CONT, bool RESET>
__global__ void __launch_bounds__(kBlock)
mountaincar_step4_kernel(MountainCarDev d, const uint32_t* __restrict__ act, float* __restrict__ obs_out,
                         float* __restrict__ rew, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    constexpr uint64_t kPerBlock = (uint64_t)kBlock * 4;
    const uint32_t lane_env = threadIdx.x * 4u, off4 = lane_env * 4u;
    const uint32_t colb = (uint32_t)(d.vel - d.pos) * 4u;  // column stride in bytes
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d.pos, 0, 0xffffffff, 0x00020000);
    bool bad = false;
    uint32_t finished = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * kPerBlock; base < d.n; base += (uint64_t)gridDim.x * kPerBlock) {  // block-uniform
        const uint64_t left = d.n - base;
        const uint32_t rem = left < kPerBlock ? (uint32_t)left : (uint32_t)kPerBlock;
        const bool in = lane_env < rem;
        const uint32_t sb = (uint32_t)base * 4u;
        const mc_u32x4v vp = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb, 0);
        const mc_u32x4v vv = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + colb, 0);
        uint4 va = CONT ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(1u, 1u, 1u, 1u);
        if (in) va = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(act + base) + off4);
        float ps[4] = {as_f32(vp.x), as_f32(vp.y), as_f32(vp.z), as_f32(vp.w)}, vs[4] = {as_f32(vv.x), as_f32(vv.y), as_f32(vv.z), as_f32(vv.w)}, r[4];
        uint32_t a[4] = {va.x, va.y, va.z, va.w}, dn[4];
        uint32_t any_done = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mountaincar_step_one<CONT>(d.p, ps[k], vs[k], a[k], r[k], dn[k], bad);
            any_done |= dn[k];
        }
        const unsigned long long m = __ballot(in && any_done);
        if (m) {  // rare: count and (RESET) restart the finished envs in place
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                finished += (uint32_t)__popcll(__ballot(in && dn[k]));
                if (RESET && in && dn[k]) mountaincar_reset_one(d, base + lane_env + k, ps[k], vs[k]);
            }
        }
        if (in) {
            __builtin_amdgcn_raw_buffer_store_b128(mc_u32x4v{as_u32(ps[0]), as_u32(ps[1]), as_u32(ps[2]), as_u32(ps[3])}, rs, off4, sb, 2);
            __builtin_amdgcn_raw_buffer_store_b128(mc_u32x4v{as_u32(vs[0]), as_u32(vs[1]), as_u32(vs[2]), as_u32(vs[3])}, rs, off4, sb + colb, 2);
            if (obs_out) {
                float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(obs_out + base) + off4);
                nt_store4(o, make_float4(ps[0], ps[1], ps[2], ps[3]));
                nt_store4(o + d.n, make_float4(vs[0], vs[1], vs[2], vs[3]));
            }
            if (rew) nt_store4(reinterpret_cast<float*>(reinterpret_cast<char*>(rew + base) + off4), make_float4(r[0], r[1], r[2], r[3]));
            if (done_out)
                __builtin_nontemporal_store(dn[0] | (dn[1] << 8) | (dn[2] << 16) | (dn[3] << 24), reinterpret_cast<uint32_t*>(done_out + base + lane_env));
            if (trunc_out) __builtin_nontemporal_store(0u, reinterpret_cast<uint32_t*>(trunc_out + base + lane_env));  // :328 truncated: false, always
        }
    }
    if ((threadIdx.x & 63) == 0 && finished)
        atomicAdd(d.done_count + ((blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) & (kDoneShards - 1)), (unsigned long long)finished);
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_INVALID_ACTION);
    }
}

// Fused K-step rollout (mgym_rollout): 4 environments per lane stay in registers for K steps; per step one 16-B
// action word in, reward / done / truncated (and optionally the observation) out.  Same per-env function and the
// same in-place auto-reset as mountaincar_step_kernel<4, CONT>, hence bit-identical to K separate steps.
typedef uint32_t mc_u32x4 __attribute__((ext_vector_type(4)));
// mgym_rollout_linear: the action comes from the observation the env holds before the step instead of a table.  Discrete(3): the index of the largest of
// the three scores s_j = (w[j][0] * position + w[j][1] * velocity) + b[j] (the first of equal ones); Box(-1, 1): the one score itself as the force (the
// step clamps it like any action).  f32, in that order, nothing fused.
struct McLinearPolicy { float w[3][2], b[3]; };
// mgym_rollout_uniform: the lane's four envs (global ids g .. g + 3, g a multiple of 4) share one Philox stream keyed by the policy seed, counter =
// (g, number of earlier mgym_rollout_uniform calls on this handle, SLOT_POLICY + block).  Discrete(3): step t takes, for env g + k, half (t % 2) of word k
// of block t / 2 (low half first), h, and the action (3 h) >> 16 — each action within 2^-16 of 1/3.  Box(-1, 1): word k of block t, top 24 bits u:
// the force u 2^-23 - 1, uniform over the multiples of 2^-23 in [-1, 1).
enum { MC_ROLL_TABLE = 0, MC_ROLL_LINEAR = 1, MC_ROLL_UNIFORM = 2 };
struct McUniformPolicy { uint64_t seed; uint32_t call; };
template <bool CONT, int POLICY = MC_ROLL_TABLE>
__global__ void __launch_bounds__(kBlock)
mountaincar_rollout_kernel(MountainCarDev d, const uint32_t* __restrict__ act, int K, float* __restrict__ obs_out, float* __restrict__ rew,
                           uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out, McLinearPolicy lin = McLinearPolicy{}, uint32_t* __restrict__ act_out = nullptr,
                           McUniformPolicy uni = McUniformPolicy{}) {
    constexpr bool LINEAR = POLICY == MC_ROLL_LINEAR;
    const uint64_t groups = d.n / 4;  // n % 4 == 0 is required by the host wrapper
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    uint32_t finished = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const uint64_t i0 = g * 4;
        float4 vp = *reinterpret_cast<const float4*>(d.pos + i0), vv = *reinterpret_cast<const float4*>(d.vel + i0);
        float ps[4] = {vp.x, vp.y, vp.z, vp.w}, vs[4] = {vv.x, vv.y, vv.z, vv.w};
        Philox4 bits{};
        for (int t = 0; t < K; ++t) {
            const uint64_t off = (uint64_t)t * d.n + i0;
            uint32_t a[4], dn[4];
            if constexpr (POLICY == MC_ROLL_UNIFORM) {
                const uint64_t gid = d.env_id_base + i0;
                if (CONT || (t & 1) == 0)
                    bits = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), uni.call, SLOT_POLICY + (uint32_t)(CONT ? t : t >> 1), (uint32_t)uni.seed, (uint32_t)(uni.seed >> 32));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (CONT) a[k] = __float_as_uint((float)(bits.w[k] >> 8) * 1.1920928955078125e-07f - 1.0f);
                    else a[k] = ((((t & 1) ? bits.w[k] >> 16 : bits.w[k] & 0xffffu)) * 3u) >> 16;
                }
                if (act_out) __builtin_nontemporal_store(mc_u32x4{a[0], a[1], a[2], a[3]}, reinterpret_cast<mc_u32x4*>(act_out + off));
            } else if constexpr (LINEAR) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float s0 = (lin.w[0][0] * ps[k] + lin.w[0][1] * vs[k]) + lin.b[0];
                    if constexpr (CONT) a[k] = __float_as_uint(s0);
                    else {
                        const float s1 = (lin.w[1][0] * ps[k] + lin.w[1][1] * vs[k]) + lin.b[1], s2 = (lin.w[2][0] * ps[k] + lin.w[2][1] * vs[k]) + lin.b[2];
                        uint32_t best = s1 > s0 ? 1u : 0u;
                        const float sb = s1 > s0 ? s1 : s0;
                        a[k] = s2 > sb ? 2u : best;
                    }
                }
                if (act_out) __builtin_nontemporal_store(mc_u32x4{a[0], a[1], a[2], a[3]}, reinterpret_cast<mc_u32x4*>(act_out + off));
            } else {
                mc_u32x4 va = __builtin_nontemporal_load(reinterpret_cast<const mc_u32x4*>(act + off));
                a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w;
            }
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mountaincar_step_one<CONT>(d.p, ps[k], vs[k], a[k], r[k], dn[k], bad);
                finished += dn[k];
                if (d.auto_reset && dn[k]) mountaincar_reset_one(d, i0 + k, ps[k], vs[k]);
            }
            if (rew) nt_store4(rew + off, make_float4(r[0], r[1], r[2], r[3]));
            if (done_out) __builtin_nontemporal_store(dn[0] | (dn[1] << 8) | (dn[2] << 16) | (dn[3] << 24), reinterpret_cast<uint32_t*>(done_out + off));
            if (trunc_out) __builtin_nontemporal_store(0u, reinterpret_cast<uint32_t*>(trunc_out + off));
            if (obs_out) {
                float* o = obs_out + (uint64_t)t * 2 * d.n + i0;
                nt_store4(o, make_float4(ps[0], ps[1], ps[2], ps[3]));
                nt_store4(o + d.n, make_float4(vs[0], vs[1], vs[2], vs[3]));
            }
        }
        nt_store4(d.pos + i0, make_float4(ps[0], ps[1], ps[2], ps[3]));
        nt_store4(d.vel + i0, make_float4(vs[0], vs[1], vs[2], vs[3]));
    }
    mountaincar_flush_counts(d, finished, bad);
}

__global__ void __launch_bounds__(kBlock)
mountaincar_reset_kernel(MountainCarDev d, const uint8_t* __restrict__ m0, const uint8_t* __restrict__ m1, int all,
                         int masks_aligned, float* __restrict__ obs_out) {
    const uint64_t groups = (d.n + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const uint64_t i0 = g * 4;
        uint32_t m = 0;
        if (all) {
            m = 0x01010101u;
        } else if (masks_aligned && i0 + 4 <= d.n) {
            if (m0) m |= *reinterpret_cast<const uint32_t*>(m0 + i0);
            if (m1) m |= *reinterpret_cast<const uint32_t*>(m1 + i0);
        } else {
            for (int k = 0; k < 4 && i0 + k < d.n; ++k) {
                uint32_t b = 0;
                if (m0) b |= m0[i0 + k];
                if (m1) b |= m1[i0 + k];
                m |= (b ? 1u : 0u) << (8 * k);
            }
        }
        if (m == 0) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint64_t i = i0 + k;
            if (i < d.n && ((m >> (8 * k)) & 0xffu)) {
                float p, v;
                mountaincar_reset_one(d, i, p, v);
                d.pos[i] = p; d.vel[i] = v;
                if (obs_out) { obs_out[i] = p; obs_out[d.n + i] = v; }
            }
        }
    }
}

__global__ void mountaincar_export_kernel(MountainCarDev d, uint32_t* __restrict__ blob) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    blob[i] = as_u32(d.pos[i]);
    blob[d.n + i] = as_u32(d.vel[i]);
    blob[2 * d.n + i] = d.episode[i];
}

__global__ void mountaincar_import_kernel(MountainCarDev d, const uint32_t* __restrict__ blob) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    d.pos[i] = as_f32(blob[i]);
    d.vel[i] = as_f32(blob[d.n + i]);
    d.episode[i] = blob[2 * d.n + i];
}

struct MountainCarEnv final : Env {
    bool continuous;
    void* base = nullptr;
    MountainCarDev dev{};

    explicit MountainCarEnv(bool c) : continuous(c) {}
    ~MountainCarEnv() override {
        if (base) (void)hipFree(base);
    }

    bool use_step4 = !getenv("MGYM_MC_OLD_KERNEL");  // A/B knob: the per-thread grid-stride kernel of round 1
    int init() override {
        obs_dim = 2;
        state_cols = 3;
        MGYM_HIP(hipMalloc(&base, 3 * n_pad * sizeof(float)));
        MGYM_HIP(hipMemsetAsync(base, 0, 3 * n_pad * sizeof(float), stream));  // mountain_car.rs:51 zeros
        float* f = static_cast<float*>(base);
        dev.pos = f; dev.vel = f + n_pad;
        dev.episode = reinterpret_cast<uint32_t*>(f + 2 * n_pad);
        dev.n = n; dev.seed = cfg.seed; dev.env_id_base = cfg.env_id_base; dev.err = d_err; dev.done_count = d_done;
        dev.auto_reset = (cfg.flags & MGYM_FLAG_AUTO_RESET) ? 1 : 0;
        MountainCarParams& p = dev.p;  // mountain_car.rs:35-40
        p.min_position = -1.2f; p.max_position = 0.6f; p.max_speed = 0.07f;
        p.goal_position = continuous ? 0.45f : 0.5f;
        p.goal_velocity = cfg.goal_velocity;
        p.force = continuous ? 0.0015f : 0.001f;
        p.gravity = 0.0025f;
        return MGYM_OK;
    }

    int reset(const uint8_t* m0, const uint8_t* m1, bool all, float* obs_out) override {
        if (n == 0) return MGYM_OK;
        int al = aligned(m0, 4) && aligned(m1, 4);
        hipLaunchKernelGGL(mountaincar_reset_kernel, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, stream, dev, m0, m1,
                           all ? 1 : 0, al, obs_out);
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int reset_deterministic(float* obs_out) override {  // mountain_car.rs:403-408: state = zeros (no reset())
        if (n == 0) return MGYM_OK;
        MGYM_HIP(hipMemsetAsync(dev.pos, 0, 2 * n_pad * sizeof(float), stream));
        if (obs_out) MGYM_HIP(hipMemsetAsync(obs_out, 0, 2 * n * sizeof(float), stream));
        return MGYM_OK;
    }

    int step(const void* actions, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0) return MGYM_OK;
        const uint32_t* act = static_cast<const uint32_t*>(actions);
        bool vec_ok = aligned(act, 16) && aligned(reward, 16) && aligned(done, 4) && aligned(trunc, 4) &&
                      (obs_out == nullptr || (aligned(obs_out, 16) && n % 4 == 0));
        dim3 gv(grid_for((n + 3) / 4)), gs(grid_for(n)), b(kBlock);
        if (vec_ok && n % 4 == 0 && n <= (1ull << 27) && use_step4) {  // the structured 16-byte kernel
            const bool rs = dev.auto_reset;
            if (continuous) { if (rs) hipLaunchKernelGGL((mountaincar_step4_kernel<true, true>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc);
                              else hipLaunchKernelGGL((mountaincar_step4_kernel<true, false>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc); }
            else { if (rs) hipLaunchKernelGGL((mountaincar_step4_kernel<false, true>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc);
                   else hipLaunchKernelGGL((mountaincar_step4_kernel<false, false>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc); }
            MGYM_HIP(hipGetLastError());
            return MGYM_OK;
        }
        if (continuous) {
            if (vec_ok) hipLaunchKernelGGL((mountaincar_step_kernel<4, true>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc);
            else hipLaunchKernelGGL((mountaincar_step_kernel<1, true>), gs, b, 0, stream, dev, act, obs_out, reward, done, trunc);
        } else {
            if (vec_ok) hipLaunchKernelGGL((mountaincar_step_kernel<4, false>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc);
            else hipLaunchKernelGGL((mountaincar_step_kernel<1, false>), gs, b, 0, stream, dev, act, obs_out, reward, done, trunc);
        }
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int rollout(const void* actions, int K, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0 || K == 0) return MGYM_OK;
        const uint32_t* act = static_cast<const uint32_t*>(actions);
        bool vec_ok = n % 4 == 0 && aligned(act, 16) && aligned(reward, 16) && aligned(done, 4) && aligned(trunc, 4) && aligned(obs_out, 16);
        if (!vec_ok) return Env::rollout(actions, K, obs_out, reward, done, trunc);  // K plain steps
        dim3 gv(grid_for(n / 4)), b(kBlock);
        if (continuous) hipLaunchKernelGGL((mountaincar_rollout_kernel<true, MC_ROLL_TABLE>), gv, b, 0, stream, dev, act, K, obs_out, reward, done, trunc, McLinearPolicy{}, (uint32_t*)nullptr, McUniformPolicy{});
        else hipLaunchKernelGGL((mountaincar_rollout_kernel<false, MC_ROLL_TABLE>), gv, b, 0, stream, dev, act, K, obs_out, reward, done, trunc, McLinearPolicy{}, (uint32_t*)nullptr, McUniformPolicy{});
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    uint32_t policy_calls = 0;  // mgym_rollout_uniform calls so far (Philox counter word of the policy stream)
    int rollout_uniform(uint64_t policy_seed, int K, void* actions_out, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0 || K == 0) return MGYM_OK;
        uint32_t* ao = static_cast<uint32_t*>(actions_out);
        if (n % 4 != 0 || !aligned(ao, 16) || !aligned(reward, 16) || !aligned(done, 4) || !aligned(trunc, 4) || !aligned(obs_out, 16)) {
            set_last_error("mgym_rollout_uniform: n_envs must be a multiple of 4 and the buffers 16-byte aligned");
            return MGYM_ERR_BAD_ARG;
        }
        const McUniformPolicy uni{policy_seed, policy_calls++};
        dim3 gv(grid_for(n / 4)), b(kBlock);
        if (continuous) hipLaunchKernelGGL((mountaincar_rollout_kernel<true, MC_ROLL_UNIFORM>), gv, b, 0, stream, dev, (const uint32_t*)nullptr, K, obs_out, reward, done, trunc, McLinearPolicy{}, ao, uni);
        else hipLaunchKernelGGL((mountaincar_rollout_kernel<false, MC_ROLL_UNIFORM>), gv, b, 0, stream, dev, (const uint32_t*)nullptr, K, obs_out, reward, done, trunc, McLinearPolicy{}, ao, uni);
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int rollout_linear(const float* policy, int K, void* actions_out, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0 || K == 0) return MGYM_OK;
        uint32_t* ao = static_cast<uint32_t*>(actions_out);
        if (n % 4 != 0 || !aligned(ao, 16) || !aligned(reward, 16) || !aligned(done, 4) || !aligned(trunc, 4) || !aligned(obs_out, 16)) {
            set_last_error("mgym_rollout_linear: n_envs must be a multiple of 4 and the buffers 16-byte aligned");
            return MGYM_ERR_BAD_ARG;
        }
        McLinearPolicy lin{};
        const int rows = continuous ? 1 : 3;   // policy: `rows` rows of (w_position, w_velocity, bias)
        for (int j = 0; j < rows; ++j) { lin.w[j][0] = policy[3 * j + 0]; lin.w[j][1] = policy[3 * j + 1]; lin.b[j] = policy[3 * j + 2]; }
        dim3 gv(grid_for(n / 4)), b(kBlock);
        if (continuous) hipLaunchKernelGGL((mountaincar_rollout_kernel<true, MC_ROLL_LINEAR>), gv, b, 0, stream, dev, (const uint32_t*)nullptr, K, obs_out, reward, done, trunc, lin, ao, McUniformPolicy{});
        else hipLaunchKernelGGL((mountaincar_rollout_kernel<false, MC_ROLL_LINEAR>), gv, b, 0, stream, dev, (const uint32_t*)nullptr, K, obs_out, reward, done, trunc, lin, ao, McUniformPolicy{});
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int observation(const float** obs, uint64_t* col_stride) override {
        *obs = dev.pos;
        *col_stride = n_pad;
        return MGYM_OK;
    }

    int get_state(void* blob) override {
        if (n == 0) return MGYM_OK;
        hipLaunchKernelGGL(mountaincar_export_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           stream, dev, static_cast<uint32_t*>(blob));
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int set_state(const void* blob) override {
        if (n == 0) return MGYM_OK;
        hipLaunchKernelGGL(mountaincar_import_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           stream, dev, static_cast<const uint32_t*>(blob));
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }
};

Env* make_mountaincar(bool continuous) { return new MountainCarEnv(continuous); }

}  // namespace mgym
