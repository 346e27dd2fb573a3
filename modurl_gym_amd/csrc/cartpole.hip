// cartpole.hip — CartPoleV1 step()/reset() as struct-of-arrays gfx950 kernels.
//
// Replaces the bodies of `impl Gym for CartPoleV1` (reference src/classic_control/cartpole.rs:
// reset :238-249, step :251-348) for n_envs independent environments.
//
// Data layout in HBM (engine-owned, one allocation, column stride n_pad words):
//   x[n] x_dot[n] theta[n] theta_dot[n]   f32   — the state, which IS the observation (:301)
//   ctr[n]                                u32   — bits 0..23 steps_since_reset (saturating; only
//                                                 ">= 500" is observable, :297), bits 24..31
//                                                 steps_beyond_terminated + 1 (0 = None, saturating)
//   episode[n]                            u32   — resets so far: word 2 of the Philox counter
// One lane owns 4 consecutive environments: every column is read and written with one
// 16-byte access per lane (1 KiB per wave instruction, fully coalesced); uniform constants
// travel in SGPRs via the kernel argument block; outputs leave with non-temporal stores.
// No MFMA: the path is element-wise.  LDS is used only by the done-mask compaction of the
// fused auto-reset / masked reset (finished envs are rare per env but present in ~every wave).
//
// Algorithmic HBM bytes per env-step (obs_out == NULL, zero-copy observation): 50
//   state 16 R + 16 W, action 4 R, reward 4 W, done 1 W, truncated 1 W, ctr 4 R + 4 W.
// Arithmetic follows cartpole.rs:264-283 in the written association, no contraction
// (-ffp-contract=off), IEEE division; sin/cos from mgym_math.h (bit-identical to glibc).
#include "common.h"
#include "mgym_math.h"
#include "philox.h"

namespace mgym {

struct CartPoleParams {
    float gravity, masspole, total_mass, length, polemass_length, force_mag, tau;
    float x_threshold, theta_threshold_radians;
    int is_euler, sutton_barto;
};

struct CartPoleDev {
    float *x, *xd, *th, *thd;
    uint32_t *ctr, *episode;
    uint64_t n;
    uint64_t seed, env_id_base;
    uint32_t* err;
    CartPoleParams p;
    int auto_reset;
};

constexpr uint32_t kStepsMask = 0x00FFFFFFu;
constexpr int kAutoResetMode = 2;  // LDS compaction: measured 11.4 us vs 15.1 us (in place) at 1 Mi envs
constexpr uint64_t kInPlaceResetFrom = 12ull << 20;  // envs: from here the fused reset is done in place (see step())
constexpr bool kNtStores = true;   // non-temporal output stores: 10.9 us vs 11.4 us at 1 Mi envs

struct CartPoleLane {
    float x, xd, th, thd;
    uint32_t ctr;
};

// reset(): cartpole.rs:238-249 — state ~ U[-0.05, 0.05)^4 sampled in f64 then cast,
// steps_since_reset = 0, steps_beyond_terminated = None.
__device__ __forceinline__ void cartpole_reset_one(const CartPoleDev& d, uint64_t i, CartPoleLane& s) {
    uint32_t ep = d.episode[i];
    uint64_t gid = d.env_id_base + i;
    Philox4 a = env_draw(d.seed, gid, ep, SLOT_RESET0);
    Philox4 b = env_draw(d.seed, gid, ep, SLOT_RESET1);
    s.x = uniform_f64_to_f32(-0.05, 0.05, a.w[0], a.w[1]);
    s.xd = uniform_f64_to_f32(-0.05, 0.05, a.w[2], a.w[3]);
    s.th = uniform_f64_to_f32(-0.05, 0.05, b.w[0], b.w[1]);
    s.thd = uniform_f64_to_f32(-0.05, 0.05, b.w[2], b.w[3]);
    s.ctr = 0u;
    d.episode[i] = ep + 1u;
}

// step(): cartpole.rs:251-348 for one environment held in registers.
__device__ __forceinline__ void cartpole_step_one(const CartPoleParams& p, CartPoleLane& s, uint32_t action,
                                                  float& reward, uint32_t& done, uint32_t& trunc, bool& bad) {
    if (action >= 2u) {  // :252 assert!(action_space.contains(&action))
        bad = true;
        reward = 0.0f; done = 0u; trunc = 0u;
        return;
    }
    float x = s.x, x_dot = s.xd, theta = s.th, theta_dot = s.thd;
    float force = (action == 0u) ? -p.force_mag : p.force_mag;  // :258-262

    float sintheta, costheta;
    mg_sincosf(theta, &sintheta, &costheta);  // :264-265

    // :267-271
    float temp = (force + p.polemass_length * theta_dot * theta_dot * sintheta) / p.total_mass;
    float thetaacc = (p.gravity * sintheta - costheta * temp) /
                     (p.length * (4.0f / 3.0f - p.masspole * costheta * costheta / p.total_mass));
    float xacc = temp - p.polemass_length * thetaacc * costheta / p.total_mass;

    if (p.is_euler) {  // :273-277
        x += p.tau * x_dot;
        x_dot += p.tau * xacc;
        theta += p.tau * theta_dot;
        theta_dot += p.tau * thetaacc;
    } else {  // :279-282 verbatim
        x_dot += 0.5f * p.tau * (xacc + temp);
        theta_dot += 0.5f * p.tau * (thetaacc + temp);
        theta += p.tau * theta_dot + 0.5f * p.tau * p.tau * thetaacc;
        theta_dot += 0.5f * p.tau * (thetaacc + temp);
    }
    s.x = x; s.xd = x_dot; s.th = theta; s.thd = theta_dot;  // :285-290

    bool terminated = x < -p.x_threshold || x > p.x_threshold || theta < -p.theta_threshold_radians ||
                      theta > p.theta_threshold_radians;  // :291-294

    uint32_t steps = s.ctr & kStepsMask;
    uint32_t sbt1 = s.ctr >> 24;  // steps_beyond_terminated + 1, 0 = None
    steps = steps < kStepsMask ? steps + 1u : steps;  // :296

    if (steps >= 500u) {  // :297-306 truncation takes precedence
        sbt1 = 1u;
        reward = 1.0f; done = 0u; trunc = 1u;
    } else if (!terminated) {  // :310-318
        reward = p.sutton_barto ? 0.0f : 1.0f; done = 0u; trunc = 0u;
    } else if (sbt1 == 0u) {  // :319-329
        sbt1 = 1u;
        reward = p.sutton_barto ? -1.0f : 1.0f; done = 1u; trunc = 0u;
    } else {  // :330-346
        reward = p.sutton_barto ? -1.0f : 0.0f;
        sbt1 = sbt1 < 255u ? sbt1 + 1u : sbt1;
        done = 1u; trunc = 0u;
    }
    s.ctr = steps | (sbt1 << 24);
}

// Fused auto-reset support: finished environments are rare per ENV (a few percent per step) but
// nearly every WAVE contains one, so resetting in place would make every wave pay two Philox
// evaluations.  Instead the block compacts its finished envs through LDS (done-mask reduction):
// lanes push (local env id) into a list, one pass of the block's first lanes draws the new
// states for the compacted list, owners pick their values up again before the vector store.
struct ResetScratch {
    uint32_t count;
    uint16_t idx[kBlock * 4];     // local env ids of finished envs
    float fresh[kBlock * 4][4];   // their new states, by list slot
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float e) {
    f32x4 v = {a, b, c, e};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void st4u(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t e) {
    u32x4 v = {a, b, c, e};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void st1u(uint32_t* p, uint32_t a) {
    if (NT) __builtin_nontemporal_store(a, p);
    else *p = a;
}

// RMODE: how MGYM_FLAG_AUTO_RESET is realised — 0 none, 1 in place per lane, 2 LDS compaction.
// NT: non-temporal stores for the output columns.
template <int VEC, int RMODE, bool NT>
__global__ void __launch_bounds__(kBlock)
cartpole_step_kernel(CartPoleDev d, const uint32_t* __restrict__ act, float* __restrict__ obs_out,
                     float* __restrict__ rew, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    __shared__ ResetScratch lds_[RMODE == 2 ? 1 : 1];
    constexpr uint64_t kPerBlock = (uint64_t)kBlock * VEC;
    bool bad = false;
    // block-uniform loop (the compaction below uses block barriers)
    for (uint64_t base = (uint64_t)blockIdx.x * kPerBlock; base < d.n; base += (uint64_t)gridDim.x * kPerBlock) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * VEC;
        CartPoleLane s[VEC];
        uint32_t a[VEC];
        float r[VEC];
        uint32_t dn[VEC], tr[VEC];
        const bool full = i0 + VEC <= d.n;
        if (VEC == 4 && full) {
            // engine columns are padded to n_pad, caller buffers were checked for 16-B alignment
            float4 vx = *reinterpret_cast<const float4*>(d.x + i0);
            float4 vxd = *reinterpret_cast<const float4*>(d.xd + i0);
            float4 vth = *reinterpret_cast<const float4*>(d.th + i0);
            float4 vthd = *reinterpret_cast<const float4*>(d.thd + i0);
            uint4 vc = *reinterpret_cast<const uint4*>(d.ctr + i0);
            uint4 va = *reinterpret_cast<const uint4*>(act + i0);
            s[0] = {vx.x, vxd.x, vth.x, vthd.x, vc.x};
            s[1 % VEC] = {vx.y, vxd.y, vth.y, vthd.y, vc.y};
            s[2 % VEC] = {vx.z, vxd.z, vth.z, vthd.z, vc.z};
            s[3 % VEC] = {vx.w, vxd.w, vth.w, vthd.w, vc.w};
            a[0] = va.x; a[1 % VEC] = va.y; a[2 % VEC] = va.z; a[3 % VEC] = va.w;
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                uint64_t i = i0 + k;
                if (i < d.n) {
                    s[k] = {d.x[i], d.xd[i], d.th[i], d.thd[i], d.ctr[i]};
                    a[k] = act[i];
                } else {
                    s[k] = {0.f, 0.f, 0.f, 0.f, 0u};
                    a[k] = 0u;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) cartpole_step_one(d.p, s[k], a[k], r[k], dn[k], tr[k], bad);

        if (RMODE == 1) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                if ((dn[k] | tr[k]) && (i0 + k < d.n)) cartpole_reset_one(d, i0 + k, s[k]);
        }
        if (RMODE == 2) {
            ResetScratch& lds = *reinterpret_cast<ResetScratch*>(&lds_[0]);
            if (threadIdx.x == 0) lds.count = 0;
            __syncthreads();
            uint32_t slot[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                slot[k] = 0xffffffffu;
                if ((dn[k] | tr[k]) && (i0 + k < d.n)) {
                    slot[k] = atomicAdd(&lds.count, 1u);
                    lds.idx[slot[k]] = (uint16_t)(threadIdx.x * VEC + k);
                }
            }
            __syncthreads();
            const uint32_t cnt = lds.count;
            for (uint32_t j = threadIdx.x; j < cnt; j += kBlock) {
                CartPoleLane f;
                cartpole_reset_one(d, base + lds.idx[j], f);
                lds.fresh[j][0] = f.x; lds.fresh[j][1] = f.xd; lds.fresh[j][2] = f.th; lds.fresh[j][3] = f.thd;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                if (slot[k] != 0xffffffffu) {
                    s[k].x = lds.fresh[slot[k]][0]; s[k].xd = lds.fresh[slot[k]][1];
                    s[k].th = lds.fresh[slot[k]][2]; s[k].thd = lds.fresh[slot[k]][3];
                    s[k].ctr = 0u;
                }
            }
        }

        if (VEC == 4 && full) {
            st4<NT>(d.x + i0, s[0].x, s[1 % VEC].x, s[2 % VEC].x, s[3 % VEC].x);
            st4<NT>(d.xd + i0, s[0].xd, s[1 % VEC].xd, s[2 % VEC].xd, s[3 % VEC].xd);
            st4<NT>(d.th + i0, s[0].th, s[1 % VEC].th, s[2 % VEC].th, s[3 % VEC].th);
            st4<NT>(d.thd + i0, s[0].thd, s[1 % VEC].thd, s[2 % VEC].thd, s[3 % VEC].thd);
            st4u<NT>(d.ctr + i0, s[0].ctr, s[1 % VEC].ctr, s[2 % VEC].ctr, s[3 % VEC].ctr);
            if (obs_out) {
                *reinterpret_cast<float4*>(obs_out + i0) = make_float4(s[0].x, s[1 % VEC].x, s[2 % VEC].x, s[3 % VEC].x);
                *reinterpret_cast<float4*>(obs_out + d.n + i0) = make_float4(s[0].xd, s[1 % VEC].xd, s[2 % VEC].xd, s[3 % VEC].xd);
                *reinterpret_cast<float4*>(obs_out + 2 * d.n + i0) = make_float4(s[0].th, s[1 % VEC].th, s[2 % VEC].th, s[3 % VEC].th);
                *reinterpret_cast<float4*>(obs_out + 3 * d.n + i0) = make_float4(s[0].thd, s[1 % VEC].thd, s[2 % VEC].thd, s[3 % VEC].thd);
            }
            if (rew) st4<NT>(rew + i0, r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
            if (done_out)
                st1u<NT>(reinterpret_cast<uint32_t*>(done_out + i0), dn[0] | (dn[1 % VEC] << 8) | (dn[2 % VEC] << 16) | (dn[3 % VEC] << 24));
            if (trunc_out)
                st1u<NT>(reinterpret_cast<uint32_t*>(trunc_out + i0), tr[0] | (tr[1 % VEC] << 8) | (tr[2 % VEC] << 16) | (tr[3 % VEC] << 24));
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                uint64_t i = i0 + k;
                if (i < d.n) {
                    d.x[i] = s[k].x; d.xd[i] = s[k].xd; d.th[i] = s[k].th; d.thd[i] = s[k].thd; d.ctr[i] = s[k].ctr;
                    if (obs_out) {
                        obs_out[i] = s[k].x; obs_out[d.n + i] = s[k].xd;
                        obs_out[2 * d.n + i] = s[k].th; obs_out[3 * d.n + i] = s[k].thd;
                    }
                    if (rew) rew[i] = r[k];
                    if (done_out) done_out[i] = (uint8_t)dn[k];
                    if (trunc_out) trunc_out[i] = (uint8_t)tr[k];
                }
            }
        }
    }
    // done-mask style wave reduction of the (rare) error bit: one atomic per wave at most
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_INVALID_ACTION);
    }
}

// Fused K-step rollout (mgym_rollout): one lane keeps its 4 environments in registers for K steps; per step it
// reads one 16-B action word and writes reward / done / truncated (and the observation if asked).  Identical to K
// launches of cartpole_step_kernel<4, RMODE, NT> by construction: same per-env function, same reset compaction.
template <int RMODE>
__global__ void __launch_bounds__(kBlock)
cartpole_rollout_kernel(CartPoleDev d, const uint32_t* __restrict__ act, int K, float* __restrict__ obs_out, float* __restrict__ rew,
                        uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    __shared__ ResetScratch lds_[1];
    constexpr uint64_t kPerBlock = (uint64_t)kBlock * 4;
    bool bad = false;
    for (uint64_t base = (uint64_t)blockIdx.x * kPerBlock; base < d.n; base += (uint64_t)gridDim.x * kPerBlock) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * 4;
        const bool in = i0 < d.n;  // n % 4 == 0 is required by the host wrapper
        CartPoleLane s[4];
        if (in) {
            float4 vx = *reinterpret_cast<const float4*>(d.x + i0), vxd = *reinterpret_cast<const float4*>(d.xd + i0);
            float4 vth = *reinterpret_cast<const float4*>(d.th + i0), vthd = *reinterpret_cast<const float4*>(d.thd + i0);
            uint4 vc = *reinterpret_cast<const uint4*>(d.ctr + i0);
            s[0] = {vx.x, vxd.x, vth.x, vthd.x, vc.x}; s[1] = {vx.y, vxd.y, vth.y, vthd.y, vc.y};
            s[2] = {vx.z, vxd.z, vth.z, vthd.z, vc.z}; s[3] = {vx.w, vxd.w, vth.w, vthd.w, vc.w};
        } else {
            for (int k = 0; k < 4; ++k) s[k] = {0.f, 0.f, 0.f, 0.f, 0u};
        }
        for (int t = 0; t < K; ++t) {
            const uint64_t off = (uint64_t)t * d.n + i0;
            uint32_t a[4] = {0u, 0u, 0u, 0u}, dn[4], tr[4];
            float r[4];
            if (in) { u32x4 va = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(act + off)); a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w; }
#pragma unroll
            for (int k = 0; k < 4; ++k) cartpole_step_one(d.p, s[k], a[k], r[k], dn[k], tr[k], bad);
            if (RMODE == 2) {  // same LDS compaction as cartpole_step_kernel
                ResetScratch& lds = lds_[0];
                if (threadIdx.x == 0) lds.count = 0;
                __syncthreads();
                uint32_t slot[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    slot[k] = 0xffffffffu;
                    if (in && (dn[k] | tr[k])) { slot[k] = atomicAdd(&lds.count, 1u); lds.idx[slot[k]] = (uint16_t)(threadIdx.x * 4 + k); }
                }
                __syncthreads();
                const uint32_t cnt = lds.count;
                for (uint32_t j = threadIdx.x; j < cnt; j += kBlock) {
                    CartPoleLane f;
                    cartpole_reset_one(d, base + lds.idx[j], f);
                    lds.fresh[j][0] = f.x; lds.fresh[j][1] = f.xd; lds.fresh[j][2] = f.th; lds.fresh[j][3] = f.thd;
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (slot[k] != 0xffffffffu) {
                        s[k].x = lds.fresh[slot[k]][0]; s[k].xd = lds.fresh[slot[k]][1]; s[k].th = lds.fresh[slot[k]][2]; s[k].thd = lds.fresh[slot[k]][3];
                        s[k].ctr = 0u;
                    }
            }
            if (in) {
                if (rew) st4<true>(rew + off, r[0], r[1], r[2], r[3]);
                if (done_out) st1u<true>(reinterpret_cast<uint32_t*>(done_out + off), dn[0] | (dn[1] << 8) | (dn[2] << 16) | (dn[3] << 24));
                if (trunc_out) st1u<true>(reinterpret_cast<uint32_t*>(trunc_out + off), tr[0] | (tr[1] << 8) | (tr[2] << 16) | (tr[3] << 24));
                if (obs_out) {
                    float* o = obs_out + (uint64_t)t * 4 * d.n + i0;
                    st4<true>(o, s[0].x, s[1].x, s[2].x, s[3].x); st4<true>(o + d.n, s[0].xd, s[1].xd, s[2].xd, s[3].xd);
                    st4<true>(o + 2 * d.n, s[0].th, s[1].th, s[2].th, s[3].th); st4<true>(o + 3 * d.n, s[0].thd, s[1].thd, s[2].thd, s[3].thd);
                }
            }
        }
        if (in) {
            st4<kNtStores>(d.x + i0, s[0].x, s[1].x, s[2].x, s[3].x); st4<kNtStores>(d.xd + i0, s[0].xd, s[1].xd, s[2].xd, s[3].xd);
            st4<kNtStores>(d.th + i0, s[0].th, s[1].th, s[2].th, s[3].th); st4<kNtStores>(d.thd + i0, s[0].thd, s[1].thd, s[2].thd, s[3].thd);
            st4u<kNtStores>(d.ctr + i0, s[0].ctr, s[1].ctr, s[2].ctr, s[3].ctr);
        }
    }
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_INVALID_ACTION);
    }
}

// Masked reset.  One lane looks at 4 consecutive mask bytes of each mask with a single dword
// load (2 B/env of traffic); masked envs are compacted through LDS exactly as in the fused
// path, so one pass of the block's first lanes evaluates Philox for all of them instead of
// every wave paying for its one or two masked lanes.  `all` skips masks and compaction.
__global__ void __launch_bounds__(kBlock)
cartpole_reset_kernel(CartPoleDev d, const uint8_t* __restrict__ m0, const uint8_t* __restrict__ m1, int all,
                      int masks_aligned, float* __restrict__ obs_out) {
    __shared__ uint32_t count;
    __shared__ uint16_t idx[kBlock * 4];
    constexpr uint64_t kPerBlock = (uint64_t)kBlock * 4;
    for (uint64_t base = (uint64_t)blockIdx.x * kPerBlock; base < d.n; base += (uint64_t)gridDim.x * kPerBlock) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * 4;
        uint32_t m = 0;
        if (i0 < d.n) {
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
        }
        uint32_t cnt;
        if (all) {
            cnt = (uint32_t)((d.n - base) < kPerBlock ? (d.n - base) : kPerBlock);
        } else {
            if (threadIdx.x == 0) count = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < d.n && ((m >> (8 * k)) & 0xffu)) idx[atomicAdd(&count, 1u)] = (uint16_t)(threadIdx.x * 4 + k);
            __syncthreads();
            cnt = count;
        }
        for (uint32_t j = threadIdx.x; j < cnt; j += kBlock) {
            const uint64_t i = base + (all ? j : (uint32_t)idx[j]);
            CartPoleLane s;
            cartpole_reset_one(d, i, s);
            d.x[i] = s.x; d.xd[i] = s.xd; d.th[i] = s.th; d.thd[i] = s.thd; d.ctr[i] = s.ctr;
            if (obs_out) {
                obs_out[i] = s.x; obs_out[d.n + i] = s.xd; obs_out[2 * d.n + i] = s.th; obs_out[3 * d.n + i] = s.thd;
            }
        }
        if (!all) __syncthreads();  // idx/count are reused by the next iteration
    }
}

// blob <-> engine columns (steps / sbt are separate blob columns)
__global__ void cartpole_export_kernel(CartPoleDev d, uint32_t* __restrict__ blob) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    blob[0 * d.n + i] = as_u32(d.x[i]);
    blob[1 * d.n + i] = as_u32(d.xd[i]);
    blob[2 * d.n + i] = as_u32(d.th[i]);
    blob[3 * d.n + i] = as_u32(d.thd[i]);
    uint32_t c = d.ctr[i];
    blob[4 * d.n + i] = c & kStepsMask;
    blob[5 * d.n + i] = (uint32_t)((int32_t)(c >> 24) - 1);  // -1 = None
    blob[6 * d.n + i] = d.episode[i];
}

__global__ void cartpole_import_kernel(CartPoleDev d, const uint32_t* __restrict__ blob) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    d.x[i] = as_f32(blob[0 * d.n + i]);
    d.xd[i] = as_f32(blob[1 * d.n + i]);
    d.th[i] = as_f32(blob[2 * d.n + i]);
    d.thd[i] = as_f32(blob[3 * d.n + i]);
    uint32_t steps = blob[4 * d.n + i];
    if (steps > kStepsMask) steps = kStepsMask;
    int32_t sbt = (int32_t)blob[5 * d.n + i];
    uint32_t sbt1 = sbt < 0 ? 0u : (sbt >= 254 ? 255u : (uint32_t)sbt + 1u);
    d.ctr[i] = steps | (sbt1 << 24);
    d.episode[i] = blob[6 * d.n + i];
}

struct CartPoleEnv final : Env {
    // population size from which the fused reset runs in place; MGYM_CARTPOLE_INPLACE_FROM overrides (tests force either mode)
    uint64_t inplace_from = getenv("MGYM_CARTPOLE_INPLACE_FROM") ? strtoull(getenv("MGYM_CARTPOLE_INPLACE_FROM"), nullptr, 0) : kInPlaceResetFrom;
    void* base = nullptr;
    CartPoleDev dev{};

    ~CartPoleEnv() override {
        if (base) (void)hipFree(base);
    }

    int init() override {
        obs_dim = 4;
        state_cols = 7;
        MGYM_HIP(hipMalloc(&base, 6 * n_pad * sizeof(float)));
        MGYM_HIP(hipMemsetAsync(base, 0, 6 * n_pad * sizeof(float), stream));
        float* f = static_cast<float*>(base);
        dev.x = f; dev.xd = f + n_pad; dev.th = f + 2 * n_pad; dev.thd = f + 3 * n_pad;
        dev.ctr = reinterpret_cast<uint32_t*>(f + 4 * n_pad);
        dev.episode = reinterpret_cast<uint32_t*>(f + 5 * n_pad);
        dev.n = n;
        dev.seed = cfg.seed;
        dev.env_id_base = cfg.env_id_base;
        dev.err = d_err;
        dev.auto_reset = (cfg.flags & MGYM_FLAG_AUTO_RESET) ? 1 : 0;
        // constructor constants, cartpole.rs:45-56, f32 arithmetic in the written order
        CartPoleParams& p = dev.p;
        const float masscart = 1.0f;
        p.gravity = 9.8f;
        p.masspole = 0.1f;
        p.total_mass = p.masspole + masscart;
        p.length = 0.5f;
        p.polemass_length = p.masspole * p.length;
        p.force_mag = 10.0f;
        p.tau = 0.02f;
        p.theta_threshold_radians = 12.0f * 2.0f * 3.14159265358979323846f / 360.0f;
        p.x_threshold = 2.4f;
        p.is_euler = cfg.is_euler;
        p.sutton_barto = cfg.sutton_barto_reward;
        // cartpole.rs:81: steps_beyond_terminated = Some(0) at construction
        if (n) {
            // fill ctr column with (1 << 24)
            MGYM_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dev.ctr), 1u << 24, n, stream));
        }
        return MGYM_OK;
    }

    int reset(const uint8_t* m0, const uint8_t* m1, bool all, float* obs_out) override {
        if (n == 0) return MGYM_OK;
        int al = aligned(m0, 4) && aligned(m1, 4);
        hipLaunchKernelGGL(cartpole_reset_kernel, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, stream, dev, m0, m1,
                           all ? 1 : 0, al, obs_out);
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int reset_deterministic(float* obs_out) override {  // cartpole.rs:437-442: self.reset()? then state = zeros
        int st = reset(nullptr, nullptr, true, nullptr);
        if (st != MGYM_OK || n == 0) return st;
        MGYM_HIP(hipMemsetAsync(dev.x, 0, 4 * n_pad * sizeof(float), stream));
        if (obs_out) MGYM_HIP(hipMemsetAsync(obs_out, 0, 4 * n * sizeof(float), stream));
        return MGYM_OK;
    }

    int step(const void* actions, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0) return MGYM_OK;
        const uint32_t* act = static_cast<const uint32_t*>(actions);
        bool vec_ok = aligned(act, 16) && aligned(reward, 16) && aligned(done, 4) && aligned(trunc, 4) &&
                      (obs_out == nullptr || (aligned(obs_out, 16) && n % 4 == 0));
        dim3 gv(grid_for((n + 3) / 4)), gs(grid_for(n)), b(kBlock);
        if (dev.auto_reset) {
            // fused reset: LDS compaction of finished envs wins while the population is Infinity-Cache sized and up to
            // ~8 Mi envs (77.8 vs 89.0 us at 8 Mi); past that each block makes several grid-stride passes and the
            // compaction's barriers cost more HBM overlap than the Philox work they save (176.6 vs 211.1 us at 16 Mi,
            // 394 vs 453 us at 32 Mi): reset in place there.  profiles/r01_cartpole_tune_variants.log
            if (vec_ok && n >= inplace_from) hipLaunchKernelGGL((cartpole_step_kernel<4, 1, kNtStores>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc);
            else if (vec_ok) hipLaunchKernelGGL((cartpole_step_kernel<4, kAutoResetMode, kNtStores>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc);
            else hipLaunchKernelGGL((cartpole_step_kernel<1, kAutoResetMode, kNtStores>), gs, b, 0, stream, dev, act, obs_out, reward, done, trunc);
        } else {
            if (vec_ok) hipLaunchKernelGGL((cartpole_step_kernel<4, 0, kNtStores>), gv, b, 0, stream, dev, act, obs_out, reward, done, trunc);
            else hipLaunchKernelGGL((cartpole_step_kernel<1, 0, kNtStores>), gs, b, 0, stream, dev, act, obs_out, reward, done, trunc);
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
        if (dev.auto_reset) hipLaunchKernelGGL((cartpole_rollout_kernel<2>), gv, b, 0, stream, dev, act, K, obs_out, reward, done, trunc);
        else hipLaunchKernelGGL((cartpole_rollout_kernel<0>), gv, b, 0, stream, dev, act, K, obs_out, reward, done, trunc);
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int observation(const float** obs, uint64_t* col_stride) override {
        *obs = dev.x;
        *col_stride = n_pad;
        return MGYM_OK;
    }

    int get_state(void* blob) override {
        if (n == 0) return MGYM_OK;
        hipLaunchKernelGGL(cartpole_export_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                           dev, static_cast<uint32_t*>(blob));
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int set_state(const void* blob) override {
        if (n == 0) return MGYM_OK;
        hipLaunchKernelGGL(cartpole_import_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                           dev, static_cast<const uint32_t*>(blob));
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }
};

Env* make_cartpole() { return new CartPoleEnv(); }

}  // namespace mgym
