// philox.h — counter-based per-environment random streams (Philox4x32-10, Salmon et al. SC'11).
//
// The reference draws from candle's Tensor::rand (cartpole.rs:240, mountain_car.rs:281) and
// rand 0.9 StdRng/ThreadRng (lunar_lander.rs:60-88,755,845-846,855-856,973-974).  A sequential
// generator per env object does not map to one-lane-per-env kernels, so each draw is a pure
// function of (seed; global env id, episode, slot): no generator state in HBM, any lane can
// produce any draw, and results are independent of how envs are sharded over GPUs.
// Distributions follow the reference (f64 53-bit for candle's f64->f32 sampling; rand's
// 23-bit UniformFloat<f32>; uniform integers by multiply-shift).
#pragma once
#include <stdint.h>
#include "mgym_math.h"

namespace mgym {

struct Philox4 { uint32_t w[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) yields both halves
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o; o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

// slots of the (env, episode) counter space
enum : uint32_t { SLOT_RESET0 = 0, SLOT_RESET1 = 1, SLOT_RESET2 = 2, SLOT_RESET3 = 3, SLOT_STEP_BASE = 16,
                  SLOT_POLICY = 0x40000000u /* on-device rollout policy: + step / 32, keyed by the policy seed */ };

__device__ __forceinline__ Philox4 env_draw(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t slot) {
    return philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), episode, slot,
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// [0,1) with 53 random bits
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    uint64_t m = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return (double)m * (1.0 / 9007199254740992.0);
}
// [0,1) with 23 random bits (rand 0.9 UniformFloat<f32>: top 23 bits as [1,2), minus 1)
__device__ __forceinline__ float u23(uint32_t w) { return (float)(w >> 9) * (1.0f / 8388608.0f); }
// candle Tensor::rand(lo, up) in f64, then .to_dtype(F32)
__device__ __forceinline__ float uniform_f64_to_f32(double lo, double hi, uint32_t w_hi, uint32_t w_lo) {
    return (float)(u53(w_hi, w_lo) * (hi - lo) + lo);
}

}  // namespace mgym
