// selftest.hip — on-device proof that the cheap CartPole instruction sequences (cartpole_math.h / cartpole_step.h)
// equal the reference-form arithmetic ON THE SILICON THAT RUNS THEM (hardware v_rcp_f32, v_fma_f32/f64, v_div_*):
// the same enumerations as tests/native/cartpole_fast_check.cpp, evaluated by gfx950 kernels.
//   out[0] sincos : every f32 with |y| < 0.75, both signs       cp_sincos_small == mg_sincosf
//   out[1] divc   : every f32 in the proven range               cp_div_const(x, 1.1f) == x / 1.1f  (IEEE v_div_* sequence)
//   out[2] div    : every f32 cos value c in [0, 1] -> divisor d, 32 numerators each (random + hard cases)
//   out[3] step   : 2^28 random guard-admitted states, all four <EULER, SB> variants: fast form == reference form
//   out[4] sincos_u : every f32 with |y| < 120, both signs: the fused lock-step mg_sincosf_u / mg_cosf_u (LunarLander,
//                   MountainCar) == mg_sincosf / mg_cosf
// Exposed as mgym_selftest_cartpole_math() (test seam; seconds on an MI355X).
#include "common.h"
#include "cartpole_step.h"

namespace mgym {

__device__ __forceinline__ uint32_t st_lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float st_rnd_float(uint32_t& s, uint32_t lo_e, uint32_t hi_e) {
    const uint32_t b = st_lcg(s), e = lo_e + (st_lcg(s) >> 8) % (hi_e - lo_e + 1u);
    return as_f32((b & 0x807fffffu) | (e << 23));
}

__device__ __forceinline__ void st_flush(unsigned long long* out, unsigned long long bad) {
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out, bad);
}

__global__ void __launch_bounds__(256) selftest_sincos_kernel(unsigned long long* out) {
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < 0x3f400000ull; u += stride) {
#pragma unroll
        for (uint32_t sg = 0; sg < 2; ++sg) {
            const float y = as_f32((uint32_t)u | (sg << 31));
            float s0, c0, s1, c1;
            mg_sincosf(y, &s0, &c0);
            cp_sincos_small(y, &s1, &c1);
            bad += (as_u32(s0) != as_u32(s1) || as_u32(c0) != as_u32(c1)) ? 1u : 0u;
        }
    }
    st_flush(out, bad);
}

__global__ void __launch_bounds__(256) selftest_sincos_u_kernel(unsigned long long* out) {
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < 0x42f00000ull; u += stride) {
#pragma unroll
        for (uint32_t sg = 0; sg < 2; ++sg) {
            const float y = as_f32((uint32_t)u | (sg << 31));
            float s0, c0, s1, c1;
            mg_sincosf(y, &s0, &c0);
            mg_sincosf_u(y, &s1, &c1);
            const float c2 = mg_cosf_u(y);
            bad += (as_u32(s0) != as_u32(s1) || as_u32(c0) != as_u32(c1) || as_u32(c2) != as_u32(c0)) ? 1u : 0u;
        }
    }
    st_flush(out, bad);
}

__global__ void __launch_bounds__(256) selftest_divc_kernel(unsigned long long* out) {
    unsigned long long bad = 0;
    constexpr CartPoleParams P = cartpole_params(1, 0);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < 0x100000000ull; u += stride) {
        const float x = as_f32((uint32_t)u);
        if (!cp_div_range_ok(x)) continue;
        // volatile copy: keeps the compiler from folding x / const into a reciprocal multiply of its own choosing
        volatile float m = P.total_mass;
        const float ref = x / m, got = cp_div_const(x, P.total_mass, P.inv_total_mass);
        bad += as_u32(ref) != as_u32(got) ? 1u : 0u;
    }
    st_flush(out, bad);
}

__global__ void __launch_bounds__(256) selftest_div_kernel(unsigned long long* out) {
    unsigned long long bad = 0;
    constexpr CartPoleParams P = cartpole_params(1, 0);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= 0x3f800000ull; u += stride) {
        const float c = as_f32((uint32_t)u);
        const float d = P.length * (4.0f / 3.0f - P.masspole * c * c / P.total_mass);  // cartpole.rs:269-270
        uint32_t rng = (uint32_t)u * 0x9E3779B9u + 12345u;
        for (int k = 0; k < 32; ++k) {
            const uint32_t bits = st_lcg(rng), e8 = 27u + (st_lcg(rng) >> 8) % 200u;  // 2^-100 .. 2^99
            float n = as_f32((bits & 0x807fffffu) | (e8 << 23));
            if (k == 0) n = 0.0f;
            if (k == 1) n = -0.0f;
            if (k == 2) n = d;
            if (k == 3) n = d * 3.0f;
            if (k == 4) n = as_f32(as_u32(d) + 1u);
            if (k == 5) n = -as_f32(as_u32(d) - 1u);
            const float ref = n / d, got = cp_div(n, d);
            bad += as_u32(ref) != as_u32(got) ? 1u : 0u;
        }
    }
    st_flush(out, bad);
}

template <bool EULER, bool SB>
__device__ __forceinline__ unsigned long long selftest_step_group(uint32_t seed, int count) {
    constexpr CartPoleParams P = cartpole_params(EULER ? 1 : 0, SB ? 1 : 0);
    unsigned long long bad = 0;
    uint32_t s = seed;
    for (int it = 0; it < count; ++it) {
        float x, xd, th, thd;
        const uint32_t mode = st_lcg(s) >> 29;
        if (mode < 5) {
            x = ((int32_t)st_lcg(s)) * (3.0f / 2147483648.0f); xd = ((int32_t)st_lcg(s)) * (4.0f / 2147483648.0f);
            th = ((int32_t)st_lcg(s)) * (0.3f / 2147483648.0f); thd = ((int32_t)st_lcg(s)) * (4.0f / 2147483648.0f);
        } else {
            x = st_rnd_float(s, 0, 200); xd = st_rnd_float(s, 0, 200);
            th = st_rnd_float(s, 0, 126); thd = st_rnd_float(s, 0, 166);
            if ((st_lcg(s) >> 30) == 0) th = as_f32(st_lcg(s) & 0x80000000u);
            if ((st_lcg(s) >> 30) == 0) thd = as_f32(st_lcg(s) & 0x80000000u);
        }
        const uint32_t action = st_lcg(s) >> 31;
        uint32_t ctr = cp_ctr_pack((st_lcg(s) >> 8) % 1030u, st_lcg(s) >> 30, st_lcg(s));
        if ((st_lcg(s) >> 28) == 0) ctr = (ctr & ~kCpStepsMask) | (498u + (st_lcg(s) >> 30));
        if (!cartpole_fast_ok(th, thd, action)) continue;
        float a0 = x, a1 = xd, a2 = th, a3 = thd, b0 = x, b1 = xd, b2 = th, b3 = thd, ra, rb;
        uint32_t ca = ctr, cb = ctr, da, ta, db, tb;
        cartpole_step_ref(P, a0, a1, a2, a3, ca, action, ra, da, ta);
        cartpole_step_fast<EULER, SB>(P, b0, b1, b2, b3, cb, action, rb, db, tb);
        const bool same = ca == cb && da == db && ta == tb && as_u32(ra) == as_u32(rb) && as_u32(a0) == as_u32(b0) &&
                          as_u32(a1) == as_u32(b1) && as_u32(a2) == as_u32(b2) && as_u32(a3) == as_u32(b3);
        bad += same ? 0u : 1u;
    }
    return bad;
}

__global__ void __launch_bounds__(256) selftest_step_kernel(unsigned long long* out, int per_thread) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    bad += selftest_step_group<true, false>(0x1234567u + tid * 2654435761u, per_thread);
    bad += selftest_step_group<true, true>(0x2345678u + tid * 2654435761u, per_thread / 4);
    bad += selftest_step_group<false, false>(0x3456789u + tid * 2654435761u, per_thread / 4);
    bad += selftest_step_group<false, true>(0x456789au + tid * 2654435761u, per_thread / 4);
    st_flush(out, bad);
}

}  // namespace mgym

using namespace mgym;

extern "C" int mgym_selftest_cartpole_math(int device, uint64_t* mismatches /* [5] */) {
    if (!mismatches) { set_last_error("mgym_selftest_cartpole_math: NULL"); return MGYM_ERR_BAD_ARG; }
    int prev = -1;
    (void)hipGetDevice(&prev);
    MGYM_HIP(hipSetDevice(device));
    unsigned long long* d = nullptr;
    MGYM_HIP(hipMalloc((void**)&d, 5 * sizeof(unsigned long long)));
    MGYM_HIP(hipMemset(d, 0, 5 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(selftest_sincos_kernel, dim3(4096), dim3(256), 0, nullptr, d + 0);
    hipLaunchKernelGGL(selftest_divc_kernel, dim3(4096), dim3(256), 0, nullptr, d + 1);
    hipLaunchKernelGGL(selftest_div_kernel, dim3(4096), dim3(256), 0, nullptr, d + 2);
    hipLaunchKernelGGL(selftest_step_kernel, dim3(4096), dim3(256), 0, nullptr, d + 3, 160);
    hipLaunchKernelGGL(selftest_sincos_u_kernel, dim3(4096), dim3(256), 0, nullptr, d + 4);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long h[5] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return hip_fail(e, "selftest", __FILE__, __LINE__);
    for (int i = 0; i < 5; ++i) mismatches[i] = h[i];
    return MGYM_OK;
}
