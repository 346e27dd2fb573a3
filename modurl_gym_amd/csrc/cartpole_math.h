// cartpole_math.h — cheaper instruction sequences for the CartPole step, each checked bit-identical to the
// reference-form arithmetic it replaces (cartpole.rs:264-271 evaluated with IEEE f32 divide and the
// glibc-equal mg_sincosf of mgym_math.h):
//   * tests/native/cartpole_fast_check.cpp (no GPU): EXHAUSTIVELY every one of the 2^32 f32 inputs of cp_sincos_small and
//     cp_div_const against mg_sincosf / IEEE `x / c`.  cp_div (n / d): exhaustive over the DIVISORS the step can form (762 602) and
//     over every 1-ulp reciprocal estimate, with SAMPLED numerators (24 random + hard values per divisor, 4.4e8 cases) plus the
//     standard argument for this division expansion (one Newton step on a 1-ulp reciprocal, quotient, two residual corrections are
//     correctly rounded while no intermediate under- or overflows: numerators in [2^-100, 2^100], which cp_div_range_ok tests and
//     DESIGN.md §4 derives from the step's guard); the whole fast step against the reference step on 7e7 guard-admitted states.
//   * tests/test_gpu_classic.py::test_cartpole_fast_math_exhaustive_on_gpu: the same enumerations on the MI355X
//     (hardware v_rcp_f32, v_fma_f64), so the proof holds for the silicon that runs the kernel.
// The step kernels use these only under a wave-uniform guard (|theta| < 0.75, |theta_dot| < 2^40, valid action)
// and fall back to the reference-form code otherwise, so results are bit-identical for EVERY input.
//
// Why they are cheaper (per env-step, f32-issue-slot equivalents; f64 ops cost two):
//   sin+cos  : 19 individually rounded f64 mul/add -> 12 f64 mul/fma               (44 -> 28 slots incl. converts)
//   x / M    : IEEE divide (11 VALU + 2 s_setreg)  -> mul + 2 fma + sign copy      (x3)
//   n / d    : IEEE divide                         -> rcp + 2 fma + mul + 4 fma + sign copy
#pragma once
#include "mgym_math.h"

namespace mgym {

MG_HD float cp_copysign(float mag, float sgn) { return as_f32((as_u32(mag) & 0x7fffffffu) | (as_u32(sgn) & 0x80000000u)); }


// |y| < 0.75 (the top-12-bit test mg_sincosf uses for its small path, abstop12(y) < abstop12(pi/4)): sin and cos
// by the same polynomials as sinf_poly(), with each  a + b*c  pair fused.  The fused forms differ from the
// individually rounded ones by < 2^-52 relative in f64, which changes the f32 rounding of the final value only if
// it lies within that distance of a rounding boundary: the exhaustive check shows this happens for NO f32 input
// below 0.75 (it does for a handful of larger ones, which is why the guard stays at 0.75).
// The |y| < 2^-12 special case of mg_sincosf (sin = y, cos = 1) is reproduced by the polynomial itself.
MG_HD void cp_sincos_small(float y, float* sp, float* cp) {
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double x = (double)y;
    const double x2 = x * x;
    const double x3 = x * x2;
    const double ts = cp_fma(x2, s3, s2);
    const double x7 = x3 * x2;
    const double sv = cp_fma(x3, s1, x);
    *sp = cp_copysign((float)cp_fma(x7, ts, sv), y);  // y = -0: the fused chain yields +0, sinf(-0) = -0
    const double x4 = x2 * x2;
    const double t2 = cp_fma(x2, c4, c3);
    const double t1 = cp_fma(x2, c1, c0);
    const double x6 = x4 * x2;
    const double cv = cp_fma(x4, c2, t1);
    *cp = (float)cp_fma(x6, t2, cv);
}
MG_HD bool cp_sincos_small_ok(float y) { return (as_u32(y) & 0x7fffffffu) < 0x3f400000u; }

// x / c for a loop-invariant positive c with r = RN(1/c) (Markstein: with a correctly rounded reciprocal and a
// faithful first quotient, ONE fused residual correction gives the correctly rounded quotient).  The sign copy
// keeps x = -0 (the fma chain turns -0 into +0).  Exhaustively checked for c = total_mass = 1.1f over all x with
// |x| in {0} U [2^-100, 2^100]; outside (incl. inf/nan) the kernels take the reference-form path.
MG_HD float cp_div_const(float x, float c, float r) {
    const float q0 = x * r;
    const float e = cp_fmaf(-c, q0, x);
    return cp_copysign(cp_fmaf(e, r, q0), x);
}

// n / d, d in [0.5, 1) (the step's divisor length * (4/3 - masspole*cos^2/total_mass) lies in [0.62, 0.67]),
// |n| in {0} U [2^-100, 2^100]: the sequence hipcc emits for an IEEE divide (reciprocal estimate, one Newton step,
// quotient, two fused residual corrections) minus the v_div_scale / v_div_fixup / denormal-mode switches that
// only matter outside that range.  `y0` is the hardware reciprocal estimate on the device (v_rcp_f32, 1 ulp) and
// the IEEE quotient on the host; the host check also runs every case with that estimate moved one ulp either way,
// so the proof covers any 1-ulp reciprocal estimate, and the GPU test covers the hardware one directly.
MG_HD float cp_rcp_estimate(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(d);
#else
    return 1.0f / d;
#endif
}
MG_HD float cp_div_from_estimate(float n, float d, float y) {
    const float e = cp_fmaf(-d, y, 1.0f);
    y = cp_fmaf(e, y, y);
    float q = n * y;
    float r = cp_fmaf(-d, q, n);
    q = cp_fmaf(r, y, q);
    r = cp_fmaf(-d, q, n);
    return cp_copysign(cp_fmaf(r, y, q), n);
}
MG_HD float cp_div(float n, float d) { return cp_div_from_estimate(n, d, cp_rcp_estimate(d)); }
// magnitude range in which cp_div_const / cp_div are proven: 0 or 2^-100 <= |x| <= 2^100
MG_HD bool cp_div_range_ok(float x) {
    const uint32_t a = as_u32(x) & 0x7fffffffu;
    return a == 0u || (a >= 0x0d800000u && a <= 0x71800000u);
}

}  // namespace mgym
