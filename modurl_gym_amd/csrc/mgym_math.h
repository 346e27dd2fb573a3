// mgym_math.h — f32 elementary functions for the gfx950 kernels.
//
// The Rust reference calls f32::sin / f32::cos / f32::tanh (cartpole.rs:264-265,
// mountain_car.rs:302, lunar_lander.rs:942-954,963), which lower to the platform libm
// (glibc on Linux).  Threshold tests on the results decide `done`, so the kernels need
// the SAME rounding as that libm, not merely a <=1-2 ulp approximation (ROCm's OCML
// sinf/cosf differ from glibc in the last bit on a few percent of inputs).
//
// These are restatements of the published algorithms glibc >= 2.28 uses:
//   sinf/cosf : ARM optimized-routines "sincosf" — evaluate in double: one
//               multiply-subtract range reduction for |x| < 120, a 192-bit 2/pi
//               table reduction above, then a degree-7/8 odd/even polynomial.
//   tanhf     : fdlibm s_tanhf.c on top of s_expm1f.c (pure f32 arithmetic).
// Every operation is an individually rounded IEEE add/mul/div (compile with
// -ffp-contract=off), so the device result is bit-identical to a non-FMA build of
// those routines; tests/test_math_host.py compiles this header for the host and
// compares against the container's glibc over all floats in the fast range.
//
// MG_HD expands to __host__ __device__ under hipcc and to nothing under g++.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MG_HD __host__ __device__ __forceinline__
#else
#define MG_HD static inline
#endif

namespace mgym {

MG_HD uint32_t as_u32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
MG_HD float as_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

// explicit fused multiply-add (the only contraction in the code base: -ffp-contract=off keeps everything else unfused)
MG_HD float cp_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MG_HD double cp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- sincosf polynomial tables (table[1] has the cosine signs flipped) ----
struct SinCosTab {
    double c0, c1, c2, c3, c4;  // cosine polynomial
    double s1, s2, s3;          // sine polynomial
};

MG_HD double sc_hpi_inv() { return 0x1.45F306DC9C883p+23; }  // 2/pi * 2^24
MG_HD double sc_hpi() { return 0x1.921FB54442D18p0; }        // pi/2
MG_HD double sc_pi63() { return 0x1.921FB54442D18p-62; }     // 2*pi / 2^64

// top 12 bits of |x| (sign dropped): cheap magnitude classes
MG_HD uint32_t abstop12(float x) { return (as_u32(x) >> 20) & 0x7ffu; }

// sine (n even) or cosine (n odd) polynomial of the reduced argument; `neg` selects
// the sign-flipped cosine coefficients (quadrants 2,3).
MG_HD float sinf_poly(double x, double x2, bool neg, int n) {
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double t1 = s2 + x2 * s3;
        double x7 = x3 * x2;
        double s = x + x3 * s1;
        return (float)(s + x7 * t1);
    } else {
        double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
               c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
        if (neg) { c0 = -c0; c1 = -c1; c2 = -c2; c3 = -c3; c4 = -c4; }
        double x4 = x2 * x2;
        double t2 = c3 + x2 * c4;
        double t1 = c0 + x2 * c1;
        double x6 = x4 * x2;
        double c = t1 + x4 * c2;
        return (float)(c + x6 * t2);
    }
}

// |x| < 120: x - n*pi/2 with n = round(x * 2/pi), computed through a 2^24-scaled
// float->int conversion so that negative values round correctly.
MG_HD double reduce_fast(double x, int* np) {
    double r = x * sc_hpi_inv();
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * sc_hpi();
}

// 2/pi to 192 bits, 8 new bits per entry.
MG_HD uint32_t inv_pio4(int i) {
    const uint32_t t[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                            0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                            0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                            0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    return t[i];
}

// |x| >= 120 (finite): 96-bit product with the matching window of 2/pi.
MG_HD double reduce_large(uint32_t xi, int* np) {
    int idx = (int)((xi >> 26) & 15);
    int shift = (int)((xi >> 23) & 7);
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * inv_pio4(idx));
    res1 = (uint64_t)xi * inv_pio4(idx + 4);
    res2 = (uint64_t)xi * inv_pio4(idx + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * sc_pi63();
}

MG_HD double sc_sign(int q) { return (q == 1 || q == 2) ? -1.0 : 1.0; }  // {1,-1,-1,1}[q&3]

#if defined(__HIPCC__)
#define MG_COLD static __host__ __device__ __attribute__((noinline))
#else
#define MG_COLD static __attribute__((noinline))
#endif
// |x| >= 120: out of line (rare; the 2/pi table reduction would otherwise cost registers on every hot path)
MG_COLD float mg_sincosf_large(float y, int want_cos) {
    if (!(abstop12(y) < abstop12(as_f32(0x7f800000u)))) return y - y;  // inf/nan -> nan
    uint32_t xi = as_u32(y);
    int sign = (int)(xi >> 31), n;
    double x = reduce_large(xi, &n);
    double s = sc_sign((n + sign) & 3);
    return sinf_poly(x * s, x * x, ((n + sign) & 2) != 0, want_cos ? (n ^ 1) : n);
}

MG_HD float mg_sinf(float y) {
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sinf_poly(x, s, false, 0);
    } else if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, &n);
        double s = sc_sign(n & 3);
        return sinf_poly(x * s, x * x, (n & 2) != 0, n);
    }
    return mg_sincosf_large(y, 0);
}

MG_HD float mg_cosf(float y) {
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sinf_poly(x, x2, false, 1);
    } else if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, &n);
        double s = sc_sign(n & 3);
        return sinf_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
    }
    return mg_sincosf_large(y, 1);
}

// sin and cos of the same argument, sharing the range reduction (glibc sincosf:
// identical per-output arithmetic to sinf/cosf).  The |x| >= pi/4 paths are kept out of line on the
// device: the environments' angles are almost always small, and inlining the table reduction into
// every kernel costs ~30 VGPRs of occupancy on the hot path.
MG_COLD void mg_sincosf_reduced(float y, float* sp, float* cp) {
    double x = y;
    int n, q;
    if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, &n);
        q = n;
    } else if (abstop12(y) < abstop12(as_f32(0x7f800000u))) {
        uint32_t xi = as_u32(y);
        x = reduce_large(xi, &n);
        q = n + (int)(xi >> 31);
    } else {
        *sp = *cp = y - y;
        return;
    }
    double s = sc_sign(q & 3);
    bool neg = (q & 2) != 0;
    double xs = x * s, x2 = x * x;
    *sp = sinf_poly(xs, x2, neg, n);
    *cp = sinf_poly(xs, x2, neg, n ^ 1);
}
MG_HD void mg_sincosf(float y, float* sp, float* cp) {
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double x = y;
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) { *sp = y; *cp = 1.0f; return; }
        *sp = sinf_poly(x, x2, false, 0);
        *cp = sinf_poly(x, x2, false, 1);
        return;
    }
    mg_sincosf_reduced(y, sp, cp);
}

// Same results as mg_sincosf, arranged for lock-step execution (LunarLander: a wave's lanes hold angles on both
// sides of pi/4 and in different quadrants, so every branch of the routine above runs for every wave).  For
// |x| < 120 the range reduction is applied unconditionally — below pi/4 it yields n = 0 and x - 0*pi/2 = x, i.e.
// exactly the small-argument path — both polynomials are evaluated once, the quadrant swaps/negates them by
// select (negating every cosine coefficient equals negating the rounded result: round-to-nearest is symmetric).
// tests/test_math_host.py checks it bit-for-bit against mg_sincosf over the floats.
MG_HD void mg_sincosf_u(float y, float* sp, float* cp) {
    if (abstop12(y) < abstop12(120.0f)) {
        // The two polynomials are evaluated with each  a + b*c  pair fused (12 f64 operations instead of 19).  The fused
        // values differ from the individually rounded ones by < 2^-52 relative, which would change the f32 rounding of the
        // result only within that distance of a rounding boundary: enumerating every f32 with |y| < 120 shows that this
        // happens for none of them (tests/native/cartpole_fast_check.cpp, group sincos_u; mgym_selftest_cartpole_math
        // repeats the enumeration on the GPU).  The range reduction stays unfused.
        int n;
        const double xr = reduce_fast((double)y, &n);
        const double x2 = xr * xr;
        const double xs = ((n & 3) == 1 || (n & 3) == 2) ? -xr : xr;
        const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
        const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
                     c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
        const double x3 = xs * x2;
        const double ts = cp_fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double sv = cp_fma(x3, s1, xs);
        const float sinv = (float)cp_fma(x7, ts, sv);
        const double x4 = x2 * x2;
        const double t2 = cp_fma(x2, c4, c3);
        const double t1 = cp_fma(x2, c1, c0);
        const double x6 = x4 * x2;
        const double cv = cp_fma(x4, c2, t1);
        const double cr = cp_fma(x6, t2, cv);
        const float cosv = (float)((n & 2) ? -cr : cr);
        const bool tiny = abstop12(y) < abstop12(0x1p-12f);
        const bool odd = (n & 1) != 0;
        *sp = tiny ? y : (odd ? cosv : sinv);
        *cp = tiny ? 1.0f : (odd ? sinv : cosv);
        return;
    }
    mg_sincosf_reduced(y, sp, cp);
}

// cosine alone, lock-step and fused like mg_sincosf_u (MountainCar: cos(3 * position), mountain_car.rs:302; the positions
// of a wave spread over several quadrants, so the branchy mg_cosf runs every path).  Bit-identical to mg_cosf for every
// f32 (|y| < 120 by the same enumeration; beyond, it IS mg_cosf's out-of-line path).
MG_HD float mg_cosf_u(float y) {
    if (abstop12(y) < abstop12(120.0f)) {
        int n;
        const double xr = reduce_fast((double)y, &n);
        const double x2 = xr * xr;
        const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
        const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
                     c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
        const double x3 = xr * x2;
        const double ts = cp_fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double sv = cp_fma(x3, s1, xr);
        const double sinv = cp_fma(x7, ts, sv);
        const double x4 = x2 * x2;
        const double t2 = cp_fma(x2, c4, c3);
        const double t1 = cp_fma(x2, c1, c0);
        const double x6 = x4 * x2;
        const double cv = cp_fma(x4, c2, t1);
        const double cosv = cp_fma(x6, t2, cv);
        const int q = n & 3;  // cos(y) = {cos, -sin, -cos, sin}[q](xr)
        double r = (q & 1) ? sinv : cosv;
        r = (q == 1 || q == 2) ? -r : r;
        return (float)r;
    }
    return mg_sincosf_large(y, 1);
}

// ---- expm1f / tanhf: fdlibm (Sun Microsystems) float versions as shipped by glibc ----
MG_HD float mg_expm1f(float x) {
    const float one = 1.0f, huge = 1.0e+30f, tiny = 1.0e-30f;
    const float o_threshold = 8.8721679688e+01f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f,
                invln2 = 1.4426950216e+00f;
    const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
                Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
    float y, hi, lo, c = 0.0f, t, e, hxs, hfx, r1;
    int32_t k;
    uint32_t hx = as_u32(x);
    uint32_t xsb = hx & 0x80000000u;
    hx &= 0x7fffffffu;

    if (hx >= 0x4195b844u) {          // |x| >= 27 ln2
        if (hx >= 0x42b17218u) {      // |x| >= 88.72
            if (hx > 0x7f800000u) return x + x;
            if (hx == 0x7f800000u) return (xsb == 0) ? x : -1.0f;
            if (x > o_threshold) return huge * huge;
        }
        if (xsb != 0) return tiny - one;
    }

    if (hx > 0x3eb17218u) {           // |x| > 0.5 ln2
        if (hx < 0x3F851592u) {       // |x| < 1.5 ln2
            if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
            else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
        } else {
            k = (int32_t)(invln2 * x + ((xsb == 0) ? 0.5f : -0.5f));
            t = (float)k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (hx < 0x33000000u) {    // |x| < 2^-25
        t = huge + x;
        return x - (t - (huge + x));
    } else {
        k = 0;
    }

    hfx = 0.5f * x;
    hxs = x * hfx;
    r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    t = 3.0f - r1 * hfx;
    e = hxs * ((r1 - t) / (6.0f - x * t));
    if (k == 0) return x - (x * e - hxs);
    e = (x * (e - c) - c);
    e -= hxs;
    if (k == -1) return 0.5f * (x - e) - 0.5f;
    if (k == 1) {
        if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
        return one + 2.0f * (x - e);
    }
    if (k <= -2 || k > 56) {
        y = one - (e - x);
        y = as_f32(as_u32(y) + ((uint32_t)k << 23));
        return y - one;
    }
    if (k < 23) {
        t = as_f32(0x3f800000u - (0x1000000u >> k));
        y = t - (e - x);
        y = as_f32(as_u32(y) + ((uint32_t)k << 23));
    } else {
        t = as_f32((uint32_t)(0x7f - k) << 23);
        y = x - (e + t);
        y += one;
        y = as_f32(as_u32(y) + ((uint32_t)k << 23));
    }
    return y;
}

MG_HD float mg_tanhf(float x) {
    const float one = 1.0f, two = 2.0f, tiny = 1.0e-30f;
    float t, z;
    uint32_t jx = as_u32(x);
    uint32_t ix = jx & 0x7fffffffu;
    bool pos = (jx >> 31) == 0;
    if (ix >= 0x7f800000u) return pos ? one / x + one : one / x - one;
    if (ix < 0x41b00000u) {           // |x| < 22
        if (ix == 0) return x;
        if (ix < 0x24000000u) return x * (one + x);
        float ax = as_f32(ix);
        if (ix >= 0x3f800000u) {
            t = mg_expm1f(two * ax);
            z = one - two / (t + two);
        } else {
            t = mg_expm1f(-two * ax);
            z = -t / (t + two);
        }
    } else {
        z = one - tiny;
    }
    return pos ? z : -z;
}

}  // namespace mgym
