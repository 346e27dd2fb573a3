// Exhaustive host-side proof that the cheaper CartPole instruction sequences (modurl_gym_amd/csrc/cartpole_math.h)
// are bit-identical to the reference-form arithmetic.  Built with -mfma (hardware fused multiply-add = the IEEE
// operation the GPU executes) and run by tests/test_oracle_classic.py.  Usage: cartpole_fast_check <threads> [stride]
//   sincos_u : every f32 with |y| < 120 (both signs): the fused lock-step mg_sincosf_u / mg_cosf_u == mg_sincosf / mg_cosf
//   sincos : every f32 with |y| < 0.75 (both signs): cp_sincos_small == mg_sincosf                (2 x 0x3f400000 inputs)
//   divc   : every f32 x in the proven range: cp_div_const(x, 1.1f) == x / 1.1f                   (2^32 inputs)
//   div    : every divisor d the step can form (d = 0.5f*(4/3 - 0.1f*c*c/1.1f), c any f32 in [-1,1]) against
//            numerators sampled per divisor (stride-able) plus hard cases: cp_div(n, d) == n / d
//   step   : the whole fast-form step against the reference-form step on random states the guard admits (live region,
//            tiny/denormal/huge magnitudes, +-0, counters around the truncation edge), all four <EULER, SB> variants
// Prints "<name> checked=<n> mismatches=<m>" per group; exit code 0 iff every m is 0.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <thread>
#include <vector>
#include "../../modurl_gym_amd/csrc/cartpole_step.h"
using namespace mgym;

static std::atomic<unsigned long> bad_su{0}, n_su{0}, bad_sc{0}, n_sc{0}, bad_dc{0}, n_dc{0}, bad_dv{0}, n_dv{0}, n_div_d{0}, bad_st{0}, n_st{0};

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
// a float with a uniformly drawn exponent in [lo_e, hi_e] (biased), random mantissa and sign
static float rnd_float(uint32_t& s, uint32_t lo_e, uint32_t hi_e) {
    const uint32_t b = lcg(s), e = lo_e + (lcg(s) >> 8) % (hi_e - lo_e + 1u);
    return as_f32((b & 0x807fffffu) | (e << 23));
}
template <bool EULER, bool SB>
static void step_group(uint32_t seed, unsigned long count, unsigned long& bad, unsigned long& checked) {
    const CartPoleParams p = cartpole_params(EULER ? 1 : 0, SB ? 1 : 0);
    uint32_t s = seed;
    for (unsigned long it = 0; it < count; ++it) {
        float x, xd, th, thd;
        const uint32_t mode = lcg(s) >> 29;
        if (mode < 5) {  // the live region of the environment
            x = ((int32_t)lcg(s)) * (3.0f / 2147483648.0f); xd = ((int32_t)lcg(s)) * (4.0f / 2147483648.0f);
            th = ((int32_t)lcg(s)) * (0.3f / 2147483648.0f); thd = ((int32_t)lcg(s)) * (4.0f / 2147483648.0f);
        } else {         // anything the guard admits: tiny / denormal / huge magnitudes
            x = rnd_float(s, 0, 200); xd = rnd_float(s, 0, 200);
            th = rnd_float(s, 0, 126); thd = rnd_float(s, 0, 166);
            if ((lcg(s) >> 30) == 0) th = as_f32(lcg(s) & 0x80000000u);   // +-0
            if ((lcg(s) >> 30) == 0) thd = as_f32(lcg(s) & 0x80000000u);
        }
        const uint32_t action = lcg(s) >> 31;
        uint32_t ctr = cp_ctr_pack((lcg(s) >> 8) % 1030u, lcg(s) >> 30, lcg(s));
        if ((lcg(s) >> 28) == 0) ctr = (ctr & ~kCpStepsMask) | (498u + (lcg(s) >> 30));   // around the truncation edge
        if (!cartpole_fast_ok(th, thd, action)) continue;
        float a[4] = {x, xd, th, thd}, b[4] = {x, xd, th, thd};
        uint32_t ca = ctr, cb = ctr, da, ta, db, tb;
        float ra, rb;
        cartpole_step_ref(p, a[0], a[1], a[2], a[3], ca, action, ra, da, ta);
        cartpole_step_fast<EULER, SB>(p, b[0], b[1], b[2], b[3], cb, action, rb, db, tb);
        bool same = ca == cb && da == db && ta == tb && as_u32(ra) == as_u32(rb);
        for (int k = 0; k < 4; ++k) same = same && as_u32(a[k]) == as_u32(b[k]);
        if (!same) {
            if (bad < 4) fprintf(stderr, "step<%d,%d> mismatch state (%a,%a,%a,%a) act %u ctr %08x: ref (%a,%a,%a,%a) r=%g d=%u t=%u ctr=%08x | fast (%a,%a,%a,%a) r=%g d=%u t=%u ctr=%08x\n",
                                 (int)EULER, (int)SB, x, xd, th, thd, action, ctr, a[0], a[1], a[2], a[3], ra, da, ta, ca, b[0], b[1], b[2], b[3], rb, db, tb, cb);
            ++bad;
        }
        ++checked;
    }
}

static void work(int tid, int nthreads, uint32_t stride) {
    unsigned long b_sc = 0, c_sc = 0, b_dc = 0, c_dc = 0;
    // ---- sincos: |y| < 0.75 ----
    for (uint64_t u = (uint64_t)tid * stride; u < 0x3f400000ull; u += (uint64_t)nthreads * stride) {
        for (int sg = 0; sg < 2; ++sg) {
            const float y = as_f32((uint32_t)u | ((uint32_t)sg << 31));
            float s0, c0, s1, c1;
            mg_sincosf(y, &s0, &c0);
            cp_sincos_small(y, &s1, &c1);
            if (as_u32(s0) != as_u32(s1) || as_u32(c0) != as_u32(c1)) {
                if (b_sc < 4) fprintf(stderr, "sincos mismatch y=%a: ref (%a,%a) fast (%a,%a)\n", y, s0, c0, s1, c1);
                ++b_sc;
            }
            ++c_sc;
        }
    }
    // ---- fused lock-step sincos / cos of the LunarLander and MountainCar kernels: every f32 with |y| < 120 ----
    {
        unsigned long b_su = 0, c_su = 0;
        for (uint64_t u = (uint64_t)tid * stride; u < 0x42f00000ull; u += (uint64_t)nthreads * stride) {
            for (int sg = 0; sg < 2; ++sg) {
                const float y = as_f32((uint32_t)u | ((uint32_t)sg << 31));
                float s0, c0, s1, c1;
                mg_sincosf(y, &s0, &c0);
                mg_sincosf_u(y, &s1, &c1);
                const float c2 = mg_cosf_u(y), c3 = mg_cosf(y);
                if (as_u32(s0) != as_u32(s1) || as_u32(c0) != as_u32(c1) || as_u32(c2) != as_u32(c3) || as_u32(c3) != as_u32(c0)) {
                    if (b_su < 4) fprintf(stderr, "sincos_u mismatch y=%a: ref (%a,%a) u (%a,%a) cos_u %a cos %a\n", y, s0, c0, s1, c1, c2, c3);
                    ++b_su;
                }
                ++c_su;
            }
        }
        bad_su += b_su; n_su += c_su;
    }
    // ---- x / total_mass ----
    const float M = 0.1f + 1.0f;  // cartpole.rs:48
    const float R = 1.0f / M;
    for (uint64_t u = (uint64_t)tid * stride; u < 0x100000000ull; u += (uint64_t)nthreads * stride) {
        const float x = as_f32((uint32_t)u);
        if (!cp_div_range_ok(x)) continue;
        const float ref = x / M, got = cp_div_const(x, M, R);
        if (as_u32(ref) != as_u32(got)) {
            if (b_dc < 4) fprintf(stderr, "div_const mismatch x=%a: ref %a fast %a\n", x, ref, got);
            ++b_dc;
        }
        ++c_dc;
    }
    bad_sc += b_sc; n_sc += c_sc; bad_dc += b_dc; n_dc += c_dc;
    // ---- n / d over every reachable divisor ----
    unsigned long b_dv = 0, c_dv = 0, nd = 0;
    uint32_t rng = 0x9E3779B9u * (uint32_t)(tid + 1);
    uint32_t last_d = 0;
    // cos(theta) takes every f32 in [-1, 1]; d depends on c*c only, enumerate c in [0, 1]
    for (uint64_t u = (uint64_t)tid; u <= 0x3f800000ull; u += (uint64_t)nthreads) {
        const float c = as_f32((uint32_t)u);
        const float d = 0.5f * (4.0f / 3.0f - 0.1f * c * c / M);  // cartpole.rs:269-270 (length * (4/3 - masspole*cos^2/total_mass))
        if (as_u32(d) == last_d) continue;  // consecutive c mostly give the same d
        last_d = as_u32(d);
        ++nd;
        for (int k = 0; k < 24; ++k) {
            rng = rng * 1664525u + 1013904223u;
            uint32_t bits = rng;
            // exponent spread over the proven range, random mantissa and sign; a few exact-quotient and boundary cases
            uint32_t ex = 0x0du * 8u + (bits >> 9) % (0x71u * 8u - 0x0du * 8u);  // biased exponent*... keep inside [2^-100, 2^100]
            uint32_t e8 = 27u + (ex % 200u);                                       // biased exponent 27..226  (2^-100 .. 2^99)
            float n = as_f32((bits & 0x807fffffu) | (e8 << 23));
            if (k == 0) n = 0.0f;
            if (k == 1) n = -0.0f;
            if (k == 2) n = d;              // quotient exactly 1
            if (k == 3) n = d * 3.0f;       // near-exact quotients
            if (k == 4) n = as_f32(as_u32(d) + 1u);
            if (k == 5) n = -as_f32(as_u32(d) - 1u);
            if (!cp_div_range_ok(n)) continue;
            const float ref = n / d, y0 = 1.0f / d;
            // any reciprocal estimate within 1 ulp (the hardware v_rcp_f32 is one such)
            const float est[3] = {y0, as_f32(as_u32(y0) + 1u), as_f32(as_u32(y0) - 1u)};
            for (int q = 0; q < 3; ++q) {
                const float got = cp_div_from_estimate(n, d, est[q]);
                if (as_u32(ref) != as_u32(got)) {
                    if (b_dv < 4) fprintf(stderr, "div mismatch n=%a d=%a est %d: ref %a fast %a\n", n, d, q, ref, got);
                    ++b_dv;
                }
                ++c_dv;
            }
        }
    }
    bad_dv += b_dv; n_dv += c_dv; n_div_d += nd;
    // ---- whole step: fast form == reference form on random states the guard admits ----
    unsigned long b_st = 0, c_st = 0;
    const unsigned long per = 40000000ul / stride / (unsigned long)nthreads + 1000ul;
    step_group<true, false>(0x1234567u + (uint32_t)tid, per, b_st, c_st);
    step_group<true, true>(0x2345678u + (uint32_t)tid, per / 4, b_st, c_st);
    step_group<false, false>(0x3456789u + (uint32_t)tid, per / 4, b_st, c_st);
    step_group<false, true>(0x456789au + (uint32_t)tid, per / 4, b_st, c_st);
    bad_st += b_st; n_st += c_st;
}

int main(int argc, char** argv) {
    const int nthreads = argc > 1 ? atoi(argv[1]) : 4;
    const uint32_t stride = argc > 2 ? (uint32_t)atoi(argv[2]) : 1;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t, nthreads, stride);
    for (auto& t : th) t.join();
    printf("sincos_u checked=%lu mismatches=%lu\n", n_su.load(), bad_su.load());
    printf("sincos_small checked=%lu mismatches=%lu\n", n_sc.load(), bad_sc.load());
    printf("div_const checked=%lu mismatches=%lu\n", n_dc.load(), bad_dc.load());
    printf("div checked=%lu mismatches=%lu divisors=%lu\n", n_dv.load(), bad_dv.load(), n_div_d.load());
    printf("step checked=%lu mismatches=%lu\n", n_st.load(), bad_st.load());
    return (bad_su.load() || bad_sc.load() || bad_dc.load() || bad_dv.load() || bad_st.load()) ? 1 : 0;
}
