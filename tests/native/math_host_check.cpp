// Host-side check of modurl_gym_amd/csrc/mgym_math.h against the container's glibc.
// Built and run by tests/test_math_host.py (no GPU).  Prints one line per function:
//   <name> total=<n> bad_fast=<n> bad_mid=<n> bad_large=<n>
// fast: |x| <= 16 (must be 0: covers every angle the environments produce);
// mid : 16 < |x| < 120 (glibc's FMA ifunc variant rounds a handful of inputs differently);
// large: |x| >= 120 (table reduction).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../modurl_gym_amd/csrc/mgym_math.h"
using namespace mgym;

int main(int argc, char** argv) {
    uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 7;
    unsigned long tot = 0, bs[3] = {0, 0, 0}, bc[3] = {0, 0, 0}, bsc = 0, bt[3] = {0, 0, 0}, bu = 0;
    for (uint64_t u = 0; u < 0x7f800000ull; u += stride) {
        for (int sg = 0; sg < 2; ++sg) {
            float f = as_f32((uint32_t)u | ((uint32_t)sg << 31));
            float af = fabsf(f);
            int cls = af <= 16.0f ? 0 : (af < 120.0f ? 1 : 2);
            float s, c;
            mg_sincosf(f, &s, &c);
            if (as_u32(sinf(f)) != as_u32(mg_sinf(f))) bs[cls]++;
            if (as_u32(cosf(f)) != as_u32(mg_cosf(f))) bc[cls]++;
            if (as_u32(s) != as_u32(mg_sinf(f)) || as_u32(c) != as_u32(mg_cosf(f))) bsc++;
            float su, cu;
            mg_sincosf_u(f, &su, &cu);
            if (as_u32(su) != as_u32(s) || as_u32(cu) != as_u32(c)) bu++;
            float th = tanhf(f), mt = mg_tanhf(f);
            if (as_u32(th) != as_u32(mt)) bt[cls]++;
            tot++;
        }
    }
    printf("sinf total=%lu bad_fast=%lu bad_mid=%lu bad_large=%lu\n", tot, bs[0], bs[1], bs[2]);
    printf("cosf total=%lu bad_fast=%lu bad_mid=%lu bad_large=%lu\n", tot, bc[0], bc[1], bc[2]);
    printf("sincosf_self total=%lu bad_fast=%lu bad_mid=0 bad_large=0\n", tot, bsc);
    printf("sincosf_uniform_vs_sincosf total=%lu bad_fast=%lu bad_mid=0 bad_large=0\n", tot, bu);
    printf("tanhf total=%lu bad_fast=%lu bad_mid=%lu bad_large=%lu\n", tot, bt[0], bt[1], bt[2]);
    return 0;
}
