/* rng.c — Philox4x32-10 restated from Salmon, Moraes, Dror, Shaw, "Parallel random
 * numbers: as easy as 1, 2, 3" (SC'11), and the uniform conversions.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference draws from candle's Tensor::rand (cartpole.rs:240, mountain_car.rs:281)
 * and rand 0.9.2 StdRng/ThreadRng (lunar_lander.rs:60-88); neither stream can be
 * reproduced here (parity unpinned), so only the DISTRIBUTIONS follow the reference:
 *   f64 uniform: 53-bit mantissa in [0,1), value*scale+low;
 *   f32 uniform: rand 0.9 UniformFloat<f32>::sample_single = (top 23 bits as [1,2)) - 1,
 *                then value0_1*scale + low.
 */
#include "oracle.h"

static void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void ora_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double ora_u53(uint32_t hi, uint32_t lo) {
    uint64_t m = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return (double)m * (1.0 / 9007199254740992.0);
}

float ora_u23(uint32_t w) {
    return (float)(w >> 9) * (1.0f / 8388608.0f);
}
