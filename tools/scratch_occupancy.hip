// tools/scratch_occupancy.hip — diagnostic: how many waves run concurrently as a function of per-lane scratch size?
// Each 64-thread block touches a private array of S bytes per lane, then spins ~100 us; 4096 blocks are launched, so
// time / 100 us = number of residency rounds = ceil(4096 / concurrent waves).
// hipcc --offload-arch=gfx950 -O3 -o tools/scratch_occupancy tools/scratch_occupancy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int WORDS>
__global__ void __launch_bounds__(64) k(unsigned* out, int spin) {
    volatile unsigned a[WORDS];
    for (int i = 0; i < WORDS; ++i) a[i] = i + threadIdx.x;
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(32); }
    unsigned s = 0;
    for (int i = 0; i < WORDS; i += 17) s += a[(i + threadIdx.x) % WORDS];
    if (s == 0xdeadbeef) out[0] = s;
}
template <int WORDS>
static void run(unsigned* d, const char* tag) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int spin = 10000;  // 100 MHz wall clock: 100 us
    hipLaunchKernelGGL(k<WORDS>, dim3(4096), dim3(64), 0, 0, d, spin); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<WORDS>, dim3(4096), dim3(64), 0, 0, d, spin); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: scratch %5d B/lane (%4d KB/wave): %.3f ms for 4096 waves of 100 us => ~%.1f rounds => ~%.0f concurrent waves\n", tag, WORDS * 4, WORDS * 4 * 64 / 1024, ms, ms / 0.1, 4096.0 / (ms / 0.1));
}
int main() {
    unsigned* d; hipMalloc((void**)&d, 64);
    run<16>(d, "S=64"); run<128>(d, "S=512"); run<256>(d, "S=1K"); run<384>(d, "S=1.5K"); run<512>(d, "S=2K"); run<728>(d, "S=2912"); run<1024>(d, "S=4K"); run<2048>(d, "S=8K");
    return 0;
}
