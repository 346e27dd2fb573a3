// tools/exec_mask_probe.hip — does a wave64 VALU instruction get cheaper when whole 16-lane groups of EXEC are off?
// One wave per block, a dependent chain of v_fma_f32 executed by the first `active` lanes only; prints cycles per instruction.
// (GPU box)  hipcc --offload-arch=gfx950 -O3 tools/exec_mask_probe.hip -o tools/_exec_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain(float* out, unsigned long long* cyc, int active, int iters) {
    float x = (float)threadIdx.x * 1e-3f, a = 1.0001f, b = 1e-7f;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, a, b);
        }
        t1 = __builtin_readcyclecounter();
        out[blockIdx.x * 64 + threadIdx.x] = x;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    const int blocks = 256;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    unsigned long long h[blocks];
    for (int active : {64, 48, 33, 32, 17, 16, 1}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, out, cyc, active, 1000);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
        printf("active lanes %2d: %.2f cycles per dependent v_fma_f32 (wave64, one wave per SIMD)\n", active, s / blocks / (1000.0 * 64));
    }
    return 0;
}
