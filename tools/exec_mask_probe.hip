// tools/exec_mask_probe.hip — does a wave64 VALU instruction get cheaper when whole 16-lane groups of EXEC are off?
// One wave per block, a dependent chain of v_fma_f32 executed by the first `active` lanes only; prints cycles per instruction.
// (GPU box)  hipcc --offload-arch=gfx950 -O3 tools/exec_mask_probe.hip -o tools/_exec_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP>
__global__ void chain(float* out, unsigned long long* cyc, int active, int iters) {
    float x[ILP];
    for (int j = 0; j < ILP; ++j) x[j] = (float)(threadIdx.x + j) * 1e-3f;
    const float a = 1.0001f, b = 1e-7f;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 64 / ILP; ++k)
#pragma unroll
                for (int j = 0; j < ILP; ++j) x[j] = __builtin_fmaf(x[j], a, b);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.0f;
        for (int j = 0; j < ILP; ++j) s += x[j];
        out[blockIdx.x * 64 + threadIdx.x] = s;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP>
void run(float* out, unsigned long long* cyc, int blocks) {
    unsigned long long h[256];
    for (int active : {64, 48, 33, 32, 17, 16, 1}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(chain<ILP>, dim3(blocks), dim3(64), 0, 0, out, cyc, active, 1000);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
        printf("%d independent chain(s), active lanes %2d: %.2f counter ticks per v_fma_f32 (wave64, one wave per SIMD)\n", ILP, active, s / blocks / (1000.0 * 64));
    }
}
int main() {
    float* out; unsigned long long* cyc;
    const int blocks = 256;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    run<1>(out, cyc, blocks); run<2>(out, cyc, blocks); run<4>(out, cyc, blocks);
    return 0;
}
