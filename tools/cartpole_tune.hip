// tools/cartpole_tune.hip — kernel-variant microbenchmark for the CartPole step (NOT product code).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/cartpole_tune.hip -o tools/cartpole_tune
#include "../modurl_gym_amd/csrc/cartpole.hip"
#include <vector>
#include <stdio.h>
namespace mgym {
void set_last_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); }
int hip_fail(hipError_t e, const char* what, const char*, int) { fprintf(stderr, "HIP %d %s\n", (int)e, what); return 4; }
Env* make_mountaincar(bool) { return nullptr; }
Env* make_lunarlander() { return nullptr; }

// ceiling: same traffic, no arithmetic
__global__ void __launch_bounds__(kBlock) copy_kernel(CartPoleDev d, const uint32_t* act, float* rew, uint8_t* dn, uint8_t* tr) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i0 = g * 4;
    if (i0 >= d.n) return;
    float4 vx = *(const float4*)(d.x + i0), vxd = *(const float4*)(d.xd + i0), vth = *(const float4*)(d.th + i0), vthd = *(const float4*)(d.thd + i0);
    uint4 vc = *(const uint4*)(d.ctr + i0), va = *(const uint4*)(act + i0);
    vx.x += 1.f; vxd.y += 1.f; vth.z += 1.f; vthd.w += 1.f; vc.x += va.x;
    *(float4*)(d.x + i0) = vx; *(float4*)(d.xd + i0) = vxd; *(float4*)(d.th + i0) = vth; *(float4*)(d.thd + i0) = vthd; *(uint4*)(d.ctr + i0) = vc;
    *(float4*)(rew + i0) = vx; *(uint32_t*)(dn + i0) = vc.y; *(uint32_t*)(tr + i0) = vc.z;
}

// software-pipelined persistent variant: each thread walks groups g, g+stride, ... and issues the loads of
// the next group before computing the current one
struct G4 { float4 x, xd, th, thd; uint4 c, a; };
__device__ __forceinline__ G4 load4(const CartPoleDev& d, const uint32_t* act, uint64_t i0) {
    G4 r; r.x = *(const float4*)(d.x + i0); r.xd = *(const float4*)(d.xd + i0); r.th = *(const float4*)(d.th + i0);
    r.thd = *(const float4*)(d.thd + i0); r.c = *(const uint4*)(d.ctr + i0); r.a = *(const uint4*)(act + i0); return r;
}
__global__ void __launch_bounds__(kBlock) pipe_kernel(CartPoleDev d, const uint32_t* __restrict__ act, float* __restrict__ rew, uint8_t* __restrict__ dn_o, uint8_t* __restrict__ tr_o) {
    const uint64_t groups = d.n / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    bool bad = false;
    G4 cur = load4(d, act, g * 4);
    for (; g < groups; g += stride) {
        uint64_t i0 = g * 4;
        G4 nxt = cur;
        if (g + stride < groups) nxt = load4(d, act, (g + stride) * 4);
        CartPoleLane s[4] = {{cur.x.x, cur.xd.x, cur.th.x, cur.thd.x, cur.c.x}, {cur.x.y, cur.xd.y, cur.th.y, cur.thd.y, cur.c.y},
                             {cur.x.z, cur.xd.z, cur.th.z, cur.thd.z, cur.c.z}, {cur.x.w, cur.xd.w, cur.th.w, cur.thd.w, cur.c.w}};
        uint32_t a[4] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w}, dn[4], tr[4]; float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { cartpole_step_one(d.p, s[k], a[k], r[k], dn[k], tr[k], bad); if (d.auto_reset && (dn[k] | tr[k])) cartpole_reset_one(d, i0 + k, s[k]); }
        *(float4*)(d.x + i0) = make_float4(s[0].x, s[1].x, s[2].x, s[3].x); *(float4*)(d.xd + i0) = make_float4(s[0].xd, s[1].xd, s[2].xd, s[3].xd);
        *(float4*)(d.th + i0) = make_float4(s[0].th, s[1].th, s[2].th, s[3].th); *(float4*)(d.thd + i0) = make_float4(s[0].thd, s[1].thd, s[2].thd, s[3].thd);
        *(uint4*)(d.ctr + i0) = make_uint4(s[0].ctr, s[1].ctr, s[2].ctr, s[3].ctr);
        *(float4*)(rew + i0) = make_float4(r[0], r[1], r[2], r[3]);
        *(uint32_t*)(dn_o + i0) = dn[0] | (dn[1] << 8) | (dn[2] << 16) | (dn[3] << 24); *(uint32_t*)(tr_o + i0) = tr[0] | (tr[1] << 8) | (tr[2] << 16) | (tr[3] << 24);
        cur = nxt;
    }
    if (__any(bad)) { if ((threadIdx.x & 63) == 0) atomicOr(d.err, 1u); }
}
}  // namespace mgym
using namespace mgym;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], 0, 0) : (1u << 20);
    int reps = argc > 2 ? atoi(argv[2]) : 50;
    CartPoleEnv env; env.cfg.struct_size = sizeof(mgym_config); env.cfg.flags = MGYM_FLAG_AUTO_RESET; env.cfg.is_euler = 1; env.cfg.seed = 1;
    uint64_t padw = argc > 3 ? strtoull(argv[3], 0, 0) : 0;
    int cap = argc > 4 ? atoi(argv[4]) : kMaxBlocks;
    env.n = n; env.n_pad = round_up(n, 1024) + padw;
    CK(hipStreamCreate(&env.stream)); CK(hipMalloc((void**)&env.d_err, 4)); CK(hipMemset(env.d_err, 0, 4));
    env.init();
    const int RING = 16;
    uint32_t* act; float* rew; uint8_t *dn, *tr;
    CK(hipMalloc((void**)&act, RING * n * 4)); CK(hipMalloc((void**)&rew, n * 4)); CK(hipMalloc((void**)&dn, n)); CK(hipMalloc((void**)&tr, n));
    { std::vector<uint32_t> h(RING * n); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 16) & 1; } CK(hipMemcpy(act, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    env.reset(nullptr, nullptr, true, nullptr);
    CK(hipStreamSynchronize(env.stream));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(env.stream, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < RING; ++k) launch(act + (uint64_t)k * n);
        CK(hipStreamEndCapture(env.stream, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, env.stream));
        CK(hipStreamSynchronize(env.stream));
        float best = 1e9, tot = 0;
        for (int rnd = 0; rnd < 5; ++rnd) {
            CK(hipEventRecord(e0, env.stream));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, env.stream));
            CK(hipEventRecord(e1, env.stream)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= (reps * RING); tot += ms; if (ms < best) best = ms;
        }
        printf("%-28s n=%llu  %.2f us/step (best %.2f)  %.0f GB/s alg  %.3e steps/s\n", name, (unsigned long long)n, tot / 5 * 1e3, best * 1e3, 50.0 * n / (tot / 5 * 1e-3) / 1e9, n / (tot / 5 * 1e-3));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    auto& d = env.dev;
    uint64_t nb = ((n + 3) / 4 + 255) / 256; if (nb > (uint64_t)cap) nb = cap;
    dim3 gv((unsigned)nb), b(256);
    printf("pad words %llu, blocks %u\n", (unsigned long long)padw, gv.x);
    for (int pass = 0; pass < 1; ++pass) {
    timeit("copy ceiling (no math)", [&](uint32_t* a) { hipLaunchKernelGGL(copy_kernel, dim3((n / 4 + 255) / 256), dim3(256), 0, env.stream, d, a, rew, dn, tr); });
    env.reset(nullptr, nullptr, true, nullptr);
    timeit("vec4 rmode1 (inline reset)", [&](uint32_t* a) { hipLaunchKernelGGL((cartpole_step_kernel<4, 1, false>), gv, b, 0, env.stream, d, a, nullptr, rew, dn, tr); });
    env.reset(nullptr, nullptr, true, nullptr);
    timeit("vec4 rmode1 NT", [&](uint32_t* a) { hipLaunchKernelGGL((cartpole_step_kernel<4, 1, true>), gv, b, 0, env.stream, d, a, nullptr, rew, dn, tr); });
    env.reset(nullptr, nullptr, true, nullptr);
    timeit("vec4 rmode2 (LDS compaction)", [&](uint32_t* a) { hipLaunchKernelGGL((cartpole_step_kernel<4, 2, false>), gv, b, 0, env.stream, d, a, nullptr, rew, dn, tr); });
    env.reset(nullptr, nullptr, true, nullptr);
    timeit("vec4 rmode2 NT", [&](uint32_t* a) { hipLaunchKernelGGL((cartpole_step_kernel<4, 2, true>), gv, b, 0, env.stream, d, a, nullptr, rew, dn, tr); });
    env.reset(nullptr, nullptr, true, nullptr);
    // population split into S independent sub-shards on S streams (fork/join inside one graph): the dependent-launch
    // gap of one chain overlaps the other chain's kernel
    for (int S : {2, 4}) {
        if (n % (1024ull * S)) continue;
        static hipStream_t ss[4]; static hipEvent_t ef, ej[4];
        for (int q = 1; q < S; ++q) if (!ss[q]) CK(hipStreamCreate(&ss[q]));
        if (!ef) { CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); for (int q = 0; q < 4; ++q) CK(hipEventCreateWithFlags(&ej[q], hipEventDisableTiming)); }
        const uint64_t h = n / S;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(env.stream, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(ef, env.stream));
        for (int q = 1; q < S; ++q) CK(hipStreamWaitEvent(ss[q], ef, 0));
        for (int k = 0; k < RING; ++k)
            for (int q = 0; q < S; ++q) {
                CartPoleDev dq = d; dq.x += q * h; dq.xd += q * h; dq.th += q * h; dq.thd += q * h; dq.ctr += q * h; dq.episode += q * h; dq.n = h; dq.env_id_base += q * h;
                hipLaunchKernelGGL((cartpole_step_kernel<4, 2, true>), dim3((unsigned)(h / 1024)), b, 0, q ? ss[q] : env.stream, dq, act + (uint64_t)k * n + q * h, nullptr, rew + q * h, dn + q * h, tr + q * h);
            }
        for (int q = 1; q < S; ++q) { CK(hipEventRecord(ej[q], ss[q])); CK(hipStreamWaitEvent(env.stream, ej[q], 0)); }
        CK(hipStreamEndCapture(env.stream, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, env.stream));
        CK(hipStreamSynchronize(env.stream));
        float tot = 0, best = 1e9;
        for (int rnd = 0; rnd < 5; ++rnd) {
            CK(hipEventRecord(e0, env.stream));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, env.stream));
            CK(hipEventRecord(e1, env.stream)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= (reps * RING); tot += ms; if (ms < best) best = ms;
        }
        printf("vec4 rmode2 NT, %d sub-shards on %d streams  n=%llu  %.2f us/step (best %.2f)  %.0f GB/s alg  %.3e steps/s\n", S, S, (unsigned long long)n, tot / 5 * 1e3, best * 1e3, 50.0 * n / (tot / 5 * 1e-3) / 1e9, n / (tot / 5 * 1e-3));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        env.reset(nullptr, nullptr, true, nullptr);
    }
    timeit("vec4 rmode0 + reset_done", [&](uint32_t* a) { hipLaunchKernelGGL((cartpole_step_kernel<4, 0, false>), gv, b, 0, env.stream, d, a, nullptr, rew, dn, tr);
        hipLaunchKernelGGL(cartpole_reset_kernel, gv, b, 0, env.stream, d, dn, tr, 0, 1, nullptr); });
    env.reset(nullptr, nullptr, true, nullptr);
    timeit("vec4 rmode0 NT + reset_done", [&](uint32_t* a) { hipLaunchKernelGGL((cartpole_step_kernel<4, 0, true>), gv, b, 0, env.stream, d, a, nullptr, rew, dn, tr);
        hipLaunchKernelGGL(cartpole_reset_kernel, gv, b, 0, env.stream, d, dn, tr, 0, 1, nullptr); });
    env.reset(nullptr, nullptr, true, nullptr);
    timeit("vec1 rmode1", [&](uint32_t* a) { hipLaunchKernelGGL((cartpole_step_kernel<1, 1, false>), dim3(grid_for(n)), b, 0, env.stream, d, a, nullptr, rew, dn, tr); });
    env.reset(nullptr, nullptr, true, nullptr);
    }
    timeit("pipelined persistent (inline reset)", [&](uint32_t* a) { hipLaunchKernelGGL(pipe_kernel, gv, b, 0, env.stream, d, a, rew, dn, tr); });
    env.reset(nullptr, nullptr, true, nullptr);
    timeit("copy ceiling again (no math)", [&](uint32_t* a) { hipLaunchKernelGGL(copy_kernel, dim3((n / 4 + 255) / 256), dim3(256), 0, env.stream, d, a, rew, dn, tr); });
    env.dev.auto_reset = 0; env.reset(nullptr, nullptr, true, nullptr);
    return 0;
}
