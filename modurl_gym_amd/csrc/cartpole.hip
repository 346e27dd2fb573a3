// cartpole.hip — CartPoleV1 step()/reset() as struct-of-arrays gfx950 kernels.
//
// Replaces the bodies of `impl Gym for CartPoleV1` (reference src/classic_control/cartpole.rs:
// reset :238-249, step :251-348) for n_envs independent environments.
//
// Data layout in HBM (engine-owned, one allocation, column stride n_pad words):
//   x[n] x_dot[n] theta[n] theta_dot[n]   f32   — the state, which IS the observation (:301)
//   ctr[n]                                u32   — steps_since_reset | steps_beyond_terminated | episode
//                                                 (bit layout in cartpole_step.h)
// One lane owns 4 consecutive environments: every column is read and written with one
// 16-byte access per lane (1 KiB per wave instruction, fully coalesced).  The engine's columns are
// addressed through ONE buffer resource — descriptor and column / pass offsets in SGPRs, one constant
// VGPR lane offset, no per-lane address arithmetic; the physics constants (cartpole.rs:45-56: none is a
// builder argument) are compile-time constants; outputs leave with non-temporal stores.
// No MFMA: the path is element-wise.
//
// Algorithmic HBM bytes per env-step (obs_out == NULL, zero-copy observation): 50
//   state 16 R + 16 W, action 4 R, reward 4 W, done 1 W, truncated 1 W, ctr 4 R + 4 W.
// There is no other per-env word: the episode counter that keys the reset draws rides in `ctr`.
//
// Arithmetic: cartpole_step.h.  A wave whose 256 environments all pass cartpole_fast_ok() (theta on the
// small-argument path, bounded theta_dot, valid action: every live environment) runs the branch-free fast
// form, proven bit-identical to the reference form; any other wave runs the reference form for that pass.
//
// Done-mask handling (fused auto-reset, MGYM_FLAG_AUTO_RESET): finished environments are rare per ENV
// (~4.5 % per step under a random policy) but nearly every WAVE holds one, so resetting in place would make
// every wave pay Philox4x32-10 (quarter-rate integer multiplies) four times over.  Instead each wave
// compacts its finished envs with __ballot + popcount prefix sums into a wave-private LDS list (no block
// barrier, no atomics), the wave's first lanes draw all new states in ONE pass, owners read theirs back.
// The same ballots feed the per-handle finished-episode counter (mgym_episode_count).
#include "common.h"
#include "cartpole_step.h"
#include "philox.h"

namespace mgym {

struct CartPoleDev {
    float* base;        // columns x, x_dot, theta, theta_dot, ctr at base + k * n_pad (one pointer: SGPR budget)
    uint64_t n_pad;
    uint64_t n;
    uint64_t seed, env_id_base;
    uint32_t* err;
    unsigned long long* done_count;  // kDoneShards partial counts (finished env-steps)
    __host__ __device__ float* x() const { return base; }
    __host__ __device__ float* xd() const { return base + n_pad; }
    __host__ __device__ float* th() const { return base + 2 * n_pad; }
    __host__ __device__ float* thd() const { return base + 3 * n_pad; }
    __host__ __device__ uint32_t* ctr() const { return reinterpret_cast<uint32_t*>(base + 4 * n_pad); }
};

constexpr bool kNtStores = true;   // non-temporal output stores: 10.9 us vs 11.4 us at 1 Mi envs (round 1)
constexpr int kWaves = kBlock / 64;

struct CartPoleLane {
    float x, xd, th, thd;
    uint32_t ctr;
};

// reset(): cartpole.rs:238-249 — state ~ U[-0.05, 0.05)^4 sampled in f64 then cast,
// steps_since_reset = 0, steps_beyond_terminated = None.  `episode` = Philox counter word 2.
__device__ __forceinline__ void cartpole_draw(uint64_t seed, uint64_t gid, uint32_t episode, float out[4]) {
    Philox4 a = env_draw(seed, gid, episode, SLOT_RESET0);
    Philox4 b = env_draw(seed, gid, episode, SLOT_RESET1);
    out[0] = uniform_f64_to_f32(-0.05, 0.05, a.w[0], a.w[1]);
    out[1] = uniform_f64_to_f32(-0.05, 0.05, a.w[2], a.w[3]);
    out[2] = uniform_f64_to_f32(-0.05, 0.05, b.w[0], b.w[1]);
    out[3] = uniform_f64_to_f32(-0.05, 0.05, b.w[2], b.w[3]);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float e) {
    f32x4 v = {a, b, c, e};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void st4u(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t e) {
    u32x4 v = {a, b, c, e};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void st1u(uint32_t* p, uint32_t a) {
    if (NT) __builtin_nontemporal_store(a, p);
    else *p = a;
}

// Wave-private scratch of the done-mask compaction: one wave's finished envs (at most 64 * VEC)
template <int VEC>
struct WaveResetScratch {
    uint32_t ent[64 * VEC];     // (episode << 12) | local env id within the wave's 64*VEC envs
    float fresh[64 * VEC][4];   // new states by list slot
};

// Steps the VEC environments of every lane of the wave (fast form when the whole wave qualifies), then — RESET —
// replaces the finished ones by fresh episodes.  `wave_first` = index of the wave's first environment.
// Returns the number of finished envs in this wave pass.  Every lane of the wave must call this (wave collectives).
template <int VEC, bool EULER, bool SB, bool RESET>
__device__ __forceinline__ uint32_t cartpole_wave_step(const CartPoleDev& d, CartPoleLane (&s)[VEC], const uint32_t (&a)[VEC], const bool (&valid)[VEC],
                                                       float (&r)[VEC], uint32_t (&dn)[VEC], uint32_t (&tr)[VEC], uint64_t wave_first,
                                                       WaveResetScratch<VEC>& lds, bool& bad) {
    constexpr CartPoleParams P = cartpole_params(EULER ? 1 : 0, SB ? 1 : 0);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < VEC; ++k) ok = ok & cartpole_fast_ok(s[k].th, s[k].thd, a[k]);  // bitwise: no short-circuit branches
    if (__all(ok)) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) cartpole_step_fast<EULER, SB>(P, s[k].x, s[k].xd, s[k].th, s[k].thd, s[k].ctr, a[k], r[k], dn[k], tr[k]);
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            if (!cartpole_step_ref(P, s[k].x, s[k].xd, s[k].th, s[k].thd, s[k].ctr, a[k], r[k], dn[k], tr[k]) && valid[k]) bad = true;
    }
    // done-mask reduction: ballots + popcounts give every finished env its slot in the wave's list
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t slot[VEC];
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const bool fin = valid[k] && (dn[k] | tr[k]);
        const unsigned long long m = __ballot(fin);
        slot[k] = fin ? total + (uint32_t)__popcll(m & below) : 0xffffffffu;
        total += (uint32_t)__popcll(m);
    }
    if (RESET && total) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            if (slot[k] != 0xffffffffu) lds.ent[slot[k]] = (s[k].ctr & (kCpEpMask << kCpEpShift)) | (uint32_t)(lane * VEC + k);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = (uint32_t)lane; j < total; j += 64u) {  // one pass unless > 64 envs of the wave finished at once
            const uint32_t e = lds.ent[j];
            float f[4];
            cartpole_draw(d.seed, d.env_id_base + wave_first + (e & 0xfffu), e >> kCpEpShift, f);
            lds.fresh[j][0] = f[0]; lds.fresh[j][1] = f[1]; lds.fresh[j][2] = f[2]; lds.fresh[j][3] = f[3];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (slot[k] != 0xffffffffu) {
                s[k].x = lds.fresh[slot[k]][0]; s[k].xd = lds.fresh[slot[k]][1];
                s[k].th = lds.fresh[slot[k]][2]; s[k].thd = lds.fresh[slot[k]][3];
                s[k].ctr = cp_ctr_after_reset(s[k].ctr);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the list is reused by the wave's next pass
    }
    return total;
}

__device__ __forceinline__ void cartpole_flush_counts(const CartPoleDev& d, uint32_t finished, bool bad) {
    // one fire-and-forget atomic per wave, spread over kDoneShards addresses; the (rare) error bit likewise
    if ((threadIdx.x & 63) == 0 && finished)
        atomicAdd(d.done_count + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kDoneShards - 1)), (unsigned long long)finished);
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_INVALID_ACTION);
    }
}

// 16-byte access at (uniform base pointer) + (32-bit per-lane byte offset): lowers to global_load/store with the
// base in SGPRs and the lane offset in one VGPR (no 64-bit per-lane address arithmetic, which was ~90 VALU
// instructions per wave pass when every column address was formed per lane)
template <typename T> __device__ __forceinline__ const T* at(const void* base, uint32_t byte_off) {
    return reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}
template <typename T> __device__ __forceinline__ T* at(void* base, uint32_t byte_off) {
    return reinterpret_cast<T*>(static_cast<char*>(base) + byte_off);
}

// VEC: envs per lane (4 = 16-byte accesses; 1 = fallback for misaligned caller buffers).
// RESET: MGYM_FLAG_AUTO_RESET.  NT: non-temporal stores for the output columns.
template <int VEC, bool EULER, bool SB, bool RESET, bool NT, int BLK>
__global__ void __launch_bounds__(BLK)
cartpole_step_kernel(CartPoleDev d, const uint32_t* __restrict__ act, float* __restrict__ obs_out,
                     float* __restrict__ rew, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    __shared__ WaveResetScratch<VEC> lds_[RESET ? BLK / 64 : 1];
    constexpr uint64_t kPerBlock = (uint64_t)BLK * VEC;
    bool bad = false;
    uint32_t finished = 0;
    WaveResetScratch<VEC>& lds = lds_[RESET ? (threadIdx.x >> 6) : 0];
    const uint32_t lane_env = threadIdx.x * VEC;     // first env of this lane within the block's slice
    const uint32_t off4 = lane_env * 4u;             // its byte offset in a 4-byte column
    const uint64_t col = d.n_pad;                    // engine column stride (words)
    const uint32_t colb = (uint32_t)col * 4u;        // ... in bytes
    // raw buffer resource over the engine's allocation (stride 0, no bounds clamp, 32-bit data format: gfx9 word 3 = 0x00020000)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d.base, 0, 0xffffffff, 0x00020000);
    for (uint64_t base = (uint64_t)blockIdx.x * kPerBlock; base < d.n; base += (uint64_t)gridDim.x * kPerBlock) {  // wave-uniform
        // uniform (scalar) bases of this block pass
        float* const bx = d.base + base;
        const uint32_t* const ba = act + base;
        const uint64_t left = d.n - base;
        const uint32_t rem = left < kPerBlock ? (uint32_t)left : (uint32_t)kPerBlock;
        CartPoleLane s[VEC];
        uint32_t a[VEC];
        bool valid[VEC];
        float r[VEC];
        uint32_t dn[VEC], tr[VEC];
        const bool in = lane_env < rem;  // the host launches VEC = 4 only when n % 4 == 0: a lane is wholly in or out
#pragma unroll
        for (int k = 0; k < VEC; ++k) valid[k] = in;
        const uint32_t sb = (uint32_t)base * 4u;  // byte offset of this pass in a column (n <= 2^27: every offset < 2^32)
        if (VEC == 4) {
            // Engine columns: buffer loads — resource descriptor and column/pass offsets in SGPRs, one constant VGPR lane
            // offset, no per-lane address arithmetic.  The columns are padded to n_pad (a multiple of the block's 1024
            // envs), so every lane of the block may load (padding holds zeros and is never stored back).
            // theta, theta_dot and the action first: the sin/cos polynomial can start while x, x_dot, ctr are in flight.
            const u32x4 vth = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + 2u * colb, 0);
            const u32x4 vthd = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + 3u * colb, 0);
            uint4 va = make_uint4(0u, 0u, 0u, 0u);
            if (in) va = *at<uint4>(ba, off4);  // caller buffer: n words exactly
            const u32x4 vx = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb, 0);
            const u32x4 vxd = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + colb, 0);
            const u32x4 vc = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + 4u * colb, 0);
            s[0] = {as_f32(vx.x), as_f32(vxd.x), as_f32(vth.x), as_f32(vthd.x), vc.x};
            s[1 % VEC] = {as_f32(vx.y), as_f32(vxd.y), as_f32(vth.y), as_f32(vthd.y), vc.y};
            s[2 % VEC] = {as_f32(vx.z), as_f32(vxd.z), as_f32(vth.z), as_f32(vthd.z), vc.z};
            s[3 % VEC] = {as_f32(vx.w), as_f32(vxd.w), as_f32(vth.w), as_f32(vthd.w), vc.w};
            a[0] = va.x; a[1 % VEC] = va.y; a[2 % VEC] = va.z; a[3 % VEC] = va.w;
        } else {
            s[0] = {0.f, 0.f, 0.f, 0.f, 0u};
            a[0] = 0u;
            if (in) {
                s[0] = {*at<float>(bx, off4), *at<float>(bx + col, off4), *at<float>(bx + 2 * col, off4), *at<float>(bx + 3 * col, off4),
                        *at<uint32_t>(bx + 4 * col, off4)};
                a[0] = *at<uint32_t>(ba, off4);
            }
        }
        const uint64_t wave_first = base + (uint64_t)(threadIdx.x & ~63) * VEC;
        finished += cartpole_wave_step<VEC, EULER, SB, RESET>(d, s, a, valid, r, dn, tr, wave_first, lds, bad);

        if (in) {
            if (VEC == 4) {
                constexpr int kAux = NT ? 2 : 0;  // cache-policy bits of the buffer store: 2 = nt
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(s[0].x), as_u32(s[1 % VEC].x), as_u32(s[2 % VEC].x), as_u32(s[3 % VEC].x)}, rs, off4, sb, kAux);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(s[0].xd), as_u32(s[1 % VEC].xd), as_u32(s[2 % VEC].xd), as_u32(s[3 % VEC].xd)}, rs, off4, sb + colb, kAux);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(s[0].th), as_u32(s[1 % VEC].th), as_u32(s[2 % VEC].th), as_u32(s[3 % VEC].th)}, rs, off4, sb + 2u * colb, kAux);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(s[0].thd), as_u32(s[1 % VEC].thd), as_u32(s[2 % VEC].thd), as_u32(s[3 % VEC].thd)}, rs, off4, sb + 3u * colb, kAux);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{s[0].ctr, s[1 % VEC].ctr, s[2 % VEC].ctr, s[3 % VEC].ctr}, rs, off4, sb + 4u * colb, kAux);
                if (obs_out) {
                    float* const bo = obs_out + base;
                    *at<float4>(bo, off4) = make_float4(s[0].x, s[1 % VEC].x, s[2 % VEC].x, s[3 % VEC].x);
                    *at<float4>(bo + d.n, off4) = make_float4(s[0].xd, s[1 % VEC].xd, s[2 % VEC].xd, s[3 % VEC].xd);
                    *at<float4>(bo + 2 * d.n, off4) = make_float4(s[0].th, s[1 % VEC].th, s[2 % VEC].th, s[3 % VEC].th);
                    *at<float4>(bo + 3 * d.n, off4) = make_float4(s[0].thd, s[1 % VEC].thd, s[2 % VEC].thd, s[3 % VEC].thd);
                }
                if (rew) st4<NT>(at<float>(rew + base, off4), r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
                if (done_out)
                    st1u<NT>(at<uint32_t>(done_out + base, lane_env), dn[0] | (dn[1 % VEC] << 8) | (dn[2 % VEC] << 16) | (dn[3 % VEC] << 24));
                if (trunc_out)
                    st1u<NT>(at<uint32_t>(trunc_out + base, lane_env), tr[0] | (tr[1 % VEC] << 8) | (tr[2 % VEC] << 16) | (tr[3 % VEC] << 24));
            } else {
                *at<float>(bx, off4) = s[0].x; *at<float>(bx + col, off4) = s[0].xd; *at<float>(bx + 2 * col, off4) = s[0].th;
                *at<float>(bx + 3 * col, off4) = s[0].thd; *at<uint32_t>(bx + 4 * col, off4) = s[0].ctr;
                if (obs_out) {
                    float* const bo = obs_out + base;
                    *at<float>(bo, off4) = s[0].x; *at<float>(bo + d.n, off4) = s[0].xd;
                    *at<float>(bo + 2 * d.n, off4) = s[0].th; *at<float>(bo + 3 * d.n, off4) = s[0].thd;
                }
                if (rew) *at<float>(rew + base, off4) = r[0];
                if (done_out) *at<uint8_t>(done_out + base, lane_env) = (uint8_t)dn[0];
                if (trunc_out) *at<uint8_t>(trunc_out + base, lane_env) = (uint8_t)tr[0];
            }
        }
    }
    cartpole_flush_counts(d, finished, bad);
}

// policy of a fused rollout: actions from the caller's [K][n] table, or drawn on the device
enum { ROLL_TABLE = 0, ROLL_UNIFORM = 1, ROLL_LINEAR = 2 };
struct LinearPolicy { float w[4], b; };   // action = ((w0 x + w1 x_dot) + w2 theta) + w3 theta_dot + b > 0 (mgym_rollout_linear)

// Fused K-step rollout (mgym_rollout / mgym_rollout_uniform): one lane keeps its 4 environments in registers for
// K steps; per step it reads one 16-B action word (or takes the next bit of its Philox policy stream) and writes
// reward / done / truncated (and the observation / the drawn actions if asked).  Identical to K launches of
// cartpole_step_kernel by construction: same per-wave function.
template <bool EULER, bool SB, bool RESET, int POLICY>
__global__ void __launch_bounds__(kBlock)
cartpole_rollout_kernel(CartPoleDev d, const uint32_t* __restrict__ act, uint32_t* __restrict__ act_out, uint64_t policy_seed, uint32_t policy_call, LinearPolicy lin,
                        int K, float* __restrict__ obs_out, float* __restrict__ rew, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    __shared__ WaveResetScratch<4> lds_[RESET ? kWaves : 1];
    constexpr uint64_t kPerBlock = (uint64_t)kBlock * 4;
    bool bad = false;
    uint32_t finished = 0;
    WaveResetScratch<4>& lds = lds_[RESET ? (threadIdx.x >> 6) : 0];
    // per-step buffers through buffer resources: descriptor + (step, pass) offset in SGPRs, one constant VGPR lane offset.
    // The host wrapper guarantees K * n * 16 < 2^32 (every byte offset fits 32 bits) and n % 4 == 0.
    const uint32_t lane_env = threadIdx.x * 4u, off4 = lane_env * 4u;
    const uint32_t colb = (uint32_t)d.n_pad * 4u, nb4 = (uint32_t)d.n * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d.base, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(POLICY == ROLL_TABLE ? act : act_out), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rew, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(done_out, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(trunc_out, 0, 0xffffffff, 0x00020000);
    for (uint64_t base = (uint64_t)blockIdx.x * kPerBlock; base < d.n; base += (uint64_t)gridDim.x * kPerBlock) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * 4;
        const uint64_t left = d.n - base;
        const bool in = lane_env < (left < kPerBlock ? (uint32_t)left : (uint32_t)kPerBlock);
        const uint32_t sb = (uint32_t)base * 4u;
        CartPoleLane s[4];
        bool valid[4] = {in, in, in, in};
        {   // engine columns are padded to the block's span: every lane may load; only in-range lanes store
            const u32x4 vx = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb, 0), vxd = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + colb, 0);
            const u32x4 vth = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + 2u * colb, 0), vthd = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + 3u * colb, 0);
            const u32x4 vc = __builtin_amdgcn_raw_buffer_load_b128(rs, off4, sb + 4u * colb, 0);
            s[0] = {as_f32(vx.x), as_f32(vxd.x), as_f32(vth.x), as_f32(vthd.x), vc.x}; s[1] = {as_f32(vx.y), as_f32(vxd.y), as_f32(vth.y), as_f32(vthd.y), vc.y};
            s[2] = {as_f32(vx.z), as_f32(vxd.z), as_f32(vth.z), as_f32(vthd.z), vc.z}; s[3] = {as_f32(vx.w), as_f32(vxd.w), as_f32(vth.w), as_f32(vthd.w), vc.w};
        }
        const uint64_t wave_first = base + (uint64_t)(threadIdx.x & ~63) * 4;
        Philox4 bits{};  // POLICY == ROLL_UNIFORM: 128 policy bits = 32 steps of this lane's 4 envs
        uint32_t so = sb;  // byte offset of (step t, this pass) in a [K][n] word buffer; byte buffers use so / 4
        for (int t = 0; t < K; ++t, so += nb4) {
            uint32_t a[4] = {0u, 0u, 0u, 0u}, dn[4], tr[4];
            float r[4];
            if (POLICY == ROLL_TABLE) {
                if (in) { const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(ra, off4, so, 2); a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w; }
            } else if (POLICY == ROLL_LINEAR) {
                // linear policy on the observation the env holds before the step, in the documented order (the file is compiled with -ffp-contract=off)
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = ((((lin.w[0] * s[k].x + lin.w[1] * s[k].xd) + lin.w[2] * s[k].th) + lin.w[3] * s[k].thd) + lin.b > 0.0f) ? 1u : 0u;
                if (in && act_out) __builtin_amdgcn_raw_buffer_store_b128(u32x4{a[0], a[1], a[2], a[3]}, ra, off4, so, 2);
            } else {
                // uniform random policy: env i takes bit (t % 32) of word (i % 4) of Philox(policy_seed; lane's first
                // global env id, policy_call, SLOT_POLICY + t / 32) — one Philox evaluation per lane per 32 steps
                if ((t & 31) == 0) bits = env_draw(policy_seed, d.env_id_base + i0, policy_call, SLOT_POLICY + (uint32_t)(t >> 5));
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = (bits.w[k] >> (t & 31)) & 1u;
                if (in && act_out) __builtin_amdgcn_raw_buffer_store_b128(u32x4{a[0], a[1], a[2], a[3]}, ra, off4, so, 2);
            }
            finished += cartpole_wave_step<4, EULER, SB, RESET>(d, s, a, valid, r, dn, tr, wave_first, lds, bad);
            if (in) {
                if (rew) __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(r[0]), as_u32(r[1]), as_u32(r[2]), as_u32(r[3])}, rr, off4, so, 2);
                if (done_out) __builtin_amdgcn_raw_buffer_store_b32(dn[0] | (dn[1] << 8) | (dn[2] << 16) | (dn[3] << 24), rd, lane_env, so >> 2, 2);
                if (trunc_out) __builtin_amdgcn_raw_buffer_store_b32(tr[0] | (tr[1] << 8) | (tr[2] << 16) | (tr[3] << 24), rt, lane_env, so >> 2, 2);
                if (obs_out) {
                    float* o = obs_out + (uint64_t)t * 4 * d.n + i0;
                    st4<true>(o, s[0].x, s[1].x, s[2].x, s[3].x); st4<true>(o + d.n, s[0].xd, s[1].xd, s[2].xd, s[3].xd);
                    st4<true>(o + 2 * d.n, s[0].th, s[1].th, s[2].th, s[3].th); st4<true>(o + 3 * d.n, s[0].thd, s[1].thd, s[2].thd, s[3].thd);
                }
            }
        }
        if (in) {
            constexpr int kAux = kNtStores ? 2 : 0;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(s[0].x), as_u32(s[1].x), as_u32(s[2].x), as_u32(s[3].x)}, rs, off4, sb, kAux);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(s[0].xd), as_u32(s[1].xd), as_u32(s[2].xd), as_u32(s[3].xd)}, rs, off4, sb + colb, kAux);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(s[0].th), as_u32(s[1].th), as_u32(s[2].th), as_u32(s[3].th)}, rs, off4, sb + 2u * colb, kAux);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{as_u32(s[0].thd), as_u32(s[1].thd), as_u32(s[2].thd), as_u32(s[3].thd)}, rs, off4, sb + 3u * colb, kAux);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{s[0].ctr, s[1].ctr, s[2].ctr, s[3].ctr}, rs, off4, sb + 4u * colb, kAux);
        }
    }
    cartpole_flush_counts(d, finished, bad);
}

// Masked reset.  One lane looks at 4 consecutive mask bytes of each mask with a single dword
// load (2 B/env of traffic); masked envs are compacted through LDS, so one pass of the block's first lanes
// evaluates Philox for all of them instead of every wave paying for its one or two masked lanes.
// `all` skips masks and compaction.
__global__ void __launch_bounds__(kBlock)
cartpole_reset_kernel(CartPoleDev d, const uint8_t* __restrict__ m0, const uint8_t* __restrict__ m1, int all,
                      int masks_aligned, float* __restrict__ obs_out) {
    __shared__ uint32_t count;
    __shared__ uint16_t idx[kBlock * 4];
    constexpr uint64_t kPerBlock = (uint64_t)kBlock * 4;
    for (uint64_t base = (uint64_t)blockIdx.x * kPerBlock; base < d.n; base += (uint64_t)gridDim.x * kPerBlock) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * 4;
        uint32_t m = 0;
        if (i0 < d.n) {
            if (all) {
                m = 0x01010101u;
            } else if (masks_aligned && i0 + 4 <= d.n) {
                if (m0) m |= *reinterpret_cast<const uint32_t*>(m0 + i0);
                if (m1) m |= *reinterpret_cast<const uint32_t*>(m1 + i0);
            } else {
                for (int k = 0; k < 4 && i0 + k < d.n; ++k) {
                    uint32_t b = 0;
                    if (m0) b |= m0[i0 + k];
                    if (m1) b |= m1[i0 + k];
                    m |= (b ? 1u : 0u) << (8 * k);
                }
            }
        }
        uint32_t cnt;
        if (all) {
            cnt = (uint32_t)((d.n - base) < kPerBlock ? (d.n - base) : kPerBlock);
        } else {
            if (threadIdx.x == 0) count = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < d.n && ((m >> (8 * k)) & 0xffu)) idx[atomicAdd(&count, 1u)] = (uint16_t)(threadIdx.x * 4 + k);
            __syncthreads();
            cnt = count;
        }
        for (uint32_t j = threadIdx.x; j < cnt; j += kBlock) {
            const uint64_t i = base + (all ? j : (uint32_t)idx[j]);
            const uint32_t c = d.ctr()[i];
            float f[4];
            cartpole_draw(d.seed, d.env_id_base + i, c >> kCpEpShift, f);
            d.x()[i] = f[0]; d.xd()[i] = f[1]; d.th()[i] = f[2]; d.thd()[i] = f[3]; d.ctr()[i] = cp_ctr_after_reset(c);
            if (obs_out) {
                obs_out[i] = f[0]; obs_out[d.n + i] = f[1]; obs_out[2 * d.n + i] = f[2]; obs_out[3 * d.n + i] = f[3];
            }
        }
        if (!all) __syncthreads();  // idx/count are reused by the next iteration
    }
}

// blob <-> engine columns (steps / sbt / episode are separate blob columns; saturations as in cartpole_step.h)
__global__ void cartpole_export_kernel(CartPoleDev d, uint32_t* __restrict__ blob) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    blob[0 * d.n + i] = as_u32(d.x()[i]);
    blob[1 * d.n + i] = as_u32(d.xd()[i]);
    blob[2 * d.n + i] = as_u32(d.th()[i]);
    blob[3 * d.n + i] = as_u32(d.thd()[i]);
    uint32_t c = d.ctr()[i];
    blob[4 * d.n + i] = c & kCpStepsMask;
    blob[5 * d.n + i] = (uint32_t)((int32_t)((c >> kCpSbtShift) & kCpSbtMask) - 1);  // -1 = None
    blob[6 * d.n + i] = c >> kCpEpShift;
}

__global__ void cartpole_import_kernel(CartPoleDev d, const uint32_t* __restrict__ blob) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    d.x()[i] = as_f32(blob[0 * d.n + i]);
    d.xd()[i] = as_f32(blob[1 * d.n + i]);
    d.th()[i] = as_f32(blob[2 * d.n + i]);
    d.thd()[i] = as_f32(blob[3 * d.n + i]);
    int32_t sbt = (int32_t)blob[5 * d.n + i];
    d.ctr()[i] = cp_ctr_pack(blob[4 * d.n + i], sbt < 0 ? 0u : (uint32_t)sbt + 1u, blob[6 * d.n + i]);
}

struct CartPoleEnv final : Env {
    void* base = nullptr;
    CartPoleDev dev{};
    bool auto_reset = false;
    int step_block = getenv("MGYM_CARTPOLE_BLOCK") ? atoi(getenv("MGYM_CARTPOLE_BLOCK")) : kBlock;  // threads per block of the step kernel (tuning knob)
    uint32_t policy_calls = 0;  // mgym_rollout_uniform calls so far (Philox counter word of the policy stream)

    ~CartPoleEnv() override {
        if (base) (void)hipFree(base);
    }

    int init() override {
        obs_dim = 4;
        state_cols = 7;
        if (n > (1ull << 27)) {  // the step kernel addresses the five columns with 32-bit byte offsets from one buffer resource
            set_last_error("mgym_create: a CartPole handle holds at most 134 217 728 environments (shard larger populations over handles)");
            return MGYM_ERR_BAD_ARG;
        }
        MGYM_HIP(hipMalloc(&base, 5 * n_pad * sizeof(float)));
        MGYM_HIP(hipMemsetAsync(base, 0, 5 * n_pad * sizeof(float), stream));
        dev.base = static_cast<float*>(base);
        dev.n_pad = n_pad;
        dev.n = n;
        dev.seed = cfg.seed;
        dev.env_id_base = cfg.env_id_base;
        dev.err = d_err;
        dev.done_count = d_done;
        auto_reset = (cfg.flags & MGYM_FLAG_AUTO_RESET) != 0;
        // cartpole.rs:81: steps_beyond_terminated = Some(0) at construction
        if (n) MGYM_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dev.ctr()), cp_ctr_pack(0u, 1u, 0u), n, stream));
        return MGYM_OK;
    }

    int reset(const uint8_t* m0, const uint8_t* m1, bool all, float* obs_out) override {
        if (n == 0) return MGYM_OK;
        int al = aligned(m0, 4) && aligned(m1, 4);
        hipLaunchKernelGGL(cartpole_reset_kernel, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, stream, dev, m0, m1,
                           all ? 1 : 0, al, obs_out);
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int reset_deterministic(float* obs_out) override {  // cartpole.rs:437-442: self.reset()? then state = zeros
        int st = reset(nullptr, nullptr, true, nullptr);
        if (st != MGYM_OK || n == 0) return st;
        MGYM_HIP(hipMemsetAsync(dev.x(), 0, 4 * n_pad * sizeof(float), stream));
        if (obs_out) MGYM_HIP(hipMemsetAsync(obs_out, 0, 4 * n * sizeof(float), stream));
        return MGYM_OK;
    }

    template <int VEC, bool RESET, int BLK>
    void launch_step_blk(const uint32_t* act, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) {
        const bool eu = cfg.is_euler, sb = cfg.sutton_barto_reward;
        uint64_t nb = (n + (uint64_t)BLK * VEC - 1) / ((uint64_t)BLK * VEC);
        const uint64_t cap = (uint64_t)kMaxBlocks * kBlock / BLK;   // same thread count as the kBlock-sized cap
        dim3 g((unsigned)(nb > cap ? cap : (nb ? nb : 1))), b(BLK);
#define MGYM_CP_LAUNCH(E, S) hipLaunchKernelGGL((cartpole_step_kernel<VEC, E, S, RESET, kNtStores, BLK>), g, b, 0, stream, dev, act, obs_out, reward, done, trunc)
        if (eu && !sb) MGYM_CP_LAUNCH(true, false);
        else if (eu) MGYM_CP_LAUNCH(true, true);
        else if (!sb) MGYM_CP_LAUNCH(false, false);
        else MGYM_CP_LAUNCH(false, true);
#undef MGYM_CP_LAUNCH
    }
    template <int VEC, bool RESET>
    void launch_step(dim3, const uint32_t* act, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) {
        if (VEC == 4 && step_block == 64) launch_step_blk<VEC, RESET, 64>(act, obs_out, reward, done, trunc);
        else if (VEC == 4 && step_block == 128) launch_step_blk<VEC, RESET, 128>(act, obs_out, reward, done, trunc);
        else launch_step_blk<VEC, RESET, kBlock>(act, obs_out, reward, done, trunc);
    }

    int step(const void* actions, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0) return MGYM_OK;
        const uint32_t* act = static_cast<const uint32_t*>(actions);
        bool vec_ok = n % 4 == 0 && aligned(act, 16) && aligned(reward, 16) && aligned(done, 4) && aligned(trunc, 4) && aligned(obs_out, 16);
        dim3 gv(grid_for((n + 3) / 4)), gs(grid_for(n));
        if (auto_reset) {
            if (vec_ok) launch_step<4, true>(gv, act, obs_out, reward, done, trunc);
            else launch_step<1, true>(gs, act, obs_out, reward, done, trunc);
        } else {
            if (vec_ok) launch_step<4, false>(gv, act, obs_out, reward, done, trunc);
            else launch_step<1, false>(gs, act, obs_out, reward, done, trunc);
        }
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    template <int POLICY>
    void launch_rollout(dim3 g, const uint32_t* act, uint32_t* act_out, uint64_t pseed, uint32_t pcall, int K, float* obs_out, float* reward,
                        uint8_t* done, uint8_t* trunc, LinearPolicy lin = LinearPolicy{}) {
        const bool eu = cfg.is_euler, sb = cfg.sutton_barto_reward, rs = auto_reset;
        dim3 b(kBlock);
#define MGYM_CP_ROLL(E, S, R) hipLaunchKernelGGL((cartpole_rollout_kernel<E, S, R, POLICY>), g, b, 0, stream, dev, act, act_out, pseed, pcall, lin, K, obs_out, reward, done, trunc)
        if (eu && !sb) { if (rs) MGYM_CP_ROLL(true, false, true); else MGYM_CP_ROLL(true, false, false); }
        else if (eu) { if (rs) MGYM_CP_ROLL(true, true, true); else MGYM_CP_ROLL(true, true, false); }
        else if (!sb) { if (rs) MGYM_CP_ROLL(false, false, true); else MGYM_CP_ROLL(false, false, false); }
        else { if (rs) MGYM_CP_ROLL(false, true, true); else MGYM_CP_ROLL(false, true, false); }
#undef MGYM_CP_ROLL
    }

    int rollout(const void* actions, int K, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0 || K == 0) return MGYM_OK;
        const uint32_t* act = static_cast<const uint32_t*>(actions);
        bool vec_ok = n % 4 == 0 && aligned(act, 16) && aligned(reward, 16) && aligned(done, 4) && aligned(trunc, 4) && aligned(obs_out, 16) &&
                      (uint64_t)K * n * 4 < (1ull << 32);  // the fused kernel addresses [K][n] word buffers with 32-bit byte offsets
        if (!vec_ok) return Env::rollout(actions, K, obs_out, reward, done, trunc);  // K plain steps
        launch_rollout<ROLL_TABLE>(dim3(grid_for(n / 4)), act, nullptr, 0, 0, K, obs_out, reward, done, trunc);
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int rollout_uniform(uint64_t policy_seed, int K, void* actions_out, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0 || K == 0) return MGYM_OK;
        uint32_t* ao = static_cast<uint32_t*>(actions_out);
        if (n % 4 != 0 || !aligned(ao, 16) || !aligned(reward, 16) || !aligned(done, 4) || !aligned(trunc, 4) || !aligned(obs_out, 16) ||
            (uint64_t)K * n * 4 >= (1ull << 32)) {
            set_last_error("mgym_rollout_uniform: n_envs must be a multiple of 4, the buffers 16-byte aligned, and K * n_envs below 2^30");
            return MGYM_ERR_BAD_ARG;
        }
        launch_rollout<ROLL_UNIFORM>(dim3(grid_for(n / 4)), nullptr, ao, policy_seed, policy_calls++, K, obs_out, reward, done, trunc);
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int rollout_linear(const float* policy, int K, void* actions_out, float* obs_out, float* reward, uint8_t* done, uint8_t* trunc) override {
        if (n == 0 || K == 0) return MGYM_OK;
        uint32_t* ao = static_cast<uint32_t*>(actions_out);
        if (n % 4 != 0 || !aligned(ao, 16) || !aligned(reward, 16) || !aligned(done, 4) || !aligned(trunc, 4) || !aligned(obs_out, 16) ||
            (uint64_t)K * n * 4 >= (1ull << 32)) {
            set_last_error("mgym_rollout_linear: n_envs must be a multiple of 4, the buffers 16-byte aligned, and K * n_envs below 2^30");
            return MGYM_ERR_BAD_ARG;
        }
        LinearPolicy lin;
        for (int q = 0; q < 4; ++q) lin.w[q] = policy[q];
        lin.b = policy[4];
        launch_rollout<ROLL_LINEAR>(dim3(grid_for(n / 4)), nullptr, ao, 0, 0, K, obs_out, reward, done, trunc, lin);
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int observation(const float** obs, uint64_t* col_stride) override {
        *obs = dev.x();
        *col_stride = n_pad;
        return MGYM_OK;
    }

    int get_state(void* blob) override {
        if (n == 0) return MGYM_OK;
        hipLaunchKernelGGL(cartpole_export_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                           dev, static_cast<uint32_t*>(blob));
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }

    int set_state(const void* blob) override {
        if (n == 0) return MGYM_OK;
        hipLaunchKernelGGL(cartpole_import_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                           dev, static_cast<const uint32_t*>(blob));
        MGYM_HIP(hipGetLastError());
        return MGYM_OK;
    }
};

Env* make_cartpole() { return new CartPoleEnv(); }

}  // namespace mgym
