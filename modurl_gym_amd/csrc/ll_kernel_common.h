// ll_kernel_common.h — what the LunarLander kernels of lunar_lander.hip and ll_roll.hip share: the staged polygon table, capacities, the
// device-built work lists of mgym_step, the per-call output pointers, wave-aggregated list appends, the LDS of a block that runs the contact
// path, the end-of-wave error / episode-count report.  (Two translation units so that the library builds in parallel.)
#pragma once
#include "common.h"
#include "ll_env.h"
#include "ll_free.h"

namespace mgym {


__device__ __forceinline__ void stage_tab(PolyTab& tab, const LLConst& k) {
    for (int t = threadIdx.x; t < 2 * kMaxPoly; t += blockDim.x) {  // blocks may be narrower than the table
        int p = t / kMaxPoly, q = t % kMaxPoly;
        tab.v[p][q] = k.poly_v[p][q];
        tab.n[p][q] = k.poly_n[p][q];
    }
    if (threadIdx.x < 2) tab.count[threadIdx.x] = k.poly_count[threadIdx.x];
    __syncthreads();
}

constexpr int kLLBlock = 64;  // one wave per block: heavy per-lane state, no intra-block cooperation
// Touching contacts one island may hold.  9 is the geometric bound of this scene: a body's polygon spans < 2 m (lander
// 1.13 m, leg diagonal 0.55 m) while terrain edges are 2 m wide, so it can touch at most two adjacent terrain edges plus
// the base edge (0,0)-(W,0) when the terrain runs at y = 0: 3 bodies x 3.  (In LDS the contact kernel keeps the first 4 per lane, see kVcNearLds.)
#ifndef LL_SOLVER_CAP
#define LL_SOLVER_CAP 9
#endif
constexpr int kSolverCap = LL_SOLVER_CAP;
constexpr int kVcNearLds = 2;  // of those, kept in LDS by the contact kernel (4 blocks per CU in either block size; the sweeps hold the first four in registers anyway); the rest in LLDev::vc_far
// the staged KEY / SEQ / TOI words of the contact cache (ll_b2.h CtHot): one LDS column per lane of the block
#define LL_HOT_DECL(BLKSZ) __shared__ uint32_t s_hot[3 * kSlots * (BLKSZ)]; const CtHot hot{(LL_LDS uint32_t*)s_hot + threadIdx.x, (uint32_t)(BLKSZ), 1u}
// the per-lane working storage of the contact path (ll_world.h WorldTmp): one LDS record per lane of the block
#define LL_TMP_DECL(BLKSZ) __shared__ WorldTmp s_tmp[(BLKSZ)]
#define LL_TMP_PTR() ((LL_LDS WorldTmp*)s_tmp + threadIdx.x)
constexpr uint32_t kWorkReset = 0x80000000u;  // worklist entry = env index | kWorkReset (reset) or plain (general step)
// Device-built lists (LLDev::work_list regions of n_pad words, lengths in LLDev::work_count):
//   L_GENERAL    envs that need the contact path this step.  Filled from BOTH ends: envs without a touching contact from
//                the front (count L_GENERAL), envs with one from the back (count L_GENERAL_T), so the waves of the
//                worklist kernel hold environments of one kind (180 joint-only sweeps vs. sweeps with contact constraints).
//   L_RESET      finished envs to reset (register-only fast path); L_RESET_SLOW: resets the fast path declined
//   L_LATE       overlapped launch order only: envs the free-flight kernel had to decline (a contact would be created)
//   L_RESET_DIRECT  staged resets: finished envs whose prepared episode does not fit (state imported, reset by the caller): reset the slow way
//   L_PREP       staged resets: envs that have just been reset and whose NEXT reset is to be prepared (+ L_PREP_SLOW: declined by the fast path)
//   L_TOI0 + r   envs whose world.step still has time-of-impact sub-steps to do after round r (see ll_toi_kernel)
constexpr int kToiRounds = 4;
//   C_NEXT       (a counter only) the length of the NEXT step's L_GENERAL while ll_epilogue_kernel is filling it; C_TICKET: its block ticket
enum { L_GENERAL = 0, L_GENERAL_T = 1, L_RESET = 2, L_RESET_SLOW = 3, L_LATE = 4, L_RESET_DIRECT = 5, L_TOI0 = 6, C_NEXT = L_TOI0 + kToiRounds, C_NEXT_T /* ... and of its touching end */, C_TICKET, C_TICKET2,
       L_COUNT, L_PREP = L_COUNT, L_PREP_SLOW, L_LISTS };  // (the counts of the lists below L_COUNT are zeroed by rebuild_list() / at the start of every call of the unfused order)

struct LLIo {
    const uint32_t* act;
    float* obs_out;
    float* rew;
    uint8_t* done_out;
    uint8_t* trunc_out;
};

__device__ __forceinline__ void ll_write_obs(const LLDev& d, const LLIo& io, uint64_t i, const float state[8]) {
    for (int q = 0; q < 8; ++q) {
        d.obs[(uint64_t)q * d.n_pad + i] = state[q];
        if (io.obs_out) io.obs_out[(uint64_t)q * d.n + i] = state[q];
    }
}

// wave-aggregated append of env indices to the worklist (done-mask style ballot + one atomic per wave)
__device__ __forceinline__ void ll_push(const LLDev& d, int which, bool want, uint32_t entry) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;  // lane 0 may be inactive in a 32-lane block's tail
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(d.work_count + which, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (want) d.work_list[(uint64_t)which * d.n_pad + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = entry;
}

// the same from the back of list `which_list` (length kept in count slot `which_count`)
__device__ __forceinline__ void ll_push_back(const LLDev& d, int which_list, int which_count, bool want, uint32_t entry) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(d.work_count + which_count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (want) d.work_list[(uint64_t)which_list * d.n_pad + (d.n_pad - 1u - (base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))))] = entry;
}

// block-aggregated append for kernels whose every wave appends (ll_classify_kernel): ONE atomic per block and list — thousands
// of per-wave atomics on one counter serialise at ~11 ns each.  All threads of the block must call it (it synchronises);
// `s_cnt` is block-shared scratch for (blockDim.x / 64 + 1) words.
__device__ __forceinline__ void ll_push_block(const LLDev& d, int which, bool want, uint32_t entry, uint32_t* s_cnt, int which_count = -1, bool from_back = false) {
    if (which_count < 0) which_count = which;   // (the epilogue fills next step's L_GENERAL under the counters C_NEXT / C_NEXT_T)
    const unsigned long long mask = __ballot(want);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();  // s_cnt may still be read by the previous call
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < nw; ++w) { const uint32_t c = s_cnt[w]; s_cnt[w] = tot; tot += c; }
        s_cnt[nw] = tot ? atomicAdd(d.work_count + which_count, tot) : 0u;
    }
    __syncthreads();
    if (want) {
        const uint32_t at = s_cnt[nw] + s_cnt[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        d.work_list[(uint64_t)which * d.n_pad + (from_back ? d.n_pad - 1u - at : at)] = entry;
    }
}

// done-mask reduction for mgym_episode_count: ballot + popcount per wave pass, one fire-and-forget atomic per wave at the end
__device__ __forceinline__ void ll_flush_done(const LLDev& d, uint32_t finished) {
    if ((threadIdx.x & 63) == 0 && finished)
        atomicAdd(d.done_count + ((blockIdx.x + (threadIdx.x >> 6)) & (kDoneShards - 1)), (unsigned long long)finished);
}

#ifndef LL_CONTACT_NUM_VGPR   // register budget of the contact kernel (arch VGPRs; the unified file holds twice that incl. AGPRs)
#define LL_CONTACT_ATTR
#else
#define LL_CONTACT_ATTR __attribute__((amdgpu_num_vgpr(LL_CONTACT_NUM_VGPR)))
#endif
// threads per block of the contact kernel: a block is ONE wave.  With the World records in LDS (BLK <= 32) BLK of its lanes
// carry an environment each; the other lanes of the wave exist only to take their share of the time-of-impact evaluations.
constexpr int ll_contact_threads(int blk) { return blk <= 32 ? 64 : blk; }

// The LDS of a block that runs the contact path:
// velocity constraints per lane kept in LDS; the others go to the global workspace (CSolverMem)
// Blocks of up to 32 lanes also keep the World record (bodies, joints, terrain heights, broad-phase boxes, the contact
// being updated, collide_edge_polygon's polygon buffer, the slot lists: 720 B per lane, all indexed at run time) in LDS
// instead of scratch: ~100 cycles per dependent access instead of >= 500 (1.45 -> 1.29 ms per step at 262 144 envs,
// scratch 2096 -> 1280 B/lane).  LDS of a 32-lane block: 2 x 32 x 124 B constraints + 4.6 KB staged contact words +
// 23 KB World + table = 38.2 KB (four blocks per CU); the sweeps hold the first four constraints in registers anyway.
template <int BLK>
struct ContactLds {
    static constexpr bool kWorldLds = BLK <= 32;
    World world[kWorldLds ? BLK : 1];
    WorldTmp tmp[BLK];
    PolyTab tab;
    VConstraint vc[kVcNearLds * BLK];
    uint32_t hot[(BLK > 32 ? 2 : 3) * kSlots * BLK];
    uint16_t task[kWorldLds ? BLK * kSlots : 1];   // time-of-impact evaluations of the wave's current pass: (owner lane << 4) | contact slot
    uint32_t late[64 * 4];                         // single-launch step: envs this wave's free-flight pass hands to its own contact path
};


__device__ __forceinline__ void ll_report(const LLDev& d, bool not_reset, uint32_t overflow, uint32_t finished) {
    ll_flush_done(d, finished);
    if (__any(not_reset) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_NOT_RESET);
    if (__any(overflow & 1u) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_CONTACT_OVERFLOW);
    if (__any(overflow & 2u) && (threadIdx.x & 63) == 0) atomicOr(d.err, DEV_ERR_SOLVER_OVERFLOW);
}

}  // namespace mgym
