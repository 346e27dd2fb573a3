// ll_roll.hip — the persistent launch of mgym_rollout for LunarLanderV3 (ll_roll.h) as a translation unit of its own, with the three host entry
// points lunar_lander.hip calls.  Replaces K consecutive calls of `impl Gym for LunarLanderV3`::step (reference src/box_2d/lunar_lander.rs:919-1167).
#include "ll_roll.h"

namespace mgym {

int ll_rollout_blocks_per_cu(int* per_cu) {
    MGYM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ll_rollout_kernel<32>, 64, 0));
    return MGYM_OK;
}
void ll_rollout_ring_init(hipStream_t s, const RollQ& q) { hipLaunchKernelGGL(ll_rollout_ring_init_kernel, dim3(256), dim3(256), 0, s, q); }
void ll_rollout_begin(hipStream_t s, const RollQ& q, uint32_t n) { hipLaunchKernelGGL(ll_rollout_begin_kernel, dim3(1), dim3(64), 0, s, q, n); }
void ll_rollout_launch(hipStream_t s, unsigned grid, const LLDev& d, const LLIo& io, const RollQ& q) {
    hipLaunchKernelGGL(ll_rollout_kernel<32>, dim3(grid), dim3(64), 0, s, d, io, q);
}
void ll_rollout_helper_launch(hipStream_t s, unsigned grid, const LLDev& d, const LLIo& io, const RollQ& q) {
    hipLaunchKernelGGL(ll_rollout_free_kernel<32>, dim3(grid), dim3(64), 0, s, d, io, q);
}

}  // namespace mgym
