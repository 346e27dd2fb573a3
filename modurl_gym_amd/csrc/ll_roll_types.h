// ll_roll_types.h — the queue descriptor, control words and statistics of mgym_rollout's persistent launch (ll_roll.h): what the host side
// (lunar_lander.hip) and the kernels (ll_roll.hip) share.
#pragma once
#include <stdint.h>

namespace mgym {

typedef uint32_t ll_u32x4 __attribute__((ext_vector_type(4)));

// RQ_CONTACT: environments that ended their last step with a TOUCHING contact (island solves with contact constraints, position solves that run out of
// their 60 iterations half of the time, 59 % take time-of-impact sub-steps); RQ_LIGHT: the other three quarters of the contact path's population
// (cached contacts that do not touch yet, hand-overs: joints-only islands that converge in 2-4 position iterations, 13 % take a sub-step).  A wave is
// as slow as its slowest lane, so the two kinds travel in separate batches — which lengthens the longest chain of a lock-step step (measured in
// round 2: not kept there) and is exactly right when nothing waits for the longest chain.
// RQ_TOI: environments whose world.step stands at a time-of-impact SUB-STEP (b2World::SolveTOI: advance the body to its earliest impact, solve the
// small island, re-open that body's contacts).  Inside a batch the lanes that sub-step (13 % of a light batch, 60 % of a touching one; up to four times)
// keep the others waiting for a third of the batch's time; here a batch takes its lanes up to the first sub-step only, stores the unfinished ones
// mid-step (ll_store(.., mid): the same resumable form the host check exercises) and queues them: a sub-step batch is 32 lanes that ALL sub-step.
enum { RQ_FREE = 0, RQ_CONTACT = 1, RQ_RESET = 2, RQ_LIGHT = 3, RQ_TOI = 4, RQ_COUNT = 5 };
// control words, each on a 128-byte line of its own.  Per queue: TAIL (ring positions handed to producers, fetch-add), HEAD (ring positions
// handed to consumers, fetch-add) and AVAIL, a counting semaphore of published entries: a consumer subtracts what it wants and gives back
// what it did not get, so taking entries costs every wave a fixed number of atomics however many waves want the same entries (a
// compare-and-swap on HEAD made 1 024 waves retry each other: 830 failed rounds per wave and launch, measured).
// HEADT: the step index of the entry taken most recently from a queue (a hint: rings hand entries out lowest step index first, roughly).  Waves serve
// the queue that is furthest BEHIND, so that all kinds of environments advance at one pace: with fixed priorities the contact path's population ran
// ahead, free-flight environments started late, and the ones among them that reached the ground in their last steps were taken through their
// remaining steps one 0.4-0.7 ms batch at a time while 900 waves idled (timeline: profiles/r04_lunarlander/rollout_timeline_*.txt).
enum { RC_AVAIL = 0 /* + queue */, RC_LIVE = 5, RC_HEADT = 6 /* + queue */, RC_HEAD = 11 /* + queue */, RC_TAIL = 16 /* + queue */, RC_CHUNK = 21, RC_ABORT = 22, RC_WORDS = 24 };
// per-launch work statistics (ticks of the 100 MHz wall clock summed over the waves; read by MGYM_LL_ROLL_STATS=1 / tools): cheap enough to stay in
enum { RS_T_TOTAL = 0, RS_T_SEED, RS_T_CONTACT, RS_N_CONTACT_BATCHES, RS_N_CONTACT_LANES, RS_T_RESET, RS_N_RESET_LANES, RS_T_FREE, RS_N_FREE_STEPS, RS_N_FREE_LANE_STEPS,
       RS_N_REFILLS, RS_T_IDLE, RS_N_SWITCHES, RS_T_FREE_QUEUE, RS_N_WAVES, RS_T_FREE_BEGIN, RS_T_FREE_SWEEPS, RS_T_FREE_FINISH, RS_T_FREE_ISSUE, RS_N_MAIN, RS_T_LIGHT, RS_N_LIGHT_BATCHES, RS_N_LIGHT_LANES, RS_N_ROTATIONS, RS_T_TOI, RS_N_TOI_BATCHES, RS_N_TOI_LANES, RS_N_HELPER_STEPS, RS_N_HELPER_LANE_STEPS, RS_COUNT = 30 };
struct RollStat { unsigned long long v[RS_COUNT]; };
struct RollQ {
    unsigned long long* ring;   // [RQ_COUNT][cap] slots {sequence << 32 | entry}; slot k starts with sequence k
    uint32_t* ctl;              // [RC_WORDS][32]
    uint32_t mask;              // cap - 1 (cap: a power of two >= n)
    uint32_t K;                 // steps of this launch (<= kRollMaxK)
    uint32_t contact_min;       // a wave that has other work takes a light-contact batch only when at least this many entries wait
    uint32_t tail_live, tail_lanes;  // once fewer environments than tail_live are still to finish, waves idle anyway: batches of at most tail_lanes lanes
                                // (a batch is as slow as it is wide, and what remains is a chain of batches)
    uint32_t toi_split;         // bit 0: light-contact batches stop at the first sub-step (RQ_TOI), bit 1: touching-contact batches too, bit 2: a sub-step batch takes ONE
                                // sub-step per visit (environments that need another come back through the queue)
    uint32_t toi_min;           // a wave that has other work takes a sub-step batch only when at least this many entries wait
    uint32_t keep_min;          // ... as long as at least this many are on board after the refill
    uint32_t keep;              // bit 0: touching-contact batches keep the environments that stay in their class (roll_contact_batch), bit 1: light ones too
    uint32_t fair;              // 1: waves serve the queue that is furthest behind (RC_HEADT); 0: fixed order touching contact, light contact, reset, free flight
    uint32_t heavy_narrow;      // lanes of a touching-contact batch while that queue is BEHIND the free-flight queue (its chain sets the pace then)
    uint32_t heavy_min, heavy_max;  // ... a touching-contact batch from this many on, of at most this many lanes: K consecutive touching steps of one
                                // environment are the launch's longest chain, and a batch is as slow as it is wide (more sub-step passes, slower lanes)
    uint32_t reset_min;         // ... and a reset batch only when at least this many finished environments wait
    uint32_t free_min;          // ... and goes into free-flight mode only when at least this many entries wait
    uint32_t helper_min;        // a free-flight helper wave (ll_rollout_free_kernel) boards when at least this many entries wait
    uint32_t refill_min;        // a resident wave refills its vacant lanes only when at least this many are vacant
    uint32_t residency;         // ... and, while a wave's worth of free-flight entries waits, trades ALL its environments for waiting ones after this many
                                // steps: the population then advances evenly (entries come out of the ring lowest step index first), and the
                                // launch does not end with a few late starters taking their K steps one 0.6 ms contact batch at a time
    unsigned long long* stat;   // [RS_COUNT] (zeroed before every launch)
    unsigned long long* trace;  // diagnosis (MGYM_LL_ROLL_TRACE=file): [grid][kRollTraceLen] events {ticks since the wave started << 8 | what it begins}; or null
    uint32_t debug;             // diagnosis (MGYM_LL_ROLL_DEBUG): 1 stop after the seed phase, 2 no free-flight mode (every environment through the contact path), 4 resident waves never switch to a contact batch
};
constexpr int kRollTraceLen = 512;
enum { RT_SEED = 1, RT_CONTACT, RT_LIGHT, RT_RESET, RT_FREE, RT_IDLE, RT_END, RT_TOI };
constexpr uint32_t kRollMaxK = 240;          // step index in the top byte of an entry (0xff.. = empty is never a valid entry)
constexpr uint32_t kRollEnvMask = 0xffffffu;
// 120 s of the 100 MHz wall clock: a wave that waits this long gives up loudly (sticky internal error, the launch drains).  The bound only has to turn a deadlock into
// an error before anything outside kills the process; it must NOT be near anything a healthy launch can see: the clock runs on while the waves do not (a launch of this
// round stood still for ~3 s in the middle of its register-only sweeps — whole-GPU, every phase of every wave 50-150 x longer — and the 3 s bound of the time turned that
// stall into an abort; profiles/r04_lunarlander/rollout_stall_record.txt).
constexpr long long kRollTimeoutTicks = 12000000000ll;


}  // namespace mgym
