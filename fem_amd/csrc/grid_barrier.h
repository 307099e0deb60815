// fem_amd/csrc/grid_barrier.h -- a barrier across every workgroup of a resident grid, XCD-hierarchical (MI355X_MICROARCH.md, price
// table row barrier-xcd; measured in this tree by tools/grid_sync_bench.hip: 6.06 us at 529 workgroups, 6.7 us with a 128-B record
// per workgroup published and a peer's read back, profiles/r06_grid_barrier_cost.txt).
//   arrive:  every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup's barrier, one lane adds to its XCC's counter
//   leader:  the XCC's last arriver writes the XCD's L2 back (release, agent scope), adds to the top counter, polls it with relaxed
//            sc1 loads until every XCC in use has arrived, acquires, and bumps the XCC's generation word
//   others:  poll that word (same L2: relaxed sc1 loads + s_sleep), then acquire once
// Counters are monotonic (no reset inside a launch).  Every spin is bounded: a grid that is not fully resident, or a peer that left,
// raises *err and every workgroup leaves its loop -- the host then takes the launch-per-kernel path.
// Words (unsigned, each on a 128-B line of its own; GridBarrier::WORDS in all, zeroed by the host before the launch):
//   [0..8) arrivals per XCC   [8] arrivals of XCC leaders   [9..17) generation per XCC   [17..25) members per XCC (census)
//   [25] census arrivals
#pragma once
#include <hip/hip_runtime.h>

namespace fem {

struct GridBarrier {
    static constexpr int LINE = 32, WORDS = 26 * LINE;
    unsigned* w;
    int* err;
};
struct GridBarrierLane {   // what a workgroup carries: its XCC, how many workgroups share it, how many XCCs are in use, barriers passed
    unsigned xcc, members, nxcc, gen;
};
__device__ __forceinline__ unsigned gb_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr long GB_SPIN_LIMIT = 1000000L;   // polls (each behind an s_sleep): of the order of a second

// once per launch: where this workgroup sits and who shares its XCC (nwg: workgroups that take part).  sh: 3 words of LDS.
__device__ __forceinline__ GridBarrierLane grid_barrier_census(const GridBarrier& b, unsigned nwg, unsigned* sh)
{
    constexpr int L = GridBarrier::LINE;
    if (threadIdx.x == 0) {
        const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;   // HW_REG_XCC_ID[3:0]
        __hip_atomic_fetch_add(b.w + (17 + x) * L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(b.w + 25 * L, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (gb_ld(b.w + 25 * L) < nwg) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > GB_SPIN_LIMIT) { *b.err = 1; break; }
        }
        unsigned n = 0;
        for (int i = 0; i < 8; i++) n += gb_ld(b.w + (17 + i) * L) != 0;
        sh[0] = x; sh[1] = gb_ld(b.w + (17 + x) * L); sh[2] = n;
    }
    __syncthreads();
    return GridBarrierLane{sh[0], sh[1], sh[2], 0u};   // (sh is next written behind the leading barrier of grid_barrier)
}
// false: the barrier gave up (*err raised by this or another workgroup); the caller leaves its loop
__device__ __forceinline__ bool grid_barrier(const GridBarrier& b, GridBarrierLane& l, unsigned* sh)
{
    constexpr int L = GridBarrier::LINE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave's stores have reached this XCD's L2
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned ok = 1;
        const unsigned target = l.gen + 1;
        const unsigned prev = __hip_atomic_fetch_add(b.w + l.xcc * L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        if (prev == l.members * target - 1) {          // the XCC's last arriver: leader of this round
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(b.w + 8 * L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (gb_ld(b.w + 8 * L) < l.nxcc * target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023) == 0 && (spins > GB_SPIN_LIMIT || *(volatile int*)b.err)) { *b.err = 1; ok = 0; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(b.w + (9 + l.xcc) * L, ok ? target : 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a leader that gave up releases its XCC too)
        } else {
            unsigned g;
            while ((g = gb_ld(b.w + (9 + l.xcc) * L)) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023) == 0 && (spins > GB_SPIN_LIMIT || *(volatile int*)b.err)) { *b.err = 1; ok = 0; break; }
            }
            if (g == 0xffffffffu) ok = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        sh[0] = ok;
    }
    l.gen++;
    __syncthreads();
    return sh[0] != 0;   // (sh[0] is next written behind the leading barrier of the next call)
}

}  // namespace fem
