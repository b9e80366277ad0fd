// hist.hpp -- histogram increments for the trace generators' multiplicity columns.
// Execution records repeat themselves (a loop's operands, small timestamp gaps, the same few lookup requests), so a wave's 64
// increments often hit a handful of table entries; plain atomics to one address serialise in L2 (the 16-bit range checker took
// 2.1 ms per 2^19 values of a guest's memory log).  hist_add lets the lanes of a wave that hold the same index elect one lane
// to add their number: one atomic per DISTINCT index per wave.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace zk {

__device__ __forceinline__ void hist_add(uint32_t* counts, uint32_t idx) {
    const unsigned lane = __lane_id();
    uint64_t todo = __ballot(1);   // the lanes that call (any divergence above is reflected here); wave-uniform from here on
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t v = __builtin_amdgcn_readlane(idx, leader);
        const uint64_t same = __ballot(idx == v) & todo;
        if ((int)lane == leader) atomicAdd(&counts[v], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

}  // namespace zk
