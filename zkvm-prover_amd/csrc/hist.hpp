// hist.hpp -- histogram increments for the trace generators' multiplicity columns.
// Execution records repeat themselves (a loop's operands, small timestamp gaps, the same few lookup requests), so a wave's 64
// increments often hit a handful of table entries; plain atomics to one address serialise in L2 (the 16-bit range checker took
// 2.1 ms per 2^19 values of a guest's memory log).  hist_add lets the lanes of a wave that hold the same index elect one lane
// to add their number: one atomic per DISTINCT index per wave.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lds_barrier.hpp"

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

// Workgroup-level front end for the generators whose rows keep requesting the same few table entries (a shift row asks for
// (q, 32 q) with q in 0..7; a division's high quotient limbs are (0, 0)): a direct-mapped cache of HOT_SLOTS (index, count)
// pairs in LDS takes the wave-merged increments, a colliding index falls through to HBM, and the cache is merged into the table
// once per workgroup.  Kernels that use it walk their rows grid-stride in a few hundred fat workgroups, so a hot entry gets
// hundreds of atomics instead of one per wave.
constexpr uint32_t HOT_SLOTS = 1024, HOT_EMPTY = 0xffffffffu;

__device__ __forceinline__ void hot_init(uint32_t* keys, uint32_t* cnts) {
    for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += blockDim.x) keys[i] = HOT_EMPTY, cnts[i] = 0;
    zk_syncthreads();
}
__device__ __forceinline__ void hot_add(uint32_t* keys, uint32_t* cnts, uint32_t* table, uint32_t idx) {
    const unsigned lane = __lane_id();
    uint64_t todo = __ballot(1);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t v = __builtin_amdgcn_readlane(idx, leader);
        const uint64_t same = __ballot(idx == v) & todo;
        if ((int)lane == leader) {
            const uint32_t c = (uint32_t)__popcll(same), slot = (v * 2654435761u) >> 22;
            const uint32_t old = atomicCAS(&keys[slot], HOT_EMPTY, v);
            if (old == HOT_EMPTY || old == v) atomicAdd(&cnts[slot], c);
            else atomicAdd(&table[v], c);
        }
        todo &= ~same;
    }
}
// every thread of the workgroup must call it (after its last hot_add)
__device__ __forceinline__ void hot_flush(const uint32_t* keys, const uint32_t* cnts, uint32_t* table) {
    zk_syncthreads();
    for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += blockDim.x)
        if (keys[i] != HOT_EMPTY && cnts[i]) atomicAdd(&table[keys[i]], cnts[i]);
}
// rows of a generator per launch: at most this many workgroups of 256 walk them grid-stride
constexpr unsigned HOT_MAX_BLOCKS = 1024;

}  // namespace zk
