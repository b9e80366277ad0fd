// lds_barrier.hpp -- the workgroup barrier of every kernel that orders LDS accesses.
//
// gfx950's s_barrier does not wait for the LDS traffic a wave has in flight: a ds_write that is still queued when its wave reaches the
// barrier can be overtaken by the ds_read another wave (on another SIMD: each has a queue of its own towards the LDS) issues behind the
// barrier -- that wave then reads what the PREVIOUS workgroup left at the address.  __syncthreads() is fence(release, workgroup) +
// s_barrier + fence(acquire, workgroup) and the release fence is what should become `s_waitcnt lgkmcnt(0)`; hipcc 7.2 lost that wait on
// the back edge of a loop whose LDS write ends the body and whose barrier opens the next iteration (the fused tree kernel of round 4:
// docs/stale_node.md; measured: tools/ubench_barrier_race.hip).  The wait is therefore stated here, in front of every barrier, where no
// compiler pass may drop it (an explicit S_WAITCNT is only ever merged with stronger ones); tools/check_barriers.py re-derives from the
// compiled assembly that no barrier of the library is reachable with an LDS write in flight (tests/test_barrier_waits_cpu.py).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void zk_syncthreads() {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // s_waitcnt lgkmcnt(0)  (gfx9 encoding: vmcnt 63 = no wait, expcnt 7 = no wait, lgkmcnt 0)
    __syncthreads();
}
