// ubench_barrier_race.hip -- does gfx950's s_barrier wait for the LDS writes a wave still has in flight?  It does not: the premise of
// docs/stale_node.md, measured outside the prover.
//
// One workgroup of 256 lanes = four waves, one per SIMD; each SIMD has a queue of its own towards the CU's LDS.  Per round every lane
// writes a word of a double-buffered array, the workgroup passes a barrier, and every lane reads the word that the NEXT wave wrote (a
// cross-wave dependency through the LDS, like rows 2 and 3 of the third layer of round 4's fused tree kernel).  In front of its word the
// odd waves queue `NOISE` LDS writes whose 64 lanes all hit one bank (64-way conflicts: tens of cycles each), so their real write sits
// deep in the queue when the wave reaches the barrier.
//   variant 0: ds_write ; s_barrier                      -- the shape hipcc 7.2 emitted on the loop's back edge
//   variant 1: ds_write ; s_waitcnt lgkmcnt(0) ; s_barrier -- what __syncthreads() stands for (csrc/lds_barrier.hpp)
// A read that returns the word of two rounds earlier counts as stale.  Expected: variant 0 stale > 0, variant 1 stale = 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_barrier_race.hip -o ubench_barrier_race && ./ubench_barrier_race [rounds = 20000] [workgroups = 1024] [noise = 16]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

// the LDS byte offset a ds_* instruction takes (a generic pointer's low half is not guaranteed to be it)
__device__ __forceinline__ unsigned lds_offset(unsigned* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)p; }

template <int WAIT>
__global__ __launch_bounds__(256) void k_race(unsigned rounds, unsigned noise, unsigned long long* stale, unsigned long long* reads) {
    __shared__ unsigned data[2][256];
    __shared__ unsigned junk[64 * 32 + 64];
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    data[0][tid] = 0xffffffffu, data[1][tid] = 0xffffffffu;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __syncthreads();
    unsigned long long my_stale = 0, my_reads = 0;
    for (unsigned r = 0; r < rounds; r++) {
        const unsigned buf = r & 1u;
        const unsigned value = (r << 8) | tid;
        if (wave & 1u)   // the odd waves bury their write behind bank-conflicting ones (every lane of the wave in bank 0: stride 32 words)
            for (unsigned k = 0; k < noise; k++) {
                const unsigned a = lds_offset(&junk[(tid & 63u) * 32u]);
                asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(k) : "memory");
            }
        const unsigned addr = lds_offset(&data[buf][tid]);
        if (WAIT) asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"v"(addr), "v"(value) : "memory");
        else asm volatile("ds_write_b32 %0, %1\n\ts_barrier" ::"v"(addr), "v"(value) : "memory");
        const unsigned from = (tid + 64u) & 255u;
        const unsigned got = data[buf][from];
        my_reads++;
        if (got != ((r << 8) | from)) my_stale++;
        // (no second barrier is needed for the double buffer when barriers work: a wave can be at most one round ahead)
    }
    atomicAdd(stale, my_stale), atomicAdd(reads, my_reads);
}

int main(int argc, char** argv) {
    const unsigned rounds = argc > 1 ? atoi(argv[1]) : 20000, wgs = argc > 2 ? atoi(argv[2]) : 1024, noise = argc > 3 ? atoi(argv[3]) : 16;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 16) != hipSuccess) return 2;
    for (int variant = 0; variant < 2; variant++) {
        (void)hipMemset(d, 0, 16);
        if (variant == 0) hipLaunchKernelGGL(k_race<0>, dim3(wgs), dim3(256), 0, 0, rounds, noise, d, d + 1);
        else hipLaunchKernelGGL(k_race<1>, dim3(wgs), dim3(256), 0, 0, rounds, noise, d, d + 1);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        unsigned long long h[2];
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        std::printf("{\"variant\": \"%s\", \"rounds\": %u, \"workgroups\": %u, \"noise_writes\": %u, \"cross_wave_reads\": %llu, \"stale_reads\": %llu}\n",
                    variant ? "ds_write ; s_waitcnt lgkmcnt(0) ; s_barrier" : "ds_write ; s_barrier", rounds, wgs, noise, h[1], h[0]);
    }
    (void)hipFree(d);
    return 0;
}
