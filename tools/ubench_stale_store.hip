// ubench_stale_store.hip -- the reproducer DESIGN.md 15 / 8 asks for: does a tree node that a workgroup stores and then COMPUTES ON from
// (the shape k_compress_top / k_compress_coop_multi had before round 4's change: a word stored under a partial EXEC mask, its registers
// reused by the next permutation's DPP code) ever differ in memory from the node the workgroup went on with -- beside memory-bound and
// LDS-heavy kernels of other streams, as in the guest flow?  Never run yet (written when the round's GPU minutes were spent).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I zkvm-prover_amd/csrc tools/ubench_stale_store.hip -o ubench_stale_store -lpthread
//   ./ubench_stale_store [iterations = 20000] [trees per launch = 64] [noise threads = 3]
// Prints, per variant ("store, then compute on" / "keep, store at the end"), the nodes whose stored value is not the hash of their
// stored children, and for the first of them whether memory holds the PREVIOUS iteration's node (a lost store) or something else.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "poseidon2_coop.hpp"

using namespace zk;

// one workgroup per tree: 128 leaves (already digests) -> 64, 32, ..., 1; layer l of tree t at digests[t][off(l)]
__device__ __forceinline__ unsigned lay_off(unsigned l) { return 256u - (256u >> l); }   // in nodes: 0, 128, 192, ...

template <bool KEEP>
__global__ __launch_bounds__(1024) void k_top(uint32_t* __restrict__ all) {
    __shared__ uint32_t buf[2][128 * 8];
    uint32_t* digests = all + (size_t)blockIdx.x * 256 * 8;
    const unsigned tid = threadIdx.x, lane = tid & 15u, grp = tid >> 4;
    const CoopConsts cc = coop_load_consts(lane);
    buf[0][tid] = digests[tid];
    __syncthreads();
    unsigned cur = 0;
    uint32_t keep[7];
#pragma unroll
    for (unsigned k = 0; k < 7; k++) {
        const unsigned n_next = 64u >> k;
        const unsigned ii = grp < n_next ? grp : 0;
        const uint32_t x = coop_permute_regs(buf[cur][16 * ii + lane], lane, cc);
        keep[k] = x;
        if (grp < n_next && lane < 8) {
            buf[cur ^ 1][8 * grp + lane] = x;
            if (!KEEP) digests[lay_off(k + 1) * 8 + 8 * grp + lane] = x;   // the old shape: store, then compute on
        }
        __syncthreads();
        cur ^= 1;
    }
    if (!KEEP || lane >= 8) return;
#pragma unroll
    for (unsigned k = 0; k < 7; k++)
        if (grp < (64u >> k)) digests[lay_off(k + 1) * 8 + 8 * grp + lane] = keep[k];
}

// every produced node against the hash of its stored children (one lane per node, plain permutation)
__global__ void k_check(const uint32_t* all, unsigned n_trees, uint32_t* report) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_trees * 127u) return;
    const unsigned t = i / 127u, j = i % 127u;   // node j of the produced layers: layer l = 1 .. 7
    unsigned l = 1, first = 0;
    while (j >= first + (128u >> l)) first += 128u >> l, l++;
    const unsigned idx = j - first;
    const uint32_t* d = all + (size_t)t * 256 * 8;
    uint32_t s[16];
    for (int q = 0; q < 16; q++) s[q] = d[(lay_off(l - 1) + 2 * idx) * 8 + q];
    poseidon2_permute_rolled(s);
    bool same = true;
    for (int q = 0; q < 8; q++) same = same && s[q] == d[(lay_off(l) + idx) * 8 + q];
    if (!same) atomicAdd(&report[0], 1u), atomicMin(&report[1], (t << 16) | (l << 8) | idx);
}

__global__ void k_fill_leaves(uint32_t* all, unsigned n_trees, uint32_t seed) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_trees * 1024u) return;
    uint32_t v = (i + 1u) * 2654435761u ^ seed * 40503u;
    v ^= v >> 15, v *= 2246822519u, v ^= v >> 13;
    all[(size_t)(i / 1024u) * 2048 + (i % 1024u)] = v % 2013265921u;
}

// noise: a streaming copy (HBM-bound) and an LDS / VALU loop, on streams of their own
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_lds(uint32_t* out, unsigned iters) {
    extern __shared__ uint32_t sh[];
    uint32_t v = threadIdx.x;
    for (unsigned i = 0; i < iters; i++) {
        sh[(threadIdx.x * 33u + i) & 8191u] = v;
        __syncthreads();
        v = v * 1664525u + sh[(threadIdx.x * 17u + i) & 8191u];
    }
    if (v == 0xdeadbeefu) out[0] = v;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const unsigned n_trees = argc > 2 ? atoi(argv[2]) : 64;
    const int noise = argc > 3 ? atoi(argv[3]) : 3;
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int t = 0; t < noise; t++)
        th.emplace_back([&, t] {
            hipStream_t s;
            hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            uint4 *a = nullptr, *b = nullptr;
            uint32_t* o = nullptr;
            const size_t n = (size_t)64 << 20;   // 1 GiB each way
            hipMalloc(&a, n * 16), hipMalloc(&b, n * 16), hipMalloc(&o, 64);
            while (!stop.load()) {
                if (t % 2 == 0) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s, a, b, n);
                else hipLaunchKernelGGL(k_lds, dim3(512), dim3(1024), 32768, s, o, 4000u);
                hipStreamSynchronize(s);
            }
            hipFree(a), hipFree(b), hipFree(o);
        });
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    uint32_t *d = nullptr, *d_rep = nullptr;
    hipMalloc(&d, (size_t)n_trees * 2048 * 4), hipMalloc(&d_rep, 8);
    std::vector<uint32_t> prev((size_t)n_trees * 2048), now((size_t)n_trees * 2048);
    for (int variant = 0; variant < 2; variant++) {
        long bad_nodes = 0, bad_iters = 0, lost = 0, other = 0;
        for (int it = 0; it < iters; it++) {
            const uint32_t init[2] = {0, 0xffffffffu};
            hipMemcpyAsync(d_rep, init, 8, hipMemcpyHostToDevice, st);
            hipLaunchKernelGGL(k_fill_leaves, dim3(n_trees * 4), dim3(256), 0, st, d, n_trees, (uint32_t)it);
            if (variant == 0) hipLaunchKernelGGL(k_top<false>, dim3(n_trees), dim3(1024), 0, st, d);
            else hipLaunchKernelGGL(k_top<true>, dim3(n_trees), dim3(1024), 0, st, d);
            hipLaunchKernelGGL(k_check, dim3((n_trees * 127 + 255) / 256), dim3(256), 0, st, d, n_trees, d_rep);
            uint32_t rep[2];
            hipMemcpyAsync(rep, d_rep, 8, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
            hipMemcpy(now.data(), d, now.size() * 4, hipMemcpyDeviceToHost);   // (512 KiB: what memory holds after this iteration)
            if (rep[0]) {
                bad_nodes += rep[0], bad_iters++;
                const unsigned t = rep[1] >> 16, l = (rep[1] >> 8) & 0xff, idx = rep[1] & 0xff;
                const size_t o = (size_t)t * 2048 + (size_t)(256u - (256u >> l) + idx) * 8;
                bool same_as_prev = it > 0;
                for (int q = 0; q < 8; q++) same_as_prev = same_as_prev && now[o + q] == prev[o + q];
                (same_as_prev ? lost : other)++;
                if (bad_iters <= 5)
                    std::printf("variant %d iteration %d: %u nodes differ, first tree %u layer %u index %u: memory holds %s\n", variant, it, rep[0], t, l, idx,
                                same_as_prev ? "the PREVIOUS iteration's node (a lost store)" : "neither this nor the previous iteration's node");
            }
            prev.swap(now);
        }
        std::printf("{\"variant\": \"%s\", \"iterations\": %d, \"trees_per_launch\": %u, \"noise_threads\": %d, \"iterations_with_a_wrong_node\": %ld, \"wrong_nodes\": %ld, "
                    "\"previous_value\": %ld, \"other_value\": %ld}\n",
                    variant ? "keep, store at the end" : "store, then compute on", iters, n_trees, noise, bad_iters, bad_nodes, lost, other);
    }
    stop.store(true);
    for (auto& x : th) x.join();
    return 0;
}
