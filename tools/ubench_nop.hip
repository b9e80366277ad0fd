// Issue-rate experiments for the modular reduction idiom on gfx950 (8 waves per SIMD, all lanes busy).
//  v0  add; subrev_co vcc; cndmask vcc          (one value, back to back -- what red_2p/madd emit)
//  v1  same with s_nop 0 after the cndmask      (what LLVM adds after an inline-asm block)
//  v2  two independent values interleaved, carries in two SGPR pairs
//  v3  add; sub; min                            (no carry)
//  v5  one value, carry in an SGPR pair (VOP3 encodings, P in an SGPR)
//  v6  v0 with P in an SGPR instead of a literal
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_nop.hip -o tools/ubench_nop ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define P 0x78000001u
#define B0(NOP)                                                                     \
    asm volatile("v_add_u32 %0, %0, %1\n\t"                                         \
                 "v_subrev_co_u32 %2, vcc, 0x78000001, %0\n\t"                      \
                 "v_cndmask_b32 %0, %2, %0, vcc\n\t" NOP                            \
                 "v_add_u32 %1, %1, %0\n\t"                                         \
                 "v_subrev_co_u32 %2, vcc, 0x78000001, %1\n\t"                      \
                 "v_cndmask_b32 %1, %2, %1, vcc\n\t" NOP                            \
                 : "+v"(x), "+v"(y), "=&v"(t)::"vcc");
// v6: v0 with P in an SGPR (4-byte VOP2 encodings throughout)
#define B6                                                                          \
    asm volatile("v_add_u32 %0, %0, %1\n\t"                                         \
                 "v_subrev_co_u32 %2, vcc, %3, %0\n\t"                              \
                 "v_cndmask_b32 %0, %2, %0, vcc\n\t"                                \
                 "v_add_u32 %1, %1, %0\n\t"                                         \
                 "v_subrev_co_u32 %2, vcc, %3, %1\n\t"                              \
                 "v_cndmask_b32 %1, %2, %1, vcc\n\t"                                \
                 : "+v"(x), "+v"(y), "=&v"(t) : "s"(P) : "vcc");
// v7 / v8: lazy Montgomery product chain with the constants P, -P^-1 in SGPRs / pinned in VGPRs (plain C)
__device__ __forceinline__ uint32_t mml(uint32_t a, uint32_t b, uint32_t pp, uint32_t mu) {
    uint64_t t = (uint64_t)a * b;
    uint32_t m = (uint32_t)t * mu;
    uint64_t s = t + (uint64_t)m * pp;
    return (uint32_t)(s >> 32);
}
// v9: two independent chains interleaved at the granularity of the (sub, cndmask) pair, both through vcc
#define B9                                                                          \
    asm volatile("v_add_u32 %0, %0, %1\n\t"                                         \
                 "v_add_u32 %2, %2, %3\n\t"                                         \
                 "v_subrev_co_u32 %4, vcc, 0x78000001, %0\n\t"                      \
                 "v_cndmask_b32 %0, %4, %0, vcc\n\t"                                \
                 "v_subrev_co_u32 %5, vcc, 0x78000001, %2\n\t"                      \
                 "v_cndmask_b32 %2, %5, %2, vcc\n\t"                                \
                 "v_add_u32 %1, %1, %0\n\t"                                         \
                 "v_add_u32 %3, %3, %2\n\t"                                         \
                 "v_subrev_co_u32 %4, vcc, 0x78000001, %1\n\t"                      \
                 "v_cndmask_b32 %1, %4, %1, vcc\n\t"                                \
                 "v_subrev_co_u32 %5, vcc, 0x78000001, %3\n\t"                      \
                 "v_cndmask_b32 %3, %5, %3, vcc\n\t"                                \
                 : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "=&v"(t), "=&v"(u)::"vcc");
// two chains (x,y) and (z,w): x += y, z += w then y += x, w += z
#define B2                                                                          \
    asm volatile("v_add_u32 %0, %0, %1\n\t"                                         \
                 "v_add_u32 %2, %2, %3\n\t"                                         \
                 "v_subrev_co_u32 %4, %6, %8, %0\n\t"                       \
                 "v_subrev_co_u32 %5, %7, %8, %2\n\t"                       \
                 "v_cndmask_b32 %0, %4, %0, %6\n\t"                                 \
                 "v_cndmask_b32 %2, %5, %2, %7\n\t"                                 \
                 "v_add_u32 %1, %1, %0\n\t"                                         \
                 "v_add_u32 %3, %3, %2\n\t"                                         \
                 "v_subrev_co_u32 %4, %6, %8, %1\n\t"                       \
                 "v_subrev_co_u32 %5, %7, %8, %3\n\t"                       \
                 "v_cndmask_b32 %1, %4, %1, %6\n\t"                                 \
                 "v_cndmask_b32 %3, %5, %3, %7\n\t"                                 \
                 : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "=&v"(t), "=&v"(u), "=&s"(c0), "=&s"(c1) : "s"(P));
#define B3                                                                          \
    asm volatile("v_add_u32 %0, %0, %1\n\t"                                         \
                 "v_subrev_u32 %2, 0x78000001, %0\n\t"                              \
                 "v_min_u32 %0, %2, %0\n\t"                                         \
                 "v_add_u32 %1, %1, %0\n\t"                                         \
                 "v_subrev_u32 %2, 0x78000001, %1\n\t"                              \
                 "v_min_u32 %1, %2, %1\n\t"                                         \
                 : "+v"(x), "+v"(y), "=&v"(t));
// one value, carry in an SGPR pair instead of vcc
#define B5                                                                          \
    asm volatile("v_add_u32 %0, %0, %1\n\t"                                         \
                 "v_subrev_co_u32 %2, %3, %4, %0\n\t"                       \
                 "v_cndmask_b32 %0, %2, %0, %3\n\t"                                 \
                 "v_add_u32 %1, %1, %0\n\t"                                         \
                 "v_subrev_co_u32 %2, %3, %4, %1\n\t"                       \
                 "v_cndmask_b32 %1, %2, %1, %3\n\t"                                 \
                 : "+v"(x), "+v"(y), "=&v"(t), "=&s"(c0) : "s"(P));
template <int V>
__global__ __launch_bounds__(256) void k(uint32_t* o, int iters) {
    uint32_t x = o[threadIdx.x] % P, y = o[threadIdx.x + 256] % P, z = o[threadIdx.x + 512] % P, w = o[threadIdx.x + 768] % P, t, u;
    uint64_t c0, c1;
    uint32_t pp = P, mu = 0x77ffffffu;
    if (V == 8) asm volatile("" : "+v"(pp), "+v"(mu));  // pin the constants in VGPRs
    for (int i = 0; i < iters; i++) {
        if (V == 0) { B0("") B0("") B0("") B0("") B0("") B0("") B0("") B0("") }
        if (V == 1) { B0("s_nop 0\n\t") B0("s_nop 0\n\t") B0("s_nop 0\n\t") B0("s_nop 0\n\t") B0("s_nop 0\n\t") B0("s_nop 0\n\t") B0("s_nop 0\n\t") B0("s_nop 0\n\t") }
        if (V == 2) { B2 B2 B2 B2 }
        if (V == 3) { B3 B3 B3 B3 B3 B3 B3 B3 }
        if (V == 5) { B5 B5 B5 B5 B5 B5 B5 B5 }
        if (V == 6) { B6 B6 B6 B6 B6 B6 B6 B6 }
        if (V == 9) { B9 B9 B9 B9 }
        if (V == 7 || V == 8) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                x = mml(x, y, pp, mu), z = mml(z, w, pp, mu);
                y = mml(y, x, pp, mu), w = mml(w, z, pp, mu);
            }
        }
    }
    o[blockIdx.x * 256 + threadIdx.x] = x + y + z + w;
}
template <int V>
static void run(const char* name, uint32_t* d) {
    const int iters = 20000, blocks = 256 * 8;  // 8 waves per SIMD
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    double instr = (double)iters * ((V == 7 || V == 8) ? 96 : 48);  // VALU instructions per wave
    printf("%-44s %.3f ms, %.3f ns per VALU instruction per SIMD\n", name, ms, ms * 1e6 / instr / 8);
}
int main() {
    uint32_t* d;
    (void)hipMalloc(&d, 4 << 24);
    (void)hipMemset(d, 1, 4 << 24);
    run<0>("v0 one value, vcc, back to back", d);
    run<1>("v1 + s_nop 0 after each cndmask", d);
    run<2>("v2 two values interleaved, SGPR carries", d);
    run<3>("v3 sub + min", d);
    run<5>("v5 one value, SGPR-pair carry", d);
    run<6>("v6 one value, vcc, P in an SGPR", d);
    run<0>("v0 again", d);
    run<6>("v6 again", d);
    run<9>("v9 two chains interleaved, vcc", d);
    run<0>("v0 again", d);
    run<9>("v9 again", d);
    run<7>("v7 Montgomery product, constants in SGPRs", d);
    run<8>("v8 Montgomery product, constants in VGPRs", d);
    run<7>("v7 again", d);
    run<8>("v8 again", d);
    return 0;
}
