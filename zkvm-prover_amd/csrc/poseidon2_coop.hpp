// poseidon2_coop.hpp -- one Poseidon2 permutation spread over 16 lanes (lane i holds state word i).
//
// The proving path has strictly serial permutation chains (the Fiat-Shamir sponge, T4; the top
// levels of every Merkle tree, K3) where one-state-per-lane leaves 63/64 of a wave idle and a
// permutation costs ~20 us of dependent instructions.  Here the 16 S-boxes of an external round
// run in parallel lanes and the linear layers use DPP cross-lane moves inside the 16-lane row
// (quad_perm rotations for the 4x4 MDS block, row_ror for the column / full sums): ~8x shorter
// dependent chain.  Same arithmetic as poseidon2.hpp (p3-poseidon2 layers, zkhash RC16).
#pragma once
#include "poseidon2.hpp"

namespace zk {

#if defined(__HIPCC__)
// diag V (Montgomery form), see poseidon2.hpp
__device__ const uint32_t POSEIDON2_DIAG_DEV[16] = {
    P - 0x1ffffffcu /* -2 */, 0x0ffffffeu /* 1 */, 0x1ffffffcu /* 2 */, 0x07ffffffu /* 1/2 */,
    0x2ffffffau /* 3 */, 0x3ffffff8u /* 4 */, P - 0x07ffffffu /* -1/2 */, P - 0x2ffffffau /* -3 */,
    P - 0x3ffffff8u /* -4 */, 1u << 24 /* 2^-8 */, 1u << 30 /* 1/4 */, 1u << 29 /* 1/8 */,
    1u << 5 /* 2^-27 */, P - (1u << 24), P - (1u << 28) /* -1/16 */, P - (1u << 5)};

// Rolled-round form of poseidon2_permute for the bulk hashing kernels: the three round groups are loops and
// the round constants come from the constant address space (scalar loads into SGPRs).  Same arithmetic.
// Measured on the 2^23 x 300 row hash: 47.0 ms against 50.3 ms for the fully unrolled form with literal
// constants (whose ~55 KB body, present twice in the kernel for full and ragged blocks, overflows the 64 KB
// instruction cache two CUs share); unrolling the internal rounds, unrolling the external rounds by two or
// merging both external groups into one body were all slower (47.2 - 48.2 ms).
static __constant__ const uint32_t POSEIDON2_RC_CONST[144] = {
#include "poseidon2_rc.inc"
};
// The bulk kernels add rc - p (mod 2^32), the form sbox7_rcs takes: the compiler would otherwise reassociate
// s + (rc - p) into two vector additions (2^23 x 300 row hash: 45.2 - 46.0 ms against 46.2 - 46.8).  One scalar load per
// constant, each just ahead of its S-box: fetching the 16 constants of an external round as one s_load_dwordx16 makes
// the compiler interleave all 16 S-boxes (84 VGPRs, 5 waves per SIMD, or spills at 8) and measured 1 - 3 % SLOWER.
struct Poseidon2RcShifted {
    uint32_t v[144];
};
constexpr Poseidon2RcShifted poseidon2_rc_shifted() {
    Poseidon2RcShifted t{};
    // [0,64) initial external rounds, [64,77) internal rounds, [80,144) final external rounds
    for (int i = 0; i < 77; i++) t.v[i] = Poseidon2Consts::RC[i] - P;
    for (int i = 0; i < 64; i++) t.v[80 + i] = Poseidon2Consts::RC[77 + i] - P;
    return t;
}
static __constant__ const Poseidon2RcShifted POSEIDON2_RCS_CONST = poseidon2_rc_shifted();
__device__ const Poseidon2RcShifted POSEIDON2_RCS_DEV = poseidon2_rc_shifted();
__device__ __forceinline__ void poseidon2_permute_rolled(uint32_t (&s)[16]) {
    typedef const __attribute__((address_space(4))) uint32_t* cptr;
    cptr rc = (cptr)POSEIDON2_RCS_CONST.v;
    p2_external_linear(s);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7_rcs(s[i], rc[r * 16 + i]);
        p2_external_linear(s);
    }
#pragma unroll 1
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7_rcs(s[0], rc[64 + r]);
        p2_internal_linear(s);
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7_rcs(s[i], rc[80 + r * 16 + i]);
        p2_external_linear(s);
    }
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// rotate within each quad: lane q reads lane (q+k) mod 4
#define ZK_QROT1 0x39
#define ZK_QROT2 0x4E
#define ZK_QROT3 0x93
// rotate right by n within the 16-lane row: lane i reads lane (i - n) mod 16
#define ZK_ROR(n) (0x120 + (n))

__device__ __forceinline__ uint32_t coop_external_linear(uint32_t x) {
    const uint32_t r1 = dpp<ZK_QROT1>(x), r2 = dpp<ZK_QROT2>(x), r3 = dpp<ZK_QROT3>(x);
    const uint32_t t = madd(madd(x, r1), madd(r2, r3));
    const uint32_t y = madd(madd(t, x), mdouble(r1));  // 2x_q + 3x_{q+1} + x_{q+2} + x_{q+3}
    uint32_t s = madd(y, dpp<ZK_ROR(8)>(y));
    s = madd(s, dpp<ZK_ROR(4)>(s));                    // sum over the four quads, same position
    return madd(y, s);
}
__device__ __forceinline__ uint32_t coop_sum16(uint32_t x) {
    uint32_t s = madd(x, dpp<ZK_ROR(8)>(x));
    s = madd(s, dpp<ZK_ROR(4)>(s));
    s = madd(s, dpp<ZK_ROR(2)>(s));
    return madd(s, dpp<ZK_ROR(1)>(s));
}
// lane = index within the 16-lane row (0..15); all 16 lanes of the row must be active
__device__ __forceinline__ uint32_t coop_permute(uint32_t x, unsigned lane) {
    x = coop_external_linear(x);
#pragma unroll 1
    for (int r = 0; r < 4; r++) x = coop_external_linear(sbox7_rcs(x, POSEIDON2_RCS_DEV.v[r * 16 + lane]));
    const uint32_t d = POSEIDON2_DIAG_DEV[lane];
#pragma unroll 1
    for (int r = 0; r < 13; r++) {
        const uint32_t sb = sbox7_rcs(x, POSEIDON2_RCS_DEV.v[64 + r]);
        x = lane == 0 ? sb : x;
        x = madd(mmul(x, d), coop_sum16(x));
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) x = coop_external_linear(sbox7_rcs(x, POSEIDON2_RCS_DEV.v[80 + r * 16 + lane]));
    return x;
}
// The same permutation with the round constants of this lane held in registers: for kernels that chain many permutations on one
// wave (the transcript), where the constant loads of the rolled form sit on the dependent chain of every round.
struct CoopConsts {
    uint32_t ext[8], in[13], diag;
};
__device__ __forceinline__ CoopConsts coop_load_consts(unsigned lane) {
    CoopConsts c;
#pragma unroll
    for (int r = 0; r < 4; r++) c.ext[r] = POSEIDON2_RCS_DEV.v[r * 16 + lane], c.ext[4 + r] = POSEIDON2_RCS_DEV.v[80 + r * 16 + lane];
#pragma unroll
    for (int r = 0; r < 13; r++) c.in[r] = POSEIDON2_RCS_DEV.v[64 + r];
    c.diag = POSEIDON2_DIAG_DEV[lane];
    return c;
}
__device__ __forceinline__ uint32_t coop_permute_regs(uint32_t x, unsigned lane, const CoopConsts& c) {
    x = coop_external_linear(x);
#pragma unroll
    for (int r = 0; r < 4; r++) x = coop_external_linear(sbox7_rcs(x, c.ext[r]));
#pragma unroll
    for (int r = 0; r < 13; r++) {
        const uint32_t sb = sbox7_rcs(x, c.in[r]);
        x = lane == 0 ? sb : x;
        x = madd(mmul(x, c.diag), coop_sum16(x));
    }
#pragma unroll
    for (int r = 0; r < 4; r++) x = coop_external_linear(sbox7_rcs(x, c.ext[4 + r]));
    return x;
}
#endif

}  // namespace zk
