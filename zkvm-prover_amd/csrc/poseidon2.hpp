// poseidon2.hpp -- Poseidon2-BabyBear width-16 permutation, x^7 S-box, 8 external + 13 internal
// rounds, on Montgomery residues.  Host + device (the host copy serves the verifier and keygen).
//
// Restates (not copies) the structure the reference links: p3-poseidon2 0.4.3 external layer
// (MDSMat4 "light" permutation) + p3-baby-bear 0.4.3 internal diagonal + zkhash-axiom 0.2.0
// RC16 round constants (Cargo.lock:5708,5545,10231).  Digest = 8 words
// (crates/types/src/proof.rs:209).  Sponge = PaddingFreeSponge<16,8,8>, compression =
// TruncatedPermutation<2,8,16>.
#pragma once
#include "babybear.hpp"

namespace zk {

struct Poseidon2Consts {
    static constexpr uint32_t RC[141] = {
#include "poseidon2_rc.inc"
    };
};

// diag V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]
// (Montgomery form; entries 0..8 are applied with adds / halving instead of a multiply)
struct Poseidon2Diag {
    // 2^-k in Montgomery form = 2^(32-k) mod p
    static constexpr uint32_t INV_2_8 = (1u << 24);           // 2^-8  * 2^32
    static constexpr uint32_t INV_4 = (1u << 30);             // 2^-2  * 2^32
    static constexpr uint32_t INV_8 = (1u << 29);             // 2^-3  * 2^32
    static constexpr uint32_t INV_16 = (1u << 28);            // 2^-4  * 2^32
    static constexpr uint32_t INV_2_27 = (1u << 5);           // 2^-27 * 2^32
};

ZK_HD uint32_t mdouble(uint32_t x) { return red_2p(x << 1); }
// x * 2^-K mod p for x in [0,p), 1 <= K <= 27, as a K-bit Montgomery step: p == 1 (mod 2^27), so
// -p^-1 == -1 (mod 2^K) and m = (-x) mod 2^K makes x + m*p divisible by 2^K; (x + m*p) >> K <= p with
// equality only for x = 0 (where m = 0 gives 0).  4 VALU instructions (and/sub, one 64-bit mad, one
// funnel shift) instead of the 5 of a general Montgomery product, and no conditional subtraction.
template <int K>
ZK_HD uint32_t mdiv_pow2(uint32_t x) {
    const uint32_t m = (0u - x) & ((1u << K) - 1u);
    const uint64_t t = (uint64_t)m * P + x;
    return (uint32_t)(t >> K);
}
ZK_HD uint32_t mhalve(uint32_t x) { return (x & 1u) ? ((x >> 1) + ((P + 1u) >> 1)) : (x >> 1); }

// x^7 for x in [0,p).  Lazy Montgomery products keep intermediates in [0, 2.02p) -- bounds, with
// c = p/2^32 = 0.46875: x2 < 1.469p, x3 < 1.689p, x4 < 2.012p -> one conditional subtraction ->
// < 1.012p, x7 < 1.801p -> one conditional subtraction -> [0,p).  Every product a*b stays below
// 2.417 p^2, the overflow limit of the 64-bit accumulate.
// (s + rc)^7 for s in [0,p) and a round constant rc in [0,p), result in [0,p).  s + (rc - p) lies in [-p, p) as a signed word
// (one addition, the constant's half is uniform), and signed Montgomery products of values that small stay below 0.97 p in
// magnitude: x2, x3, x4, x7 need no conditional step until the final canon_signed -- 1 + 12 + 2 instructions against the
// 3 + 12 + 4 of sbox7(madd(s, rc)).
// rcs = rc - p (mod 2^32), a constant prepared once: s + rcs is the signed word s + rc - p
ZK_HD uint32_t sbox7_rcs(uint32_t s, uint32_t rcs) {
    const int32_t x = (int32_t)(s + rcs);
    const int32_t x2 = smml(x, x), x3 = smml(x2, x), x4 = smml(x2, x2);
    return canon_signed(smml(x3, x4));
}
ZK_HD uint32_t sbox7_rc(uint32_t s, uint32_t rc) { return sbox7_rcs(s, rc - P); }
ZK_HD uint32_t sbox7(uint32_t x) {
    uint32_t x2 = mmul_lazy(x, x);
    uint32_t x3 = mmul_lazy(x2, x);
    uint32_t x4 = red_2p(mmul_lazy(x2, x2));
    return red_2p(mmul_lazy(x3, x4));
}

// M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on each 4-block, then add column sums.
ZK_HD void p2_external_linear(uint32_t s[16]) {
#pragma unroll
    for (int b = 0; b < 16; b += 4) {
        uint32_t x0 = s[b], x1 = s[b + 1], x2 = s[b + 2], x3 = s[b + 3];
        uint32_t t01 = madd(x0, x1), t23 = madd(x2, x3);
        uint32_t t0123 = madd(t01, t23);
        uint32_t t01123 = madd(t0123, x1), t01233 = madd(t0123, x3);
        s[b + 3] = madd(madd(t01233, x0), x0);  // + 2 x0 as two additions: one instruction fewer than doubling first
        s[b + 1] = madd(madd(t01123, x2), x2);
        s[b + 0] = madd(t01123, t01);
        s[b + 2] = madd(t01233, t23);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t sum = madd(madd(s[k], s[4 + k]), madd(s[8 + k], s[12 + k]));
#pragma unroll
        for (int b = 0; b < 16; b += 4) s[b + k] = madd(s[b + k], sum);
    }
}

ZK_HD void p2_internal_linear(uint32_t s[16]) {
    uint32_t sum = 0;
#ifdef ZK_P2_MODULAR_SUM
    {
        uint32_t a0 = madd(s[0], s[1]), a1 = madd(s[2], s[3]), a2 = madd(s[4], s[5]), a3 = madd(s[6], s[7]);
        uint32_t a4 = madd(s[8], s[9]), a5 = madd(s[10], s[11]), a6 = madd(s[12], s[13]), a7 = madd(s[14], s[15]);
        sum = madd(madd(madd(a0, a1), madd(a2, a3)), madd(madd(a4, a5), madd(a6, a7)));
    }
#else
    {
        // Lazy sum of the 16 lanes: pairs fit 32 bits (2p < 2^32), the eight pair sums are added in 64 bits (< 16p < 2^35)
        // and folded once with 2^32 == 2^28 - 2 (mod p): 30 instructions instead of the 45 of fifteen modular additions.
        const uint32_t a0 = s[0] + s[1], a1 = s[2] + s[3], a2 = s[4] + s[5], a3 = s[6] + s[7];
        const uint32_t a4 = s[8] + s[9], a5 = s[10] + s[11], a6 = s[12] + s[13], a7 = s[14] + s[15];
        const uint64_t t = (uint64_t)a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        const uint64_t x = (uint64_t)(uint32_t)(t >> 32) * 0x0FFFFFFEu + (uint32_t)t;  // < 7 * 2^28 + 2^32 < 2^33
        // x >= 2^32 leaves a low word below 2^31, so adding 2^28 - 2 cannot wrap
        const uint32_t y = (uint32_t)x + ((uint32_t)(x >> 32) ? 0x0FFFFFFEu : 0u);     // < 2^32 < 3p
        sum = red_2p(red_2p(y));
    }
#endif
    uint32_t d;
    s[0] = msub(sum, mdouble(s[0]));                          // -2
    s[1] = madd(sum, s[1]);                                   //  1
    s[2] = madd(madd(sum, s[2]), s[2]);                       //  2
    s[3] = madd(sum, mhalve(s[3]));                           //  1/2
    d = madd(sum, s[4]); s[4] = madd(madd(d, s[4]), s[4]);    //  3
    s[5] = madd(sum, mdouble(mdouble(s[5])));                 //  4
    s[6] = msub(sum, mhalve(s[6]));                           // -1/2
    d = mdouble(s[7]); s[7] = msub(sum, madd(d, s[7]));       // -3
    s[8] = msub(sum, mdouble(mdouble(s[8])));                 // -4
    s[9] = madd(sum, mdiv_pow2<8>(s[9]));                     //  1/2^8
    s[10] = madd(sum, mdiv_pow2<2>(s[10]));                   //  1/4
    s[11] = madd(sum, mdiv_pow2<3>(s[11]));                   //  1/8
    s[12] = madd(sum, mdiv_pow2<27>(s[12]));                  //  1/2^27
    s[13] = msub(sum, mdiv_pow2<8>(s[13]));                   // -1/2^8
    s[14] = msub(sum, mdiv_pow2<4>(s[14]));                   // -1/16
    s[15] = msub(sum, mdiv_pow2<27>(s[15]));                  // -1/2^27
}

ZK_HD void poseidon2_permute(uint32_t s[16]) {
    p2_external_linear(s);
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7_rc(s[i], Poseidon2Consts::RC[r * 16 + i]);
        p2_external_linear(s);
    }
#pragma unroll
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7_rc(s[0], Poseidon2Consts::RC[64 + r]);
        p2_internal_linear(s);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7_rc(s[i], Poseidon2Consts::RC[77 + r * 16 + i]);
        p2_external_linear(s);
    }
}

// The host's permutation: the AVX-512 form (csrc/poseidon2_avx512.cpp, the whole state in one register: about half the scalar code's
// time) where the CPU has it -- the verifier, the aggregation witness generator and the transcript's long absorptions hash on the host.
// (-DZK_NO_HOST_AVX512: builds of single sources without csrc/poseidon2_avx512.cpp, e.g. the sanitizer build of the verifier)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__) && !defined(ZK_NO_HOST_AVX512)
#define ZK_HAVE_HOST_AVX512 1
void poseidon2_permute_avx512(uint32_t s[16]);
void poseidon2_permute16_avx512(uint32_t* t);   // sixteen independent permutations: t[16 w + k] = word w of instance k
// sixteen states side by side (transposed): the vector form where the CPU has it, else sixteen scalar permutations
inline void poseidon2_permute16_host(uint32_t* t) {
    static const bool fast = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    if (fast) return poseidon2_permute16_avx512(t);
    for (int k = 0; k < 16; k++) {
        uint32_t s[16];
        for (int w = 0; w < 16; w++) s[w] = t[16 * w + k];
        poseidon2_permute(s);
        for (int w = 0; w < 16; w++) t[16 * w + k] = s[w];
    }
}
inline void poseidon2_permute_host(uint32_t s[16]) {
    static const bool fast = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    if (fast) poseidon2_permute_avx512(s);
    else poseidon2_permute(s);
}
#else
inline void poseidon2_permute_host(uint32_t s[16]) { poseidon2_permute(s); }
inline void poseidon2_permute16_host(uint32_t* t) {
    for (int k = 0; k < 16; k++) {
        uint32_t s[16];
        for (int w = 0; w < 16; w++) s[w] = t[16 * w + k];
        poseidon2_permute(s);
        for (int w = 0; w < 16; w++) t[16 * w + k] = s[w];
    }
}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_P2_PERMUTE_HERE(s) poseidon2_permute(s)
#else
#define ZK_P2_PERMUTE_HERE(s) poseidon2_permute_host(s)
#endif

// host-side helpers (sponge over a slice, 2-to-1 compression) for the verifier / keygen
ZK_HD void p2_hash_slice(const uint32_t* in, size_t len, uint32_t out[8]) {
    uint32_t s[16];
    for (int i = 0; i < 16; i++) s[i] = 0;
    size_t i = 0;
    while (i < len) {
        size_t n = len - i < 8 ? len - i : 8;
        for (size_t k = 0; k < n; k++) s[k] = in[i + k];
        ZK_P2_PERMUTE_HERE(s);
        i += n;
    }
    for (int k = 0; k < 8; k++) out[k] = s[k];
}
ZK_HD void p2_compress(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]) {
    uint32_t s[16];
    for (int k = 0; k < 8; k++) {
        s[k] = l[k];
        s[8 + k] = r[k];
    }
    ZK_P2_PERMUTE_HERE(s);
    for (int k = 0; k < 8; k++) out[k] = s[k];
}

}  // namespace zk
