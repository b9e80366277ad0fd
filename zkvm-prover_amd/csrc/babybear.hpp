// babybear.hpp -- BabyBear (p = 2^31 - 2^27 + 1) in Montgomery form, R = 2^32, and its
// quartic extension F[x]/(x^4 - 11).  Shared by the gfx950 kernels and by the host-side
// planner/verifier of libzkhip (never by oracle/, which has its own canonical-form code).
//
// Matches the memory convention of the reference's field crate: p3-baby-bear / p3-monty-31
// 0.4.3 store elements as Montgomery u32 in [0,p) (Cargo.lock:5545,5685; canonical values
// only appear at serialization, crates/prover/src/prover/mod.rs:136-137).
//
// gfx950 cost model (tools/ubench_valu.hip, measured): v_mad_u64_u32 4.2, v_mul_lo_u32 4.4,
// v_mul_hi_u32 4.1, v_add/sub 2.3, v_min_u32 4.1 issue cycles per wave64 instruction.  The
// cheapest Montgomery product is therefore two v_mad_u64_u32 + one v_mul_lo_u32:
//     t = a*b;  m = lo(t) * (-p^-1 mod 2^32);  s = t + m*p;  r = hi(s)   (r < 2p)
// followed by one conditional subtraction (v_sub + v_min).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

constexpr uint32_t P = 0x78000001u;          // 2013265921
constexpr uint32_t MONTY_MU = 0x88000001u;   // p * MU == 1 (mod 2^32)
constexpr uint32_t MONTY_NEG_MU = 0u - MONTY_MU;  // -p^-1 mod 2^32 = 0x77ffffff
constexpr uint32_t MONTY_ONE = 0x0ffffffeu;  // 2^32 mod p
constexpr uint32_t MONTY_R2 = 1172168163u;   // 2^64 mod p (checked in tests)
constexpr uint32_t GEN_2_27_CANON = 0x1a427a41u;  // generator of the 2^27 subgroup (canonical)
constexpr uint32_t FIELD_GEN_CANON = 31u;          // multiplicative generator (canonical)

// ---- raw Montgomery arithmetic on u32 ------------------------------------------------------
// reduce x in [0, 2p) to [0, p).  (For x in [p, 2^32) it returns x - p, so inputs slightly above
// 2p come back slightly above p: the lazy S-box relies on that.)
ZK_HD uint32_t red_2p(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_sub_co + v_cndmask are full-rate VOP2 (2.3 cycles each); v_min_u32 is half rate (4.1)
    uint32_t y;
    // P as a 32-bit literal ("i"): measured 5 % faster in the hash kernel than holding it in an SGPR ("s"),
    // although the encoding is 4 bytes longer.
    asm("v_subrev_co_u32 %0, vcc, %2, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "=&v"(y) : "v"(x), "i"(P) : "vcc");
    return y;
#else
    uint32_t y = x - P;
    return y < x ? y : x;  // unsigned min(x, x-p): if x < p, x-p wraps to a huge value
#endif
}
// product in [0, 2p) given a*b < 2.417 p^2  (a*b + 2^32 p must fit 64 bits)
ZK_HD uint32_t mmul_lazy(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t m = (uint32_t)t * MONTY_NEG_MU;
    uint64_t s = t + (uint64_t)m * P;
    return (uint32_t)(s >> 32);
}
ZK_HD uint32_t mmul(uint32_t a, uint32_t b) { return red_2p(mmul_lazy(a, b)); }
// Signed Montgomery product: d == a * b * 2^-32 (mod p) with |d| <= |a*b| / 2^32 + p/2, for any int32 a, b (|a*b| <= 2^62 keeps
// t - m*p inside 64 bits).  Operands in [-p, p] give |d| < 0.97 p: a chain of products needs NO conditional step between
// them, where the unsigned form (result in [0, 2p)) has to be brought below the 2.417 p^2 product bound again.  Same three
// instructions (v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32).  Sums of two such values do not fit 32 bits (2p > 2^31), so
// everything additive stays in the unsigned [0, p) form; canon_signed() is the way back (2 full-rate instructions).
ZK_HD int32_t smml(int32_t a, int32_t b) {
    const int64_t t = (int64_t)a * b;
    const int32_t m = (int32_t)((uint32_t)t * MONTY_MU);
    const int64_t s = t - (int64_t)m * (int32_t)P;  // low word == 0
    return (int32_t)(s >> 32);
}
// d in (-p, p) -> [0, p)
ZK_HD uint32_t canon_signed(int32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t y;
    asm("v_add_co_u32 %0, vcc, %2, %1\n\tv_cndmask_b32 %0, %1, %0, vcc" : "=&v"(y) : "v"(d), "i"(P) : "vcc");
    return y;
#else
    return d < 0 ? (uint32_t)(d + (int32_t)P) : (uint32_t)d;
#endif
}
ZK_HD uint32_t madd(uint32_t a, uint32_t b) { return red_2p(a + b); }
ZK_HD uint32_t msub(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d, e;
    asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_add_u32 %1, %4, %0\n\tv_cndmask_b32 %0, %0, %1, vcc"
        : "=&v"(d), "=&v"(e)
        : "v"(a), "v"(b), "i"(P)
        : "vcc");
    return d;
#else
    uint32_t d = a - b;
    uint32_t e = d + P;
    return e < d ? e : d;  // if a < b, d wrapped (huge) and d+p wraps back to the small right value
#endif
}
// [0, p) -> the representative in (-p/2, p/2]
ZK_HD int32_t center_signed(uint32_t x) { return x > (P >> 1) ? (int32_t)(x - P) : (int32_t)x; }
// signed Montgomery step of a 64-bit sum of products: d == t * 2^-32 (mod p), |d| <= |t| / 2^32 + p/2 (|t| < 1.21 p^2 keeps it in 32 bits)
ZK_HD int32_t smred64(int64_t t) {
    const int32_t m = (int32_t)((uint32_t)t * MONTY_MU);
    const int64_t s = t - (int64_t)m * (int32_t)P;
    return (int32_t)(s >> 32);
}
// d in (-2p, 2p) as far as it fits a signed word -> [0, p)
ZK_HD uint32_t canon_signed_wide(int32_t d) {
    if (d < 0) d += (int32_t)P;
    if (d < 0) d += (int32_t)P;
    if (d >= (int32_t)P) d -= (int32_t)P;
    return (uint32_t)d;
}
ZK_HD uint32_t mneg(uint32_t a) { return a == 0 ? 0 : P - a; }
ZK_HD uint32_t to_monty(uint32_t canon) { return mmul(canon, MONTY_R2); }
ZK_HD uint32_t from_monty(uint32_t m) { return mmul(m, 1u); }
ZK_HD uint32_t mpow(uint32_t a, uint64_t e) {
    uint32_t r = MONTY_ONE;
    while (e) {
        if (e & 1) r = mmul(r, a);
        a = mmul(a, a);
        e >>= 1;
    }
    return r;
}
ZK_HD uint32_t minv(uint32_t a) { return mpow(a, P - 2); }
// generator of the order-2^bits subgroup, Montgomery form
ZK_HD uint32_t two_adic_generator(unsigned bits) {
    uint32_t g = to_monty(GEN_2_27_CANON);
    for (unsigned i = bits; i < 27; i++) g = mmul(g, g);
    return g;
}
ZK_HD uint32_t bitrev32(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

// ---- extension field element: 4 Montgomery coefficients -------------------------------------
struct Ext {
    uint32_t c[4];
};
constexpr uint32_t EXT_W_CANON = 11;

ZK_HD Ext ext_zero() { return Ext{{0, 0, 0, 0}}; }
ZK_HD Ext ext_one() { return Ext{{MONTY_ONE, 0, 0, 0}}; }
ZK_HD Ext ext_from_base(uint32_t a) { return Ext{{a, 0, 0, 0}}; }
ZK_HD Ext ext_add(const Ext& a, const Ext& b) {
    return Ext{{madd(a.c[0], b.c[0]), madd(a.c[1], b.c[1]), madd(a.c[2], b.c[2]), madd(a.c[3], b.c[3])}};
}
ZK_HD Ext ext_sub(const Ext& a, const Ext& b) {
    return Ext{{msub(a.c[0], b.c[0]), msub(a.c[1], b.c[1]), msub(a.c[2], b.c[2]), msub(a.c[3], b.c[3])}};
}
ZK_HD Ext ext_neg(const Ext& a) { return Ext{{mneg(a.c[0]), mneg(a.c[1]), mneg(a.c[2]), mneg(a.c[3])}}; }
ZK_HD Ext ext_mul_base(const Ext& a, uint32_t b) {
    return Ext{{mmul(a.c[0], b), mmul(a.c[1], b), mmul(a.c[2], b), mmul(a.c[3], b)}};
}
// Montgomery-reduce a sum of up to four products of residues (t < 4 p^2 < 2^64) to [0,p).
// The low word of t + m*p is zero by construction, so the shift is done on the halves to
// avoid the 65-bit intermediate.
ZK_HD uint32_t mred64(uint64_t t) {
    uint32_t m = (uint32_t)t * MONTY_NEG_MU;
    uint64_t s = (uint64_t)(uint32_t)t + (uint64_t)m * P;  // < 2^32 + 2^32 p ; low word == 0
    uint64_t r = (t >> 32) + (s >> 32);                    // (t + m p) / 2^32 < 2.875 p
    if (r >= 2ull * P) r -= 2ull * P;
    return red_2p((uint32_t)r);
}
// Split accumulator for long sums of Montgomery products (sum_k a_k * b_k with hundreds of terms): groups of four
// products are summed in 64 bits (4 p^2 < 2^64) and the group sum is banked as two halves -- hi += t >> 32,
// lo += t & (2^32 - 1), four full-rate instructions -- instead of being Montgomery-reduced and added modularly (13).
// One reduction at the end: sum * 2^-32 = hi + lo * 2^-32 (mod p).  Good for < 2^31 groups (hi, lo < 2^63).
struct LazyAcc {
    uint64_t hi = 0, lo = 0;
    ZK_HD void add_group(uint64_t t) {
        hi += t >> 32;
        lo += (uint32_t)t;
    }
    // the Montgomery-form value of the banked sum: (hi * 2^32 + lo) * 2^-32 mod p
    ZK_HD uint32_t reduce() const {
        // hi < 2^63: split once more so that every piece satisfies mred64's bound (< 0.94 * 2^64)
        const uint32_t h = mmul(mred64(hi & 0xffffffffull), MONTY_R2);                     // (hi mod 2^32) mod p
        const uint32_t hh = mmul(mmul(mred64(hi >> 32), MONTY_R2), MONTY_R2);              // (hi >> 32) * 2^32 mod p
        const uint32_t l = madd(mred64(lo & 0xffffffffull), mmul(mred64(lo >> 32), MONTY_R2));  // lo * 2^-32 mod p
        return madd(madd(h, hh), l);
    }
};
constexpr uint32_t EXT_W_MONTY = 939524073u;  // 11 * 2^32 mod p
ZK_HD Ext ext_mul(const Ext& a, const Ext& b) {
    // schoolbook, delayed reduction: every coefficient sums <= 4 products in 64 bits.
    uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
    uint32_t b0 = b.c[0], b1 = b.c[1], b2 = b.c[2], b3 = b.c[3];
    uint32_t r4 = mred64((uint64_t)a1 * b3 + (uint64_t)a2 * b2 + (uint64_t)a3 * b1);
    uint32_t r5 = mred64((uint64_t)a2 * b3 + (uint64_t)a3 * b2);
    uint32_t r6 = mred64((uint64_t)a3 * b3);
    uint32_t c0 = mred64((uint64_t)a0 * b0);
    uint32_t c1 = mred64((uint64_t)a0 * b1 + (uint64_t)a1 * b0);
    uint32_t c2 = mred64((uint64_t)a0 * b2 + (uint64_t)a1 * b1 + (uint64_t)a2 * b0);
    uint32_t c3 = mred64((uint64_t)a0 * b3 + (uint64_t)a1 * b2 + (uint64_t)a2 * b1 + (uint64_t)a3 * b0);
    return Ext{{madd(c0, mmul(r4, EXT_W_MONTY)), madd(c1, mmul(r5, EXT_W_MONTY)),
                madd(c2, mmul(r6, EXT_W_MONTY)), c3}};
}
ZK_HD Ext ext_sqr(const Ext& a) { return ext_mul(a, a); }
ZK_HD Ext ext_pow(Ext a, uint64_t e) {
    Ext r = ext_one();
    while (e) {
        if (e & 1) r = ext_mul(r, a);
        a = ext_sqr(a);
        e >>= 1;
    }
    return r;
}
// Frobenius x -> x^p on F[x]/(x^4-11): coefficient i scaled by z^i, z = 11^((p-1)/4)
ZK_HD Ext ext_frobenius(const Ext& a) {
    const uint32_t z = to_monty(1728404513u);  // DTH_ROOT
    uint32_t z2 = mmul(z, z), z3 = mmul(z2, z);
    return Ext{{a.c[0], mmul(a.c[1], z), mmul(a.c[2], z2), mmul(a.c[3], z3)}};
}
ZK_HD Ext ext_inv(const Ext& a) {
    Ext f1 = ext_frobenius(a), f2 = ext_frobenius(f1), f3 = ext_frobenius(f2);
    Ext t = ext_mul(ext_mul(f1, f2), f3);
    Ext n = ext_mul(t, a);  // norm, in the base field
    return ext_mul_base(t, minv(n.c[0]));
}
ZK_HD bool ext_eq(const Ext& a, const Ext& b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}

}  // namespace zk
