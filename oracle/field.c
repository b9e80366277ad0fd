/*
 * oracle/field.c -- BabyBear and its quartic extension, canonical form.
 * TEST INFRASTRUCTURE (see zk_oracle.h).  The extension arithmetic is pinned by the FRI fold triples of the
 * reference's stored proofs (tests/test_ref_vectors_cpu.py).
 *
 * Follows the published definitions of p3-baby-bear / p3-monty-31 / p3-field
 * 0.4.3 (Cargo.lock:5545,5605,5685): p = 2^31 - 2^27 + 1
 * (scripts/compress_bn254.py:10), multiplicative generator 31, two-adic
 * generator of order 2^27 = 0x1a427a41, extension F[x]/(x^4 - 11).
 * The reference stores Montgomery u32 (as_canonical_u32 at
 * crates/prover/src/prover/mod.rs:136-137); field values are
 * representation-independent so the oracle works on canonical integers.
 */
#include "zk_oracle.h"

uint32_t ora_add(uint32_t a, uint32_t b) {
    uint32_t s = a + b; /* < 2^32 since a,b < 2^31 */
    return s >= ORA_P ? s - ORA_P : s;
}
uint32_t ora_sub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + ORA_P - b; }
uint32_t ora_mul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % ORA_P); }

uint32_t ora_pow(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = ora_mul(r, a);
        a = ora_mul(a, a);
        e >>= 1;
    }
    return r;
}
uint32_t ora_inv(uint32_t a) { return ora_pow(a, ORA_P - 2); }

uint32_t ora_two_adic_generator(unsigned bits) {
    /* p3-baby-bear: TWO_ADIC_GENERATORS[27] = 0x1a427a41 (= 31^15); SURVEY.md A.1 */
    uint32_t g = 0x1a427a41u;
    for (unsigned i = bits; i < 27; i++) g = ora_mul(g, g);
    return g;
}

/* (a0 + a1 x + a2 x^2 + a3 x^3)(b0 + ...), x^4 = 11 (p3 BinomialExtensionField<_,4>, W = 11) */
void ora_ext_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]) {
    uint64_t c[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) c[i + j] = (c[i + j] + (uint64_t)a[i] * b[j]) % ORA_P;
    uint32_t r[4];
    for (int k = 0; k < 4; k++) {
        uint64_t v = c[k];
        if (k + 4 < 7) v += 11ull * c[k + 4];
        r[k] = (uint32_t)(v % ORA_P);
    }
    for (int k = 0; k < 4; k++) out[k] = r[k];
}

/* inverse by Fermat in the extension: a^(p^4 - 2).  Slow but obviously right;
 * p^4-2 does not fit u64, so use a^-1 = conj(a) / Norm(a) via Frobenius:
 * Norm(a) = a * a^p * a^(p^2) * a^(p^3) in F.  Frobenius on x^4=11: x -> x * 11^((p-1)/4). */
static void ext_frob(const uint32_t a[4], uint32_t out[4]) {
    uint32_t z = ora_pow(11, (ORA_P - 1) / 4); /* DTH_ROOT = 1728404513, SURVEY.md A.1 */
    uint32_t zp = 1;
    for (int i = 0; i < 4; i++) {
        out[i] = ora_mul(a[i], zp);
        zp = ora_mul(zp, z);
    }
}
void ora_ext_inv(const uint32_t a[4], uint32_t out[4]) {
    uint32_t f1[4], f2[4], f3[4], t[4], n[4];
    ext_frob(a, f1);
    ext_frob(f1, f2);
    ext_frob(f2, f3);
    ora_ext_mul(f1, f2, t);
    ora_ext_mul(t, f3, t); /* t = a^(p+p^2+p^3) */
    ora_ext_mul(t, a, n);  /* norm, lies in F: n[1..3] == 0 */
    uint32_t ni = ora_inv(n[0]);
    for (int i = 0; i < 4; i++) out[i] = ora_mul(t[i], ni);
}
