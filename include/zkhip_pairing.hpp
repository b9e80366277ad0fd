// zkhip_pairing.hpp -- the PAIRING extension's phantom hint: the final-exponentiation witness (SURVEY.md 8(f) f3;
// `[app_vm_config.pairing] supported_curves = ["Bn254"]`, crates/circuits/chunk-circuit/openvm.toml:35-36).  OpenVM's pairing extension
// (openvm-pairing, un-vendored) brings NO chip: its executor is a phantom sub-executor, PairingPhantom::HintFinalExp, that leaves a residue
// witness (c, u) in the hint stream; the guest -- whose Miller loop runs over the fp2 / modular intrinsics -- then checks
//     c^lambda = f u,     lambda = 6 x + 2 + p - p^2 + p^3,
// with one exponentiation instead of the final exponentiation (Novakovic, Eagen, "On Proving Pairings", 2024).  The circuit proves nothing
// about a hint; a wrong one only makes the guest's check fail.
//
// Here: phantom kind 2 (include/zkhip_vm.hpp).  Operand buffer: word 0 = the pairing curve (0 = Bn254), then f as OpenVM's
// SexticExtField<Fp2>: six Fp2 coefficients (a_i, b_i) of w^i, 8 little-endian words per base-field element (96 words).  Pushed: c, then
// u, in the same layout (192 words).  DEVIATION from OpenVM, stated: its sub-executor takes the points (P_i, Q_i) and runs the Miller loop
// itself; this one takes the Miller loop's OUTPUT f, which the guest has anyway.
//
// Curve 1 = Bls12_381 (`crates/circuits/batch-circuit/openvm.toml:25-26`), 12 words per base-field element (144 words in, 288 pushed):
// (c, s) with c^lambda = f s, lambda = p + |x|.  There (p^12 - 1) / r = 27 * ((|x| + 1) / 3) * C with C coprime to the other two factors
// and to lambda: the scaling factor s = f^(e - 1) (e = 1 mod C, e = 0 mod the rest) takes f to its component of order C, and
// c = (f s)^(1 / lambda mod C) -- two exponentiations, no search (gnark's / OpenVM's construction, by the Chinese remainder theorem).
//
// Arithmetic: Fp by Montgomery products on four (Bn254) or six (Bls12_381) 64-bit limbs; Fp12 = Fp[w] / (w^12 - 18 w^6 + 82) (Bn254: w^6 = 9 + u,
// u^2 = -1) or Fp[w] / (w^12 - 2 w^6 + 2) (Bls12_381: w^6 = 1 + u), twelve
// coefficients, schoolbook products.  The rule that makes (c, u) unique (tests/pairing_util.py states the same rule with Python integers
// and derives every constant from x): tau = w^((p^12 - 1) / 27) generates the 27-part; u = tau^j for the smallest j in {0, 1, 2} that
// makes y = f u a cube; c = y^A tau^k, A the exponent that takes y to the lambda-th root of its component of order coprime to 3 r, k the
// smallest exponent with tau^(k lambda) = the 27-part of y.  Host only.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>

namespace zkhip {
namespace pairing {
#include "zkhip_pairing_constants.inc"

// a base field: N limbs of 64 bits; the Fp12 above it: w^12 = A w^6 - B, w^6 = T + u
struct Curve {
    int n;                 // limbs (4 or 6)
    const uint64_t* p;     // the modulus (6 limbs, zero-padded)
    const uint64_t* r2;    // 2^(128 n) mod p
    uint64_t ninv;         // -p^-1 mod 2^64
    uint64_t a, b, t;
};
inline const Curve& bn254() {
    static const Curve c{4, BN254_P, BN254_R2, BN254_NINV, 18, 82, 9};
    return c;
}
inline const Curve& bls12_381() {
    static const Curve c{6, BLS_P, BLS_R2, BLS_NINV, 2, 2, 1};
    return c;
}

struct Fp {
    uint64_t l[6];   // Montgomery form, < p (limbs beyond the curve's n are zero)
};
inline bool fp_eq(const Fp& a, const Fp& b) { return memcmp(a.l, b.l, sizeof a.l) == 0; }
inline bool geq_p(const Curve& c, const uint64_t* t) {
    for (int k = c.n - 1; k >= 0; k--)
        if (t[k] != c.p[k]) return t[k] > c.p[k];
    return true;
}
inline void sub_p(const Curve& c, uint64_t* t) {
    unsigned __int128 br = 0;
    for (int k = 0; k < c.n; k++) {
        const unsigned __int128 d = (unsigned __int128)t[k] - c.p[k] - br;
        t[k] = (uint64_t)d, br = (d >> 64) & 1;
    }
}
inline Fp fp_add(const Curve& c, const Fp& a, const Fp& b) {
    Fp r{};
    unsigned __int128 cy = 0;
    for (int k = 0; k < c.n; k++) cy += (unsigned __int128)a.l[k] + b.l[k], r.l[k] = (uint64_t)cy, cy >>= 64;
    if (cy || geq_p(c, r.l)) sub_p(c, r.l);
    return r;
}
inline Fp fp_sub(const Curve& c, const Fp& a, const Fp& b) {
    Fp r{};
    unsigned __int128 br = 0;
    for (int k = 0; k < c.n; k++) {
        const unsigned __int128 d = (unsigned __int128)a.l[k] - b.l[k] - br;
        r.l[k] = (uint64_t)d, br = (d >> 64) & 1;
    }
    if (br) {
        unsigned __int128 cy = 0;
        for (int k = 0; k < c.n; k++) cy += (unsigned __int128)r.l[k] + c.p[k], r.l[k] = (uint64_t)cy, cy >>= 64;
    }
    return r;
}
inline Fp fp_mul(const Curve& c, const Fp& a, const Fp& b) {   // CIOS Montgomery product
    const int n = c.n;
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        unsigned __int128 cy = 0;
        for (int j = 0; j < n; j++) cy += (unsigned __int128)a.l[i] * b.l[j] + t[j], t[j] = (uint64_t)cy, cy >>= 64;
        cy += t[n], t[n] = (uint64_t)cy, t[n + 1] = (uint64_t)(cy >> 64);
        const uint64_t m = t[0] * c.ninv;
        cy = (unsigned __int128)m * c.p[0] + t[0], cy >>= 64;
        for (int j = 1; j < n; j++) cy += (unsigned __int128)m * c.p[j] + t[j], t[j - 1] = (uint64_t)cy, cy >>= 64;
        cy += t[n], t[n - 1] = (uint64_t)cy, t[n] = t[n + 1] + (uint64_t)(cy >> 64);
    }
    Fp r{};
    for (int k = 0; k < n; k++) r.l[k] = t[k];
    if (t[n] || geq_p(c, r.l)) sub_p(c, r.l);
    return r;
}
inline Fp fp_from_words(const Curve& c, const uint32_t* w, bool* reduced) {   // 2 n canonical little-endian words -> Montgomery
    Fp a{};
    for (int k = 0; k < c.n; k++) a.l[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
    if (reduced) *reduced = !geq_p(c, a.l);
    Fp r2{};
    memcpy(r2.l, c.r2, 8 * (size_t)c.n);
    return fp_mul(c, a, r2);
}
inline void fp_to_words(const Curve& c, const Fp& a, uint32_t* w) {
    Fp one{};
    one.l[0] = 1;
    const Fp v = fp_mul(c, a, one);
    for (int k = 0; k < c.n; k++) w[2 * k] = (uint32_t)v.l[k], w[2 * k + 1] = (uint32_t)(v.l[k] >> 32);
}
inline Fp fp_small(const Curve& c, uint64_t v) {
    uint32_t w[12] = {(uint32_t)v, (uint32_t)(v >> 32)};
    return fp_from_words(c, w, nullptr);
}

using Fp12 = std::array<Fp, 12>;   // coefficients of w^0 .. w^11
inline Fp12 f12_one(const Curve& c) {
    Fp12 r{};
    r[0] = fp_small(c, 1);
    return r;
}
inline bool f12_eq(const Fp12& a, const Fp12& b) {
    for (int k = 0; k < 12; k++)
        if (!fp_eq(a[k], b[k])) return false;
    return true;
}
inline Fp12 f12_mul(const Curve& c, const Fp12& a, const Fp12& b) {
    const Fp ca = fp_small(c, c.a), cb = fp_small(c, c.b);
    Fp t[23];
    memset(t, 0, sizeof t);
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) t[i + j] = fp_add(c, t[i + j], fp_mul(c, a[i], b[j]));
    for (int k = 22; k >= 12; k--) {   // w^12 = A w^6 - B
        t[k - 6] = fp_add(c, t[k - 6], fp_mul(c, ca, t[k]));
        t[k - 12] = fp_sub(c, t[k - 12], fp_mul(c, cb, t[k]));
    }
    Fp12 r;
    for (int k = 0; k < 12; k++) r[k] = t[k];
    return r;
}
template <size_t L>
inline Fp12 f12_pow(const Curve& c, const Fp12& a, const uint64_t (&e)[L]) {
    Fp12 r = f12_one(c);
    bool started = false;
    for (int k = (int)L - 1; k >= 0; k--)
        for (int bit = 63; bit >= 0; bit--) {
            if (started) r = f12_mul(c, r, r);
            if ((e[k] >> bit) & 1) r = started ? f12_mul(c, r, a) : a, started = true;
        }
    return r;
}
inline Fp12 f12_pow_small(const Curve& c, const Fp12& a, unsigned e) {
    Fp12 r = f12_one(c);
    for (unsigned k = 0; k < e; k++) r = f12_mul(c, r, a);
    return r;
}
// OpenVM's SexticExtField<Fp2> layout <-> flat coefficients: (a_k + b_k u) w^k with u = w^6 - T
inline bool f12_load(const Curve& c, const uint32_t* words, Fp12* f) {
    const Fp t = fp_small(c, c.t);
    const int ew = 2 * c.n;
    for (int k = 0; k < 6; k++) {
        bool ra = true, rb = true;
        const Fp a = fp_from_words(c, words + 2 * ew * k, &ra), b = fp_from_words(c, words + 2 * ew * k + ew, &rb);
        if (!ra || !rb) return false;
        (*f)[k] = fp_sub(c, a, fp_mul(c, t, b)), (*f)[k + 6] = b;
    }
    return true;
}
inline void f12_store(const Curve& c, const Fp12& v, uint32_t* dst) {
    const Fp t = fp_small(c, c.t);
    const int ew = 2 * c.n;
    for (int q = 0; q < 6; q++) {
        fp_to_words(c, fp_add(c, v[q], fp_mul(c, t, v[q + 6])), dst + 2 * ew * q);
        fp_to_words(c, v[q + 6], dst + 2 * ew * q + ew);
    }
}

// Bn254: f (96 words) -> (c, u) as 192 words; false with a reason for an f outside the subgroup or unreduced coefficients
inline bool final_exp_hint_bn254(const uint32_t f_words[96], uint32_t out[192], const char** why) {
    const Curve& cv = bn254();
    Fp12 f;
    if (!f12_load(cv, f_words, &f)) return *why = "a coefficient of f is not reduced", false;
    const Fp12 one = f12_one(cv);
    if (!f12_eq(f12_pow(cv, f, BN254_H), one)) return *why = "f does not lie in the subgroup of order (p^12 - 1) / r (not a Miller loop's output)", false;
    Fp12 w{};
    w[1] = fp_small(cv, 1);
    const Fp12 tau = f12_pow(cv, w, BN254_E_27);
    Fp12 u = one, y = f;
    int j = 0;
    for (; j < 3; j++) {
        if (f12_eq(f12_pow(cv, y, BN254_E_CUBE), one)) break;
        u = f12_mul(cv, u, tau), y = f12_mul(cv, f, u);
    }
    if (j == 3) return *why = "no cubic residue among f, f tau, f tau^2", false;
    const Fp12 c_u = f12_pow(cv, y, BN254_A_U), y_t = f12_pow(cv, y, BN254_P_T);
    Fp12 tk = one;
    int k = 0;
    for (; k < 27; k++) {
        if (f12_eq(f12_pow_small(cv, tk, BN254_LAMBDA_MOD_27), y_t)) break;
        tk = f12_mul(cv, tk, tau);
    }
    if (k == 27) return *why = "the 27-part has no lambda-th root", false;
    f12_store(cv, f12_mul(cv, c_u, tk), out), f12_store(cv, u, out + 96);
    return true;
}
// Bls12_381: f (144 words) -> (c, s) as 288 words
inline bool final_exp_hint_bls12_381(const uint32_t f_words[144], uint32_t out[288], const char** why) {
    const Curve& cv = bls12_381();
    Fp12 f;
    if (!f12_load(cv, f_words, &f)) return *why = "a coefficient of f is not reduced", false;
    if (!f12_eq(f12_pow(cv, f, BLS_H), f12_one(cv))) return *why = "f does not lie in the subgroup of order (p^12 - 1) / r (not a Miller loop's output)", false;
    f12_store(cv, f12_pow(cv, f, BLS_E_C), out), f12_store(cv, f12_pow(cv, f, BLS_E_S), out + 144);
    return true;
}

}  // namespace pairing
}  // namespace zkhip
