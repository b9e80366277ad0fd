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
// itself; this one takes the Miller loop's OUTPUT f, which the guest has anyway.  BLS12-381 (the batch circuit's curve) is refused: with
// lambda = p + |x| the cofactor gcd(lambda / r, (p^12 - 1) / r) has a 46-bit prime factor and the witness needs that paper's other
// construction -- not built.
//
// Arithmetic: Fp by Montgomery products on four 64-bit limbs; Fp12 = Fp[w] / (w^12 - 18 w^6 + 82) (w^6 = 9 + u, u^2 = -1), twelve
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

struct Fp {
    uint64_t l[4];   // Montgomery form, < p
};
inline bool fp_eq(const Fp& a, const Fp& b) { return memcmp(a.l, b.l, 32) == 0; }
inline bool geq_p(const uint64_t* t) {
    for (int k = 3; k >= 0; k--)
        if (t[k] != BN254_P[k]) return t[k] > BN254_P[k];
    return true;
}
inline Fp fp_add(const Fp& a, const Fp& b) {
    Fp r;
    unsigned __int128 c = 0;
    for (int k = 0; k < 4; k++) c += (unsigned __int128)a.l[k] + b.l[k], r.l[k] = (uint64_t)c, c >>= 64;
    if (c || geq_p(r.l)) {
        unsigned __int128 br = 0;
        for (int k = 0; k < 4; k++) {
            const unsigned __int128 d = (unsigned __int128)r.l[k] - BN254_P[k] - br;
            r.l[k] = (uint64_t)d, br = (d >> 64) & 1;
        }
    }
    return r;
}
inline Fp fp_sub(const Fp& a, const Fp& b) {
    Fp r;
    unsigned __int128 br = 0;
    for (int k = 0; k < 4; k++) {
        const unsigned __int128 d = (unsigned __int128)a.l[k] - b.l[k] - br;
        r.l[k] = (uint64_t)d, br = (d >> 64) & 1;
    }
    if (br) {
        unsigned __int128 c = 0;
        for (int k = 0; k < 4; k++) c += (unsigned __int128)r.l[k] + BN254_P[k], r.l[k] = (uint64_t)c, c >>= 64;
    }
    return r;
}
inline Fp fp_mul(const Fp& a, const Fp& b) {   // CIOS Montgomery product
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; j++) c += (unsigned __int128)a.l[i] * b.l[j] + t[j], t[j] = (uint64_t)c, c >>= 64;
        c += t[4], t[4] = (uint64_t)c, t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * BN254_NINV;
        c = (unsigned __int128)m * BN254_P[0] + t[0], c >>= 64;
        for (int j = 1; j < 4; j++) c += (unsigned __int128)m * BN254_P[j] + t[j], t[j - 1] = (uint64_t)c, c >>= 64;
        c += t[4], t[3] = (uint64_t)c, t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fp r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(r.l)) {
        unsigned __int128 br = 0;
        for (int k = 0; k < 4; k++) {
            const unsigned __int128 d = (unsigned __int128)r.l[k] - BN254_P[k] - br;
            r.l[k] = (uint64_t)d, br = (d >> 64) & 1;
        }
    }
    return r;
}
inline Fp fp_from_words(const uint32_t w[8], bool* reduced) {   // canonical little-endian words -> Montgomery
    Fp a;
    for (int k = 0; k < 4; k++) a.l[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
    if (reduced) *reduced = !geq_p(a.l);
    Fp r2;
    memcpy(r2.l, BN254_R2, 32);
    return fp_mul(a, r2);
}
inline void fp_to_words(const Fp& a, uint32_t w[8]) {
    const Fp one{{1, 0, 0, 0}};
    const Fp c = fp_mul(a, one);
    for (int k = 0; k < 4; k++) w[2 * k] = (uint32_t)c.l[k], w[2 * k + 1] = (uint32_t)(c.l[k] >> 32);
}
inline Fp fp_small(uint64_t v) {
    uint32_t w[8] = {(uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0};
    return fp_from_words(w, nullptr);
}

using Fp12 = std::array<Fp, 12>;   // coefficients of w^0 .. w^11
inline Fp12 f12_one() {
    Fp12 r{};
    r[0] = fp_small(1);
    return r;
}
inline bool f12_eq(const Fp12& a, const Fp12& b) {
    for (int k = 0; k < 12; k++)
        if (!fp_eq(a[k], b[k])) return false;
    return true;
}
inline Fp12 f12_mul(const Fp12& a, const Fp12& b) {
    static const Fp c18 = fp_small(18), c82 = fp_small(82);
    Fp t[23];
    memset(t, 0, sizeof t);
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) t[i + j] = fp_add(t[i + j], fp_mul(a[i], b[j]));
    for (int k = 22; k >= 12; k--) {   // w^12 = 18 w^6 - 82
        t[k - 6] = fp_add(t[k - 6], fp_mul(c18, t[k]));
        t[k - 12] = fp_sub(t[k - 12], fp_mul(c82, t[k]));
    }
    Fp12 r;
    for (int k = 0; k < 12; k++) r[k] = t[k];
    return r;
}
template <size_t L>
inline Fp12 f12_pow(const Fp12& a, const uint64_t (&e)[L]) {
    Fp12 r = f12_one();
    bool started = false;
    for (int k = (int)L - 1; k >= 0; k--)
        for (int bit = 63; bit >= 0; bit--) {
            if (started) r = f12_mul(r, r);
            if ((e[k] >> bit) & 1) r = started ? f12_mul(r, a) : a, started = true;
        }
    return r;
}
inline Fp12 f12_pow_small(const Fp12& a, unsigned e) {
    Fp12 r = f12_one();
    for (unsigned k = 0; k < e; k++) r = f12_mul(r, a);
    return r;
}

// f (96 words: OpenVM's sextic layout) -> (c, u) as 192 words; false with a reason for an f outside the subgroup or unreduced coefficients
inline bool final_exp_hint_bn254(const uint32_t f_words[96], uint32_t out[192], const char** why) {
    const Fp nine = fp_small(9);
    Fp12 f;
    for (int k = 0; k < 6; k++) {
        bool ra = true, rb = true;
        const Fp a = fp_from_words(f_words + 16 * k, &ra), b = fp_from_words(f_words + 16 * k + 8, &rb);
        if (!ra || !rb) return *why = "a coefficient of f is not reduced", false;
        f[k] = fp_sub(a, fp_mul(nine, b)), f[k + 6] = b;   // (a + b u) w^k with u = w^6 - 9
    }
    const Fp12 one = f12_one();
    if (!f12_eq(f12_pow(f, BN254_H), one)) return *why = "f does not lie in the subgroup of order (p^12 - 1) / r (not a Miller loop's output)", false;
    Fp12 w{};
    w[1] = fp_small(1);
    const Fp12 tau = f12_pow(w, BN254_E_27);
    Fp12 u = one, y = f;
    int j = 0;
    for (; j < 3; j++) {
        if (f12_eq(f12_pow(y, BN254_E_CUBE), one)) break;
        u = f12_mul(u, tau), y = f12_mul(f, u);
    }
    if (j == 3) return *why = "no cubic residue among f, f tau, f tau^2", false;
    const Fp12 c_u = f12_pow(y, BN254_A_U), y_t = f12_pow(y, BN254_P_T);
    Fp12 tk = one, c{};
    int k = 0;
    for (; k < 27; k++) {
        if (f12_eq(f12_pow_small(tk, BN254_LAMBDA_MOD_27), y_t)) break;
        tk = f12_mul(tk, tau);
    }
    if (k == 27) return *why = "the 27-part has no lambda-th root", false;
    c = f12_mul(c_u, tk);
    auto put = [&](const Fp12& v, uint32_t* dst) {
        for (int q = 0; q < 6; q++) {
            fp_to_words(fp_add(v[q], fp_mul(nine, v[q + 6])), dst + 16 * q);
            fp_to_words(v[q + 6], dst + 16 * q + 8);
        }
    };
    put(c, out), put(u, out + 96);
    return true;
}

}  // namespace pairing
}  // namespace zkhip
