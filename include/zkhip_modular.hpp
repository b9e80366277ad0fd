// zkhip_modular.hpp -- modular multiplication, addition and subtraction over a 256-bit modulus (crates/circuits/chunk-circuit/openvm.toml:8-59
// lists `modular` with the secp256k1 / bn254 / bls12-381 field and scalar moduli: the arithmetic under the EVM guest's ecrecover and pairing
// precompiles).  OpenVM's chips (openvm-algebra-circuit ModularMulDiv over its mod-builder, un-vendored) state r = a b mod P as an integer
// identity on byte limbs with range-checked carries; this is the same statement in this repository's own columns, ONE multiplication
// per row:
//   a[32] b[32] q[32] r[32]   byte limbs (little-endian), each looked up pairwise in the 8-bit bitwise table
//   carry (cx, cy)[62]        limb position k of a b - q P - r sums to 256 c_k - c_{k-1} with c_k = cx_k + 256 cy_k - 2^14, (cx_k, cy_k) in the
//                             range-tuple table (cx < 256, cy < 2048): every quantity stays far below the field's characteristic, so the 63
//                             limb equations hold over the integers and a b = q P + r exactly (the last carry is zero)
//   marker[32], diff          r < P: the most significant limb where r and P differ is marked, P - r there is in 1..255
//   real, is_add, is_sub      rows beyond the records are zero; a real row multiplies unless one of the flags is set: then the limb
//                             identity is a + b = q P + r, resp. a - b + q P = r, in the same columns (OpenVM's ModularAddSub)
//   is_div, marker2[32], diff2
//                             a division x / y is the multiplication row (a, b, r) = (x / y, y, x): the same identity a b = q P + r read
//                             from the other side (OpenVM's ModularMulDiv does the same); the quotient x / y sits in the a columns, so
//                             a < P is enforced there as well (second marker set), and inside the VM the word indices of a and r swap.
//                             As in OpenVM the caller guarantees y != 0 (mod P): the executor refuses the call otherwise
//   is_eq, eq, inv            an equality test (OpenVM's ModularIsEqual) is a subtraction row with is_eq set on top: r = (a - b) mod P is
//                             canonical, eq = [r = 0] through the sum S of r's limbs (bytes: S = 0 only for r = 0): eq S = 0 and
//                             (is_eq - eq)(S inv - 1) = 0; inside the VM the result word is eq instead of r
// The modulus is a constant of the AIR (one chip per modulus, as OpenVM instantiates one per configured modulus).  a and b are bytes
// but not required to be reduced; r is the canonical residue.  Every constraint has degree <= 3.
// External parity: Python's integers (tests/golden/modular_kat.json: random and edge operands for the secp256k1 and bn254 field and
// scalar moduli) and the secp256k1 generator's curve equation.  Header-only; device generator: csrc/modular.hip.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

#include "zkhip_air.hpp"

namespace zkhip {
namespace modular {
using air::AirBuilder;
using air::Expr;
using air::Kind;

constexpr size_t LIMBS = 32, N_CARRY = 2 * LIMBS - 2;
constexpr size_t COL_A = 0, COL_B = 32, COL_Q = 64, COL_R = 96, COL_CX = 128, COL_CY = COL_CX + N_CARRY, COL_MARK = COL_CY + N_CARRY, COL_DIFF = COL_MARK + LIMBS,
                 COL_REAL = COL_DIFF + 1, COL_IS_ADD = COL_REAL + 1, COL_IS_SUB = COL_REAL + 2, COL_IS_DIV = COL_REAL + 3, COL_MARK2 = COL_IS_DIV + 1,
                 COL_DIFF2 = COL_MARK2 + LIMBS, COL_IS_EQ = COL_DIFF2 + 1, COL_EQ = COL_IS_EQ + 1, COL_INV = COL_IS_EQ + 2, WIDTH = COL_IS_EQ + 3;
static_assert(WIDTH == 325, "modular chip width");
enum Op : uint32_t { OP_MUL, OP_ADD, OP_SUB, OP_DIV, OP_IS_EQ, N_OPS };
constexpr int64_t CARRY_OFFSET = 1 << 14;
using Modulus = std::array<uint8_t, LIMBS>;   // little-endian bytes

// AirBuilder(WIDTH, 0); bitwise_bus: (x, y, 0, 0) byte pairs; tuple_bus: (cx, cy)
inline void modmul_air(AirBuilder& b, const Modulus& P, uint32_t bitwise_bus, uint32_t tuple_bus) {
    const Expr real = b.var(COL_REAL), zero = b.constant(0);
    const Expr is_add = b.var(COL_IS_ADD), is_sub = b.var(COL_IS_SUB), is_div = b.var(COL_IS_DIV), is_mul = real - is_add - is_sub, q_sign = real - is_sub * 2;
    // (is_mul covers the division rows: the same limb identity)
    for (const Expr& f : {real, is_add, is_sub, is_div, is_mul, is_mul - is_div}) b.assert_zero(f * (f - 1));
    {   // equality test on top of a subtraction row
        const Expr is_eq = b.var(COL_IS_EQ), eq = b.var(COL_EQ);
        Expr sum = b.var(COL_R);
        for (size_t i = 1; i < LIMBS; i++) sum = sum + b.var(COL_R + i);
        for (const Expr& f : {is_eq, eq, is_sub - is_eq, is_eq - eq}) b.assert_zero(f * (f - 1));
        b.assert_zero(eq * sum);
        b.assert_zero((is_eq - eq) * (sum * b.var(COL_INV) - 1));
    }
    auto carry = [&](size_t k) { return b.var(COL_CX + k) + b.var(COL_CY + k) * 256 - CARRY_OFFSET; };
    for (size_t k = 0; k <= N_CARRY; k++) {   // limb positions 0 .. 62 of  [a b | a + b | a - b]  -  (+ | + | -) q P  -  r
        Expr prod = b.constant(0), qp = b.constant(0);
        for (size_t i = 0; i < LIMBS; i++) {
            if (k < i || k - i >= LIMBS) continue;
            const size_t j = k - i;
            prod = prod + b.var(COL_A + i) * b.var(COL_B + j);
            if (P[j]) qp = qp + b.var(COL_Q + i) * (int64_t)P[j];
        }
        Expr s = is_mul * prod - q_sign * qp;
        if (k < LIMBS) s = s + is_add * (b.var(COL_A + k) + b.var(COL_B + k)) + is_sub * (b.var(COL_A + k) - b.var(COL_B + k)) - real * b.var(COL_R + k);
        Expr cs = b.constant(0);
        if (k > 0) cs = cs + carry(k - 1);
        if (k < N_CARRY) cs = cs - carry(k) * 256;   // the carry out of position 62 is zero
        b.assert_zero(s + real * cs);
    }
    // r < P
    Expr n_marked = b.constant(0), diff = b.constant(0);
    for (size_t i = 0; i < LIMBS; i++) {
        const Expr m = b.var(COL_MARK + i);
        b.assert_zero(m * (m - 1));
        n_marked = n_marked + m;
        diff = diff + m * (b.constant(P[i]) - b.var(COL_R + i));
    }
    b.assert_zero(n_marked - real);
    Expr above = b.constant(0);   // markers at limbs above i
    for (size_t ii = LIMBS; ii-- > 0;) {
        b.assert_zero((real - above - b.var(COL_MARK + ii)) * (b.var(COL_R + ii) - b.constant(P[ii])));   // equal below no marker yet
        above = above + b.var(COL_MARK + ii);
    }
    b.assert_zero(b.var(COL_DIFF) - diff);
    b.push_interaction(bitwise_bus, {b.var(COL_DIFF) - real, zero, zero, zero}, real, Kind::Send);   // P - r at the marked limb is 1..255
    // a < P on the division rows (the quotient x / y is canonical)
    {
        Expr n2 = b.constant(0), diff2 = b.constant(0), above2 = b.constant(0);
        for (size_t i = 0; i < LIMBS; i++) {
            const Expr m = b.var(COL_MARK2 + i);
            b.assert_zero(m * (m - 1));
            n2 = n2 + m;
            diff2 = diff2 + m * (b.constant(P[i]) - b.var(COL_A + i));
        }
        b.assert_zero(n2 - is_div);
        for (size_t ii = LIMBS; ii-- > 0;) {
            b.assert_zero((is_div - above2 - b.var(COL_MARK2 + ii)) * (b.var(COL_A + ii) - b.constant(P[ii])));
            above2 = above2 + b.var(COL_MARK2 + ii);
        }
        b.assert_zero(b.var(COL_DIFF2) - diff2);
        b.push_interaction(bitwise_bus, {b.var(COL_DIFF2) - is_div, zero, zero, zero}, is_div, Kind::Send);
    }
    for (size_t base : {COL_A, COL_B, COL_Q, COL_R})
        for (size_t i = 0; i < LIMBS; i += 2) b.push_interaction(bitwise_bus, {b.var(base + i), b.var(base + i + 1), zero, zero}, real, Kind::Send);
    for (size_t k = 0; k < N_CARRY; k++) b.push_interaction(tuple_bus, {b.var(COL_CX + k), b.var(COL_CY + k)}, real, Kind::Send);
}

// The chip inside the VM (include/zkhip_vm_circuit.hpp): the same AIR plus a timestamp column; the 24 words of a call -- a, b (read) and
// r (written) -- are received from the modular adapter on `word_bus` as (timestamp, word index 0..23, low half, high half, operation).
// AirBuilder(VM_WIDTH, 0)
constexpr size_t COL_TS = WIDTH, VM_WIDTH = WIDTH + 1;
inline void modmul_vm_air(AirBuilder& b, const Modulus& P, uint32_t bitwise_bus, uint32_t tuple_bus, uint32_t word_bus) {
    modmul_air(b, P, bitwise_bus, tuple_bus);
    const Expr ts = b.var(COL_TS), real = b.var(COL_REAL), is_div = b.var(COL_IS_DIV), is_eq = b.var(COL_IS_EQ), eq = b.var(COL_EQ);
    const Expr op = b.var(COL_IS_ADD) + b.var(COL_IS_SUB) * 2 + is_div * 3 + is_eq * 2;   // an equality test is a subtraction row: 2 + 2 = 4
    // words 0..7 the first operand, 8..15 the second, 16..23 the result: (a, b, r) -- for a division (r, b, a)
    auto half = [&](size_t base, size_t k, size_t h) { return b.var(base + 4 * k + 2 * h) + b.var(base + 4 * k + 2 * h + 1) * 256; };
    for (size_t k = 0; k < 8; k++) {
        const Expr a_lo = half(COL_A, k, 0), a_hi = half(COL_A, k, 1), r_lo = half(COL_R, k, 0), r_hi = half(COL_R, k, 1);
        const Expr sw_lo = is_div * (r_lo - a_lo), sw_hi = is_div * (r_hi - a_hi);
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)k), a_lo + sw_lo, a_hi + sw_hi, op}, real, Kind::Receive);
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)(8 + k)), half(COL_B, k, 0), half(COL_B, k, 1), op}, real, Kind::Receive);
        // the result: r -- a for a division, the equality bit for a test (is_div and is_eq exclude one another: is_div <= is_mul, is_eq <= is_sub)
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)(16 + k)), r_lo - sw_lo - is_eq * r_lo + (k == 0 ? eq : b.constant(0)), r_hi - sw_hi - is_eq * r_hi, op}, real,
                           Kind::Receive);
    }
}

// 256-bit helpers on little-endian 32-bit words (host; the tests' expected values come from Python, not from here)
struct U256 {
    uint32_t w[8];
};
inline Modulus modulus_bytes(const U256& p) {
    Modulus m;
    for (size_t i = 0; i < LIMBS; i++) m[i] = (uint8_t)(p.w[i / 4] >> (8 * (i % 4)));
    return m;
}
// (q, r) = divmod(a b, p) by binary long division: r < p; q fits 256 bits when a, b < p (returns false otherwise)
inline bool mulmod(const U256& a, const U256& b, const U256& p, U256* q, U256* r) {
    uint32_t prod[16] = {};
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.w[i] * b.w[j] + prod[i + j];
            prod[i + j] = (uint32_t)c, c >>= 32;
        }
        prod[i + 8] = (uint32_t)c;
    }
    uint32_t rem[9] = {}, quo[16] = {};
    for (int bit = 511; bit >= 0; bit--) {
        for (int k = 8; k > 0; k--) rem[k] = (rem[k] << 1) | (rem[k - 1] >> 31);
        rem[0] = (rem[0] << 1) | ((prod[bit / 32] >> (bit % 32)) & 1u);
        bool ge = rem[8] != 0;
        if (!ge) {
            ge = true;
            for (int k = 7; k >= 0; k--)
                if (rem[k] != p.w[k]) {
                    ge = rem[k] > p.w[k];
                    break;
                }
        }
        if (ge) {
            uint64_t br = 0;
            for (int k = 0; k < 9; k++) {
                const uint64_t d = (uint64_t)rem[k] - (k < 8 ? p.w[k] : 0u) - br;
                rem[k] = (uint32_t)d, br = (d >> 32) & 1u;
            }
            quo[bit / 32] |= 1u << (bit % 32);
        }
    }
    bool fits = true;
    for (int k = 8; k < 16; k++) fits = fits && quo[k] == 0;
    for (int k = 0; k < 8; k++) q->w[k] = quo[k], r->w[k] = rem[k];
    return fits;
}

// (q, r) with a + b = q P + r (OP_ADD) or a - b + q P = r (OP_SUB; needs |a - b| < P), r < P; false if there is no such pair
inline bool addsubmod(uint32_t op, const U256& a, const U256& b, const U256& p, U256* q, U256* r) {
    uint32_t num[9] = {};
    *q = U256{};
    auto ge = [&](const uint32_t* x) {   // x (9 words) >= p
        if (x[8]) return true;
        for (int k = 7; k >= 0; k--)
            if (x[k] != p.w[k]) return x[k] > p.w[k];
        return true;
    };
    auto sub_p = [&](uint32_t* x) {
        uint64_t br = 0;
        for (int k = 0; k < 9; k++) {
            const uint64_t d = (uint64_t)x[k] - (k < 8 ? p.w[k] : 0u) - br;
            x[k] = (uint32_t)d, br = (d >> 32) & 1u;
        }
    };
    if (op == OP_ADD) {
        uint64_t c = 0;
        for (int k = 0; k < 8; k++) c += (uint64_t)a.w[k] + b.w[k], num[k] = (uint32_t)c, c >>= 32;
        num[8] = (uint32_t)c;
        while (ge(num)) {
            sub_p(num);
            if (++q->w[0] == 0) return false;
        }
    } else {
        uint64_t br = 0;
        for (int k = 0; k < 8; k++) {
            const uint64_t d = (uint64_t)a.w[k] - b.w[k] - br;
            num[k] = (uint32_t)d, br = (d >> 32) & 1u;
        }
        if (br) {   // a < b: one P brings the difference back (if it does not, the operands are too far apart)
            uint64_t c = 0;
            for (int k = 0; k < 8; k++) c += (uint64_t)num[k] + p.w[k], num[k] = (uint32_t)c, c >>= 32;
            if (!c) return false;
            q->w[0] = 1;
        }
        if (ge(num)) return false;
    }
    for (int k = 0; k < 8; k++) r->w[k] = num[k];
    return true;
}

// x / y mod p for x < p and y invertible (binary extended Euclid on nine words for the inverse); false otherwise
inline bool divmod_p(const U256& x, const U256& y, const U256& p, U256* out) {
    auto less = [](const U256& u, const U256& v) {
        for (int k = 7; k >= 0; k--)
            if (u.w[k] != v.w[k]) return u.w[k] < v.w[k];
        return false;
    };
    if (!(p.w[0] & 1u) || !less(x, p)) return false;
    U256 yr, q;
    {   // y mod p
        U256 one{};
        one.w[0] = 1;
        if (!mulmod(y, one, p, &q, &yr)) return false;
    }
    bool zero = true;
    for (uint32_t w : yr.w) zero = zero && w == 0;
    if (zero) return false;
    struct W9 {
        uint32_t w[9];
    };
    auto from = [](const U256& v) {
        W9 r{};
        for (int k = 0; k < 8; k++) r.w[k] = v.w[k];
        return r;
    };
    auto is_one = [](const W9& v) {
        uint32_t o = v.w[0] ^ 1u;
        for (int k = 1; k < 9; k++) o |= v.w[k];
        return o == 0;
    };
    auto zero9 = [](const W9& v) {
        uint32_t o = 0;
        for (int k = 0; k < 9; k++) o |= v.w[k];
        return o == 0;
    };
    auto shr1 = [](W9& v) {
        for (int k = 0; k < 8; k++) v.w[k] = (v.w[k] >> 1) | (v.w[k + 1] << 31);
        v.w[8] >>= 1;
    };
    auto add = [](W9& v, const W9& o) {
        uint64_t c = 0;
        for (int k = 0; k < 9; k++) c += (uint64_t)v.w[k] + o.w[k], v.w[k] = (uint32_t)c, c >>= 32;
    };
    auto sub = [](W9& v, const W9& o) {
        uint64_t br = 0;
        for (int k = 0; k < 9; k++) {
            const uint64_t d = (uint64_t)v.w[k] - o.w[k] - br;
            v.w[k] = (uint32_t)d, br = (d >> 32) & 1u;
        }
    };
    auto ge = [](const W9& v, const W9& o) {
        for (int k = 8; k >= 0; k--)
            if (v.w[k] != o.w[k]) return v.w[k] > o.w[k];
        return true;
    };
    const W9 P9 = from(p);
    W9 u = from(yr), v = P9, x1 = from(x), x2{};   // invariants: u = x1' y, v = x2' y (mod p) scaled by x: x1 ends as x / y
    auto halve = [&](W9& t, W9& c) {
        while (!(t.w[0] & 1u)) {
            shr1(t);
            if (c.w[0] & 1u) add(c, P9);
            shr1(c);
        }
    };
    for (int guard = 0; guard < 2048 && !is_one(u) && !is_one(v); guard++) {
        if (zero9(u) || zero9(v)) return false;
        halve(u, x1), halve(v, x2);
        if (ge(u, v)) {
            sub(u, v);
            if (!ge(x1, x2)) add(x1, P9);
            sub(x1, x2);
        } else {
            sub(v, u);
            if (!ge(x2, x1)) add(x2, P9);
            sub(x2, x1);
        }
    }
    if (!is_one(u) && !is_one(v)) return false;
    const W9& r = is_one(u) ? x1 : x2;
    for (int k = 0; k < 8; k++) out->w[k] = r.w[k];
    return true;
}

}  // namespace modular
}  // namespace zkhip
