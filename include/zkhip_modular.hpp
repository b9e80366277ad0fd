// zkhip_modular.hpp -- modular multiplication, addition and subtraction over a modulus of up to 256 or up to 384 bits (crates/circuits/chunk-circuit/openvm.toml:8-59
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
// LIMBS: everything above is written for L = 32 byte limbs; a modulus above 2^256 (the BLS12-381 base field of the reference's BATCH
// circuit: crates/circuits/batch-circuit/openvm.toml:18-36) gets L = 48 limbs -- the same columns with 48 in place of 32 and 94 carries
// (`Cols`; OpenVM likewise instantiates its chips with 32 or 48 limbs per configured modulus).  The carry bounds hold for both: a limb
// position sums at most 48 byte products (< 2^21.6) next to the quotient's, the carries stay inside (-2^14, 2^19 - 2^14).
// The modulus is a constant of the AIR (one chip per modulus, as OpenVM instantiates one per configured modulus).  a and b are bytes
// but not required to be reduced; r is the canonical residue.  Every constraint has degree <= 3.
// External parity: Python's integers (tests/golden/modular_kat.json: random and edge operands for the secp256k1 and bn254 field and
// scalar moduli) and the secp256k1 generator's curve equation.  Header-only; device generator: csrc/modular.hip.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

#include "zkhip_air.hpp"

namespace zkhip {
namespace modular {
using air::AirBuilder;
using air::Expr;
using air::Kind;

constexpr size_t MAX_LIMBS = 48, MAX_WORDS = 12;
// column layout for L limbs
struct Cols {
    size_t L, N_CARRY, A, B, Q, R, CX, CY, MARK, DIFF, REAL, IS_ADD, IS_SUB, IS_DIV, MARK2, DIFF2, IS_EQ, EQ, INV, WIDTH, TS, VM_WIDTH;
    constexpr explicit Cols(size_t l)
        : L(l), N_CARRY(2 * l - 2), A(0), B(l), Q(2 * l), R(3 * l), CX(4 * l), CY(CX + N_CARRY), MARK(CY + N_CARRY), DIFF(MARK + l), REAL(DIFF + 1), IS_ADD(REAL + 1),
          IS_SUB(REAL + 2), IS_DIV(REAL + 3), MARK2(IS_DIV + 1), DIFF2(MARK2 + l), IS_EQ(DIFF2 + 1), EQ(IS_EQ + 1), INV(IS_EQ + 2), WIDTH(IS_EQ + 3), TS(WIDTH),
          VM_WIDTH(WIDTH + 1) {}
};
// (the 32-limb layout under its round-3 names)
constexpr Cols C32(32);
constexpr size_t LIMBS = 32, N_CARRY = C32.N_CARRY;
constexpr size_t COL_A = C32.A, COL_B = C32.B, COL_Q = C32.Q, COL_R = C32.R, COL_CX = C32.CX, COL_CY = C32.CY, COL_MARK = C32.MARK, COL_DIFF = C32.DIFF,
                 COL_REAL = C32.REAL, COL_IS_ADD = C32.IS_ADD, COL_IS_SUB = C32.IS_SUB, COL_IS_DIV = C32.IS_DIV, COL_MARK2 = C32.MARK2, COL_DIFF2 = C32.DIFF2,
                 COL_IS_EQ = C32.IS_EQ, COL_EQ = C32.EQ, COL_INV = C32.INV, WIDTH = C32.WIDTH;
static_assert(WIDTH == 325, "modular chip width");
static_assert(Cols(48).WIDTH == 485, "modular chip width, 48 limbs");
enum Op : uint32_t { OP_MUL, OP_ADD, OP_SUB, OP_DIV, OP_IS_EQ, N_OPS };
constexpr int64_t CARRY_OFFSET = 1 << 14;
// little-endian bytes of a modulus or coefficient, `limbs` (32 or 48) of them
struct Modulus {
    std::array<uint8_t, MAX_LIMBS> b{};
    uint32_t limbs = 32;
    uint8_t operator[](size_t i) const { return b[i]; }
    uint8_t& operator[](size_t i) { return b[i]; }
    uint8_t* data() { return b.data(); }
    const uint8_t* data() const { return b.data(); }
    size_t size() const { return limbs; }
    const uint8_t* begin() const { return b.data(); }
    const uint8_t* end() const { return b.data() + limbs; }
    bool operator==(const Modulus& o) const { return limbs == o.limbs && b == o.b; }
    bool operator!=(const Modulus& o) const { return !(*this == o); }
    bool operator<(const Modulus& o) const { return limbs != o.limbs ? limbs < o.limbs : b < o.b; }
};

// AirBuilder(WIDTH, 0); bitwise_bus: (x, y, 0, 0) byte pairs; tuple_bus: (cx, cy)
inline void modmul_air(AirBuilder& b, const Modulus& P, uint32_t bitwise_bus, uint32_t tuple_bus) {
    const Cols C(P.limbs);
    const size_t LIMBS = C.L, N_CARRY = C.N_CARRY;
    const size_t COL_A = C.A, COL_B = C.B, COL_Q = C.Q, COL_R = C.R, COL_CX = C.CX, COL_CY = C.CY, COL_MARK = C.MARK, COL_DIFF = C.DIFF, COL_REAL = C.REAL,
                 COL_IS_ADD = C.IS_ADD, COL_IS_SUB = C.IS_SUB, COL_IS_DIV = C.IS_DIV, COL_MARK2 = C.MARK2, COL_DIFF2 = C.DIFF2, COL_IS_EQ = C.IS_EQ, COL_EQ = C.EQ,
                 COL_INV = C.INV;
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
    for (size_t k = 0; k <= N_CARRY; k++) {   // limb positions 0 .. 2 L - 2 of  [a b | a + b | a - b]  -  (+ | + | -) q P  -  r
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
        if (k < N_CARRY) cs = cs - carry(k) * 256;   // the carry out of the last position is zero
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

// The chip inside the VM (include/zkhip_vm_circuit.hpp): the same AIR plus a timestamp column; the 3 L / 4 words of a call -- a, b (read)
// and r (written) -- are received from the modular adapter on `word_bus` as (timestamp, word index, low half, high half, operation).
// AirBuilder(Cols(P.limbs).VM_WIDTH, 0)
constexpr size_t COL_TS = WIDTH, VM_WIDTH = WIDTH + 1;   // (32 limbs)
inline void modmul_vm_air(AirBuilder& b, const Modulus& P, uint32_t bitwise_bus, uint32_t tuple_bus, uint32_t word_bus) {
    modmul_air(b, P, bitwise_bus, tuple_bus);
    const Cols C(P.limbs);
    const size_t NW = C.L / 4;
    const Expr ts = b.var(C.TS), real = b.var(C.REAL), is_div = b.var(C.IS_DIV), is_eq = b.var(C.IS_EQ), eq = b.var(C.EQ);
    const Expr op = b.var(C.IS_ADD) + b.var(C.IS_SUB) * 2 + is_div * 3 + is_eq * 2;   // an equality test is a subtraction row: 2 + 2 = 4
    // words 0..NW-1 the first operand, the next NW the second, the last NW the result: (a, b, r) -- for a division (r, b, a)
    auto half = [&](size_t base, size_t k, size_t h) { return b.var(base + 4 * k + 2 * h) + b.var(base + 4 * k + 2 * h + 1) * 256; };
    for (size_t k = 0; k < NW; k++) {
        const Expr a_lo = half(C.A, k, 0), a_hi = half(C.A, k, 1), r_lo = half(C.R, k, 0), r_hi = half(C.R, k, 1);
        const Expr sw_lo = is_div * (r_lo - a_lo), sw_hi = is_div * (r_hi - a_hi);
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)k), a_lo + sw_lo, a_hi + sw_hi, op}, real, Kind::Receive);
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)(NW + k)), half(C.B, k, 0), half(C.B, k, 1), op}, real, Kind::Receive);
        // the result: r -- a for a division, the equality bit for a test (is_div and is_eq exclude one another: is_div <= is_mul, is_eq <= is_sub)
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)(2 * NW + k)), r_lo - sw_lo - is_eq * r_lo + (k == 0 ? eq : b.constant(0)), r_hi - sw_hi - is_eq * r_hi, op}, real,
                           Kind::Receive);
    }
}

// big-integer helpers on little-endian 32-bit words, 12 words = 384 bits of capacity (host; the tests' expected values come from Python,
// not from here).  A modulus below 2^256 has 8 words / 32 limbs, one above 12 words / 48 limbs.
struct UInt {
    uint32_t w[MAX_WORDS];
};
using U256 = UInt;   // (the round-3 name)
inline size_t words_of(const UInt& p) { return (p.w[8] | p.w[9] | p.w[10] | p.w[11]) ? 12 : 8; }
inline UInt load_words(const void* src, size_t n_words) {
    UInt v{};
    memcpy(v.w, src, 4 * n_words);
    return v;
}
inline Modulus modulus_bytes(const UInt& p, size_t limbs = 0) {
    Modulus m;
    m.limbs = (uint32_t)(limbs ? limbs : 4 * words_of(p));
    for (size_t i = 0; i < m.limbs; i++) m[i] = (uint8_t)(p.w[i / 4] >> (8 * (i % 4)));
    return m;
}
// (q, r) = divmod(a b, p) by binary long division: r < p; q fits the modulus's width when a, b < p (returns false otherwise)
inline bool mulmod(const UInt& a, const UInt& b, const UInt& p, UInt* q, UInt* r) {
    constexpr int N = MAX_WORDS;
    uint32_t prod[2 * N] = {};
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        for (int j = 0; j < N; j++) {
            c += (uint64_t)a.w[i] * b.w[j] + prod[i + j];
            prod[i + j] = (uint32_t)c, c >>= 32;
        }
        prod[i + N] = (uint32_t)c;
    }
    uint32_t rem[N + 1] = {}, quo[2 * N] = {};
    for (int bit = 64 * N - 1; bit >= 0; bit--) {
        for (int k = N; k > 0; k--) rem[k] = (rem[k] << 1) | (rem[k - 1] >> 31);
        rem[0] = (rem[0] << 1) | ((prod[bit / 32] >> (bit % 32)) & 1u);
        bool ge = rem[N] != 0;
        if (!ge) {
            ge = true;
            for (int k = N - 1; k >= 0; k--)
                if (rem[k] != p.w[k]) {
                    ge = rem[k] > p.w[k];
                    break;
                }
        }
        if (ge) {
            uint64_t br = 0;
            for (int k = 0; k <= N; k++) {
                const uint64_t d = (uint64_t)rem[k] - (k < N ? p.w[k] : 0u) - br;
                rem[k] = (uint32_t)d, br = (d >> 32) & 1u;
            }
            quo[bit / 32] |= 1u << (bit % 32);
        }
    }
    bool fits = true;
    for (size_t k = words_of(p); k < 2 * (size_t)N; k++) fits = fits && quo[k] == 0;
    for (int k = 0; k < N; k++) q->w[k] = quo[k], r->w[k] = rem[k];
    return fits;
}

// (q, r) with a + b = q P + r (OP_ADD) or a - b + q P = r (OP_SUB; needs |a - b| < P), r < P; false if there is no such pair
inline bool addsubmod(uint32_t op, const UInt& a, const UInt& b, const UInt& p, UInt* q, UInt* r) {
    constexpr int N = MAX_WORDS;
    uint32_t num[N + 1] = {};
    *q = UInt{};
    auto ge = [&](const uint32_t* x) {   // x (N + 1 words) >= p
        if (x[N]) return true;
        for (int k = N - 1; k >= 0; k--)
            if (x[k] != p.w[k]) return x[k] > p.w[k];
        return true;
    };
    auto sub_p = [&](uint32_t* x) {
        uint64_t br = 0;
        for (int k = 0; k <= N; k++) {
            const uint64_t d = (uint64_t)x[k] - (k < N ? p.w[k] : 0u) - br;
            x[k] = (uint32_t)d, br = (d >> 32) & 1u;
        }
    };
    if (op == OP_ADD) {
        uint64_t c = 0;
        for (int k = 0; k < N; k++) c += (uint64_t)a.w[k] + b.w[k], num[k] = (uint32_t)c, c >>= 32;
        num[N] = (uint32_t)c;
        while (ge(num)) {
            sub_p(num);
            if (++q->w[0] == 0) return false;
        }
    } else {
        uint64_t br = 0;
        for (int k = 0; k < N; k++) {
            const uint64_t d = (uint64_t)a.w[k] - b.w[k] - br;
            num[k] = (uint32_t)d, br = (d >> 32) & 1u;
        }
        if (br) {   // a < b: one P brings the difference back (if it does not, the operands are too far apart)
            uint64_t c = 0;
            for (int k = 0; k < N; k++) c += (uint64_t)num[k] + p.w[k], num[k] = (uint32_t)c, c >>= 32;
            if (!c) return false;
            q->w[0] = 1;
        }
        if (ge(num)) return false;
    }
    for (int k = 0; k < N; k++) r->w[k] = num[k];
    return true;
}

// x / y mod p for x < p and y invertible (binary extended Euclid on N + 1 words for the inverse); false otherwise
inline bool divmod_p(const UInt& x, const UInt& y, const UInt& p, UInt* out) {
    constexpr int N = MAX_WORDS;
    auto less = [](const UInt& u, const UInt& v) {
        for (int k = N - 1; k >= 0; k--)
            if (u.w[k] != v.w[k]) return u.w[k] < v.w[k];
        return false;
    };
    if (!(p.w[0] & 1u) || !less(x, p)) return false;
    UInt yr, q;
    {   // y mod p
        UInt one{};
        one.w[0] = 1;
        if (!mulmod(y, one, p, &q, &yr)) return false;
    }
    bool zero = true;
    for (uint32_t w : yr.w) zero = zero && w == 0;
    if (zero) return false;
    struct W {
        uint32_t w[N + 1];
    };
    auto from = [](const UInt& v) {
        W r{};
        for (int k = 0; k < N; k++) r.w[k] = v.w[k];
        return r;
    };
    auto is_one = [](const W& v) {
        uint32_t o = v.w[0] ^ 1u;
        for (int k = 1; k <= N; k++) o |= v.w[k];
        return o == 0;
    };
    auto zero_w = [](const W& v) {
        uint32_t o = 0;
        for (int k = 0; k <= N; k++) o |= v.w[k];
        return o == 0;
    };
    auto shr1 = [](W& v) {
        for (int k = 0; k < N; k++) v.w[k] = (v.w[k] >> 1) | (v.w[k + 1] << 31);
        v.w[N] >>= 1;
    };
    auto add = [](W& v, const W& o) {
        uint64_t c = 0;
        for (int k = 0; k <= N; k++) c += (uint64_t)v.w[k] + o.w[k], v.w[k] = (uint32_t)c, c >>= 32;
    };
    auto sub = [](W& v, const W& o) {
        uint64_t br = 0;
        for (int k = 0; k <= N; k++) {
            const uint64_t d = (uint64_t)v.w[k] - o.w[k] - br;
            v.w[k] = (uint32_t)d, br = (d >> 32) & 1u;
        }
    };
    auto ge = [](const W& v, const W& o) {
        for (int k = N; k >= 0; k--)
            if (v.w[k] != o.w[k]) return v.w[k] > o.w[k];
        return true;
    };
    const W PW = from(p);
    W u = from(yr), v = PW, x1 = from(x), x2{};   // invariants: u = x1' y, v = x2' y (mod p) scaled by x: x1 ends as x / y
    auto halve = [&](W& t, W& c) {
        while (!(t.w[0] & 1u)) {
            shr1(t);
            if (c.w[0] & 1u) add(c, PW);
            shr1(c);
        }
    };
    for (int guard = 0; guard < 4096 && !is_one(u) && !is_one(v); guard++) {
        if (zero_w(u) || zero_w(v)) return false;
        halve(u, x1), halve(v, x2);
        if (ge(u, v)) {
            sub(u, v);
            if (!ge(x1, x2)) add(x1, PW);
            sub(x1, x2);
        } else {
            sub(v, u);
            if (!ge(x2, x1)) add(x2, PW);
            sub(x2, x1);
        }
    }
    if (!is_one(u) && !is_one(v)) return false;
    const W& r = is_one(u) ? x1 : x2;
    *out = UInt{};
    for (int k = 0; k < N; k++) out->w[k] = r.w[k];
    return true;
}

}  // namespace modular
}  // namespace zkhip
