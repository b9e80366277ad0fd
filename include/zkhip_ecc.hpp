// zkhip_ecc.hpp -- short-Weierstrass point addition and doubling over a 256-bit prime field (SURVEY.md 8(f) f3;
// crates/circuits/chunk-circuit/openvm.toml:38-59 lists `[[app_vm_config.ecc.supported_curves]]` secp256k1, P-256 and bn254 G1: the
// arithmetic under the EVM guest's ecrecover / p256verify / bn254 add-mul precompiles).  OpenVM's chips (openvm-ecc-circuit EcAddNe /
// EcDouble over its mod-builder, un-vendored) state the chord-and-tangent formulas as integer identities on byte limbs with
// range-checked carries; this is the same statement in this repository's own columns, ONE point operation per row:
//   x1 y1 x2 y2 [32 each]   the operands' coordinates, byte limbs (little-endian); a doubling ignores (x2, y2)
//   l x3 y3 [32 each]       the slope and the result
//   q1 q2 q3 [33 each] + sign bit each
//                           signed quotients of the three identities
//                             add:     l (x2 - x1) - (y2 - y1) = q1 P          double:  2 l y1 - 3 x1^2 - a = q1 P
//                             l^2 - x1 - x2' - x3 = q2 P                       (x2' = x2, resp. x1)
//                             l (x1 - x3) - y1 - y3 = q3 P
//   carry (cx, cy)[3][63]   limb position k of an identity sums to 256 c_k - c_{k-1} with c_k = cx_k + 256 cy_k - 2^18, (cx_k, cy_k) in the
//                           range-tuple table (cx < 256, cy < 2048).  Every limb is a looked-up byte and every carry is bounded, so each limb
//                           equation is below 2^27 in absolute value and holds over the integers; the 64 of them telescope to the identity
//   marker[2][32], diff[2]  x3 < P and y3 < P (the most significant limb that differs from P's is marked, P - limb there is in 1..255):
//                           the result written back to memory is the canonical residue
//   real, is_double         rows beyond the records are zero
// The modulus P and the coefficient a are constants of the AIR (one chip per configured curve, as OpenVM instantiates them).  As for
// OpenVM's EcAddNe the caller guarantees x1 != x2 (mod P) for an addition and y1 != 0 for a doubling: the slope is then the unique
// solution of the first identity; the executor (include/zkhip_vm.hpp) refuses a call that breaks this.  Every constraint has degree <= 3.
// External parity: Python's integers (tests/golden/ecc_kat.json: multiples of the three curves' standard generators -- among them the
// published 2G and 3G of secp256k1 -- added and doubled with the textbook formulas).  Header-only; device generator: csrc/ecc.hip.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

#include "zkhip_air.hpp"
#include "zkhip_modular.hpp"

namespace zkhip {
namespace ecc {
using air::AirBuilder;
using air::Expr;
using air::Kind;
using modular::Modulus;
using modular::U256;

constexpr size_t N_EQ = 3;
// column layout for L limbs (32: a prime below 2^256; 48: below 2^384 -- BLS12-381 G1 of the reference's batch circuit)
struct Cols {
    size_t L, Q_LIMBS, N_POS, N_CARRY, X1, Y1, X2, Y2, LAM, X3, Y3, Q, QS, CX, CY, MARK, DIFF, REAL, IS_DOUBLE, WIDTH, TS, VM_WIDTH, CALL_WORDS, RECORD_WORDS;
    constexpr explicit Cols(size_t l)
        : L(l), Q_LIMBS(l + 1), N_POS(2 * l), N_CARRY(2 * l - 1), X1(0), Y1(l), X2(2 * l), Y2(3 * l), LAM(4 * l), X3(5 * l), Y3(6 * l), Q(7 * l), QS(Q + N_EQ * Q_LIMBS),
          CX(QS + N_EQ), CY(CX + N_EQ * N_CARRY), MARK(CY + N_EQ * N_CARRY), DIFF(MARK + 2 * l), REAL(DIFF + 2), IS_DOUBLE(REAL + 1), WIDTH(REAL + 2), TS(WIDTH),
          VM_WIDTH(WIDTH + 1), CALL_WORDS(6 * (l / 4)), RECORD_WORDS(1 + 5 * (l / 4)) {}
};
constexpr Cols C32(32);
constexpr size_t LIMBS = 32, Q_LIMBS = C32.Q_LIMBS, N_POS = C32.N_POS, N_CARRY = C32.N_CARRY;
constexpr size_t COL_X1 = C32.X1, COL_Y1 = C32.Y1, COL_X2 = C32.X2, COL_Y2 = C32.Y2, COL_L = C32.LAM, COL_X3 = C32.X3, COL_Y3 = C32.Y3, COL_Q = C32.Q, COL_QS = C32.QS,
                 COL_CX = C32.CX, COL_CY = C32.CY, COL_MARK = C32.MARK, COL_DIFF = C32.DIFF, COL_REAL = C32.REAL, COL_IS_DOUBLE = C32.IS_DOUBLE, WIDTH = C32.WIDTH;
static_assert(WIDTH == 772, "ecc chip width");
static_assert(Cols(48).WIDTH == 1156, "ecc chip width, 48 limbs");
enum Op : uint32_t { OP_ADD_NE, OP_DOUBLE, N_OPS };
constexpr int64_t CARRY_OFFSET = 1 << 18;
constexpr size_t CALL_WORDS = C32.CALL_WORDS;     // x1 y1 | x2 y2 (read) | x3 y3 (written), L / 4 32-bit words each
constexpr size_t RECORD_WORDS = C32.RECORD_WORDS; // op | x1 y1 x2 y2 | slope

// AirBuilder(WIDTH, 0); bitwise_bus: (x, y, 0, 0) byte pairs; tuple_bus: (cx, cy)
inline void ec_air(AirBuilder& b, const Modulus& P, const Modulus& A, uint32_t bitwise_bus, uint32_t tuple_bus) {
    const Cols C(P.limbs);
    const size_t LIMBS = C.L, Q_LIMBS = C.Q_LIMBS, N_POS = C.N_POS, N_CARRY = C.N_CARRY;
    const size_t COL_X1 = C.X1, COL_Y1 = C.Y1, COL_X2 = C.X2, COL_Y2 = C.Y2, COL_L = C.LAM, COL_X3 = C.X3, COL_Y3 = C.Y3, COL_Q = C.Q, COL_QS = C.QS, COL_CX = C.CX,
                 COL_CY = C.CY, COL_MARK = C.MARK, COL_DIFF = C.DIFF, COL_REAL = C.REAL, COL_IS_DOUBLE = C.IS_DOUBLE;
    const Expr real = b.var(COL_REAL), is_dbl = b.var(COL_IS_DOUBLE), is_add = real - is_dbl, zero = b.constant(0);
    for (const Expr& f : {real, is_dbl, is_add}) b.assert_zero(f * (f - 1));
    auto limb = [&](size_t base, size_t i) { return b.var(base + i); };
    auto carry = [&](size_t e, size_t k) { return b.var(COL_CX + e * N_CARRY + k) + b.var(COL_CY + e * N_CARRY + k) * 256 - CARRY_OFFSET; };
    Expr q_sign[N_EQ];
    for (size_t e = 0; e < N_EQ; e++) {
        const Expr qs = b.var(COL_QS + e);
        q_sign[e] = real - qs * 2;
        b.assert_zero(qs * (qs - real));   // a bit, and zero off the real rows
    }
    // (position by position, the three identities side by side, and the quotient limbs grouped by the modulus byte they meet: the
    // builder shares equal subexpressions, and a product shared between far-apart constraints would stay live in between --
    // csrc/air_compile.hpp keeps at most 96 intermediates, 64 without raising the kernel's LDS limit)
    for (size_t k = 0; k < N_POS; k++)
        for (size_t e : {0, 2, 1}) {   // (identities 1 and 3 share the products l_i x1_j)
            Expr s = b.constant(0), qp = b.constant(0);
            for (unsigned v = 1; v < 256; v++) {
                std::vector<size_t> idx;
                for (size_t i = 0; i < Q_LIMBS; i++)
                    if (k >= i && k - i < LIMBS && P[k - i] == v) idx.push_back(i);
                if (idx.empty()) continue;
                // (the sum starts with its last and its first limb: that pair is different at every position, so no partial sum is
                // shared with -- and kept live until -- a position far away)
                Expr group = b.var(COL_Q + e * Q_LIMBS + idx.back());
                for (size_t t = 0; t + 1 < idx.size(); t++) group = group + b.var(COL_Q + e * Q_LIMBS + idx[t]);
                qp = qp + group * (int64_t)v;
            }
            Expr p_add = b.constant(0), p_dbl = b.constant(0), p_any = b.constant(0);
            for (size_t i = 0; i < LIMBS; i++) {
                if (k < i || k - i >= LIMBS) continue;
                const size_t j = k - i;
                const Expr l = limb(COL_L, i);
                if (e == 0) {
                    p_add = p_add + l * limb(COL_X2, j) - l * limb(COL_X1, j);
                    p_dbl = p_dbl + l * limb(COL_Y1, j) * 2 - limb(COL_X1, i) * limb(COL_X1, j) * 3;
                } else if (e == 1) {
                    p_any = p_any + l * limb(COL_L, j);
                } else {
                    p_any = p_any + l * limb(COL_X1, j) - l * limb(COL_X3, j);
                }
            }
            if (e == 0) {
                s = is_add * p_add + is_dbl * p_dbl;
                if (k < LIMBS) s = s - is_add * (limb(COL_Y2, k) - limb(COL_Y1, k)) - is_dbl * (int64_t)A[k];
            } else if (e == 1) {
                s = p_any;
                if (k < LIMBS) s = s - limb(COL_X1, k) - is_add * limb(COL_X2, k) - is_dbl * limb(COL_X1, k) - limb(COL_X3, k);
            } else {
                s = p_any;
                if (k < LIMBS) s = s - limb(COL_Y1, k) - limb(COL_Y3, k);
            }
            Expr cs = b.constant(0);
            if (k > 0) cs = cs + carry(e, k - 1);
            if (k < N_CARRY) cs = cs - carry(e, k) * 256;   // the carry out of the last position is zero
            b.assert_zero(s - q_sign[e] * qp + real * cs);
        }
    // x3 < P, y3 < P
    const size_t out[2] = {COL_X3, COL_Y3};
    for (size_t o = 0; o < 2; o++) {
        Expr n_marked = b.constant(0), diff = b.constant(0), above = b.constant(0);
        for (size_t i = 0; i < LIMBS; i++) {
            const Expr m = b.var(COL_MARK + o * LIMBS + i);
            b.assert_zero(m * (m - 1));
            n_marked = n_marked + m;
            diff = diff + m * (b.constant(P[i]) - limb(out[o], i));
        }
        b.assert_zero(n_marked - real);
        for (size_t ii = LIMBS; ii-- > 0;) {
            b.assert_zero((real - above - b.var(COL_MARK + o * LIMBS + ii)) * (limb(out[o], ii) - b.constant(P[ii])));
            above = above + b.var(COL_MARK + o * LIMBS + ii);
        }
        b.assert_zero(b.var(COL_DIFF + o) - diff);
        b.push_interaction(bitwise_bus, {b.var(COL_DIFF + o) - real, zero, zero, zero}, real, Kind::Send);
    }
    for (size_t base : {COL_X1, COL_Y1, COL_X2, COL_Y2, COL_L, COL_X3, COL_Y3})
        for (size_t i = 0; i < LIMBS; i += 2) b.push_interaction(bitwise_bus, {b.var(base + i), b.var(base + i + 1), zero, zero}, real, Kind::Send);
    for (size_t e = 0; e < N_EQ; e++) {
        for (size_t i = 0; i + 1 < Q_LIMBS; i += 2)
            b.push_interaction(bitwise_bus, {b.var(COL_Q + e * Q_LIMBS + i), b.var(COL_Q + e * Q_LIMBS + i + 1), zero, zero}, real, Kind::Send);
        b.push_interaction(bitwise_bus, {b.var(COL_Q + e * Q_LIMBS + Q_LIMBS - 1), zero, zero, zero}, real, Kind::Send);
        for (size_t k = 0; k < N_CARRY; k++) b.push_interaction(tuple_bus, {b.var(COL_CX + e * N_CARRY + k), b.var(COL_CY + e * N_CARRY + k)}, real, Kind::Send);
    }
}

// The chip inside the VM (include/zkhip_vm_circuit.hpp): the same AIR plus a timestamp column; the 6 L / 4 words of a call -- (x1, y1),
// (x2, y2) read and (x3, y3) written -- are received from the curve's adapter on `word_bus` as (timestamp, word index, low half,
// high half, operation).  AirBuilder(Cols(P.limbs).VM_WIDTH, 0)
constexpr size_t COL_TS = WIDTH, VM_WIDTH = WIDTH + 1;   // (32 limbs)
inline void ec_vm_air(AirBuilder& b, const Modulus& P, const Modulus& A, uint32_t bitwise_bus, uint32_t tuple_bus, uint32_t word_bus) {
    ec_air(b, P, A, bitwise_bus, tuple_bus);
    const Cols C(P.limbs);
    const size_t NW = C.L / 4;
    const Expr ts = b.var(C.TS), real = b.var(C.REAL), op = b.var(C.IS_DOUBLE);
    const size_t base[6] = {C.X1, C.Y1, C.X2, C.Y2, C.X3, C.Y3};
    for (size_t o = 0; o < 6; o++)
        for (size_t k = 0; k < NW; k++) {
            const size_t c = base[o] + 4 * k;
            b.push_interaction(word_bus, {ts, b.constant((uint32_t)(NW * o + k)), b.var(c) + b.var(c + 1) * 256, b.var(c + 2) + b.var(c + 3) * 256, op}, real, Kind::Receive);
        }
}

// ---- host arithmetic (the executor's; the tests' expected values come from Python, not from here) ----
struct Curve {
    U256 p, a;
};
inline bool is_zero(const U256& x) {
    uint32_t o = 0;
    for (uint32_t w : x.w) o |= w;
    return o == 0;
}
inline bool less(const U256& x, const U256& y) {
    for (int k = (int)modular::MAX_WORDS - 1; k >= 0; k--)
        if (x.w[k] != y.w[k]) return x.w[k] < y.w[k];
    return false;
}
inline U256 mod_add(const U256& x, const U256& y, const U256& p) {
    U256 q, r;
    modular::addsubmod(modular::OP_ADD, x, y, p, &q, &r);
    return r;
}
inline U256 mod_sub(const U256& x, const U256& y, const U256& p) {   // x, y < p
    U256 q, r;
    modular::addsubmod(modular::OP_SUB, x, y, p, &q, &r);
    return r;
}
inline U256 mod_mul(const U256& x, const U256& y, const U256& p) {
    U256 q, r;
    modular::mulmod(x, y, p, &q, &r);
    return r;
}
// x^-1 mod p for odd p and 0 < x < p with gcd(x, p) = 1; false if there is none
inline bool mod_inv(const U256& x, const U256& p, U256* out) {
    if (is_zero(x) || !(p.w[0] & 1u) || !less(x, p)) return false;
    U256 one{};
    one.w[0] = 1;
    return modular::divmod_p(one, x, p, out);
}
// the slope and the result of one operation; false if an operand is not reduced or the slope does not exist (x1 = x2, resp. y1 = 0)
inline bool ec_op(uint32_t op, const Curve& c, const U256& x1, const U256& y1, const U256& x2, const U256& y2, U256* l, U256* x3, U256* y3) {
    if (op >= N_OPS || !less(x1, c.p) || !less(y1, c.p)) return false;
    if (op == OP_ADD_NE && (!less(x2, c.p) || !less(y2, c.p))) return false;
    U256 num, den, inv;
    if (op == OP_ADD_NE) {
        num = mod_sub(y2, y1, c.p), den = mod_sub(x2, x1, c.p);
    } else {
        const U256 xx = mod_mul(x1, x1, c.p);
        num = mod_add(mod_add(mod_add(xx, xx, c.p), xx, c.p), c.a, c.p), den = mod_add(y1, y1, c.p);
    }
    if (!mod_inv(den, c.p, &inv)) return false;
    *l = mod_mul(num, inv, c.p);
    const U256& xo = op == OP_ADD_NE ? x2 : x1;
    *x3 = mod_sub(mod_sub(mod_mul(*l, *l, c.p), x1, c.p), xo, c.p);
    *y3 = mod_sub(mod_mul(*l, mod_sub(x1, *x3, c.p), c.p), y1, c.p);
    return true;
}

}  // namespace ecc
}  // namespace zkhip
