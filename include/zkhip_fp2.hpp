// zkhip_fp2.hpp -- arithmetic in the quadratic extension Fp[u] / (u^2 + 1) of a 256-bit prime field (SURVEY.md 8(f) f3;
// crates/circuits/chunk-circuit/openvm.toml:30-33 lists `[app_vm_config.fp2] supported_moduli = [["Bn254Fp2", "<p>"]]`: the field under
// bn254's G2 and the pairing precompile).  OpenVM's chips (openvm-algebra-circuit Fp2AddSub / Fp2MulDiv over its mod-builder,
// un-vendored) state the component formulas as integer identities on byte limbs with range-checked carries; this is the same statement in
// this repository's own columns, ONE operation per row on (a0 + a1 u), (b0 + b1 u) -> (r0 + r1 u):
//   a0 a1 b0 b1 r0 r1 [32 each]   byte limbs (little-endian)
//   q0 q1 [33 each] + sign bit    signed quotients of the two component identities
//                                   mul:  a0 b0 - a1 b1 - r0 = q0 P        a0 b1 + a1 b0 - r1 = q1 P
//                                   add:  a_e + b_e - r_e = q_e P          sub:  a_e - b_e - r_e = q_e P        (e = 0, 1)
//   carry (cx, cy)[2][63]         as in include/zkhip_ecc.hpp: limb position k sums to 256 c_k - c_{k-1}, c_k = cx_k + 256 cy_k - 2^18
//   marker[2][32], diff[2]        r0 < P, r1 < P
//   marker2[2][32], diff2[2]      a division x / y is the multiplication row (a, b, r) = (x / y, y, x) (as in include/zkhip_modular.hpp):
//                                 the quotient sits in the a columns, so a0 < P and a1 < P are enforced on those rows; inside the VM the
//                                 word indices of a and r swap.  As in OpenVM the caller guarantees y != 0
//   real, is_add, is_sub, is_div  rows beyond the records are zero
// The modulus is a constant of the AIR (one chip per configured field).  Every constraint has degree <= 3.
// External parity: Python's integers (tests/golden/fp2_kat.json, bn254's Fp2).  Header-only; device generator: csrc/fp2.hip.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

#include "zkhip_air.hpp"
#include "zkhip_ecc.hpp"
#include "zkhip_modular.hpp"

namespace zkhip {
namespace fp2 {
using air::AirBuilder;
using air::Expr;
using air::Kind;
using modular::Modulus;
using modular::U256;

// column layout for L limbs (32: a prime below 2^256; 48: below 2^384 -- BLS12-381's Fp2 of the reference's batch circuit)
struct Cols {
    size_t L, Q_LIMBS, N_POS, N_CARRY, A0, A1, B0, B1, R0, R1, Q, QS, CX, CY, MARK, DIFF, MARK2, DIFF2, REAL, IS_ADD, IS_SUB, IS_DIV, WIDTH, TS, VM_WIDTH, RECORD_WORDS;
    constexpr explicit Cols(size_t l)
        : L(l), Q_LIMBS(l + 1), N_POS(2 * l), N_CARRY(2 * l - 1), A0(0), A1(l), B0(2 * l), B1(3 * l), R0(4 * l), R1(5 * l), Q(6 * l), QS(Q + 2 * Q_LIMBS), CX(QS + 2),
          CY(CX + 2 * N_CARRY), MARK(CY + 2 * N_CARRY), DIFF(MARK + 2 * l), MARK2(DIFF + 2), DIFF2(MARK2 + 2 * l), REAL(DIFF2 + 2), IS_ADD(REAL + 1), IS_SUB(REAL + 2),
          IS_DIV(REAL + 3), WIDTH(REAL + 4), TS(WIDTH), VM_WIDTH(WIDTH + 1), RECORD_WORDS(1 + l) {}
};
constexpr Cols C32(32);
constexpr size_t LIMBS = 32, Q_LIMBS = C32.Q_LIMBS, N_POS = C32.N_POS, N_CARRY = C32.N_CARRY;
constexpr size_t COL_A0 = C32.A0, COL_A1 = C32.A1, COL_B0 = C32.B0, COL_B1 = C32.B1, COL_R0 = C32.R0, COL_R1 = C32.R1, COL_Q = C32.Q, COL_QS = C32.QS, COL_CX = C32.CX,
                 COL_CY = C32.CY, COL_MARK = C32.MARK, COL_DIFF = C32.DIFF, COL_MARK2 = C32.MARK2, COL_DIFF2 = C32.DIFF2, COL_REAL = C32.REAL, COL_IS_ADD = C32.IS_ADD,
                 COL_IS_SUB = C32.IS_SUB, COL_IS_DIV = C32.IS_DIV, WIDTH = C32.WIDTH;
static_assert(WIDTH == 648, "fp2 chip width");
static_assert(Cols(48).WIDTH == 968, "fp2 chip width, 48 limbs");
enum Op : uint32_t { OP_MUL, OP_ADD, OP_SUB, OP_DIV, N_OPS };
constexpr int64_t CARRY_OFFSET = 1 << 18;
constexpr size_t RECORD_WORDS = C32.RECORD_WORDS;   // op | a0 a1 | b0 b1   (a division's record holds the quotient x / y in the a slot)

// AirBuilder(WIDTH, 0); bitwise_bus: (x, y, 0, 0) byte pairs; tuple_bus: (cx, cy)
inline void fp2_air(AirBuilder& b, const Modulus& P, uint32_t bitwise_bus, uint32_t tuple_bus) {
    const Cols C(P.limbs);
    const size_t LIMBS = C.L, Q_LIMBS = C.Q_LIMBS, N_POS = C.N_POS, N_CARRY = C.N_CARRY;
    const size_t COL_A0 = C.A0, COL_A1 = C.A1, COL_B0 = C.B0, COL_B1 = C.B1, COL_R0 = C.R0, COL_R1 = C.R1, COL_Q = C.Q, COL_QS = C.QS, COL_CX = C.CX, COL_CY = C.CY,
                 COL_MARK = C.MARK, COL_DIFF = C.DIFF, COL_MARK2 = C.MARK2, COL_DIFF2 = C.DIFF2, COL_REAL = C.REAL, COL_IS_ADD = C.IS_ADD, COL_IS_SUB = C.IS_SUB,
                 COL_IS_DIV = C.IS_DIV;
    const Expr real = b.var(COL_REAL), is_add = b.var(COL_IS_ADD), is_sub = b.var(COL_IS_SUB), is_div = b.var(COL_IS_DIV), is_mul = real - is_add - is_sub,
               zero = b.constant(0);
    // (is_mul covers the division rows: the same limb identities)
    for (const Expr& f : {real, is_add, is_sub, is_div, is_mul, is_mul - is_div}) b.assert_zero(f * (f - 1));
    auto carry = [&](size_t e, size_t k) { return b.var(COL_CX + e * N_CARRY + k) + b.var(COL_CY + e * N_CARRY + k) * 256 - CARRY_OFFSET; };
    Expr q_sign[2];
    for (size_t e = 0; e < 2; e++) {
        const Expr qs = b.var(COL_QS + e);
        q_sign[e] = real - qs * 2;
        b.assert_zero(qs * (qs - real));
    }
    const size_t A[2] = {COL_A0, COL_A1}, B[2] = {COL_B0, COL_B1}, R[2] = {COL_R0, COL_R1};
    // (position by position, the two identities side by side; quotient sums start with their outermost limbs: see include/zkhip_ecc.hpp)
    for (size_t k = 0; k < N_POS; k++)
        for (size_t e = 0; e < 2; e++) {
            Expr qp = b.constant(0);
            for (unsigned v = 1; v < 256; v++) {
                std::vector<size_t> idx;
                for (size_t i = 0; i < Q_LIMBS; i++)
                    if (k >= i && k - i < LIMBS && P[k - i] == v) idx.push_back(i);
                if (idx.empty()) continue;
                Expr group = b.var(COL_Q + e * Q_LIMBS + idx.back());
                for (size_t t = 0; t + 1 < idx.size(); t++) group = group + b.var(COL_Q + e * Q_LIMBS + idx[t]);
                qp = qp + group * (int64_t)v;
            }
            Expr prod = b.constant(0);
            for (size_t i = 0; i < LIMBS; i++) {
                if (k < i || k - i >= LIMBS) continue;
                const size_t j = k - i;
                if (e == 0) prod = prod + b.var(COL_A0 + i) * b.var(COL_B0 + j) - b.var(COL_A1 + i) * b.var(COL_B1 + j);
                else prod = prod + b.var(COL_A0 + i) * b.var(COL_B1 + j) + b.var(COL_A1 + i) * b.var(COL_B0 + j);
            }
            Expr s = is_mul * prod;
            if (k < LIMBS) s = s + is_add * (b.var(A[e] + k) + b.var(B[e] + k)) + is_sub * (b.var(A[e] + k) - b.var(B[e] + k)) - real * b.var(R[e] + k);
            Expr cs = b.constant(0);
            if (k > 0) cs = cs + carry(e, k - 1);
            if (k < N_CARRY) cs = cs - carry(e, k) * 256;
            b.assert_zero(s - q_sign[e] * qp + real * cs);
        }
    // r0, r1 < P on every row; a0, a1 < P on the division rows
    auto below_p = [&](size_t col, size_t mark, size_t diffc, const Expr& on) {
        Expr n_marked = b.constant(0), diff = b.constant(0), above = b.constant(0);
        for (size_t i = 0; i < LIMBS; i++) {
            const Expr m = b.var(mark + i);
            b.assert_zero(m * (m - 1));
            n_marked = n_marked + m;
            diff = diff + m * (b.constant(P[i]) - b.var(col + i));
        }
        b.assert_zero(n_marked - on);
        for (size_t ii = LIMBS; ii-- > 0;) {
            b.assert_zero((on - above - b.var(mark + ii)) * (b.var(col + ii) - b.constant(P[ii])));
            above = above + b.var(mark + ii);
        }
        b.assert_zero(b.var(diffc) - diff);
        b.push_interaction(bitwise_bus, {b.var(diffc) - on, zero, zero, zero}, on, Kind::Send);
    };
    for (size_t e = 0; e < 2; e++) below_p(R[e], COL_MARK + e * LIMBS, COL_DIFF + e, real);
    for (size_t e = 0; e < 2; e++) below_p(A[e], COL_MARK2 + e * LIMBS, COL_DIFF2 + e, is_div);
    for (size_t base : {COL_A0, COL_A1, COL_B0, COL_B1, COL_R0, COL_R1})
        for (size_t i = 0; i < LIMBS; i += 2) b.push_interaction(bitwise_bus, {b.var(base + i), b.var(base + i + 1), zero, zero}, real, Kind::Send);
    for (size_t e = 0; e < 2; e++) {
        for (size_t i = 0; i + 1 < Q_LIMBS; i += 2)
            b.push_interaction(bitwise_bus, {b.var(COL_Q + e * Q_LIMBS + i), b.var(COL_Q + e * Q_LIMBS + i + 1), zero, zero}, real, Kind::Send);
        b.push_interaction(bitwise_bus, {b.var(COL_Q + e * Q_LIMBS + Q_LIMBS - 1), zero, zero, zero}, real, Kind::Send);
        for (size_t k = 0; k < N_CARRY; k++) b.push_interaction(tuple_bus, {b.var(COL_CX + e * N_CARRY + k), b.var(COL_CY + e * N_CARRY + k)}, real, Kind::Send);
    }
}

// The chip inside the VM: + a timestamp column; the 6 L / 4 words of a call -- a, b read and r written, two components each -- are
// received from the field's adapter on `word_bus` as (timestamp, word index, low half, high half, operation); for a division the
// first operand's words are the r columns and the result's the a columns.  AirBuilder(Cols(P.limbs).VM_WIDTH, 0)
constexpr size_t COL_TS = WIDTH, VM_WIDTH = WIDTH + 1;   // (32 limbs)
inline void fp2_vm_air(AirBuilder& b, const Modulus& P, uint32_t bitwise_bus, uint32_t tuple_bus, uint32_t word_bus) {
    fp2_air(b, P, bitwise_bus, tuple_bus);
    const Cols C(P.limbs);
    const size_t NW = C.L / 4;
    const Expr ts = b.var(C.TS), real = b.var(C.REAL), is_div = b.var(C.IS_DIV), op = b.var(C.IS_ADD) + b.var(C.IS_SUB) * 2 + is_div * 3;
    auto half = [&](size_t base, size_t k, size_t h) { return b.var(base + 4 * k + 2 * h) + b.var(base + 4 * k + 2 * h + 1) * 256; };
    const size_t A[2] = {C.A0, C.A1}, B[2] = {C.B0, C.B1}, R[2] = {C.R0, C.R1};
    for (size_t e = 0; e < 2; e++)
        for (size_t k = 0; k < NW; k++) {
            const Expr a_lo = half(A[e], k, 0), a_hi = half(A[e], k, 1), r_lo = half(R[e], k, 0), r_hi = half(R[e], k, 1);
            const Expr sw_lo = is_div * (r_lo - a_lo), sw_hi = is_div * (r_hi - a_hi);
            b.push_interaction(word_bus, {ts, b.constant((uint32_t)(NW * e + k)), a_lo + sw_lo, a_hi + sw_hi, op}, real, Kind::Receive);
            b.push_interaction(word_bus, {ts, b.constant((uint32_t)(2 * NW + NW * e + k)), half(B[e], k, 0), half(B[e], k, 1), op}, real, Kind::Receive);
            b.push_interaction(word_bus, {ts, b.constant((uint32_t)(4 * NW + NW * e + k)), r_lo - sw_lo, r_hi - sw_hi, op}, real, Kind::Receive);
        }
}

// ---- host arithmetic (the executor's; the tests' expected values come from Python, not from here) ----
struct Elem {
    U256 c0, c1;
};
// r = a op b in Fp[u] / (u^2 + 1); components of a and b below p; false if a division has no quotient (b = 0) or an operand is not reduced
inline bool fp2_op(uint32_t op, const U256& p, const Elem& a, const Elem& b, Elem* r) {
    using namespace ecc;
    if (op >= N_OPS || !less(a.c0, p) || !less(a.c1, p) || !less(b.c0, p) || !less(b.c1, p) || !(p.w[0] & 1u)) return false;
    auto mul = [&](const Elem& x, const Elem& y) {
        return Elem{mod_sub(mod_mul(x.c0, y.c0, p), mod_mul(x.c1, y.c1, p), p), mod_add(mod_mul(x.c0, y.c1, p), mod_mul(x.c1, y.c0, p), p)};
    };
    switch (op) {
        case OP_ADD: *r = Elem{mod_add(a.c0, b.c0, p), mod_add(a.c1, b.c1, p)}; return true;
        case OP_SUB: *r = Elem{mod_sub(a.c0, b.c0, p), mod_sub(a.c1, b.c1, p)}; return true;
        case OP_MUL: *r = mul(a, b); return true;
        default: {   // a / b = a conj(b) / (b0^2 + b1^2)
            const U256 norm = mod_add(mod_mul(b.c0, b.c0, p), mod_mul(b.c1, b.c1, p), p);
            U256 inv;
            if (!mod_inv(norm, p, &inv)) return false;
            const U256 zero{};
            const Elem conj{b.c0, mod_sub(zero, b.c1, p)}, t = mul(a, conj);
            *r = Elem{mod_mul(t.c0, inv, p), mod_mul(t.c1, inv, p)};
            return true;
        }
    }
}

}  // namespace fp2
}  // namespace zkhip
