// zkhip_int256.hpp -- 256-bit ALU chip (crates/circuits/chunk-circuit/openvm.toml:16-17 enables `bigint`: the EVM's
// 256-bit words).  OpenVM's chip is the RV32 base ALU core instantiated with 32 limbs (openvm-bigint-circuit Rv32BaseAlu256, un-vendored);
// this is the 32-limb form of include/zkhip_chips.hpp's ALU core: ONE operation per row,
//   a[32] (result) | b[32] | c[32] | is_add is_sub is_xor is_or is_and       byte limbs, little-endian; 101 columns
// ADD / SUB through their carry chains (carry_i = (b_i + c_i + carry_{i-1} - a_i) / 256 boolean; SUB with a and b exchanged; the carry
// out of limb 31 is dropped: arithmetic modulo 2^256); XOR / OR / AND -- and the byte range of every result limb -- through the bitwise
// lookup: per limb (x, y, x ^ y, 1) with (x, y) = (b_i, c_i) for the bitwise opcodes and (a_i, a_i) for ADD / SUB, where x ^ y = a_i
// (XOR), 2 a_i - b_i - c_i (OR), b_i + c_i - 2 a_i (AND), 0 (ADD / SUB).  Degree 3.
// External parity: Python's integers (tests/golden/int256_kat.json).  Header-only; device generator: csrc/int256.hip.
#pragma once
#include <cstdint>
#include <vector>

#include "zkhip_air.hpp"

namespace zkhip {
namespace int256 {
using air::AirBuilder;
using air::Expr;
using air::Kind;

constexpr size_t LIMBS = 32, COL_A = 0, COL_B = 32, COL_C = 64, COL_FLAGS = 96, WIDTH = 101;
enum Op : uint32_t { OP_ADD, OP_SUB, OP_XOR, OP_OR, OP_AND, N_OPS };

// AirBuilder(WIDTH, 0)
inline void alu256_air(AirBuilder& b, uint32_t bitwise_bus) {
    Expr f[N_OPS];
    Expr ok = b.constant(0);
    for (size_t i = 0; i < N_OPS; i++) {
        f[i] = b.var(COL_FLAGS + i);
        b.assert_zero(f[i] * (f[i] - 1));
        ok = ok + f[i];
    }
    b.assert_zero(ok * (ok - 1));
    const int64_t inv256 = 2005401601;   // 256^-1 modulo the BabyBear prime
    Expr carry_add = b.constant(0), carry_sub = b.constant(0);
    for (size_t i = 0; i < LIMBS; i++) {
        const Expr ai = b.var(COL_A + i), bi = b.var(COL_B + i), ci = b.var(COL_C + i);
        carry_add = (bi + ci - ai + carry_add) * inv256;
        carry_sub = (ai + ci - bi + carry_sub) * inv256;
        b.assert_zero(f[OP_ADD] * (carry_add * (carry_add - 1)));
        b.assert_zero(f[OP_SUB] * (carry_sub * (carry_sub - 1)));
    }
    const Expr bitwise = f[OP_XOR] + f[OP_OR] + f[OP_AND], one = b.constant(1);
    for (size_t i = 0; i < LIMBS; i++) {
        const Expr ai = b.var(COL_A + i), bi = b.var(COL_B + i), ci = b.var(COL_C + i);
        const Expr x = bitwise * bi + (1 - bitwise) * ai, y = bitwise * ci + (1 - bitwise) * ai;
        const Expr z = f[OP_XOR] * ai + f[OP_OR] * (ai * 2 - bi - ci) + f[OP_AND] * (bi + ci - ai * 2);
        b.push_interaction(bitwise_bus, {x, y, z, one}, ok, Kind::Send);
    }
}

// The chip inside the VM: + a timestamp column; the 24 words of a call (b, c read; a written) are received from the int256 adapter on
// `word_bus` as (timestamp, word index, low half, high half, opcode).  AirBuilder(VM_WIDTH, 0)
constexpr size_t COL_TS = WIDTH, VM_WIDTH = WIDTH + 1;
inline void alu256_vm_air(AirBuilder& b, uint32_t bitwise_bus, uint32_t word_bus) {
    alu256_air(b, bitwise_bus);
    const Expr ts = b.var(COL_TS);
    Expr ok = b.constant(0), op = b.constant(0);
    for (size_t i = 0; i < N_OPS; i++) ok = ok + b.var(COL_FLAGS + i), op = op + b.var(COL_FLAGS + i) * (int64_t)i;
    const size_t base[3] = {COL_B, COL_C, COL_A};
    for (size_t o = 0; o < 3; o++)
        for (size_t k = 0; k < 8; k++) {
            const size_t c = base[o] + 4 * k;
            b.push_interaction(word_bus, {ts, b.constant((uint32_t)(8 * o + k)), b.var(c) + b.var(c + 1) * 256, b.var(c + 2) + b.var(c + 3) * 256, op}, ok, Kind::Receive);
        }
}

// ---- 256-bit multiplication (the low 256 bits of b c: OpenVM's Rv32Multiplication256) -- a chip of its own: per limb position k < 32
//   sum_{i + j = k} b_i c_j + carry_{k-1} = a_k + 256 carry_k,   carry_k = cx_k + 256 cy_k with (cx_k, cy_k) in the range-tuple table
// (a carry stays below 32 * 255^2 / 255 < 2^13; everything stays far below the field's characteristic, so the equations hold over the
// integers).  Columns: a[32] | b[32] | c[32] | cx[32] | cy[32] | real; bytes looked up pairwise in the bitwise table.  Degree 3.
constexpr uint32_t OP_MUL = 5;
constexpr size_t MUL_COL_CX = 96, MUL_COL_CY = 128, MUL_COL_REAL = 160, MUL_WIDTH = 161;
inline void mul256_air(AirBuilder& b, uint32_t bitwise_bus, uint32_t tuple_bus) {
    const Expr real = b.var(MUL_COL_REAL), zero = b.constant(0);
    b.assert_zero(real * (real - 1));
    auto carry = [&](size_t k) { return b.var(MUL_COL_CX + k) + b.var(MUL_COL_CY + k) * 256; };
    for (size_t k = 0; k < LIMBS; k++) {
        Expr s = b.constant(0);
        for (size_t i = 0; i <= k; i++) s = s + b.var(COL_B + i) * b.var(COL_C + k - i);
        if (k > 0) s = s + carry(k - 1);
        b.assert_zero(real * (s - b.var(COL_A + k) - carry(k) * 256));
    }
    for (size_t base : {COL_A, COL_B, COL_C})
        for (size_t i = 0; i < LIMBS; i += 2) b.push_interaction(bitwise_bus, {b.var(base + i), b.var(base + i + 1), zero, zero}, real, Kind::Send);
    for (size_t k = 0; k < LIMBS; k++) b.push_interaction(tuple_bus, {b.var(MUL_COL_CX + k), b.var(MUL_COL_CY + k)}, real, Kind::Send);
}
// inside the VM: + a timestamp column; the call's 24 words arrive on the ALU chip's word bus with opcode OP_MUL
constexpr size_t MUL_COL_TS = MUL_WIDTH, MUL_VM_WIDTH = MUL_WIDTH + 1;
inline void mul256_vm_air(AirBuilder& b, uint32_t bitwise_bus, uint32_t tuple_bus, uint32_t word_bus) {
    mul256_air(b, bitwise_bus, tuple_bus);
    const Expr ts = b.var(MUL_COL_TS), real = b.var(MUL_COL_REAL);
    const size_t base[3] = {COL_B, COL_C, COL_A};
    for (size_t o = 0; o < 3; o++)
        for (size_t k = 0; k < 8; k++) {
            const size_t c = base[o] + 4 * k;
            b.push_interaction(word_bus, {ts, b.constant((uint32_t)(8 * o + k)), b.var(c) + b.var(c + 1) * 256, b.var(c + 2) + b.var(c + 3) * 256, b.constant(OP_MUL)}, real,
                               Kind::Receive);
        }
}
// ---- 256-bit comparisons (OpenVM's Rv32LessThan256 and the equality its Rv32BranchEqual256 decides) -- a chip of its own, the RV32
// less-than core (include/zkhip_chips.hpp) at 32 limbs with an equality output: ONE comparison per row,
//   b[32] | c[32] | t | marker[32] | diff | b_msb c_msb | is_sltu is_slt is_eq                    103 columns
// The most significant limb where b and c differ is marked; there (c_i - b_i) (2 t - 1) = diff is in 1..255 (one range lookup), so t says
// which operand is smaller; limbs above the marker are equal; no marker = equal operands (then t = 0).  SLT reads the top limbs as
// signed bytes (b_msb = b_31 or b_31 - 256, looked up shifted by 128).  Result word: t for SLTU / SLT, 1 - (a marker exists) for EQ.
// b and c are looked up pairwise in the bitwise table (the comparison of limbs needs honest bytes).  Degree 3.
constexpr uint32_t OP_SLTU = 6, OP_SLT = 7, OP_EQ = 8;
constexpr size_t CMP_COL_B = 0, CMP_COL_C = 32, CMP_COL_T = 64, CMP_COL_MARK = 65, CMP_COL_DIFF = 97, CMP_COL_BMSB = 98, CMP_COL_CMSB = 99, CMP_COL_FLAGS = 100,
                 CMP_WIDTH = 103;
// the result limb 0 (the other limbs of the result word are zero)
inline Expr cmp256_air(AirBuilder& b, uint32_t bitwise_bus) {
    const Expr is_sltu = b.var(CMP_COL_FLAGS), is_slt = b.var(CMP_COL_FLAGS + 1), is_eq = b.var(CMP_COL_FLAGS + 2), real = is_sltu + is_slt + is_eq;
    const Expr t = b.var(CMP_COL_T), dv = b.var(CMP_COL_DIFF), bm = b.var(CMP_COL_BMSB), cm = b.var(CMP_COL_CMSB), zero = b.constant(0);
    for (const Expr& f : {is_sltu, is_slt, is_eq, real, t}) b.assert_zero(f * (f - 1));
    const size_t top[2] = {CMP_COL_B + LIMBS - 1, CMP_COL_C + LIMBS - 1};
    const Expr msb[2] = {bm, cm};
    for (int o = 0; o < 2; o++) {
        const Expr d = b.var(top[o]) - msb[o];   // 0, or 256 for a negative signed operand
        b.assert_zero(d * (d - 256));
        b.assert_zero((1 - is_slt) * d);
    }
    const Expr sign = t * 2 - 1;
    Expr prefix = b.constant(0);
    bool first = true;
    for (size_t ii = LIMBS; ii-- > 0;) {
        const Expr m = b.var(CMP_COL_MARK + ii);
        b.assert_zero(m * (m - 1));
        const Expr diff = ((ii == LIMBS - 1 ? cm : b.var(CMP_COL_C + ii)) - (ii == LIMBS - 1 ? bm : b.var(CMP_COL_B + ii))) * sign;
        prefix = first ? m : prefix + m;
        first = false;
        b.assert_zero((1 - prefix) * diff);   // no marker at or above this limb: equal here
        b.assert_zero(m * (dv - diff));
    }
    b.assert_zero(prefix * (prefix - 1));
    b.assert_zero((1 - prefix) * t);          // equal operands are not less
    b.assert_zero((1 - real) * prefix);       // rows beyond the records carry no marker
    b.push_interaction(bitwise_bus, {bm + is_slt * 128, cm + is_slt * 128, zero, zero}, real, Kind::Send);
    b.push_interaction(bitwise_bus, {dv - 1, zero, zero, zero}, prefix, Kind::Send);
    for (size_t base : {CMP_COL_B, CMP_COL_C})
        for (size_t i = 0; i < LIMBS; i += 2) b.push_interaction(bitwise_bus, {b.var(base + i), b.var(base + i + 1), zero, zero}, real, Kind::Send);
    return is_eq * (1 - prefix) + (is_sltu + is_slt) * t;
}
// inside the VM: + a timestamp column; the call's 24 words arrive on the ALU chip's word bus with the comparison's opcode.
// 256-BIT BRANCHES (round 6; OpenVM's Rv32BranchEqual256 / Rv32BranchLessThan256, crates/circuits/chunk-circuit/openvm.toml:17-18): the same
// comparison decides a branch.  Opcodes 12 beq, 13 bne, 14 bltu, 15 blt, 16 bgeu, 17 bge: the row is the comparison's (equality for 12 / 13,
// unsigned less-than for 14 / 16, signed for 15 / 17; the result word is still written, as the comparison's), plus
//   is_br   the call is a branch            neg   the branch is taken when the comparison is FALSE (bne, bgeu, bge)
//   taken   = out XOR neg: sent to the ecall chip on the branch bus with the call's timestamp -- there it chooses between pc + 4 and
//           pc + a2 (include/zkhip_vm_circuit.hpp ecall_air).
constexpr uint32_t OP_BEQ = 12, OP_BNE = 13, OP_BLTU = 14, OP_BLT = 15, OP_BGEU = 16, OP_BGE = 17;
inline bool is_branch_op(uint32_t op) { return op >= OP_BEQ && op <= OP_BGE; }
inline bool branch_negates(uint32_t op) { return op == OP_BNE || op == OP_BGEU || op == OP_BGE; }
// the comparison a branch opcode rests on (OP_EQ / OP_SLTU / OP_SLT); other opcodes unchanged
inline uint32_t compare_op_of(uint32_t op) { return op == OP_BEQ || op == OP_BNE ? 8u : op == OP_BLTU || op == OP_BGEU ? 6u : op == OP_BLT || op == OP_BGE ? 7u : op; }
//   opc     the call's opcode as the adapter announces it (a column, so that the 24 word-bus messages stay of degree 1)
constexpr size_t CMP_COL_TS = CMP_WIDTH, CMP_COL_BR = CMP_WIDTH + 1, CMP_COL_NEG = CMP_WIDTH + 2, CMP_COL_TAKEN = CMP_WIDTH + 3, CMP_COL_OPC = CMP_WIDTH + 4,
                 CMP_VM_WIDTH = CMP_WIDTH + 5;
inline void cmp256_vm_air(AirBuilder& b, uint32_t bitwise_bus, uint32_t word_bus, uint32_t branch_bus) {
    const Expr out = cmp256_air(b, bitwise_bus);
    const Expr ts = b.var(CMP_COL_TS), is_sltu = b.var(CMP_COL_FLAGS), is_slt = b.var(CMP_COL_FLAGS + 1), is_eq = b.var(CMP_COL_FLAGS + 2);
    const Expr is_br = b.var(CMP_COL_BR), neg = b.var(CMP_COL_NEG), taken = b.var(CMP_COL_TAKEN);
    const Expr real = is_sltu + is_slt + is_eq, zero = b.constant(0);
    for (const Expr& f : {is_br, neg, taken}) b.assert_zero(f * (f - 1));
    b.assert_zero(is_br * (1 - real));    // a branch is a real comparison row
    b.assert_zero(neg * (1 - is_br));     // only a branch negates
    b.assert_zero(taken * (1 - is_br));
    b.assert_zero(taken - is_br * out - neg + out * neg * 2);     // taken = out XOR neg on a branch row, 0 elsewhere (neg implies is_br); degree 3
    // the opcode the adapter announces on the word bus: 6 / 7 / 8 for a comparison, 14 + 2 neg / 15 + 2 neg / 12 + neg for a branch
    const Expr op = b.var(CMP_COL_OPC);
    b.assert_zero(op - (is_sltu * (int64_t)OP_SLTU + is_slt * (int64_t)OP_SLT + is_eq * (int64_t)OP_EQ +
                        is_br * (is_sltu * (int64_t)(OP_BLTU - OP_SLTU) + is_slt * (int64_t)(OP_BLT - OP_SLT) + is_eq * (int64_t)(OP_BEQ - OP_EQ)) + neg * (is_sltu + is_slt) * 2 + neg * is_eq));
    b.push_interaction(branch_bus, {ts, taken}, is_br, Kind::Send);
    const size_t base[2] = {CMP_COL_B, CMP_COL_C};
    for (size_t o = 0; o < 2; o++)
        for (size_t k = 0; k < 8; k++) {
            const size_t c = base[o] + 4 * k;
            b.push_interaction(word_bus, {ts, b.constant((uint32_t)(8 * o + k)), b.var(c) + b.var(c + 1) * 256, b.var(c + 2) + b.var(c + 3) * 256, op}, real, Kind::Receive);
        }
    for (size_t k = 0; k < 8; k++) b.push_interaction(word_bus, {ts, b.constant((uint32_t)(16 + k)), k == 0 ? out : zero, zero, op}, real, Kind::Receive);
}
// 0 / 1 (host)
inline uint32_t cmp256(uint32_t op, const uint32_t b[8], const uint32_t c[8]) {
    bool eq = true, lt = false;
    for (int i = 7; i >= 0; i--)
        if (b[i] != c[i]) {
            eq = false, lt = b[i] < c[i];
            break;
        }
    op = compare_op_of(op);   // (a branch opcode: the comparison it rests on)
    if (op == OP_EQ) return eq ? 1u : 0u;
    if (op == OP_SLT && ((b[7] ^ c[7]) >> 31)) return b[7] >> 31;   // different signs: the negative one is smaller
    return lt ? 1u : 0u;
}
// a 256-bit branch: taken?
inline bool branch256_taken(uint32_t op, const uint32_t b[8], const uint32_t c[8]) { return (cmp256(op, b, c) != 0) != branch_negates(op); }
// ---- 256-bit shifts (OpenVM's Rv32Shift256: SLL / SRL / SRA by c mod 256) -- a chip of its own, in two steps per row:
//   shift amount   c0 = bit_shift + 8 limb_shift with one-hot markers bm[8], lm[32] (c0, c1 looked up as bytes: the low half word of c
//                  splits uniquely; the other halves of c only pass through to the bus); mult = 2^bit_shift = sum bm_i 2^i
//   bit step       t = b shifted by bit_shift bits, limb by limb with carries cy_k < mult (looked up as (cy_k, mult - 1 - cy_k)):
//                    left :  t_k + 256 cy_k = b_k mult + cy_{k-1}                          (the carry out of limb 31 is dropped)
//                    right:  t'_m mult + cy_m = b_m + 256 cy_{m+1},  cy_32 = sign (mult - 1)   (sign = b's top bit for SRA, else 0)
//                  a right shift stores t' REVERSED (column k holds t'_{31-k}): then both directions select limbs the same way
//   limb step      [left] a_i, [right] a_{31-i}  =  sum_{j <= i} lm_j t_{i-j}  +  [right] 255 sign (sum_{j > i} lm_j)
// so every product lm_j t_k occurs once (OpenVM's chip spells out all 32 x 32 (limb shift, limb) cases for both directions).
// Columns: a[32] | b[32] | t[32] | cy[32] | c0 c1 c_hi0 | c words 1..7 as halves [14] | bm[8] | lm[32] | sign | is_sll is_srl is_sra   189
// Lookups: 32 carries, t and b pairwise (the limb equations need honest bytes; a inherits from t), (c0, c1), the sign bit of b_31 as the
// XOR (b_31, 128, b_31 + 128 - 256 sign) for SRA.  Degree 3.
constexpr uint32_t OP_SLL = 9, OP_SRL = 10, OP_SRA = 11, N_INT256_OPS = 18;   // (12 .. 17: the branches, above)
constexpr size_t SH_COL_A = 0, SH_COL_B = 32, SH_COL_T = 64, SH_COL_CY = 96, SH_COL_C0 = 128, SH_COL_C1 = 129, SH_COL_CHI0 = 130, SH_COL_CW = 131, SH_COL_BM = 145,
                 SH_COL_LM = 153, SH_COL_SIGN = 185, SH_COL_FLAGS = 186, SH_WIDTH = 189;
inline void shift256_air(AirBuilder& b, uint32_t bitwise_bus) {
    const Expr sll = b.var(SH_COL_FLAGS), srl = b.var(SH_COL_FLAGS + 1), sra = b.var(SH_COL_FLAGS + 2), real = sll + srl + sra, right = srl + sra;
    const Expr sign = b.var(SH_COL_SIGN), zero = b.constant(0), one = b.constant(1);
    for (const Expr& f : {sll, srl, sra, real, sign}) b.assert_zero(f * (f - 1));
    b.assert_zero(sign * (1 - sra));
    Expr sbm = b.var(SH_COL_BM), mult = b.var(SH_COL_BM), amount = b.var(SH_COL_BM + 1), slm = b.var(SH_COL_LM);
    b.assert_zero(b.var(SH_COL_BM) * (b.var(SH_COL_BM) - 1));
    b.assert_zero(b.var(SH_COL_LM) * (b.var(SH_COL_LM) - 1));
    for (size_t i = 1; i < 8; i++) {
        const Expr m = b.var(SH_COL_BM + i);
        b.assert_zero(m * (m - 1));
        sbm = sbm + m, mult = mult + m * (int64_t)(1 << i);
        if (i > 1) amount = amount + m * (int64_t)i;
    }
    for (size_t j = 1; j < LIMBS; j++) {
        const Expr m = b.var(SH_COL_LM + j);
        b.assert_zero(m * (m - 1));
        slm = slm + m, amount = amount + m * (int64_t)(8 * j);
    }
    b.assert_zero(sbm - real);
    b.assert_zero(slm - real);
    b.assert_zero(b.var(SH_COL_C0) - amount);
    auto cy = [&](size_t k) { return b.var(SH_COL_CY + k); };
    auto t = [&](size_t k) { return b.var(SH_COL_T + k); };
    // bit step
    for (size_t k = 0; k < LIMBS; k++) {
        Expr left = t(k) + cy(k) * 256 - b.var(SH_COL_B + k) * mult;
        if (k > 0) left = left - cy(k - 1);
        const size_t m = LIMBS - 1 - k;
        const Expr in = k == 0 ? sign * (mult - 1) : cy(m + 1);
        const Expr rgt = t(k) * mult + cy(m) - b.var(SH_COL_B + m) - in * 256;
        b.assert_zero(sll * left + right * rgt);
    }
    // limb step (from the top: the sums of the markers above limb i grow one term at a time)
    const Expr fill = right * sign * 255;
    Expr above = b.constant(0);
    bool any_above = false;
    for (size_t ii = LIMBS; ii-- > 0;) {
        Expr sel = b.var(SH_COL_LM) * t(ii);
        for (size_t j = 1; j <= ii; j++) sel = sel + b.var(SH_COL_LM + j) * t(ii - j);
        Expr rhs = sel;
        if (any_above) rhs = rhs + fill * above;
        b.assert_zero(sll * b.var(SH_COL_A + ii) + right * b.var(SH_COL_A + LIMBS - 1 - ii) - rhs);
        above = any_above ? above + b.var(SH_COL_LM + ii) : b.var(SH_COL_LM + ii);
        any_above = true;
    }
    for (size_t k = 0; k < LIMBS; k++) b.push_interaction(bitwise_bus, {cy(k), mult - 1 - cy(k), zero, zero}, real, Kind::Send);
    for (size_t base : {SH_COL_T, SH_COL_B})
        for (size_t i = 0; i < LIMBS; i += 2) b.push_interaction(bitwise_bus, {b.var(base + i), b.var(base + i + 1), zero, zero}, real, Kind::Send);
    b.push_interaction(bitwise_bus, {b.var(SH_COL_C0), b.var(SH_COL_C1), zero, zero}, real, Kind::Send);
    const Expr top = b.var(SH_COL_B + LIMBS - 1);
    b.push_interaction(bitwise_bus, {top, b.constant(128), top + 128 - sign * 256, one}, sra, Kind::Send);
}
// inside the VM: + a timestamp column; the call's 24 words arrive on the ALU chip's word bus with the shift's opcode
constexpr size_t SH_COL_TS = SH_WIDTH, SH_VM_WIDTH = SH_WIDTH + 1;
inline void shift256_vm_air(AirBuilder& b, uint32_t bitwise_bus, uint32_t word_bus) {
    shift256_air(b, bitwise_bus);
    const Expr ts = b.var(SH_COL_TS), sll = b.var(SH_COL_FLAGS), srl = b.var(SH_COL_FLAGS + 1), sra = b.var(SH_COL_FLAGS + 2), real = sll + srl + sra;
    const Expr op = sll * (int64_t)OP_SLL + srl * (int64_t)OP_SRL + sra * (int64_t)OP_SRA;
    auto word = [&](size_t base, size_t k, Expr* lo, Expr* hi) { *lo = b.var(base + 4 * k) + b.var(base + 4 * k + 1) * 256, *hi = b.var(base + 4 * k + 2) + b.var(base + 4 * k + 3) * 256; };
    for (size_t k = 0; k < 8; k++) {
        Expr lo, hi;
        word(SH_COL_B, k, &lo, &hi);
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)k), lo, hi, op}, real, Kind::Receive);
        if (k == 0) lo = b.var(SH_COL_C0) + b.var(SH_COL_C1) * 256, hi = b.var(SH_COL_CHI0);
        else lo = b.var(SH_COL_CW + 2 * (k - 1)), hi = b.var(SH_COL_CW + 2 * (k - 1) + 1);
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)(8 + k)), lo, hi, op}, real, Kind::Receive);
        word(SH_COL_A, k, &lo, &hi);
        b.push_interaction(word_bus, {ts, b.constant((uint32_t)(16 + k)), lo, hi, op}, real, Kind::Receive);
    }
}
// a <- b shifted by c mod 256 (host)
inline void shift256(uint32_t op, const uint32_t b[8], const uint32_t c[8], uint32_t a[8]) {
    const unsigned s = c[0] & 255u, ws = s / 32, bs = s % 32;
    const uint32_t fill = op == OP_SRA && (b[7] >> 31) ? 0xffffffffu : 0u;
    for (int i = 0; i < 8; i++) {
        if (op == OP_SLL) {
            const int k = i - (int)ws;
            const uint32_t lo = k >= 0 ? b[k] : 0u, below = k - 1 >= 0 ? b[k - 1] : 0u;
            a[i] = bs ? (lo << bs) | (below >> (32 - bs)) : lo;
        } else {
            const unsigned k = (unsigned)i + ws;
            const uint32_t lo = k < 8 ? b[k] : fill, up = k + 1 < 8 ? b[k + 1] : fill;
            a[i] = bs ? (lo >> bs) | (up << (32 - bs)) : lo;
        }
    }
}
inline void mul256(const uint32_t b[8], const uint32_t c[8], uint32_t a[8]) {
    uint32_t t[8] = {};
    for (int i = 0; i < 8; i++) {
        uint64_t carry = 0;
        for (int j = 0; i + j < 8; j++) {
            carry += (uint64_t)b[i] * c[j] + t[i + j];
            t[i + j] = (uint32_t)carry, carry >>= 32;
        }
    }
    for (int i = 0; i < 8; i++) a[i] = t[i];
}

// a <- b op c on little-endian 32-bit words (host)
inline void alu256(uint32_t op, const uint32_t b[8], const uint32_t c[8], uint32_t a[8]) {
    uint64_t carry = 0, borrow = 0;
    for (int i = 0; i < 8; i++) {
        switch (op) {
            case OP_ADD: carry += (uint64_t)b[i] + c[i], a[i] = (uint32_t)carry, carry >>= 32; break;
            case OP_SUB: {
                const uint64_t d = (uint64_t)b[i] - c[i] - borrow;
                a[i] = (uint32_t)d, borrow = (d >> 32) & 1u;
                break;
            }
            case OP_XOR: a[i] = b[i] ^ c[i]; break;
            case OP_OR: a[i] = b[i] | c[i]; break;
            default: a[i] = b[i] & c[i]; break;
        }
    }
}

}  // namespace int256
}  // namespace zkhip
