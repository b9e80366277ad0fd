// zkhip_chips.hpp -- the AIRs of the VM chips whose traces the device generates (include/zkhip.h: zkhip_*_tracegen), built with
// the C++ AIR builder (include/zkhip_air.hpp).  Each function emits WORD FOR WORD the program of its Python twin in
// zkvm-prover_amd/air.py (tests/test_air_builder_cpp.py compares them), so a proving key made from these programs is the key the
// Python-driven tests make.  What the chips are, and which OpenVM chip each follows, is documented at the Python definitions and
// at the generators in include/zkhip.h; here only the construction order matters: node numbers follow creation order, and C++
// leaves the evaluation order of a binary operator's operands unspecified, so every sub-expression is bound to a name in the
// order Python evaluates it.
#pragma once
#include "zkhip_air.hpp"

namespace zkhip {
namespace chips {
using air::AirBuilder;
using air::Expr;
using air::Kind;

constexpr uint32_t RANGE_BUS = 5, RANGE_TUPLE_BUS = 6, PROGRAM_BUS = 8, BITWISE_BUS = 9, MEMORY_BUS = 1;
constexpr int64_t INV_256 = 2005401601;  // 256^-1 mod p
constexpr int64_t INV_2 = 1006632961;     // 2^-1 mod p

// air.range_table_air(bus)
inline void range_table_air(AirBuilder& b, uint32_t bus = RANGE_BUS) {
    b.when_first_row(b.prep(0));
    const Expr p1 = b.prep(0, 1);
    const Expr p0 = b.prep(0);
    const Expr d = p1 - p0;
    b.when_transition(d - 1);
    const Expr f = b.prep(0);
    const Expr c = b.var(0);
    b.push_interaction(bus, {f}, c, Kind::Receive);
}

// air.range_tuple_table_air(size_x, size_y, bus): the sizes only shape the preprocessed trace, not the constraints
inline void range_tuple_table_air(AirBuilder& b, uint32_t bus = RANGE_TUPLE_BUS) {
    b.when_first_row(b.prep(0));
    b.when_first_row(b.prep(1));
    {
        const Expr y1 = b.prep(1, 1);
        const Expr y0 = b.prep(1);
        const Expr d = y1 - y0;
        const Expr d1 = d - 1;
        const Expr y1b = b.prep(1, 1);
        b.when_transition(d1 * y1b);
    }
    {
        const Expr x1 = b.prep(0, 1);
        const Expr x0 = b.prep(0);
        const Expr l = x1 - x0;
        const Expr x1b = b.prep(0, 1);
        const Expr x0b = b.prep(0);
        const Expr r0 = x1b - x0b;
        const Expr r = r0 - 1;
        b.when_transition(l * r);
    }
    const Expr f0 = b.prep(0);
    const Expr f1 = b.prep(1);
    const Expr c = b.var(0);
    b.push_interaction(bus, {f0, f1}, c, Kind::Receive);
}

// air.bitwise_lookup_air(bits, bus): AirBuilder(2, 0, prep_width = 3)
inline void bitwise_lookup_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    {   // (Python evaluates the count argument before push_interaction lifts the integer fields)
        const Expr x = b.prep(0);
        const Expr y = b.prep(1);
        const Expr c = b.var(0);
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {x, y, z0, z0}, c, Kind::Receive);
    }
    {
        const Expr x = b.prep(0);
        const Expr y = b.prep(1);
        const Expr z = b.prep(2);
        const Expr c = b.var(1);
        const Expr one = b.constant(1);
        b.push_interaction(bus, {x, y, z, one}, c, Kind::Receive);
    }
}

// air.program_air(bus): AirBuilder(10, 0) with set_cached_width(9)
inline void program_air(AirBuilder& b, uint32_t bus = PROGRAM_BUS) {
    std::vector<Expr> f;
    for (size_t c = 0; c < 9; c++) f.push_back(b.var(c));
    const Expr cnt = b.var(9);
    b.push_interaction(bus, f, cnt, Kind::Receive);
}

// air.exec_frame_air(bus): AirBuilder(10, 0)
inline void exec_frame_air(AirBuilder& b, uint32_t bus = PROGRAM_BUS) {
    const Expr ok = b.var(9);
    const Expr t = ok - 1;
    b.assert_zero(ok * t);
    std::vector<Expr> f;
    for (size_t c = 0; c < 9; c++) f.push_back(b.var(c));
    b.push_interaction(bus, f, ok, Kind::Send);
}

// air.rv32_alu_core_air(bus): AirBuilder(18, 0)
inline void rv32_alu_core_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    Expr a_[4], b_[4], c_[4], fl[5];
    for (int i = 0; i < 4; i++) a_[i] = b.var(i);
    for (int i = 0; i < 4; i++) b_[i] = b.var(4 + i);
    for (int i = 0; i < 4; i++) c_[i] = b.var(8 + i);
    for (int i = 0; i < 5; i++) fl[i] = b.var(12 + i);
    const Expr f_add = fl[0], f_sub = fl[1], f_xor = fl[2], f_or = fl[3], f_and = fl[4];
    const Expr ok = b.var(17);
    for (const Expr& f : {f_add, f_sub, f_xor, f_or, f_and, ok}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    {
        const Expr s1 = f_add + f_sub;
        const Expr s2 = s1 + f_xor;
        const Expr s3 = s2 + f_or;
        const Expr s4 = s3 + f_and;
        b.assert_zero(s4 - ok);
    }
    Expr carry_add, carry_sub;
    for (int i = 0; i < 4; i++) {
        {
            const Expr s = b_[i] + c_[i];
            const Expr d = s - a_[i];
            const Expr prev = i == 0 ? b.constant(0) : carry_add;
            const Expr t = d + prev;
            carry_add = t * INV_256;
        }
        {
            const Expr s = a_[i] + c_[i];
            const Expr d = s - b_[i];
            const Expr prev = i == 0 ? b.constant(0) : carry_sub;
            const Expr t = d + prev;
            carry_sub = t * INV_256;
        }
        {
            const Expr t = carry_add - 1;
            const Expr m = carry_add * t;
            b.assert_zero(f_add * m);
        }
        {
            const Expr t = carry_sub - 1;
            const Expr m = carry_sub * t;
            b.assert_zero(f_sub * m);
        }
    }
    const Expr bw0 = f_xor + f_or;
    const Expr bitwise = bw0 + f_and;
    for (int i = 0; i < 4; i++) {
        const Expr x1 = bitwise * b_[i];
        const Expr nb = 1 - bitwise;
        const Expr x2 = nb * a_[i];
        const Expr x = x1 + x2;
        const Expr y1 = bitwise * c_[i];
        const Expr nb2 = 1 - bitwise;
        const Expr y2 = nb2 * a_[i];
        const Expr y = y1 + y2;
        const Expr z1 = f_xor * a_[i];
        const Expr a2 = a_[i] * 2;
        const Expr o1 = a2 - b_[i];
        const Expr o2 = o1 - c_[i];
        const Expr z2 = f_or * o2;
        const Expr z12 = z1 + z2;
        const Expr bc = b_[i] + c_[i];
        const Expr a2b = a_[i] * 2;
        const Expr n1 = bc - a2b;
        const Expr z3 = f_and * n1;
        const Expr z = z12 + z3;
        const Expr one = b.constant(1);
        b.push_interaction(bus, {x, y, z, one}, ok, Kind::Send);
    }
}

// air.rv32_lt_core_air(bus): AirBuilder(18, 0)
inline void rv32_lt_core_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    Expr bl[4], cl[4], mk[4];
    for (int i = 0; i < 4; i++) bl[i] = b.var(i);
    for (int i = 0; i < 4; i++) cl[i] = b.var(4 + i);
    const Expr cmp = b.var(8), slt = b.var(9), sltu = b.var(10), bm = b.var(11), cm = b.var(12);
    for (int i = 0; i < 4; i++) mk[i] = b.var(13 + i);
    const Expr dv = b.var(17);
    const Expr ok = slt + sltu;
    for (const Expr& f : {slt, sltu, ok, cmp, mk[0], mk[1], mk[2], mk[3]}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    for (int k = 0; k < 2; k++) {
        const Expr limb = k == 0 ? bl[3] : cl[3], f = k == 0 ? bm : cm;
        const Expr d = limb - f;
        const Expr d256 = d - 256;
        b.assert_zero(d * d256);
        const Expr ns = 1 - slt;
        b.assert_zero(ns * d);
    }
    const Expr c2 = cmp * 2;
    const Expr sign = c2 - 1;
    Expr prefix;
    for (int i = 3; i >= 0; i--) {
        const Expr hi = i == 3 ? cm : cl[i], lo = i == 3 ? bm : bl[i];
        const Expr d0 = hi - lo;
        const Expr diff = d0 * sign;
        prefix = i == 3 ? mk[i] : prefix + mk[i];
        const Expr np = 1 - prefix;
        b.assert_zero(np * diff);
        const Expr dd = dv - diff;
        b.assert_zero(mk[i] * dd);
    }
    {
        const Expr t = prefix - 1;
        b.assert_zero(prefix * t);
    }
    {
        const Expr np = 1 - prefix;
        b.assert_zero(np * cmp);
    }
    {
        const Expr nok = 1 - ok;
        b.assert_zero(nok * prefix);
    }
    {
        const Expr s1 = slt * 128;
        const Expr x = bm + s1;
        const Expr s2 = slt * 128;
        const Expr y = cm + s2;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {x, y, z0, z0}, ok, Kind::Send);
    }
    {
        const Expr x = dv - 1;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {x, z0, z0, z0}, prefix, Kind::Send);
    }
}

// air.rv32_branch_eq_core_air(): AirBuilder(17, 0)
inline void rv32_branch_eq_core_air(AirBuilder& b) {
    Expr al[4], bl[4], mk[4];
    for (int i = 0; i < 4; i++) al[i] = b.var(i);
    for (int i = 0; i < 4; i++) bl[i] = b.var(4 + i);
    const Expr taken = b.var(8), imm = b.var(9), beq = b.var(10), bne = b.var(11);
    for (int i = 0; i < 4; i++) mk[i] = b.var(12 + i);
    const Expr inc = b.var(16);
    const Expr ok = beq + bne;
    for (const Expr& f : {beq, bne, ok, taken}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    const Expr e0 = taken * beq;
    const Expr nt = 1 - taken;
    const Expr e1 = nt * bne;
    const Expr eq = e0 + e1;
    Expr total = eq;
    for (int i = 0; i < 4; i++) {
        const Expr d = al[i] - bl[i];
        b.assert_zero(eq * d);
        const Expr dm = d * mk[i];
        total = total + dm;
    }
    {
        const Expr t = total - 1;
        b.assert_zero(ok * t);
    }
    {
        const Expr ti = taken * imm;
        const Expr x = inc - ti;
        const Expr nt2 = 1 - taken;
        const Expr f = nt2 * 4;
        const Expr y = x - f;
        b.assert_zero(ok * y);
    }
}

// air.rv32_branch_lt_core_air(bus): AirBuilder(23, 0)
inline void rv32_branch_lt_core_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    Expr al[4], bl[4], mk[4];
    for (int i = 0; i < 4; i++) al[i] = b.var(i);
    for (int i = 0; i < 4; i++) bl[i] = b.var(4 + i);
    const Expr cmp = b.var(8), taken = b.var(9), imm = b.var(10);
    const Expr blt = b.var(11), bltu = b.var(12), bge = b.var(13), bgeu = b.var(14);
    const Expr am = b.var(15), bm = b.var(16);
    for (int i = 0; i < 4; i++) mk[i] = b.var(17 + i);
    const Expr dv = b.var(21), inc = b.var(22);
    const Expr sgn = blt + bge;
    const Expr ge = bge + bgeu;
    const Expr ok0 = blt + bltu;
    const Expr ok1 = ok0 + bge;
    const Expr ok = ok1 + bgeu;
    for (const Expr& f : {blt, bltu, bge, bgeu, ok, cmp, taken, mk[0], mk[1], mk[2], mk[3]}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    {
        const Expr s = cmp + ge;
        const Expr cg = cmp * ge;
        const Expr cg2 = cg * 2;
        const Expr r = s - cg2;
        b.assert_zero(taken - r);
    }
    for (int k = 0; k < 2; k++) {
        const Expr limb = k == 0 ? al[3] : bl[3], f = k == 0 ? am : bm;
        const Expr d = limb - f;
        const Expr d256 = d - 256;
        b.assert_zero(d * d256);
        const Expr ns = 1 - sgn;
        b.assert_zero(ns * d);
    }
    const Expr c2 = cmp * 2;
    const Expr sign = c2 - 1;
    Expr prefix;
    for (int i = 3; i >= 0; i--) {
        const Expr hi = i == 3 ? bm : bl[i], lo = i == 3 ? am : al[i];
        const Expr d0 = hi - lo;
        const Expr diff = d0 * sign;
        prefix = i == 3 ? mk[i] : prefix + mk[i];
        const Expr np = 1 - prefix;
        b.assert_zero(np * diff);
        const Expr dd = dv - diff;
        b.assert_zero(mk[i] * dd);
    }
    {
        const Expr t = prefix - 1;
        b.assert_zero(prefix * t);
    }
    {
        const Expr np = 1 - prefix;
        b.assert_zero(np * cmp);
    }
    {
        const Expr nok = 1 - ok;
        b.assert_zero(nok * prefix);
    }
    {
        const Expr ti = taken * imm;
        const Expr x = inc - ti;
        const Expr nt = 1 - taken;
        const Expr f = nt * 4;
        const Expr y = x - f;
        b.assert_zero(ok * y);
    }
    {
        const Expr s1 = sgn * 128;
        const Expr x = am + s1;
        const Expr s2 = sgn * 128;
        const Expr y = bm + s2;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {x, y, z0, z0}, ok, Kind::Send);
    }
    {
        const Expr x = dv - 1;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {x, z0, z0, z0}, prefix, Kind::Send);
    }
}

// air.rv32_jal_lui_core_air(bus): AirBuilder(9, 0)
inline void rv32_jal_lui_core_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    const Expr pc = b.var(0), imm = b.var(1);
    Expr rd[4];
    for (int i = 0; i < 4; i++) rd[i] = b.var(2 + i);
    const Expr jal = b.var(6), lui = b.var(7), inc = b.var(8);
    const Expr ok = jal + lui;
    for (const Expr& f : {jal, lui, ok}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    b.assert_zero(lui * rd[0]);
    {
        const Expr t1 = rd[2] * 256;
        const Expr t2 = rd[1] + t1;
        const Expr t3 = rd[3] * 65536;
        const Expr t4 = t2 + t3;
        const Expr t5 = imm * 16;
        const Expr t6 = t4 - t5;
        b.assert_zero(lui * t6);
    }
    {
        const Expr u1 = rd[1] * 256;
        const Expr u2 = rd[0] + u1;
        const Expr u3 = rd[2] * 65536;
        const Expr u4 = u2 + u3;
        const Expr u5 = rd[3] * 16777216;
        const Expr u6 = u4 + u5;
        const Expr u7 = u6 - pc;
        const Expr u8 = u7 - 4;
        b.assert_zero(jal * u8);
    }
    {
        const Expr v1 = jal * imm;
        const Expr v2 = inc - v1;
        const Expr v3 = lui * 4;
        const Expr v4 = v2 - v3;
        b.assert_zero(ok * v4);
    }
    {
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {rd[0], rd[1], z0, z0}, ok, Kind::Send);
    }
    {
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {rd[2], rd[3], z0, z0}, ok, Kind::Send);
    }
    {
        const Expr x = rd[3] * 4;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {x, z0, z0, z0}, jal, Kind::Send);
    }
}

// air.rv32_auipc_core_air(bus): AirBuilder(14, 0)
inline void rv32_auipc_core_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    const Expr pc = b.var(0), imm = b.var(1);
    Expr pl[4], il[3], rd[4];
    for (int i = 0; i < 4; i++) pl[i] = b.var(2 + i);
    for (int i = 0; i < 3; i++) il[i] = b.var(6 + i);
    for (int i = 0; i < 4; i++) rd[i] = b.var(9 + i);
    const Expr ok = b.var(13);
    {
        const Expr t = ok - 1;
        b.assert_zero(ok * t);
    }
    {
        const Expr u1 = pl[1] * 256;
        const Expr u2 = pl[0] + u1;
        const Expr u3 = pl[2] * 65536;
        const Expr u4 = u2 + u3;
        const Expr u5 = pl[3] * 16777216;
        const Expr u6 = u4 + u5;
        const Expr u7 = u6 - pc;
        b.assert_zero(ok * u7);
    }
    {
        const Expr u1 = il[1] * 256;
        const Expr u2 = il[0] + u1;
        const Expr u3 = il[2] * 65536;
        const Expr u4 = u2 + u3;
        const Expr u5 = imm * 16;
        const Expr u6 = u4 - u5;
        b.assert_zero(ok * u6);
    }
    {
        const Expr d = rd[0] - pl[0];
        b.assert_zero(ok * d);
    }
    Expr carry;
    for (int i = 1; i < 4; i++) {
        const Expr s = pl[i] + il[i - 1];
        const Expr d = s - rd[i];
        const Expr prev = i == 1 ? b.constant(0) : carry;
        const Expr t = d + prev;
        carry = t * INV_256;
        const Expr c1 = carry - 1;
        const Expr m = carry * c1;
        b.assert_zero(ok * m);
    }
    const Expr xs[5] = {pl[0], pl[2], il[0], il[2], rd[2]}, ys[5] = {pl[1], pl[3], il[1], rd[1], rd[3]};
    for (int k = 0; k < 5; k++) {
        const Expr y = k == 1 ? ys[k] * 4 : ys[k];   // the top pc limb below 2^6
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {xs[k], y, z0, z0}, ok, Kind::Send);
    }
}

// air.rv32_jalr_core_air(bus): AirBuilder(20, 0)
inline void rv32_jalr_core_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    const Expr pc = b.var(0), imm = b.var(1);
    const Expr il0 = b.var(2), il1 = b.var(3), sign = b.var(4);
    Expr rs[4], rd[4], t[4];
    for (int i = 0; i < 4; i++) rs[i] = b.var(5 + i);
    for (int i = 0; i < 4; i++) rd[i] = b.var(9 + i);
    for (int i = 0; i < 4; i++) t[i] = b.var(13 + i);
    const Expr lsb = b.var(17), to_pc = b.var(18), ok = b.var(19);
    for (const Expr& f : {ok, sign, lsb}) {
        const Expr m = f - 1;
        b.assert_zero(f * m);
    }
    {
        const Expr u1 = il1 * 256;
        const Expr u2 = il0 + u1;
        const Expr u3 = u2 - imm;
        b.assert_zero(ok * u3);
    }
    Expr ext[4];
    ext[0] = il0;
    {
        const Expr s240 = sign * 240;
        ext[1] = il1 + s240;
    }
    ext[2] = sign * 255;
    ext[3] = sign * 255;
    Expr carry;
    for (int i = 0; i < 4; i++) {
        const Expr s = rs[i] + ext[i];
        const Expr d = s - t[i];
        const Expr prev = i == 0 ? b.constant(0) : carry;
        const Expr e = d + prev;
        carry = e * INV_256;
        const Expr c1 = carry - 1;
        const Expr m = carry * c1;
        b.assert_zero(ok * m);
    }
    {
        const Expr u1 = t[1] * 256;
        const Expr u2 = t[0] + u1;
        const Expr u3 = t[2] * 65536;
        const Expr u4 = u2 + u3;
        const Expr u5 = t[3] * 16777216;
        const Expr u6 = u4 + u5;
        const Expr u7 = u6 - lsb;
        const Expr u8 = u7 - to_pc;
        b.assert_zero(ok * u8);
    }
    {
        const Expr u1 = rd[1] * 256;
        const Expr u2 = rd[0] + u1;
        const Expr u3 = rd[2] * 65536;
        const Expr u4 = u2 + u3;
        const Expr u5 = rd[3] * 16777216;
        const Expr u6 = u4 + u5;
        const Expr u7 = u6 - pc;
        const Expr u8 = u7 - 4;
        b.assert_zero(ok * u8);
    }
    {
        const Expr s8 = sign * 8;
        const Expr d = il1 - s8;
        const Expr y = d * 32;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {il0, y, z0, z0}, ok, Kind::Send);
    }
    {
        const Expr d = t[0] - lsb;
        const Expr x = d * INV_2;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {x, t[1], z0, z0}, ok, Kind::Send);
    }
    {
        const Expr y = t[3] * 4;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {t[2], y, z0, z0}, ok, Kind::Send);
    }
    {
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {rd[0], rd[1], z0, z0}, ok, Kind::Send);
    }
    {
        const Expr y = rd[3] * 4;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {rd[2], y, z0, z0}, ok, Kind::Send);
    }
}

// air.rv32_shift_core_air(bus): AirBuilder(32, 0)
inline void rv32_shift_core_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    Expr a_[4], b_[4], bm[8], lm[4], cy[4];
    for (int i = 0; i < 4; i++) a_[i] = b.var(i);
    for (int i = 0; i < 4; i++) b_[i] = b.var(4 + i);
    const Expr c0 = b.var(8), sll = b.var(9), srl = b.var(10), sra = b.var(11);
    for (int i = 0; i < 8; i++) bm[i] = b.var(12 + i);
    for (int i = 0; i < 4; i++) lm[i] = b.var(20 + i);
    for (int i = 0; i < 4; i++) cy[i] = b.var(24 + i);
    const Expr sign = b.var(28), q = b.var(29), ml = b.var(30), mr = b.var(31);
    const Expr ok0 = sll + srl;
    const Expr ok = ok0 + sra;
    const Expr right = srl + sra;
    {
        std::vector<Expr> fs{sll, srl, sra, ok, sign};
        for (int i = 0; i < 8; i++) fs.push_back(bm[i]);
        for (int i = 0; i < 4; i++) fs.push_back(lm[i]);
        for (const Expr& f : fs) {
            const Expr t = f - 1;
            b.assert_zero(f * t);
        }
    }
    Expr sbm = bm[0], slm = lm[0], mult = bm[0], bs, ls;
    for (int i = 1; i < 8; i++) {
        sbm = sbm + bm[i];
        const Expr m2 = bm[i] * (int64_t)(1 << i);
        mult = mult + m2;
        const Expr mi = bm[i] * (int64_t)i;
        bs = i == 1 ? mi : bs + mi;
    }
    for (int j = 1; j < 4; j++) {
        slm = slm + lm[j];
        const Expr mj = lm[j] * (int64_t)j;
        ls = j == 1 ? mj : ls + mj;
    }
    b.assert_zero(sbm - ok);
    b.assert_zero(slm - ok);
    {
        const Expr t1 = c0 - bs;
        const Expr l8 = ls * 8;
        const Expr t2 = t1 - l8;
        const Expr q32 = q * 32;
        b.assert_zero(t2 - q32);
    }
    {
        const Expr m = sll * mult;
        b.assert_zero(ml - m);
    }
    {
        const Expr m = right * mult;
        b.assert_zero(mr - m);
    }
    {
        const Expr ns = 1 - sra;
        b.assert_zero(sign * ns);
    }
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            if (i < j) {
                const Expr m = a_[i] * sll;
                b.assert_zero(lm[j] * m);
            } else {
                const int k = i - j;
                const Expr e1 = b_[k] * ml;
                const Expr e2 = cy[k] * sll;
                const Expr e3 = e2 * 256;
                Expr exp = e1 - e3;
                if (k > 0) {
                    const Expr e4 = cy[k - 1] * sll;
                    exp = exp + e4;
                }
                const Expr as = a_[i] * sll;
                const Expr d = as - exp;
                b.assert_zero(lm[j] * d);
            }
            if (i + j > 3) {
                const Expr ar = a_[i] * right;
                const Expr sr = sign * right;
                const Expr s255 = sr * 255;
                const Expr d = ar - s255;
                b.assert_zero(lm[j] * d);
            } else {
                const int k = i + j;
                Expr nxt;
                if (k == 3) {
                    const Expr t = mr - right;
                    nxt = sign * t;
                } else {
                    nxt = cy[k + 1] * right;
                }
                const Expr am = a_[i] * mr;
                const Expr n256 = nxt * 256;
                const Expr d1 = am - n256;
                const Expr bc = b_[k] - cy[k];
                const Expr br = bc * right;
                const Expr d2 = d1 - br;
                b.assert_zero(lm[j] * d2);
            }
        }
    for (int i = 0; i < 4; i++) {
        const Expr m = ml + mr;
        const Expr m1 = m - 1;
        const Expr y = m1 - cy[i];
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {cy[i], y, z0, z0}, ok, Kind::Send);
    }
    {
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {a_[0], a_[1], z0, z0}, ok, Kind::Send);
        b.push_interaction(bus, {a_[2], a_[3], z0, z0}, ok, Kind::Send);
        const Expr q32 = q * 32;
        b.push_interaction(bus, {q, q32, z0, z0}, ok, Kind::Send);
    }
    {
        const Expr s1 = b_[3] + 128;
        const Expr s256 = sign * 256;
        const Expr z = s1 - s256;
        const Expr c128 = b.constant(128);
        const Expr one = b.constant(1);
        b.push_interaction(bus, {b_[3], c128, z, one}, sra, Kind::Send);
    }
}

// air.rv32_mul_core_air(bus): AirBuilder(13, 0)
inline void rv32_mul_core_air(AirBuilder& b, uint32_t bus = RANGE_TUPLE_BUS) {
    Expr a_[4], b_[4], c_[4];
    for (int i = 0; i < 4; i++) a_[i] = b.var(i);
    for (int i = 0; i < 4; i++) b_[i] = b.var(4 + i);
    for (int i = 0; i < 4; i++) c_[i] = b.var(8 + i);
    const Expr ok = b.var(12);
    {
        const Expr t = ok - 1;
        b.assert_zero(ok * t);
    }
    Expr carry;
    for (int i = 0; i < 4; i++) {
        // acc = carry (or 0); for k: acc = b_k * c_{i-k} + acc
        Expr acc;
        bool have = i > 0;
        if (have) acc = carry;
        for (int k = 0; k <= i; k++) {
            const Expr m = b_[k] * c_[i - k];
            acc = have ? m + acc : m + 0;
            have = true;
        }
        const Expr d = acc - a_[i];
        carry = d * INV_256;
        b.push_interaction(bus, {a_[i], carry}, ok, Kind::Send);
    }
}

// air.rv32_mulh_core_air(tuple_bus, bitwise_bus): AirBuilder(21, 0)
inline void rv32_mulh_core_air(AirBuilder& b, uint32_t tuple_bus = RANGE_TUPLE_BUS, uint32_t bitwise_bus = BITWISE_BUS) {
    Expr a_[4], b_[4], c_[4], am[4];
    for (int i = 0; i < 4; i++) a_[i] = b.var(i);
    for (int i = 0; i < 4; i++) b_[i] = b.var(4 + i);
    for (int i = 0; i < 4; i++) c_[i] = b.var(8 + i);
    for (int i = 0; i < 4; i++) am[i] = b.var(12 + i);
    const Expr bs = b.var(16), cs = b.var(17), mulh = b.var(18), mulhsu = b.var(19), mulhu = b.var(20);
    const Expr ok0 = mulh + mulhsu;
    const Expr ok = ok0 + mulhu;
    for (const Expr& f : {mulh, mulhsu, mulhu, ok, bs, cs}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    b.assert_zero(mulhu * bs);
    {
        const Expr u = mulhu + mulhsu;
        b.assert_zero(u * cs);
    }
    const Expr b_ext = bs * 255;
    const Expr c_ext = cs * 255;
    Expr carry;
    for (int i = 0; i < 4; i++) {
        Expr acc;
        bool have = i > 0;
        if (have) acc = carry;
        for (int k = 0; k <= i; k++) {
            const Expr m = b_[k] * c_[i - k];
            acc = have ? m + acc : m + 0;
            have = true;
        }
        const Expr d = acc - am[i];
        carry = d * INV_256;
        b.push_interaction(tuple_bus, {am[i], carry}, ok, Kind::Send);
    }
    for (int j = 0; j < 4; j++) {
        Expr acc = carry;
        for (int k = j + 1; k < 4; k++) {
            const Expr m = b_[k] * c_[4 + j - k];
            acc = m + acc;
        }
        for (int k = 0; k <= j; k++) {
            const Expr m1 = b_[k] * c_ext;
            const Expr m2 = c_[k] * b_ext;
            const Expr m = m1 + m2;
            acc = m + acc;
        }
        const Expr d = acc - a_[j];
        carry = d * INV_256;
        b.push_interaction(tuple_bus, {a_[j], carry}, ok, Kind::Send);
    }
    {
        const Expr s = bs * 128;
        const Expr d = b_[3] - s;
        const Expr x = d * 2;
        const Expr z0 = b.constant(0);
        const Expr cnt = mulh + mulhsu;
        b.push_interaction(bitwise_bus, {x, z0, z0, z0}, cnt, Kind::Send);
    }
    {
        const Expr s = cs * 128;
        const Expr d = c_[3] - s;
        const Expr x = d * 2;
        const Expr z0 = b.constant(0);
        b.push_interaction(bitwise_bus, {x, z0, z0, z0}, mulh, Kind::Send);
    }
}

// air.mmcs_path_air(hash_bus, claims_bus): AirBuilder(39, 0) -- in-circuit verification of mixed-height Merkle openings
inline void mmcs_path_air(AirBuilder& b, uint32_t hash_bus, uint32_t claims_bus) {
    Expr root[8], par[8], a_[8], b_[8], n_root[8], n_par[8];
    for (int i = 0; i < 8; i++) root[i] = b.var(i);
    for (int i = 0; i < 8; i++) par[i] = b.var(8 + i);
    for (int i = 0; i < 8; i++) a_[i] = b.var(16 + i);
    for (int i = 0; i < 8; i++) b_[i] = b.var(24 + i);
    const Expr bit = b.var(32), inj = b.var(33), first = b.var(34), last = b.var(35), real = b.var(36), idx = b.var(37), lvl = b.var(38);
    for (int i = 0; i < 8; i++) n_root[i] = b.next(i);
    for (int i = 0; i < 8; i++) n_par[i] = b.next(8 + i);
    const Expr n_bit = b.next(32), n_inj = b.next(33), n_first = b.next(34), n_real = b.next(36), n_idx = b.next(37), n_lvl = b.next(38);
    for (const Expr& f : {bit, inj, first, last, real}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    b.assert_zero(inj * bit);
    {
        const Expr nr = 1 - real;
        b.assert_zero(first * nr);
    }
    {
        const Expr nr = 1 - real;
        b.assert_zero(last * nr);
    }
    b.assert_zero(last * inj);
    const Expr link = real - last;
    b.when_first_row(real - first);
    b.when_last_row(link);
    {
        const Expr nn = 1 - n_real;
        b.when_transition(link * nn);
    }
    b.when_transition(link * n_first);
    {
        const Expr m = last * n_real;
        const Expr nf = 1 - n_first;
        b.when_transition(m * nf);
    }
    {
        const Expr nr = 1 - real;
        b.when_transition(nr * n_real);
    }
    for (int i = 0; i < 8; i++) {
        {
            const Expr d = par[i] - root[i];
            b.assert_zero(first * d);
        }
        {
            const Expr d = n_root[i] - root[i];
            b.when_transition(link * d);
        }
        {
            const Expr t1 = n_par[i] - a_[i];
            const Expr t2 = b_[i] - a_[i];
            const Expr t3 = bit * t2;
            const Expr t4 = t1 - t3;
            b.when_transition(link * t4);
        }
    }
    {
        const Expr d = idx - bit;
        b.assert_zero(first * d);
    }
    {
        const Expr d = lvl - 1;
        const Expr e = d + inj;
        b.assert_zero(first * e);
    }
    {
        const Expr w = 2 - n_inj;
        const Expr m = idx * w;
        const Expr d = n_idx - m;
        const Expr e = d - n_bit;
        b.when_transition(link * e);
    }
    {
        const Expr d = n_lvl - lvl;
        const Expr e = d - 1;
        const Expr f = e + n_inj;
        b.when_transition(link * f);
    }
    {
        std::vector<Expr> msg;
        for (int i = 0; i < 8; i++) msg.push_back(a_[i]);
        for (int i = 0; i < 8; i++) msg.push_back(b_[i]);
        for (int i = 0; i < 8; i++) msg.push_back(par[i]);
        b.push_interaction(hash_bus, msg, real, Kind::Send);
    }
    {
        std::vector<Expr> msg(root, root + 8);
        msg.push_back(lvl), msg.push_back(idx);
        for (int i = 0; i < 8; i++) msg.push_back(b_[i]);
        b.push_interaction(claims_bus, msg, inj, Kind::Send);
    }
    {
        std::vector<Expr> msg(root, root + 8);
        msg.push_back(lvl), msg.push_back(idx);
        for (int i = 0; i < 8; i++) {
            const Expr t = b_[i] - a_[i];
            const Expr m = bit * t;
            msg.push_back(a_[i] + m);
        }
        b.push_interaction(claims_bus, msg, last, Kind::Send);
    }
}

// air.var_range_table_air(bus): AirBuilder(1, 0, prep_width = 2) -- OpenVM VariableRangeCheckerChip
inline void var_range_table_air(AirBuilder& b, uint32_t bus = 7) {
    const Expr v = b.prep(0), bits = b.prep(1);
    b.push_interaction(bus, {v, bits}, b.var(0), Kind::Receive);
}

// air.castf_air(bus): AirBuilder(6, 0) -- native CASTF: x to limbs of 8, 8, 8, 6 bits through the variable range checker
inline void castf_air(AirBuilder& b, uint32_t bus = 7) {
    const Expr x = b.var(0);
    Expr limb[4];
    for (int i = 0; i < 4; i++) limb[i] = b.var(1 + i);
    const Expr ok = b.var(5);
    {
        const Expr t = ok - 1;
        b.assert_zero(ok * t);
    }
    {
        const Expr u1 = limb[1] * 256;
        const Expr u2 = limb[0] + u1;
        const Expr u3 = limb[2] * 65536;
        const Expr u4 = u2 + u3;
        const Expr u5 = limb[3] * 16777216;
        const Expr u6 = u4 + u5;
        const Expr u7 = u6 - x;
        b.assert_zero(ok * u7);
    }
    for (int i = 0; i < 4; i++) {
        const Expr bits = b.constant(i < 3 ? 8 : 6);
        b.push_interaction(bus, {limb[i], bits}, ok, Kind::Send);
    }
}

// air.duplex_air(hash_bus, io_bus): AirBuilder(50, 0) -- the DuplexChallenger in-circuit, one row per duplexing
inline void duplex_air(AirBuilder& b, uint32_t hash_bus, uint32_t io_bus) {
    Expr st_in[16], st_out[16], f[8], s_[8], n_in[16], n_f[8];
    for (int i = 0; i < 16; i++) st_in[i] = b.var(i);
    for (int i = 0; i < 16; i++) st_out[i] = b.var(16 + i);
    for (int i = 0; i < 8; i++) f[i] = b.var(32 + i);
    for (int i = 0; i < 8; i++) s_[i] = b.var(40 + i);
    const Expr seq = b.var(48), real = b.var(49);
    for (int i = 0; i < 16; i++) n_in[i] = b.next(i);
    for (int i = 0; i < 8; i++) n_f[i] = b.next(32 + i);
    const Expr n_seq = b.next(48), n_real = b.next(49);
    for (int i = 0; i < 17; i++) {
        const Expr x = i < 8 ? f[i] : i < 16 ? s_[i - 8] : real;
        const Expr t = x - 1;
        b.assert_zero(x * t);
    }
    for (int j = 0; j < 8; j++) {
        {
            const Expr nr = 1 - real;
            b.assert_zero(nr * f[j]);
        }
        {
            const Expr nr = 1 - real;
            b.assert_zero(nr * s_[j]);
        }
    }
    for (int j = 0; j < 7; j++) {
        {
            const Expr nf = 1 - f[j];
            b.assert_zero(f[j + 1] * nf);
        }
        {
            const Expr ns = 1 - s_[j + 1];
            b.assert_zero(s_[j] * ns);
        }
    }
    {
        const Expr nr = 1 - real;
        b.when_transition(nr * n_real);
    }
    b.when_first_row(seq);
    {
        const Expr d = n_seq - seq;
        const Expr e = d - 1;
        b.when_transition(n_real * e);
    }
    for (int j = 0; j < 16; j++) {
        if (j < 8) {
            {
                const Expr nf = 1 - f[j];
                b.when_first_row(nf * st_in[j]);
            }
            {
                const Expr nf = 1 - n_f[j];
                const Expr m = n_real * nf;
                const Expr d = n_in[j] - st_out[j];
                b.when_transition(m * d);
            }
        } else {
            b.when_first_row(st_in[j]);
            const Expr d = n_in[j] - st_out[j];
            b.when_transition(n_real * d);
        }
    }
    {
        std::vector<Expr> msg(st_in, st_in + 16);
        msg.insert(msg.end(), st_out, st_out + 16);
        b.push_interaction(hash_bus, msg, real, Kind::Send);
    }
    for (int j = 0; j < 8; j++) {
        const Expr lane = b.constant(j);
        const Expr kind = b.constant(0);
        b.push_interaction(io_bus, {seq, lane, st_in[j], kind}, f[j], Kind::Send);
    }
    for (int j = 0; j < 8; j++) {
        const Expr lane = b.constant(j);
        const Expr kind = b.constant(1);
        b.push_interaction(io_bus, {seq, lane, st_out[j], kind}, s_[j], Kind::Send);
    }
}

// air.duplex_io_air(io_bus): AirBuilder(5, 0)
inline void duplex_io_air(AirBuilder& b, uint32_t io_bus) {
    std::vector<Expr> msg;
    for (int i = 0; i < 4; i++) msg.push_back(b.var(i));
    b.push_interaction(io_bus, msg, b.var(4), Kind::Receive);
}

// air._ext_mul_exprs(x, y): coordinates of x * y in F[X] / (X^4 - 11), node for node as the Python helper builds them
inline void ext_mul_exprs(const Expr x[4], const Expr y[4], Expr out[4]) {
    {
        const Expr t0 = x[0] * y[0];
        const Expr t1 = x[1] * y[3];
        const Expr t2 = x[2] * y[2];
        const Expr s1 = t1 + t2;
        const Expr t3 = x[3] * y[1];
        const Expr s2 = s1 + t3;
        const Expr w = s2 * 11;
        out[0] = t0 + w;
    }
    {
        const Expr t0 = x[0] * y[1];
        const Expr t1 = x[1] * y[0];
        const Expr s0 = t0 + t1;
        const Expr t2 = x[2] * y[3];
        const Expr t3 = x[3] * y[2];
        const Expr s1 = t2 + t3;
        const Expr w = s1 * 11;
        out[1] = s0 + w;
    }
    {
        const Expr t0 = x[0] * y[2];
        const Expr t1 = x[1] * y[1];
        const Expr s0 = t0 + t1;
        const Expr t2 = x[2] * y[0];
        const Expr s1 = s0 + t2;
        const Expr t3 = x[3] * y[3];
        const Expr w = t3 * 11;
        out[2] = s1 + w;
    }
    {
        const Expr t0 = x[0] * y[3];
        const Expr t1 = x[1] * y[2];
        const Expr s0 = t0 + t1;
        const Expr t2 = x[2] * y[1];
        const Expr s1 = s0 + t2;
        const Expr t3 = x[3] * y[0];
        out[3] = s1 + t3;
    }
}

// air.field_arith_air(): AirBuilder(8, 0) -- native base-field ADD / SUB / MUL / DIV
inline void field_arith_air(AirBuilder& b) {
    const Expr a_ = b.var(0), b_ = b.var(1), c_ = b.var(2), add = b.var(3), sub = b.var(4), mul = b.var(5), div = b.var(6), inv = b.var(7);
    const Expr ok0 = add + sub;
    const Expr ok1 = ok0 + mul;
    const Expr ok = ok1 + div;
    for (const Expr& f : {add, sub, mul, div, ok}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    {
        const Expr d = a_ - b_;
        const Expr e = d - c_;
        b.assert_zero(add * e);
    }
    {
        const Expr d = a_ - b_;
        const Expr e = d + c_;
        b.assert_zero(sub * e);
    }
    {
        const Expr m = b_ * c_;
        const Expr e = a_ - m;
        b.assert_zero(mul * e);
    }
    {
        const Expr m = a_ * c_;
        const Expr e = b_ - m;
        b.assert_zero(div * e);
    }
    {
        const Expr m = c_ * inv;
        const Expr e = m - 1;
        b.assert_zero(div * e);
    }
}

// air.field_ext_air(): AirBuilder(20, 0) -- native extension-field ADD / SUB / MUL / DIV
inline void field_ext_air(AirBuilder& b) {
    Expr x[4], y[4], z[4], inv[4], xy[4], xi[4], yi[4];
    for (int i = 0; i < 4; i++) x[i] = b.var(i);
    for (int i = 0; i < 4; i++) y[i] = b.var(4 + i);
    for (int i = 0; i < 4; i++) z[i] = b.var(8 + i);
    const Expr add = b.var(12), sub = b.var(13), mul = b.var(14), div = b.var(15);
    for (int i = 0; i < 4; i++) inv[i] = b.var(16 + i);
    const Expr ok0 = add + sub;
    const Expr ok1 = ok0 + mul;
    const Expr ok = ok1 + div;
    for (const Expr& f : {add, sub, mul, div, ok}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    ext_mul_exprs(x, y, xy);
    ext_mul_exprs(x, inv, xi);
    ext_mul_exprs(y, inv, yi);
    for (int i = 0; i < 4; i++) {
        {
            const Expr d = z[i] - x[i];
            const Expr e = d - y[i];
            b.assert_zero(add * e);
        }
        {
            const Expr d = z[i] - x[i];
            const Expr e = d + y[i];
            b.assert_zero(sub * e);
        }
        {
            const Expr e = z[i] - xy[i];
            b.assert_zero(mul * e);
        }
        {
            const Expr e = z[i] - xi[i];
            b.assert_zero(div * e);
        }
        {
            const Expr e = yi[i] - (i == 0 ? 1 : 0);
            b.assert_zero(div * e);
        }
    }
}

// air.fri_fold_air(): AirBuilder(18, 0) -- one arity-2 FRI folding step per row
inline void fri_fold_air(AirBuilder& b, int point_bus = -1) {   // AirBuilder(19, 0); with point_bus the row sends (k, x_inv) there
    Expr e0[4], e1[4], beta[4], folded[4], d[4];
    for (int i = 0; i < 4; i++) e0[i] = b.var(i);
    for (int i = 0; i < 4; i++) e1[i] = b.var(4 + i);
    for (int i = 0; i < 4; i++) beta[i] = b.var(8 + i);
    const Expr xinv = b.var(12);
    for (int i = 0; i < 4; i++) folded[i] = b.var(13 + i);
    const Expr real = b.var(17), k = b.var(18);
    {
        const Expr t = real - 1;
        b.assert_zero(real * t);
    }
    for (int i = 0; i < 4; i++) d[i] = e0[i] - e1[i];
    auto mul = [&](int i, int j) { return beta[i] * d[j]; };
    Expr prod[4];
    {
        const Expr t0 = mul(0, 0);
        const Expr t1 = mul(1, 3);
        const Expr t2 = mul(2, 2);
        const Expr s1 = t1 + t2;
        const Expr t3 = mul(3, 1);
        const Expr s2 = s1 + t3;
        const Expr w = s2 * 11;
        prod[0] = t0 + w;
    }
    {
        const Expr t0 = mul(0, 1);
        const Expr t1 = mul(1, 0);
        const Expr s0 = t0 + t1;
        const Expr t2 = mul(2, 3);
        const Expr t3 = mul(3, 2);
        const Expr s1 = t2 + t3;
        const Expr w = s1 * 11;
        prod[1] = s0 + w;
    }
    {
        const Expr t0 = mul(0, 2);
        const Expr t1 = mul(1, 1);
        const Expr s0 = t0 + t1;
        const Expr t2 = mul(2, 0);
        const Expr s1 = s0 + t2;
        const Expr t3 = mul(3, 3);
        const Expr w = t3 * 11;
        prod[2] = s1 + w;
    }
    {
        const Expr t0 = mul(0, 3);
        const Expr t1 = mul(1, 2);
        const Expr s0 = t0 + t1;
        const Expr t2 = mul(2, 1);
        const Expr s1 = s0 + t2;
        const Expr t3 = mul(3, 0);
        prod[3] = s1 + t3;
    }
    for (int i = 0; i < 4; i++) {
        const Expr f2 = folded[i] * 2;
        const Expr a = f2 - e0[i];
        const Expr c = a - e1[i];
        const Expr xp = xinv * prod[i];
        b.assert_zero(c - xp);
    }
    if (point_bus >= 0) b.push_interaction((uint32_t)point_bus, {k, xinv}, real, Kind::Send);
}

// air.domain_point_air(point_bus): AirBuilder(54, 0) -- x^-1 of a FRI pair from the bits of its index
inline void domain_point_air(AirBuilder& b, uint32_t point_bus) {
    constexpr int NB = 26;
    constexpr uint64_t P = zkhip::air::P;
    // W_j^-1, W_j = the generator of the subgroup of order 2^(j + 2) (air.domain_point_inverse_roots)
    auto powmod = [](uint64_t base, uint64_t e) {
        uint64_t r = 1;
        for (base %= P; e; e >>= 1, base = base * base % P)
            if (e & 1) r = r * base % P;
        return r;
    };
    const Expr k = b.var(0);
    Expr bit[NB], acc[NB];
    for (int j = 0; j < NB; j++) bit[j] = b.var(1 + j);
    for (int j = 0; j < NB; j++) acc[j] = b.var(1 + NB + j);
    const Expr mult = b.var(1 + 2 * NB);
    Expr total;
    for (int j = 0; j < NB; j++) {
        {
            const Expr t = bit[j] - 1;
            b.assert_zero(bit[j] * t);
        }
        const Expr term = bit[j] * (int64_t)(1u << j);
        total = j == 0 ? term : total + term;
        const int64_t winv = (int64_t)powmod(powmod(0x1A427A41ull, 1ull << (27 - (j + 2))), P - 2);
        const Expr f0 = bit[j] * (winv - 1);
        const Expr factor = f0 + 1;
        if (j == 0) {
            b.assert_zero(acc[j] - factor);
        } else {
            const Expr m = acc[j - 1] * factor;
            b.assert_zero(acc[j] - m);
        }
    }
    b.assert_zero(k - total);
    b.push_interaction(point_bus, {k, acc[NB - 1]}, mult, Kind::Receive);
}

// air.mmcs_claims_air(claims_bus): AirBuilder(19, 0)
inline void mmcs_claims_air(AirBuilder& b, uint32_t claims_bus) {
    std::vector<Expr> msg;
    for (int i = 0; i < 18; i++) msg.push_back(b.var(i));
    b.push_interaction(claims_bus, msg, b.var(18), Kind::Receive);
}

// air.rv32_divrem_core_air(tuple_bus, bitwise_bus): AirBuilder(41, 0)
inline void rv32_divrem_core_air(AirBuilder& b, uint32_t tuple_bus = RANGE_TUPLE_BUS, uint32_t bitwise_bus = BITWISE_BUS) {
    Expr bl[4], cl[4], ql[4], rl[4], ca[4], ra[4], mk[4];
    for (int i = 0; i < 4; i++) bl[i] = b.var(i);
    for (int i = 0; i < 4; i++) cl[i] = b.var(4 + i);
    for (int i = 0; i < 4; i++) ql[i] = b.var(8 + i);
    for (int i = 0; i < 4; i++) rl[i] = b.var(12 + i);
    for (int i = 0; i < 4; i++) ca[i] = b.var(16 + i);
    for (int i = 0; i < 4; i++) ra[i] = b.var(20 + i);
    const Expr b_sign = b.var(24), c_sign = b.var(25), q_sign = b.var(26), r_sign = b.var(27);
    const Expr kc = b.var(28), kr = b.var(29), zd = b.var(30), cinv = b.var(31);
    for (int i = 0; i < 4; i++) mk[i] = b.var(32 + i);
    const Expr diff = b.var(36);
    const Expr div = b.var(37), divu = b.var(38), rem = b.var(39), remu = b.var(40);
    const Expr ok0 = div + divu;
    const Expr ok1 = ok0 + rem;
    const Expr ok = ok1 + remu;
    const Expr sgn = div + rem;
    for (const Expr& f : {div, divu, rem, remu, ok, b_sign, c_sign, q_sign, r_sign, kc, kr, zd, mk[0], mk[1], mk[2], mk[3]}) {
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    for (const Expr& f : {b_sign, c_sign, q_sign, r_sign}) {
        const Expr ns = 1 - sgn;
        b.assert_zero(ns * f);
    }
    {
        const Expr nok = 1 - ok;
        b.assert_zero(nok * zd);
    }
    const Expr cs0 = cl[0] + cl[1];
    const Expr cs1 = cs0 + cl[2];
    const Expr csum = cs1 + cl[3];
    for (int i = 0; i < 4; i++) {
        b.assert_zero(zd * cl[i]);
        const Expr d = ql[i] - 255;
        b.assert_zero(zd * d);
    }
    {
        const Expr nz = ok - zd;
        const Expr pr = csum * cinv;
        const Expr m1 = pr - 1;
        b.assert_zero(nz * m1);
    }
    const Expr b_ext = b_sign * 255;
    const Expr c_ext = c_sign * 255;
    const Expr q_ext = q_sign * 255;
    const Expr r_ext = r_sign * 255;
    Expr carry;
    for (int i = 0; i < 4; i++) {
        Expr acc;
        bool have = i > 0;
        if (have) acc = carry;
        for (int k = 0; k <= i; k++) {
            const Expr m = cl[k] * ql[i - k];
            acc = have ? m + acc : m + 0;
            have = true;
        }
        const Expr s = acc + rl[i];
        const Expr d = s - bl[i];
        carry = d * INV_256;
        b.push_interaction(tuple_bus, {ql[i], carry}, ok, Kind::Send);
    }
    for (int j = 0; j < 4; j++) {
        Expr acc = carry;
        for (int k = j + 1; k < 4; k++) {
            const Expr m = cl[k] * ql[4 + j - k];
            acc = m + acc;
        }
        for (int k = 0; k <= j; k++) {
            const Expr m1 = cl[k] * q_ext;
            const Expr m2 = ql[k] * c_ext;
            const Expr m = m1 + m2;
            acc = m + acc;
        }
        const Expr s = acc + r_ext;
        const Expr d = s - b_ext;
        carry = d * INV_256;
        b.push_interaction(tuple_bus, {rl[j], carry}, ok, Kind::Send);
    }
    for (int g = 0; g < 2; g++) {
        const Expr* x = g == 0 ? cl : rl;
        const Expr* xa = g == 0 ? ca : ra;
        const Expr sg = g == 0 ? c_sign : r_sign, k = g == 0 ? kc : kr;
        for (int i = 0; i < 4; i++) {
            const Expr ns = 1 - sg;
            const Expr d = x[i] - xa[i];
            b.assert_zero(ns * d);
        }
        {
            const Expr t1 = x[1] * 256;
            const Expr t2 = x[0] + t1;
            const Expr t3 = t2 + xa[0];
            const Expr t4 = xa[1] * 256;
            const Expr t5 = t3 + t4;
            const Expr t6 = k * 65536;
            const Expr t7 = t5 - t6;
            b.assert_zero(sg * t7);
        }
        {
            const Expr t1 = x[3] * 256;
            const Expr t2 = x[2] + t1;
            const Expr t3 = t2 + xa[2];
            const Expr t4 = xa[3] * 256;
            const Expr t5 = t3 + t4;
            const Expr t6 = t5 + k;
            const Expr t7 = t6 - 65536;
            b.assert_zero(sg * t7);
        }
    }
    {
        const Expr nb = 1 - b_sign;
        b.assert_zero(r_sign * nb);
    }
    for (int i = 0; i < 4; i++) {
        const Expr nr = 1 - r_sign;
        const Expr m = b_sign * nr;
        b.assert_zero(m * rl[i]);
    }
    Expr prefix;
    for (int i = 3; i >= 0; i--) {
        const Expr d = ca[i] - ra[i];
        prefix = i == 3 ? mk[i] : prefix + mk[i];
        const Expr nz = 1 - zd;
        const Expr np = nz - prefix;
        b.assert_zero(np * d);
        const Expr dd = diff - d;
        b.assert_zero(mk[i] * dd);
    }
    {
        const Expr nz = ok - zd;
        b.assert_zero(prefix - nz);
    }
    {
        const Expr s1 = b_sign * 128;
        const Expr d1 = bl[3] - s1;
        const Expr x = d1 * 2;
        const Expr s2 = c_sign * 128;
        const Expr d2 = cl[3] - s2;
        const Expr y = d2 * 2;
        const Expr z0 = b.constant(0);
        b.push_interaction(bitwise_bus, {x, y, z0, z0}, sgn, Kind::Send);
    }
    const Expr xs[4] = {ca[0], ca[2], ra[0], ra[2]}, ys[4] = {ca[1], ca[3], ra[1], ra[3]};
    for (int k = 0; k < 4; k++) {
        const Expr z0 = b.constant(0);
        b.push_interaction(bitwise_bus, {xs[k], ys[k], z0, z0}, ok, Kind::Send);
    }
    {
        const Expr x = diff - 1;
        const Expr z0 = b.constant(0);
        b.push_interaction(bitwise_bus, {x, z0, z0, z0}, prefix, Kind::Send);
    }
}

// air.rv32_loadstore_core_air(bus): AirBuilder(33, 0); cases in the order of air.RV32_LOADSTORE_CASES
inline void rv32_loadstore_core_air(AirBuilder& b, uint32_t bus = BITWISE_BUS) {
    enum Kind_ { LW, LHU, LBU, SW, SH, SB, LH, LB };
    static const struct { Kind_ kind; int s; } CASES[20] = {{LW, 0}, {LHU, 0}, {LHU, 2}, {LBU, 0}, {LBU, 1}, {LBU, 2}, {LBU, 3}, {SW, 0}, {SH, 0}, {SH, 2},
                                                           {SB, 0}, {SB, 1}, {SB, 2}, {SB, 3}, {LH, 0}, {LH, 2}, {LB, 0}, {LB, 1}, {LB, 2}, {LB, 3}};
    Expr rd[4], pv[4], wr[4], fl[20];
    for (int i = 0; i < 4; i++) rd[i] = b.var(i);
    for (int i = 0; i < 4; i++) pv[i] = b.var(4 + i);
    for (int i = 0; i < 4; i++) wr[i] = b.var(8 + i);
    for (int i = 0; i < 20; i++) fl[i] = b.var(12 + i);
    const Expr sign = b.var(32);
    Expr ok = fl[0];
    for (int i = 1; i < 20; i++) ok = ok + fl[i];
    for (int i = 0; i < 22; i++) {
        const Expr f = i < 20 ? fl[i] : i == 20 ? ok : sign;
        const Expr t = f - 1;
        b.assert_zero(f * t);
    }
    Expr sgn = fl[14];
    for (int i = 15; i < 20; i++) sgn = sgn + fl[i];
    {
        const Expr ns = 1 - sgn;
        b.assert_zero(sign * ns);
    }
    const Expr ext = sign * 255;
    for (int i = 0; i < 4; i++) {
        Expr acc;
        bool have = false;
        for (int c = 0; c < 20; c++) {
            const Kind_ kind = CASES[c].kind;
            const int s = CASES[c].s;
            Expr t;
            bool term = true;
            switch (kind) {
                case LW: case SW: t = rd[i]; break;
                case LHU: case LH:
                    if (i < 2) t = rd[s + i];
                    else if (kind == LH) t = ext;
                    else term = false;
                    break;
                case LBU: case LB:
                    if (i == 0) t = rd[s];
                    else if (kind == LB) t = ext;
                    else term = false;
                    break;
                case SH: t = (s <= i && i < s + 2) ? rd[i - s] : pv[i]; break;
                default: t = i == s ? rd[0] : pv[i]; break;
            }
            if (!term) continue;
            const Expr m = fl[c] * t;
            acc = have ? acc + m : m;
            have = true;
        }
        b.assert_zero(wr[i] - acc);
    }
    Expr top;
    for (int c = 14; c < 20; c++) {
        const Expr t = fl[c] * rd[CASES[c].kind == LH ? CASES[c].s + 1 : CASES[c].s];
        top = c == 14 ? t : top + t;
    }
    {
        const Expr s128 = sign * 128;
        const Expr d = top - s128;
        const Expr x = d * 2;
        const Expr z0 = b.constant(0);
        b.push_interaction(bus, {x, z0, z0, z0}, sgn, Kind::Send);
    }
}

// air.memory_access_air(range_bus, memory_bus): AirBuilder(10, 0)
inline void memory_access_air(AirBuilder& b, uint32_t range_bus = RANGE_BUS, uint32_t memory_bus = MEMORY_BUS) {
    Expr v[10];
    for (int i = 0; i < 10; i++) v[i] = b.var(i);
    const Expr as_ = v[0], ptr = v[1], pd = v[2], pts = v[3], d = v[4], ts = v[5], rd = v[6], ok = v[7], lo = v[8], hi = v[9];
    {
        const Expr t = ok - 1;
        b.assert_zero(ok * t);
    }
    {
        const Expr t = rd - 1;
        b.assert_zero(rd * t);
    }
    {
        const Expr nok = 1 - ok;
        b.assert_zero(nok * rd);
    }
    {
        const Expr t = d - pd;
        b.assert_zero(rd * t);
    }
    {
        const Expr t1 = ts - pts;
        const Expr t2 = t1 - 1;
        const Expr t3 = t2 - lo;
        const Expr h = hi * (int64_t)(1 << 16);
        const Expr t4 = t3 - h;
        b.assert_zero(ok * t4);
    }
    b.push_interaction(range_bus, {lo}, ok, Kind::Send);
    b.push_interaction(range_bus, {hi}, ok, Kind::Send);
    {
        const Expr h8 = hi * 8;   // gap_hi < 2^13: the gap cannot stand for a negative difference modulo p
        b.push_interaction(range_bus, {h8}, ok, Kind::Send);
    }
    b.push_interaction(range_bus, {d}, ok, Kind::Send);
    b.push_interaction(memory_bus, {as_, ptr, pd, pts}, ok, Kind::Receive);
    b.push_interaction(memory_bus, {as_, ptr, d, ts}, ok, Kind::Send);
}

// air.memory_boundary_air(pointer_bits, range_bus, memory_bus): AirBuilder(8, 0)
inline void memory_boundary_air(AirBuilder& b, unsigned pointer_bits = 27, uint32_t range_bus = RANGE_BUS, uint32_t memory_bus = MEMORY_BUS) {
    Expr v[8];
    for (int i = 0; i < 8; i++) v[i] = b.var(i);
    const Expr as_ = v[0], ptr = v[1], init = v[2], fin = v[3], ts = v[4], ok = v[5], lo = v[6], hi = v[7];
    const Expr ok_n = b.var(5, 1);
    {
        const Expr t = ok - 1;
        b.assert_zero(ok * t);
    }
    {
        const Expr nok = 1 - ok;
        b.when_transition(ok_n * nok);
    }
    const Expr k1 = as_ * (int64_t)(1ll << pointer_bits);
    const Expr key = k1 + ptr;
    const Expr a1 = b.var(0, 1);
    const Expr kn1 = a1 * (int64_t)(1ll << pointer_bits);
    const Expr p1 = b.var(1, 1);
    const Expr key_n = kn1 + p1;
    {
        const Expr t1 = key_n - key;
        const Expr t2 = t1 - 1;
        const Expr t3 = t2 - lo;
        const Expr h = hi * (int64_t)(1 << 16);
        const Expr t4 = t3 - h;
        b.when_transition(ok_n * t4);
    }
    b.push_interaction(range_bus, {lo}, ok, Kind::Send);
    b.push_interaction(range_bus, {hi}, ok, Kind::Send);
    {
        const Expr h8 = hi * 8;   // keys and gaps below 2^29: two rows cannot carry one key
        b.push_interaction(range_bus, {h8}, ok, Kind::Send);
    }
    {
        const Expr z0 = b.constant(0);
        b.push_interaction(memory_bus, {as_, ptr, init, z0}, ok, Kind::Send);
    }
    b.push_interaction(memory_bus, {as_, ptr, fin, ts}, ok, Kind::Receive);
}

}  // namespace chips
}  // namespace zkhip
