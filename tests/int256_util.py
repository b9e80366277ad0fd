"""Shared by the 256-bit ALU chip's CPU and GPU tests: records, the oracle twin, the AIR set with the bitwise table."""
import ctypes as C

import numpy as np

import zkvm_prover_amd as z
from zkvm_prover_amd import air

WIDTH, BITWISE_BUS, MUL_WIDTH, TUPLE_BUS, SX, SY = 101, 9, 161, 6, 256, 32
NOPV = np.zeros(0, np.uint32)


def words(v):
    return [(int(v) >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def records(cases):
    """[(op, b, c)] -> [n, 17] words"""
    return np.array([[op] + words(b) + words(c) for op, b, c in cases], dtype=np.uint32).reshape(-1, 17)


def ora_alu(ora, op, b, c):
    l = ora.lib()
    l.ora_int256_alu.restype = None
    l.ora_int256_alu.argtypes = [C.c_uint32] + [C.POINTER(C.c_uint8)] * 3
    bb = np.frombuffer(int(b).to_bytes(32, "little"), dtype=np.uint8).copy()
    cb = np.frombuffer(int(c).to_bytes(32, "little"), dtype=np.uint8).copy()
    a = np.zeros(32, np.uint8)
    u8 = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint8))  # noqa: E731
    l.ora_int256_alu(op, u8(bb), u8(cb), u8(a))
    return int.from_bytes(a.tobytes(), "little")


def ora_trace(ora, cases, log_height):
    """(trace [101, N], xor counts [65536], bad)"""
    l = ora.lib()
    l.ora_int256_alu_trace.restype = C.c_size_t
    l.ora_int256_alu_trace.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    recs = np.ascontiguousarray(records(cases)) if len(cases) else np.zeros((0, 17), np.uint32)
    tr = np.zeros((WIDTH, 1 << log_height), np.uint32)
    xc = np.zeros(1 << 16, np.uint32)
    bad = l.ora_int256_alu_trace(ora.p32(recs), len(cases), log_height, ora.p32(tr), ora.p32(xc))
    return tr, xc, bad


def instance(trace, xc, log_height):
    program, width = z.int256_alu_air(BITWISE_BUS)
    bitwise = np.stack([np.zeros(1 << 16, np.uint32), xc])
    return [dict(program=program, log_height=log_height, width=width, n_pvs=0, trace=trace, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8, BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, trace=bitwise, pvs=NOPV, prep=air.bitwise_lookup_prep(8))]


def ora_mul_trace(ora, pairs, log_height):
    """(trace [161, N], bitwise range counts, tuple counts, bad) of the multiplication chip for [(b, c)]"""
    l = ora.lib()
    l.ora_mul256_trace.restype = C.c_size_t
    l.ora_mul256_trace.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32), C.c_uint32]
    recs = np.array([words(b) + words(c) for b, c in pairs], dtype=np.uint32).reshape(-1, 16)
    tr = np.zeros((MUL_WIDTH, 1 << log_height), np.uint32)
    bw, tup = np.zeros(1 << 16, np.uint32), np.zeros(SX * SY, np.uint32)
    bad = l.ora_mul256_trace(ora.p32(recs), 16, 0, len(pairs), log_height, ora.p32(tr), ora.p32(bw), ora.p32(tup), SY)
    return tr, bw, tup, bad


def mul_instance(trace, bw, tup, log_height):
    program, width = z.int256_mul_air(BITWISE_BUS, TUPLE_BUS)
    return [dict(program=program, log_height=log_height, width=width, n_pvs=0, trace=trace, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8, BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, trace=np.stack([bw, np.zeros(1 << 16, np.uint32)]), pvs=NOPV,
                 prep=air.bitwise_lookup_prep(8)),
            dict(program=air.range_tuple_table_air(SX, SY, TUPLE_BUS).program(), log_height=13, width=1, n_pvs=0, trace=tup.reshape(1, -1), pvs=NOPV,
                 prep=air.range_tuple_prep(SX, SY))]


CMP_WIDTH = 103


def cmp_result(op, b, c):
    """Python's answer: op 6 b < c unsigned, 7 b < c as two's-complement 256-bit integers, 8 b == c"""
    sgn = lambda v: v - (1 << 256) if v >> 255 else v  # noqa: E731
    return int(b < c) if op == 6 else int(sgn(b) < sgn(c)) if op == 7 else int(b == c)


def cmp_twin_trace(cases, log_height):
    """the tests' twin of zkhip_int256_cmp_tracegen for [(op, b, c)]: (trace [103, N] canonical, bitwise range counts [65536])"""
    P = 2013265921
    tr = np.zeros((CMP_WIDTH, 1 << log_height), np.uint32)
    bw = np.zeros(1 << 16, np.uint32)
    for row, (op, b, c) in enumerate(cases):
        bl, cl = list(int(b).to_bytes(32, "little")), list(int(c).to_bytes(32, "little"))
        tr[0:32, row], tr[32:64, row] = bl, cl
        for i in range(0, 32, 2):
            bw[bl[i] * 256 + bl[i + 1]] += 1
            bw[cl[i] * 256 + cl[i + 1]] += 1
        sb, sc = list(bl), list(cl)
        if op == 7:
            sb[31], sc[31] = (bl[31] - 256 if bl[31] >= 128 else bl[31]), (cl[31] - 256 if cl[31] >= 128 else cl[31])
        mark = next((i for i in range(31, -1, -1) if sb[i] != sc[i]), -1)
        lt = mark >= 0 and sb[mark] < sc[mark]
        if mark >= 0:
            diff = abs(sc[mark] - sb[mark])
            tr[65 + mark, row], tr[97, row] = 1, diff
            bw[(diff - 1) * 256] += 1
        tr[64, row] = int(lt)
        tr[98, row], tr[99, row] = sb[31] % P, sc[31] % P
        shift = 128 if op == 7 else 0
        bw[(sb[31] + shift) * 256 + (sc[31] + shift)] += 1
        tr[100 + (op - 6), row] = 1
    return tr, bw


def cmp_instance(trace, bw, log_height):
    program, width = z.int256_cmp_air(BITWISE_BUS)
    return [dict(program=program, log_height=log_height, width=width, n_pvs=0, trace=trace, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8, BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, trace=np.stack([bw, np.zeros(1 << 16, np.uint32)]), pvs=NOPV,
                 prep=air.bitwise_lookup_prep(8))]


SH_WIDTH = 189


def shift_result(op, b, c):
    """Python's answer: op 9 b << s, 10 b >> s, 11 the arithmetic shift of b as a two's-complement 256-bit integer, s = c mod 256"""
    s_, M = c % 256, 1 << 256
    if op == 9:
        return (b << s_) % M
    if op == 10:
        return b >> s_
    return ((b - M if b >> 255 else b) >> s_) % M


def shift_twin_trace(cases, log_height):
    """the tests' twin of zkhip_int256_shift_tracegen for [(op, b, c)]: (trace [189, N] canonical, bitwise range counts, xor counts)"""
    tr = np.zeros((SH_WIDTH, 1 << log_height), np.uint32)
    bw, xc = np.zeros(1 << 16, np.uint32), np.zeros(1 << 16, np.uint32)
    for row, (op, b, c) in enumerate(cases):
        bl, cl = list(int(b).to_bytes(32, "little")), list(int(c).to_bytes(32, "little"))
        amount = cl[0]
        bs, ls, mult = amount & 7, amount >> 3, 1 << (amount & 7)
        left = op == 9
        sign = bl[31] >> 7 if op == 11 else 0
        t, cy = [0] * 32, [0] * 32
        if left:
            carry = 0
            for k in range(32):
                v = bl[k] * mult + carry
                t[k], carry = v & 255, v >> 8
                cy[k] = carry
        else:
            inn = sign * (mult - 1)
            for m in range(31, -1, -1):
                v = bl[m] + 256 * inn
                t[31 - m], cy[m] = v >> bs, v & (mult - 1)
                inn = cy[m]
        for i in range(32):
            sel = t[i - ls] if i >= ls else (0 if left else 255 * sign)
            tr[(i if left else 31 - i), row] = sel
            tr[32 + i, row], tr[64 + i, row], tr[96 + i, row] = bl[i], t[i], cy[i]
            bw[cy[i] * 256 + (mult - 1 - cy[i])] += 1
        for i in range(0, 32, 2):
            bw[t[i] * 256 + t[i + 1]] += 1
            bw[bl[i] * 256 + bl[i + 1]] += 1
        tr[128, row], tr[129, row], tr[130, row] = cl[0], cl[1], cl[2] + 256 * cl[3]
        bw[cl[0] * 256 + cl[1]] += 1
        for k in range(1, 8):
            tr[131 + 2 * (k - 1), row], tr[132 + 2 * (k - 1), row] = cl[4 * k] + 256 * cl[4 * k + 1], cl[4 * k + 2] + 256 * cl[4 * k + 3]
        tr[145 + bs, row], tr[153 + ls, row], tr[185, row] = 1, 1, sign
        if op == 11:
            xc[bl[31] * 256 + 128] += 1
        tr[186 + (op - 9), row] = 1
    return tr, bw, xc


def shift_instance(trace, bw, xc, log_height):
    program, width = z.int256_shift_air(BITWISE_BUS)
    return [dict(program=program, log_height=log_height, width=width, n_pvs=0, trace=trace, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8, BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, trace=np.stack([bw, xc]), pvs=NOPV,
                 prep=air.bitwise_lookup_prep(8))]
