"""Shared by the modular-multiplication chip's CPU and GPU tests: records, the oracle twin, the AIR set with its two lookup tables."""
import ctypes as C

import numpy as np

import zkvm_prover_amd as z
from zkvm_prover_amd import air

WIDTH, BITWISE_BUS, TUPLE_BUS, SX, SY = 325, 9, 6, 256, 128
NOPV = np.zeros(0, np.uint32)


def to_bytes(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)


def records_bytes(pairs):
    """[(a, b)] -> [n, 64] bytes (what the oracle takes); .view('<u4') gives the device records"""
    return np.stack([np.concatenate([to_bytes(a), to_bytes(b)]) for a, b in pairs])


def ora_modmul(ora, a, b, p):
    l = ora.lib()
    l.ora_modmul.restype = C.c_int
    l.ora_modmul.argtypes = [C.POINTER(C.c_uint8)] * 5
    q, r = np.zeros(32, np.uint8), np.zeros(32, np.uint8)
    u8 = lambda x: np.ascontiguousarray(x).ctypes.data_as(C.POINTER(C.c_uint8))  # noqa: E731
    ab, bb, pb = to_bytes(a).copy(), to_bytes(b).copy(), to_bytes(p).copy()
    rc = l.ora_modmul(u8(ab), u8(bb), u8(pb), u8(q), u8(r))
    return rc, int.from_bytes(q.tobytes(), "little"), int.from_bytes(r.tobytes(), "little")


def ora_addsub(ora, op, a, b, p):
    l = ora.lib()
    l.ora_modaddsub.restype = C.c_int
    l.ora_modaddsub.argtypes = [C.c_uint] + [C.POINTER(C.c_uint8)] * 5
    q, r = np.zeros(32, np.uint8), np.zeros(32, np.uint8)
    u8 = lambda x: np.ascontiguousarray(x).ctypes.data_as(C.POINTER(C.c_uint8))  # noqa: E731
    ab, bb, pb = to_bytes(a).copy(), to_bytes(b).copy(), to_bytes(p).copy()
    rc = l.ora_modaddsub(op, u8(ab), u8(bb), u8(pb), u8(q), u8(r))
    return rc, int.from_bytes(q.tobytes(), "little"), int.from_bytes(r.tobytes(), "little")


def ora_trace(ora, pairs, p, log_height, ops=None):
    """(trace [288, N], bitwise range counts [65536], tuple counts [SX * SY], bad); ops: per pair 0 mul (default), 1 add, 2 sub"""
    l = ora.lib()
    l.ora_modular_trace.restype = C.c_size_t
    l.ora_modular_trace.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint8), C.c_uint, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]
    recs = np.ascontiguousarray(records_bytes(pairs)) if len(pairs) else np.zeros((0, 64), np.uint8)
    pb = to_bytes(p).copy()
    tr = np.zeros((WIDTH, 1 << log_height), np.uint32)
    bw, tup = np.zeros(1 << 16, np.uint32), np.zeros(SX * SY, np.uint32)
    opv = None if ops is None else np.ascontiguousarray(ops, dtype=np.uint32)
    bad = l.ora_modular_trace(recs.ctypes.data_as(C.POINTER(C.c_uint8)), None if opv is None else ora.p32(opv), len(pairs), pb.ctypes.data_as(C.POINTER(C.c_uint8)),
                              log_height, ora.p32(tr), ora.p32(bw), ora.p32(tup), SY)
    return tr, bw, tup, bad


def instance(p, trace, bw, tup, log_height):
    """the chip with the two tables it looks up in (8-bit bitwise table, 256 x 128 range-tuple table)"""
    program, width = z.modmul_air(p, BITWISE_BUS, TUPLE_BUS)
    bitwise = np.stack([bw, np.zeros(1 << 16, np.uint32)])
    return [dict(program=program, log_height=log_height, width=width, n_pvs=0, trace=trace, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8, BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, trace=bitwise, pvs=NOPV, prep=air.bitwise_lookup_prep(8)),
            dict(program=air.range_tuple_table_air(SX, SY, TUPLE_BUS).program(), log_height=15, width=1, n_pvs=0, trace=tup.reshape(1, -1), pvs=NOPV,
                 prep=air.range_tuple_prep(SX, SY))]


def cols(L):
    """column offsets of the modular chip for L limbs (include/zkhip_modular.hpp `Cols`)"""
    nc = 2 * L - 2
    c = dict(L=L, N_CARRY=nc, A=0, B=L, Q=2 * L, R=3 * L, CX=4 * L, CY=4 * L + nc, MARK=4 * L + 2 * nc)
    c["DIFF"] = c["MARK"] + L
    c["REAL"], c["IS_ADD"], c["IS_SUB"], c["IS_DIV"] = c["DIFF"] + 1, c["DIFF"] + 2, c["DIFF"] + 3, c["DIFF"] + 4
    c["MARK2"] = c["IS_DIV"] + 1
    c["DIFF2"] = c["MARK2"] + L
    c["IS_EQ"], c["EQ"], c["INV"], c["WIDTH"] = c["DIFF2"] + 1, c["DIFF2"] + 2, c["DIFF2"] + 3, c["DIFF2"] + 4
    return c


def py_trace(rows, p, log_height):
    """A Python twin of the modular chip's columns for ANY limb count (32 below 2^256, 48 above), written from the header's description
    of the columns, with Python's integers: rows = [(op, a, b)], op 0 mul, 1 add, 2 sub, 3 div (a / b), 4 is_eq.
    Returns (trace [WIDTH, N] canonical, bitwise range counts [65536], tuple counts [SX * SY])."""
    P_BB = 2013265921
    L = 32 if p < 1 << 256 else 48
    c = cols(L)
    N = 1 << log_height
    tr = np.zeros((c["WIDTH"], N), np.uint32)
    bw, tup = np.zeros(1 << 16, np.uint32), np.zeros(SX * SY, np.uint32)
    pb = list(int(p).to_bytes(L, "little"))
    for row, (op_in, a, b) in enumerate(rows):
        is_div, is_eq = op_in == 3, op_in == 4
        if is_div:                      # the row is the product (a / b) b = a: the quotient in the a columns
            a = a * pow(b, -1, p) % p
        op = 0 if is_div else 2 if is_eq else op_in
        if op == 0:
            q, r = divmod(a * b, p)
            sign = 1
        elif op == 1:
            q, r = divmod(a + b, p)
            sign = 1
        else:
            r = (a - b) % p
            q = (r - (a - b)) // p      # a - b + q p = r
            sign = -1
        ab, bb, qb, rb = (list(int(v).to_bytes(L, "little")) for v in (a, b, q, r))
        for i in range(L):
            tr[c["A"] + i, row], tr[c["B"] + i, row], tr[c["Q"] + i, row], tr[c["R"] + i, row] = ab[i], bb[i], qb[i], rb[i]
        for x in (ab, bb, qb, rb):
            for i in range(0, L, 2):
                bw[x[i] * 256 + x[i + 1]] += 1
        carry = 0
        for k in range(2 * L - 1):
            s = carry
            for i in range(L):
                j = k - i
                if 0 <= j < L:
                    if op == 0:
                        s += ab[i] * bb[j]
                    s -= sign * qb[i] * pb[j]
            if k < L:
                s += (ab[k] + bb[k]) if op == 1 else (ab[k] - bb[k]) if op == 2 else 0
                s -= rb[k]
            assert s % 256 == 0
            carry = s // 256
            if k < c["N_CARRY"]:
                v = carry + (1 << 14)
                assert 0 <= v < 256 * SY
                tr[c["CX"] + k, row], tr[c["CY"] + k, row] = v & 255, v >> 8
                tup[(v & 255) * SY + (v >> 8)] += 1
            else:
                assert carry == 0
        mark = max(i for i in range(L) if rb[i] != pb[i])
        tr[c["MARK"] + mark, row] = 1
        diff = pb[mark] - rb[mark]
        tr[c["DIFF"], row], tr[c["REAL"], row], tr[c["IS_ADD"], row], tr[c["IS_SUB"], row] = diff, 1, int(op == 1), int(op == 2)
        bw[((diff - 1) & 255) * 256] += 1
        if is_div:
            mark2 = max(i for i in range(L) if ab[i] != pb[i])
            assert ab[mark2] < pb[mark2]
            tr[c["MARK2"] + mark2, row] = 1
            tr[c["DIFF2"], row] = pb[mark2] - ab[mark2]
            bw[((pb[mark2] - ab[mark2] - 1) & 255) * 256] += 1
        tr[c["IS_DIV"], row] = int(is_div)
        limb_sum = sum(rb)
        eq = is_eq and limb_sum == 0
        tr[c["IS_EQ"], row], tr[c["EQ"], row] = int(is_eq), int(eq)
        tr[c["INV"], row] = pow(limb_sum, -1, P_BB) if is_eq and not eq else 0
    return tr, bw, tup
