"""Shared by the Fp2 chip's CPU and GPU tests: records, the CPU twin of the device generator (Python integers: tests only), the AIR set with
its two lookup tables."""
import numpy as np

import zkvm_prover_amd as z
from zkvm_prover_amd import air

WIDTH, BITWISE_BUS, TUPLE_BUS, SX, SY = 648, 9, 6, 256, 2048
Q_LIMBS, N_POS, N_CARRY, CARRY_OFFSET, RECORD_WORDS = 33, 64, 63, 1 << 18, 33
COL_Q, COL_QS, COL_CX, COL_CY, COL_MARK, COL_DIFF, COL_MARK2, COL_DIFF2, COL_REAL = 192, 258, 260, 386, 512, 576, 578, 642, 644
NOPV = np.zeros(0, np.uint32)


def words(v, n=8):
    return [(int(v) >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def record(op, a, b, n=8):
    """the device record of a call: for a division (a, b) is (quotient, divisor); n = words per component (12 for a 48-limb modulus)"""
    return [op] + words(a[0], n) + words(a[1], n) + words(b[0], n) + words(b[1], n)


def cols(L):
    """column offsets of the chip for L limbs (include/zkhip_fp2.hpp `Cols`)"""
    ql, nc = L + 1, 2 * L - 1
    c = dict(L=L, Q_LIMBS=ql, N_POS=2 * L, N_CARRY=nc, Q=6 * L, QS=6 * L + 2 * ql)
    c["CX"] = c["QS"] + 2
    c["CY"] = c["CX"] + 2 * nc
    c["MARK"] = c["CY"] + 2 * nc
    c["DIFF"] = c["MARK"] + 2 * L
    c["MARK2"] = c["DIFF"] + 2
    c["DIFF2"] = c["MARK2"] + 2 * L
    c["REAL"] = c["DIFF2"] + 2
    c["WIDTH"] = c["REAL"] + 4
    return c


def twin_trace(calls, p, log_height):
    """calls: [(op, (a0, a1), (b0, b1))] as the RECORDS hold them -> (trace [WIDTH, N] canonical, bitwise range counts, tuple counts);
    32 limbs (648 columns) for a modulus below 2^256, 48 (968 columns) above"""
    NL = 32 if p < 1 << 256 else 48
    C = cols(NL)
    WIDTH, Q_LIMBS, N_POS, N_CARRY = C["WIDTH"], C["Q_LIMBS"], C["N_POS"], C["N_CARRY"]
    COL_Q, COL_QS, COL_CX, COL_CY, COL_MARK, COL_DIFF, COL_MARK2, COL_DIFF2, COL_REAL = (C[k] for k in ("Q", "QS", "CX", "CY", "MARK", "DIFF", "MARK2", "DIFF2", "REAL"))
    N = 1 << log_height
    tr = np.zeros((WIDTH, N), np.uint32)
    bw, tup = np.zeros(1 << 16, np.uint32), np.zeros(SX * SY, np.uint32)
    pb = p.to_bytes(NL, "little")
    for row, (op_in, a, b) in enumerate(calls):
        is_div = op_in == 3
        op = 0 if is_div else op_in
        if op == 0:
            v = [a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]]
        elif op == 1:
            v = [a[0] + b[0], a[1] + b[1]]
        else:
            v = [a[0] - b[0], a[1] - b[1]]
        r = [x % p for x in v]
        qs = [(x - y) // p for x, y in zip(v, r)]
        L = [x.to_bytes(NL, "little") for x in (a[0], a[1], b[0], b[1], r[0], r[1])]
        for o, limbs in enumerate(L):
            tr[NL * o:NL * o + NL, row] = np.frombuffer(limbs, dtype=np.uint8)
            for i in range(0, NL, 2):
                bw[limbs[i] * 256 + limbs[i + 1]] += 1
        A = (L[0], L[1])
        B = (L[2], L[3])
        R = (L[4], L[5])
        for e, q in enumerate(qs):
            ql = abs(q).to_bytes(Q_LIMBS, "little")
            neg = q < 0
            tr[COL_Q + e * Q_LIMBS:COL_Q + (e + 1) * Q_LIMBS, row] = np.frombuffer(ql, dtype=np.uint8)
            tr[COL_QS + e, row] = 1 if neg else 0
            for i in range(0, Q_LIMBS, 2):
                bw[ql[i] * 256 + (ql[i + 1] if i + 1 < Q_LIMBS else 0)] += 1
            c = 0
            for k in range(N_POS):
                s = c
                for i in range(Q_LIMBS):
                    j = k - i
                    if j < 0 or j >= NL:
                        continue
                    s -= (-1 if neg else 1) * ql[i] * pb[j]
                    if i >= NL or op != 0:
                        continue
                    s += A[0][i] * B[0][j] - A[1][i] * B[1][j] if e == 0 else A[0][i] * B[1][j] + A[1][i] * B[0][j]
                if k < NL:
                    if op == 1:
                        s += A[e][k] + B[e][k]
                    if op == 2:
                        s += A[e][k] - B[e][k]
                    s -= R[e][k]
                assert s % 256 == 0
                c = s // 256
                if k < N_CARRY:
                    val = c + CARRY_OFFSET
                    assert 0 <= val < SX * SY
                    tr[COL_CX + e * N_CARRY + k, row], tr[COL_CY + e * N_CARRY + k, row] = val & 255, val >> 8
                    tup[(val & 255) * SY + (val >> 8)] += 1
                else:
                    assert c == 0
        for which, (limbs2, mcol, dcol, on) in enumerate(((R, COL_MARK, COL_DIFF, True), (A, COL_MARK2, COL_DIFF2, is_div))):
            if not on:
                continue
            for e in range(2):
                x = limbs2[e]
                assert int.from_bytes(x, "little") < p
                mark = max(i for i in range(NL) if x[i] != pb[i])
                tr[mcol + NL * e + mark, row] = 1
                diff = pb[mark] - x[mark]
                tr[dcol + e, row] = diff
                bw[(diff - 1) * 256] += 1
        tr[COL_REAL, row], tr[COL_REAL + 1, row], tr[COL_REAL + 2, row], tr[COL_REAL + 3, row] = 1, op_in == 1, op_in == 2, is_div
    return tr, bw, tup


def instance(p, trace, bw, tup, log_height):
    program, width = z.fp2_air(p, BITWISE_BUS, TUPLE_BUS)
    bitwise = np.stack([bw, np.zeros(1 << 16, np.uint32)])
    return [dict(program=program, log_height=log_height, width=width, n_pvs=0, trace=trace, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8, BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, trace=bitwise, pvs=NOPV, prep=air.bitwise_lookup_prep(8)),
            dict(program=air.range_tuple_table_air(SX, SY, TUPLE_BUS).program(), log_height=19, width=1, n_pvs=0, trace=tup.reshape(1, -1), pvs=NOPV,
                 prep=air.range_tuple_prep(SX, SY))]
