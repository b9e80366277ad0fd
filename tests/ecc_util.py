"""Shared by the elliptic-curve chip's CPU and GPU tests: records, the CPU twin of the device generator (Python integers: tests only),
the AIR set with its two lookup tables."""
import numpy as np

import zkvm_prover_amd as z
from zkvm_prover_amd import air

WIDTH, BITWISE_BUS, TUPLE_BUS, SX, SY = 772, 9, 6, 256, 2048
Q_LIMBS, N_POS, N_CARRY, CARRY_OFFSET, RECORD_WORDS = 33, 64, 63, 1 << 18, 41
COL_Q, COL_QS, COL_CX, COL_CY, COL_MARK, COL_DIFF, COL_REAL, COL_DBL = 224, 323, 326, 515, 704, 768, 770, 771
NOPV = np.zeros(0, np.uint32)


def words(v, n=8):
    return [(int(v) >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def record(op, p1, p2, slope, n=8):
    """n = words per coordinate: 8, or 12 for a modulus above 2^256 (48 limbs)"""
    return [op] + words(p1[0], n) + words(p1[1], n) + words(p2[0], n) + words(p2[1], n) + words(slope, n)


def cols(L):
    """column offsets of the chip for L limbs (include/zkhip_ecc.hpp `Cols`)"""
    ql, nc = L + 1, 2 * L - 1
    c = dict(L=L, Q_LIMBS=ql, N_POS=2 * L, N_CARRY=nc, Q=7 * L, QS=7 * L + 3 * ql)
    c["CX"] = c["QS"] + 3
    c["CY"] = c["CX"] + 3 * nc
    c["MARK"] = c["CY"] + 3 * nc
    c["DIFF"] = c["MARK"] + 2 * L
    c["REAL"], c["DBL"], c["WIDTH"] = c["DIFF"] + 2, c["DIFF"] + 3, c["DIFF"] + 4
    return c


def slope_of(op, p, a, p1, p2):
    if op == 1:
        return (3 * p1[0] * p1[0] + a) * pow(2 * p1[1], -1, p) % p
    return (p2[1] - p1[1]) * pow(p2[0] - p1[0], -1, p) % p


def twin_trace(calls, p, a, log_height):
    """calls: [(op, (x1, y1), (x2, y2), slope)] -> (trace [WIDTH, N] canonical, bitwise range counts [65536], tuple counts [SX * SY]);
    32 limbs (772 columns) for a modulus below 2^256, 48 (1156 columns) above"""
    NL = 32 if p < 1 << 256 else 48
    C = cols(NL)
    WIDTH, Q_LIMBS, N_POS, N_CARRY = C["WIDTH"], C["Q_LIMBS"], C["N_POS"], C["N_CARRY"]
    COL_Q, COL_QS, COL_CX, COL_CY, COL_MARK, COL_DIFF, COL_REAL, COL_DBL = C["Q"], C["QS"], C["CX"], C["CY"], C["MARK"], C["DIFF"], C["REAL"], C["DBL"]
    N = 1 << log_height
    tr = np.zeros((WIDTH, N), np.uint32)
    bw, tup = np.zeros(1 << 16, np.uint32), np.zeros(SX * SY, np.uint32)
    pb = p.to_bytes(NL, "little")
    ab = a.to_bytes(NL, "little")
    for row, (op, p1, p2, lam) in enumerate(calls):
        x1, y1, x2, y2 = p1[0], p1[1], p2[0], p2[1]
        dbl = op == 1
        v1 = 2 * lam * y1 - 3 * x1 * x1 - a if dbl else lam * (x2 - x1) - (y2 - y1)
        assert v1 % p == 0
        x3 = (lam * lam - x1 - (x1 if dbl else x2)) % p
        v2 = lam * lam - x1 - (x1 if dbl else x2) - x3
        y3 = (lam * (x1 - x3) - y1) % p
        v3 = lam * (x1 - x3) - y1 - y3
        qs = [v // p for v in (v1, v2, v3)]
        L = [v.to_bytes(NL, "little") for v in (x1, y1, x2, y2, lam, x3, y3)]
        for o, limbs in enumerate(L):
            tr[NL * o:NL * o + NL, row] = np.frombuffer(limbs, dtype=np.uint8)
            for i in range(0, NL, 2):
                bw[limbs[i] * 256 + limbs[i + 1]] += 1
        X1, Y1, X2, Y2, LM, X3, Y3 = L
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
                    if i >= NL:
                        continue
                    if e == 0:
                        s += 2 * LM[i] * Y1[j] - 3 * X1[i] * X1[j] if dbl else LM[i] * (X2[j] - X1[j])
                    elif e == 1:
                        s += LM[i] * LM[j]
                    else:
                        s += LM[i] * (X1[j] - X3[j])
                if k < NL:
                    if e == 0:
                        s -= ab[k] if dbl else Y2[k] - Y1[k]
                    elif e == 1:
                        s -= X1[k] + (X1[k] if dbl else X2[k]) + X3[k]
                    else:
                        s -= Y1[k] + Y3[k]
                assert s % 256 == 0
                c = s // 256
                if k < N_CARRY:
                    v = c + CARRY_OFFSET
                    assert 0 <= v < SX * SY
                    tr[COL_CX + e * N_CARRY + k, row], tr[COL_CY + e * N_CARRY + k, row] = v & 255, v >> 8
                    tup[(v & 255) * SY + (v >> 8)] += 1
                else:
                    assert c == 0
        for o, limbs in enumerate((X3, Y3)):
            mark = max(i for i in range(NL) if limbs[i] != pb[i])
            tr[COL_MARK + NL * o + mark, row] = 1
            diff = pb[mark] - limbs[mark]
            assert 1 <= diff <= 255
            tr[COL_DIFF + o, row] = diff
            bw[(diff - 1) * 256] += 1
        tr[COL_REAL, row], tr[COL_DBL, row] = 1, 1 if dbl else 0
    return tr, bw, tup


def instance(p, a, trace, bw, tup, log_height):
    """the chip with the two tables it looks up in (8-bit bitwise table, 256 x 2048 range-tuple table)"""
    program, width = z.ec_air(p, a, BITWISE_BUS, TUPLE_BUS)
    bitwise = np.stack([bw, np.zeros(1 << 16, np.uint32)])
    return [dict(program=program, log_height=log_height, width=width, n_pvs=0, trace=trace, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8, BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, trace=bitwise, pvs=NOPV, prep=air.bitwise_lookup_prep(8)),
            dict(program=air.range_tuple_table_air(SX, SY, TUPLE_BUS).program(), log_height=19, width=1, n_pvs=0, trace=tup.reshape(1, -1), pvs=NOPV,
                 prep=air.range_tuple_prep(SX, SY))]
