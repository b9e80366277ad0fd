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
