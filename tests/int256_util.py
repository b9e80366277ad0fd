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
