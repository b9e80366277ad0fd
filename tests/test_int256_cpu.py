"""CPU: the 256-bit ALU chip (include/zkhip_int256.hpp) -- parity anchored OUTSIDE this repository.
  * the oracle's byte-wise ALU (oracle/int256.c) and the product's host function (zkhip_int256_alu_host) against Python's integers
    (tests/golden/int256_kat.json: add, sub, xor, or, and modulo 2^256 incl. the carry chain's edge cases);
  * the chip's trace (oracle twin of zkhip_int256_alu_tracegen) with the bitwise table satisfies the AIRs and balances the bus; a
    changed cell breaks a constraint or the bus; the oracle proves the set and both verifiers accept."""
import json
import os

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import int256_util as iu
import vm2_util as v2

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = (1, 0, 4, 3, 3)


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "int256_kat.json")) as f:
        return [(c["op"], int(c["b"], 16), int(c["c"], 16), int(c["a"], 16)) for c in json.load(f)["cases"]]


def test_alu_against_python_integers(ora, kat):
    M = 1 << 256
    for op, b, c, a in kat:
        assert a == [(b + c) % M, (b - c) % M, b ^ c, b | c, b & c, (b * c) % M][op]
        assert z.int256_alu_host(op, b, c) == a
        if op < 5:
            assert iu.ora_alu(ora, op, b, c) == a


def test_trace_satisfies_the_air_and_the_bus_balances(ora, kat):
    cases = [(op, b, c) for op, b, c, _ in kat[::3] if op < 5][:60]
    tr, xc, bad = iu.ora_trace(ora, cases, 6)
    assert bad == 0
    inst = iu.instance(tr, xc, 6)
    for d in inst:
        assert air.quotient_chunks(d["program"]) <= 2
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    want = {(op, b, c): a for op, b, c, a in kat}
    for row, (op, b, c) in enumerate(cases):
        assert bytes(tr[0:32, row].astype(np.uint8)) == want[(op, b, c)].to_bytes(32, "little")
    rng = np.random.default_rng(3)
    for _ in range(60):
        t2 = tr.copy()
        col, r_ = int(rng.integers(0, iu.WIDTH)), int(rng.integers(0, len(cases)))
        t2[col, r_] = (int(t2[col, r_]) + 1) % 2013265921
        i2 = iu.instance(t2, xc, 6)
        assert air.check_trace(i2[0]["program"], t2, iu.NOPV) != [] or v2.bus_imbalance(i2) != {}, (col, r_)


def test_multiplication_chip(ora, kat):
    """the 256-bit multiplication chip: the oracle twin's products are Python's low 256 bits, the trace with its two tables satisfies
    the AIRs and balances both buses, a changed cell is caught, the oracle proves the set and both verifiers accept"""
    pairs = [(b, c) for op, b, c, _ in kat if op == 5][:14]
    tr, bw, tup, bad = iu.ora_mul_trace(ora, pairs, 4)
    assert bad == 0
    for row, (b, c) in enumerate(pairs):
        assert bytes(tr[0:32, row].astype(np.uint8)) == ((b * c) % (1 << 256)).to_bytes(32, "little")
    inst = iu.mul_instance(tr, bw, tup, 4)
    for d in inst:
        assert air.quotient_chunks(d["program"]) <= 2
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    rng = np.random.default_rng(4)
    for _ in range(40):
        t2 = tr.copy()
        col, r_ = int(rng.integers(0, iu.MUL_WIDTH)), int(rng.integers(0, len(pairs)))
        t2[col, r_] = (int(t2[col, r_]) + 1) % 2013265921
        i2 = iu.mul_instance(t2, bw, tup, 4)
        assert air.check_trace(i2[0]["program"], t2, iu.NOPV) != [] or v2.bus_imbalance(i2) != {}, (col, r_)
    proof = ora.stark_prove(PARAMS, inst)
    assert ora.stark_verify(PARAMS, inst, proof) == 0
    vk = []
    for d in inst:
        v = {k: d[k] for k in ("program", "log_height", "width", "n_pvs")}
        if d.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(PARAMS, d)
        vk.append(v)
    assert z.verify(PARAMS, vk, [iu.NOPV] * 3, proof.tobytes()) == 0


@pytest.fixture(scope="module")
def cmp_kat():
    with open(os.path.join(HERE, "golden", "int256_kat.json")) as f:
        return [(c["op"], int(c["b"], 16), int(c["c"], 16), int(c["a"], 16)) for c in json.load(f)["cmp"]]


def test_comparison_chip(cmp_kat):
    """Rv32LessThan256 / equality as a chip of its own: the host function against Python's integers (unsigned, signed, equal; operands
    that differ in one bit, in the sign only, not at all); the tests' twin trace with the bitwise table satisfies the AIRs and balances the
    bus; a flipped answer, a marker on a lower differing limb and an equality claimed for different operands have no trace."""
    cases = cmp_kat
    for op, b, c, a in cases:
        assert iu.cmp_result(op, b, c) == a and z.int256_alu_host(op, b, c) == a
    sel = cases[:13] + cases[49:62] + cases[98:111] + cases[20:30] + cases[70:80] + cases[120:125]
    tr, bw = iu.cmp_twin_trace([(op, b, c) for op, b, c, _ in sel], 6)
    inst = iu.cmp_instance(tr, bw, 6)
    for d in inst:
        assert air.quotient_chunks(d["program"]) <= 2
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    program = inst[0]["program"]
    # the answer flipped (t, for a less-than row with different operands)
    k = next(i for i, (op, b, c, a) in enumerate(sel) if op == 6 and b != c)
    t2 = tr.copy()
    t2[64, k] ^= 1
    i2 = iu.cmp_instance(t2, bw, 6)
    assert air.check_trace(program, t2, iu.NOPV) != [] or v2.bus_imbalance(i2) != {}
    # the marker moved to a lower differing limb
    k = next(i for i, (op, b, c, a) in enumerate(sel) if bin(b ^ c).count("1") > 40)
    marked = int(np.nonzero(tr[65:97, k])[0][0])
    lower = next(i for i in range(marked - 1, -1, -1) if tr[i, k] != tr[32 + i, k])
    t3 = tr.copy()
    t3[65 + marked, k], t3[65 + lower, k] = 0, 1
    t3[97, k] = abs(int(tr[32 + lower, k]) - int(tr[lower, k]))
    assert air.check_trace(program, t3, iu.NOPV) != []
    # "equal" claimed for different operands: the marker removed
    k = next(i for i, (op, b, c, a) in enumerate(sel) if op == 8 and b != c)
    t4 = tr.copy()
    t4[65:97, k], t4[97, k], t4[64, k] = 0, 0, 0
    assert air.check_trace(program, t4, iu.NOPV) != []


def test_shift_chip():
    """Rv32Shift256 as a chip of its own: the host function against Python's integers (left, logical and arithmetic right; amounts 0,
    1, 255, whole limbs, beyond 255 -- only c mod 256 counts); the tests' twin trace with the bitwise table satisfies the AIRs and
    balances the bus; a result limb changed, another limb shift, and a logical shift of a negative word passed off as arithmetic have no
    trace."""
    with open(os.path.join(HERE, "golden", "int256_kat.json")) as f:
        cases = [(c["op"], int(c["b"], 16), int(c["c"], 16), int(c["a"], 16)) for c in json.load(f)["shift"]]
    for op, b, c, a in cases:
        assert iu.shift_result(op, b, c) == a and z.int256_alu_host(op, b, c) == a
    sel = cases[:20] + cases[42:62] + cases[84:104]
    tr, bw, xc = iu.shift_twin_trace([(op, b, c) for op, b, c, _ in sel], 6)
    for row, (_, _, _, a) in enumerate(sel):
        assert bytes(tr[0:32, row].astype(np.uint8)) == a.to_bytes(32, "little")
    inst = iu.shift_instance(tr, bw, xc, 6)
    for d in inst:
        assert air.quotient_chunks(d["program"]) <= 2
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    program = inst[0]["program"]
    rng = np.random.default_rng(4)
    for _ in range(30):
        t2 = tr.copy()
        c_, r_ = int(rng.integers(0, 128)), int(rng.integers(0, len(sel)))       # a result, operand, shifted or carry limb
        t2[c_, r_] = (int(t2[c_, r_]) + 1) % 2013265921
        i2 = iu.shift_instance(t2, bw, xc, 6)
        assert air.check_trace(program, t2, iu.NOPV) != [] or v2.bus_imbalance(i2) != {}, (c_, r_)
    k = next(i for i, (op, b, c, a) in enumerate(sel) if 8 <= c % 256 < 240 and b)
    t3 = tr.copy()
    ls = int(np.nonzero(tr[153:185, k])[0][0])
    t3[153 + ls, k], t3[153 + ls + 1, k] = 0, 1                                   # one limb further
    assert air.check_trace(program, t3, iu.NOPV) != []
    k = next(i for i, (op, b, c, a) in enumerate(sel) if op == 10 and b >> 255 and c % 256)
    t4 = tr.copy()
    t4[187, k], t4[188, k] = 0, 1                                                 # srl relabelled sra with sign = 0
    i4 = iu.shift_instance(t4, bw, xc, 6)
    assert air.check_trace(program, t4, iu.NOPV) != [] or v2.bus_imbalance(i4) != {}


def test_oracle_proves_the_chip_with_its_table(ora):
    rng = np.random.default_rng(5)
    cases = [(int(rng.integers(0, 5)), int.from_bytes(rng.bytes(32), "little"), int.from_bytes(rng.bytes(32), "little")) for _ in range(7)]
    tr, xc, bad = iu.ora_trace(ora, cases, 3)
    inst = iu.instance(tr, xc, 3)
    proof = ora.stark_prove(PARAMS, inst)
    assert ora.stark_verify(PARAMS, inst, proof) == 0
    vk = []
    for d in inst:
        v = {k: d[k] for k in ("program", "log_height", "width", "n_pvs")}
        if d.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(PARAMS, d)
        vk.append(v)
    assert z.verify(PARAMS, vk, [iu.NOPV] * 2, proof.tobytes()) == 0
