"""CPU: the Fp2 chip (include/zkhip_fp2.hpp) -- parity anchored OUTSIDE this repository.
  * the product's host function (zkhip_fp2_host: the executor's arithmetic) against Python's integers, and the published generator of
    bn254's G2 satisfies the twist equation y^2 = x^3 + 3 / (9 + u) when evaluated through it (tests/golden/fp2_kat.json);
  * the chip's trace (the tests' twin of zkhip_fp2_tracegen) with its lookup tables satisfies the AIRs and balances both buses; the results
    in the trace are Python's; a changed cell breaks a constraint or a bus; a division row without the canonicity markers of its quotient
    is refused; the host function refuses a zero divisor and components that are not reduced."""
import json
import os

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import fp2_util as fu
import vm2_util as v2

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "fp2_kat.json")) as f:
        d = json.load(f)
    cases = [(c["op"], (int(c["a0"], 16), int(c["a1"], 16)), (int(c["b0"], 16), int(c["b1"], 16)), (int(c["r0"], 16), int(c["r1"], 16))) for c in d["cases"]]
    return int(d["p"], 16), {k: int(v, 16) for k, v in d["g2"].items()}, cases


def test_host_function_against_python_integers(kat):
    p, g2, cases = kat
    for op, a, b, r in cases:
        assert z.fp2_host(op, p, a, b) == r
    x, y = (g2["x0"], g2["x1"]), (g2["y0"], g2["y1"])
    x3 = z.fp2_host(0, p, z.fp2_host(0, p, x, x), x)
    assert z.fp2_host(0, p, y, y) == z.fp2_host(1, p, x3, z.fp2_host(3, p, (3, 0), (9, 1)))   # the twist equation at EIP-197's generator
    assert z.fp2_host(3, p, x, (0, 0)) is None and z.fp2_host(0, p, (p, 0), x) is None and z.fp2_host(4, p, x, y) is None


def test_trace_satisfies_the_air_and_the_buses_balance(kat):
    p, _, cases = kat
    # the records: a division's record holds (quotient, divisor)
    calls = [(op, r if op == 3 else a, b) for op, a, b, r in cases[::2]]
    tr, bw, tup = fu.twin_trace(calls, p, 6)
    inst = fu.instance(p, tr, bw, tup, 6)
    for d in inst:
        assert air.quotient_chunks(d["program"]) <= 2
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    for row, (op, a, b, r) in enumerate(cases[::2]):
        res_cols = (0, 32) if op == 3 else (128, 160)          # a division's result sits in the a columns, its dividend in the r columns
        got = tuple(int.from_bytes(bytes(tr[c:c + 32, row].astype(np.uint8)), "little") for c in res_cols)
        assert got == r
        if op == 3:
            assert tuple(int.from_bytes(bytes(tr[c:c + 32, row].astype(np.uint8)), "little") for c in (128, 160)) == a
    program = inst[0]["program"]
    rng = np.random.default_rng(5)
    for _ in range(40):
        t2 = tr.copy()
        c, r_ = int(rng.integers(0, fu.COL_MARK)), int(rng.integers(0, len(calls)))
        t2[c, r_] = (int(t2[c, r_]) + 1) % 2013265921
        i2 = fu.instance(p, t2, bw, tup, 6)
        assert air.check_trace(program, t2, fu.NOPV) != [] or v2.bus_imbalance(i2) != {}, (c, r_)
    k = next(i for i, c in enumerate(calls) if c[0] == 3)
    t3 = tr.copy()
    t3[fu.COL_MARK2:fu.COL_DIFF2 + 2, k] = 0                # a division whose quotient is not shown to be canonical
    assert air.check_trace(program, t3, fu.NOPV) != []
