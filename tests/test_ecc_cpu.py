"""CPU: the elliptic-curve chip (include/zkhip_ecc.hpp) -- parity anchored OUTSIDE this repository.
  * the product's host function (zkhip_ec_host: the executor's arithmetic) against Python's integers and the published multiples of the
    secp256k1 / P-256 / bn254 generators (tests/golden/ecc_kat.json);
  * the chip's trace (the tests' twin of zkhip_ec_tracegen) with its lookup tables satisfies the AIRs and balances both buses; a changed
    cell breaks a constraint or a bus; a wrong result, a non-canonical result (x3 + P) and a wrong slope have no trace; the host function
    refuses operands that are not reduced, equal abscissae and a doubling of a point of order two."""
import json
import os

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import ecc_util as eu
import vm2_util as v2

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "ecc_kat.json")) as f:
        return json.load(f)


def curve_cases(kat, name):
    c = kat["curves"][name]
    p, a = int(c["p"], 16), int(c["a"], 16)
    cases = [(k["op"], (int(k["x1"], 16), int(k["y1"], 16)), (int(k["x2"], 16), int(k["y2"], 16)), int(k["slope"], 16), (int(k["x3"], 16), int(k["y3"], 16)))
             for k in kat["cases"] if k["curve"] == name]
    return p, a, cases


def test_host_function_against_python_integers(kat):
    for name in kat["curves"]:
        p, a, cases = curve_cases(kat, name)
        for op, p1, p2, lam, r in cases:
            assert z.ec_host(op, p, a, p1, p2) == (lam, r[0], r[1])
        g = (int(kat["curves"][name]["gx"], 16), int(kat["curves"][name]["gy"], 16))
        assert z.ec_host(0, p, a, g, g) is None                       # equal abscissae: no chord
        assert z.ec_host(0, p, a, g, (g[0], p - g[1])) is None        # the inverse point
        assert z.ec_host(1, p, a, (g[0], 0), g) is None               # no tangent slope
        assert z.ec_host(1, p, a, (g[0] + p, g[1]), g) is None if g[0] + p < 1 << 256 else True   # not reduced
        assert z.ec_host(2, p, a, g, g) is None


@pytest.mark.parametrize("name", ["secp256k1", "p256", "bn254"])
def test_trace_satisfies_the_air_and_the_buses_balance(kat, name):
    p, a, cases = curve_cases(kat, name)
    calls = [(op, p1, p2, lam) for op, p1, p2, lam, _ in cases]
    tr, bw, tup = eu.twin_trace(calls, p, a, 4)
    inst = eu.instance(p, a, tr, bw, tup, 4)
    for d in inst:
        assert air.quotient_chunks(d["program"]) <= 2
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    for row, (_, _, _, _, r) in enumerate(cases):
        assert bytes(tr[160:192, row].astype(np.uint8)) == r[0].to_bytes(32, "little") and bytes(tr[192:224, row].astype(np.uint8)) == r[1].to_bytes(32, "little")
    if name != "secp256k1":
        return
    rng = np.random.default_rng(3)
    for _ in range(40):   # any changed cell of a real row: a constraint fails or a lookup has no table entry
        t2 = tr.copy()
        c, r_ = int(rng.integers(0, eu.WIDTH)), int(rng.integers(0, len(calls)))
        t2[c, r_] = (int(t2[c, r_]) + 1) % 2013265921
        i2 = eu.instance(p, a, t2, bw, tup, 4)
        assert air.check_trace(i2[0]["program"], t2, eu.NOPV) != [] or v2.bus_imbalance(i2) != {}, (c, r_)


def test_forged_rows_have_no_trace(kat):
    p, a, cases = curve_cases(kat, "secp256k1")
    op, p1, p2, lam, r = next(c for c in cases if c[0] == 0)
    program = z.ec_air(p, a, eu.BITWISE_BUS, eu.TUPLE_BUS)[0]
    good, _, _ = eu.twin_trace([(op, p1, p2, lam)], p, a, 1)
    assert air.check_trace(program, good, eu.NOPV) == []
    # another slope: the first identity has no integer quotient -- the twin refuses, and with the honest row's quotients a limb equation fails
    with pytest.raises(AssertionError):
        eu.twin_trace([(op, p1, p2, (lam + 1) % p)], p, a, 1)
    t2 = good.copy()
    t2[128, 0] = (int(t2[128, 0]) + 1) % 256
    assert air.check_trace(program, t2, eu.NOPV) != []
    # the chord operation relabelled as a doubling
    t3 = good.copy()
    t3[eu.COL_DBL, 0] = 1
    assert air.check_trace(program, t3, eu.NOPV) != []
    # x3 + P (where it fits 256 bits) with the quotient one smaller satisfies the second identity but not x3 < P
    bn_p, bn_a, bn_cases = curve_cases(kat, "bn254")
    op, p1, p2, lam, r = next(c for c in bn_cases if c[0] == 0)
    bn_program = z.ec_air(bn_p, bn_a, eu.BITWISE_BUS, eu.TUPLE_BUS)[0]
    tr, _, _ = eu.twin_trace([(op, p1, p2, lam)], bn_p, bn_a, 1)
    forged = tr.copy()
    x3f = r[0] + bn_p
    forged[160:192, 0] = np.frombuffer(x3f.to_bytes(32, "little"), dtype=np.uint8)
    pb = bn_p.to_bytes(32, "little")
    # recompute identities 2 and 3 for the forged abscissa so that only the range argument is left to object
    x1, y1, x2 = p1[0], p1[1], p2[0]
    y3 = (lam * (x1 - x3f) - y1) % bn_p
    forged[192:224, 0] = np.frombuffer(y3.to_bytes(32, "little"), dtype=np.uint8)
    for e, v in ((1, lam * lam - x1 - x2 - x3f), (2, lam * (x1 - x3f) - y1 - y3)):
        assert v % bn_p == 0
        q = v // bn_p
        ql = abs(q).to_bytes(33, "little")
        forged[eu.COL_Q + 33 * e:eu.COL_Q + 33 * (e + 1), 0] = np.frombuffer(ql, dtype=np.uint8)
        forged[eu.COL_QS + e, 0] = 1 if q < 0 else 0
        L = [bytes(forged[32 * o:32 * o + 32, 0].astype(np.uint8)) for o in range(7)]
        c = 0
        for k in range(64):
            s = c
            for i in range(33):
                j = k - i
                if 0 <= j < 32:
                    s -= (-1 if q < 0 else 1) * ql[i] * pb[j]
                    if i < 32:
                        s += L[4][i] * L[4][j] if e == 1 else L[4][i] * (L[0][j] - L[5][j])
            if k < 32:
                s -= L[0][k] + L[2][k] + L[5][k] if e == 1 else L[1][k] + L[6][k]
            assert s % 256 == 0
            c = s // 256
            if k < 63:
                forged[eu.COL_CX + 63 * e + k, 0], forged[eu.COL_CY + 63 * e + k, 0] = (c + (1 << 18)) & 255, (c + (1 << 18)) >> 8
        assert c == 0
    # markers for y3 recomputed; x3's markers cannot be made right
    forged[eu.COL_MARK + 32:eu.COL_MARK + 64, 0] = 0
    ylimbs = y3.to_bytes(32, "little")
    mark = max(i for i in range(32) if ylimbs[i] != pb[i])
    forged[eu.COL_MARK + 32 + mark, 0], forged[eu.COL_DIFF + 1, 0] = 1, pb[mark] - ylimbs[mark]
    failing = air.check_trace(bn_program, forged, eu.NOPV)
    assert failing != []
    # ... and with the forged limbs restored to the canonical ones the same recomputation is accepted
    assert air.check_trace(bn_program, tr, eu.NOPV) == []
