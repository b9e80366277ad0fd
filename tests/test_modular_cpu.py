"""CPU: the modular-multiplication chip (include/zkhip_modular.hpp) -- parity anchored OUTSIDE this repository.
  * the oracle's byte-wise divmod (oracle/modular.c) and the product's host function (zkhip_modmul_host) against Python's integers for the
    secp256k1 / bn254 field and scalar moduli (tests/golden/modular_kat.json); the published secp256k1 generator satisfies the curve
    equation when computed through them;
  * the chip's trace (oracle twin of zkhip_modmul_tracegen) with its lookup tables satisfies the AIRs and balances both buses; r and q in
    the trace are Python's; a changed cell breaks a constraint or a bus; a non-canonical residue (r + P) has no trace; the oracle proves
    the set and both verifiers accept."""
import json
import os

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import modular_util as mu
import vm2_util as v2

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = (1, 0, 4, 3, 3)


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "modular_kat.json")) as f:
        return json.load(f)


def test_divmod_against_python_integers(ora, kat):
    mod = {k: int(v, 16) for k, v in kat["moduli"].items()}
    for c in kat["cases"]:
        p, a, b, q, r = mod[c["modulus"]], int(c["a"], 16), int(c["b"], 16), int(c["q"], 16), int(c["r"], 16)
        assert (q, r) == divmod(a * b, p)
        assert mu.ora_modmul(ora, a, b, p) == (0, q, r)
        assert z.modmul_host(a, b, p) == (q, r)
    # the published generator of secp256k1 lies on y^2 = x^3 + 7
    p = mod["secp256k1_p"]
    gx, gy = int(kat["secp256k1_generator"]["x"], 16), int(kat["secp256k1_generator"]["y"], 16)
    y2 = z.modmul_host(gy, gy, p)[1]
    x3 = z.modmul_host(z.modmul_host(gx, gx, p)[1], gx, p)[1]
    assert y2 == (x3 + 7) % p and mu.ora_modmul(ora, gy, gy, p)[2] == y2


def test_addition_and_subtraction(ora, kat):
    """ModularAddSub in the multiplication chip's columns: the oracle's byte-wise add / sub and the product's host function against
    Python's integers; a trace that mixes the three operations satisfies the AIR and balances both buses; a sub row relabelled add fails."""
    mod = {k: int(v, 16) for k, v in kat["moduli"].items()}
    for c in kat["addsub"]:
        p, op, a, b, q, r = mod[c["modulus"]], c["op"], int(c["a"], 16), int(c["b"], 16), int(c["q"], 16), int(c["r"], 16)
        assert (a + b == q * p + r) if op == 1 else (a - b + q * p == r) and r < p
        assert mu.ora_addsub(ora, op, a, b, p) == (0, q, r) and z.modular_host(op, a, b, p) == (q, r)
    p = mod["secp256k1_n"]
    rows = [(c["op"], int(c["a"], 16), int(c["b"], 16)) for c in kat["addsub"] if c["modulus"] == "secp256k1_n"][:10]
    rows += [(0, int(c["a"], 16), int(c["b"], 16)) for c in kat["cases"] if c["modulus"] == "secp256k1_n"][:4]
    tr, bw, tup, bad = mu.ora_trace(ora, [(a, b) for _, a, b in rows], p, 4, ops=[o for o, _, _ in rows])
    assert bad == 0
    inst = mu.instance(p, tr, bw, tup, 4)
    for d in inst:
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    for row, (op, a, b) in enumerate(rows):
        want = [a * b % p, (a + b) % p, (a - b) % p][op]
        assert bytes(tr[96:128, row].astype(np.uint8)) == want.to_bytes(32, "little")
    k = next(i for i, (op, a, b) in enumerate(rows) if op == 2 and a != b)
    t2 = tr.copy()
    t2[286, k], t2[287, k] = 1, 0
    assert air.check_trace(inst[0]["program"], t2, mu.NOPV) != []


def test_division(ora, kat):
    """ModularMulDiv's division in the multiplication chip's columns: the host function against Python's integers; a division row is the
    product (x / y) y = x with the quotient in the a columns; a trace mixing divisions with the other operations satisfies the AIR and
    balances both buses; a quotient that is not reduced ((x / y) + P, where it fits) has no trace; the host function refuses a zero
    divisor and a dividend that is not reduced."""
    mod = {k: int(v, 16) for k, v in kat["moduli"].items()}
    for c in kat["div"]:
        p, x, y, r = mod[c["modulus"]], int(c["a"], 16), int(c["b"], 16), int(c["r"], 16)
        assert r * y % p == x and r < p
        assert z.modular_host(3, x, y, p) == (0, r)
    p = mod["bn254_p"]
    lib = z.load_library()
    import ctypes as C
    from zkvm_prover_amd._binding import _int_words, _u32p
    def refused(x, y):
        w = [np.array(_int_words(v), dtype=np.uint32) for v in (x, y, p)]
        q, r = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
        return lib.zkhip_modular_host(3, _u32p(w[0]), _u32p(w[1]), _u32p(w[2]), _u32p(q), _u32p(r)) != 0
    assert refused(5, 0) and refused(5, p) and refused(p + 1, 3) and not refused(5, 3)
    divs = [(int(c["a"], 16), int(c["b"], 16), int(c["r"], 16)) for c in kat["div"] if c["modulus"] == "bn254_p"]
    rows = [(3, r, y) for _, y, r in divs[:9]]                      # the record of a division: (quotient, divisor)
    rows += [(c["op"], int(c["a"], 16), int(c["b"], 16)) for c in kat["addsub"] if c["modulus"] == "bn254_p"][:3]
    rows += [(0, int(c["a"], 16), int(c["b"], 16)) for c in kat["cases"] if c["modulus"] == "bn254_p"][:3]
    tr, bw, tup, bad = mu.ora_trace(ora, [(a, b) for _, a, b in rows], p, 4, ops=[o for o, _, _ in rows])
    assert bad == 0
    inst = mu.instance(p, tr, bw, tup, 4)
    for d in inst:
        assert air.quotient_chunks(d["program"]) <= 2
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    for row, (x, y, r) in enumerate(divs[:9]):
        assert bytes(tr[96:128, row].astype(np.uint8)) == x.to_bytes(32, "little") and bytes(tr[0:32, row].astype(np.uint8)) == r.to_bytes(32, "little")
        assert tr[288, row] == 1
    # the quotient r + P (bn254: fits 256 bits) multiplies to the same x modulo P, but is not canonical: the oracle twin refuses the
    # record, and the row it writes (without a marker) breaks the AIR
    x, y, r = divs[8]
    forged, _, _, bad = mu.ora_trace(ora, [(r + p, y)], p, 1, ops=[3])
    assert bad == 1 and air.check_trace(inst[0]["program"], forged, mu.NOPV) != []
    # a division row relabelled as a plain multiplication is fine by the AIR alone (it IS a product) -- inside the VM the word bus tells
    # them apart (tests/test_vm2_cpu.py)
    t2 = tr.copy()
    t2[288, 0], t2[289:322, 0] = 0, 0
    assert air.check_trace(inst[0]["program"], t2, mu.NOPV) == []


def test_equality_test(ora, kat):
    """ModularIsEqual on top of a subtraction row: the host function's bit is Python's; equal and different operands in one trace satisfy
    the AIR and balance both buses; "equal" claimed for different operands and "different" claimed for equal ones have no trace."""
    p = int(kat["moduli"]["secp256k1_n"], 16)
    pairs = [(5, 5), (5, 6), (0, 0), (p - 1, p - 1), (p - 1, 0), (0, p - 1), (1 << 200, 1 << 200), (1 << 200, (1 << 200) + 256)]
    for a, b in pairs:
        assert z.modular_host(4, a, b, p) == (0, int(a == b))
    rows = [(4, a, b) for a, b in pairs] + [(2, 9, 4), (1, 9, 4), (0, 9, 4)]
    tr, bw, tup, bad = mu.ora_trace(ora, [(a, b) for _, a, b in rows], p, 4, ops=[o for o, _, _ in rows])
    assert bad == 0
    inst = mu.instance(p, tr, bw, tup, 4)
    for d in inst:
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    for row, (a, b) in enumerate(pairs):
        assert tr[322, row] == 1 and tr[287, row] == 1 and tr[323, row] == int(a == b)
    program = inst[0]["program"]
    t2 = tr.copy()
    t2[323, 1], t2[324, 1] = 1, 0                      # 5 == 6 claimed
    assert air.check_trace(program, t2, mu.NOPV) != []
    t3 = tr.copy()
    t3[323, 0] = 0                                     # 5 != 5 claimed: no inverse of a zero sum
    for inv in (0, 1, 12345):
        t3[324, 0] = inv
        assert air.check_trace(program, t3, mu.NOPV) != []
    t4 = tr.copy()
    t4[323, 8] = 1                                     # the bit set on a plain subtraction row
    assert air.check_trace(program, t4, mu.NOPV) != []


@pytest.mark.parametrize("name", ["secp256k1_p", "bn254_r"])
def test_trace_satisfies_the_air_and_the_buses_balance(ora, kat, name):
    p = int(kat["moduli"][name], 16)
    pairs = [(int(c["a"], 16), int(c["b"], 16)) for c in kat["cases"] if c["modulus"] == name][:13]
    tr, bw, tup, bad = mu.ora_trace(ora, pairs, p, 4)
    assert bad == 0
    inst = mu.instance(p, tr, bw, tup, 4)
    for d in inst:
        assert air.quotient_chunks(d["program"]) <= 2
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == []
    assert v2.bus_imbalance(inst) == {}
    for row, (a, b) in enumerate(pairs):
        q, r = divmod(a * b, p)
        assert bytes(tr[64:96, row].astype(np.uint8)) == q.to_bytes(32, "little") and bytes(tr[96:128, row].astype(np.uint8)) == r.to_bytes(32, "little")
    rng = np.random.default_rng(3)
    for _ in range(60):   # any changed cell of a real row: a constraint fails or a lookup has no table entry
        t2 = tr.copy()
        c, r_ = int(rng.integers(0, mu.WIDTH)), int(rng.integers(0, len(pairs)))
        t2[c, r_] = (int(t2[c, r_]) + 1) % 2013265921
        i2 = mu.instance(p, t2, bw, tup, 4)
        assert air.check_trace(i2[0]["program"], t2, mu.NOPV) != [] or v2.bus_imbalance(i2) != {}, (c, r_)
    # the residue r + P with the quotient q - 1 satisfies the integer identity but not r < P
    a, b = pairs[8]
    q, r = divmod(a * b, p)
    if r + p < 1 << 256 and q > 0:
        forged, _, _, _ = mu.ora_trace(ora, [(a, b)], p, 4)
        forged[64:96, 0] = np.frombuffer((q - 1).to_bytes(32, "little"), dtype=np.uint8)
        forged[96:128, 0] = np.frombuffer((r + p).to_bytes(32, "little"), dtype=np.uint8)
        # (carries recomputed for the forged limbs so that only the range argument is left to object)
        pb = p.to_bytes(32, "little")
        c = 0
        for k in range(63):
            s = c + sum(int(forged[i, 0]) * int(forged[32 + k - i, 0]) - int(forged[64 + i, 0]) * pb[k - i] for i in range(32) if 0 <= k - i < 32)
            s -= int(forged[96 + k, 0]) if k < 32 else 0
            assert s % 256 == 0
            c = s // 256
            if k < 62:
                forged[128 + k, 0], forged[190 + k, 0] = (c + (1 << 14)) & 255, (c + (1 << 14)) >> 8
        assert c == 0
        assert air.check_trace(inst[0]["program"], forged, mu.NOPV) != []


def test_oracle_proves_the_chip_with_its_tables(ora, kat):
    p = int(kat["moduli"]["bn254_p"], 16)
    rng = np.random.default_rng(5)
    pairs = [(int.from_bytes(rng.bytes(32), "little") % p, int.from_bytes(rng.bytes(32), "little") % p) for _ in range(6)]
    tr, bw, tup, bad = mu.ora_trace(ora, pairs, p, 3)
    inst = mu.instance(p, tr, bw, tup, 3)
    proof = ora.stark_prove(PARAMS, inst)
    assert ora.stark_verify(PARAMS, inst, proof) == 0
    vk = []
    for d in inst:
        v = {k: d[k] for k in ("program", "log_height", "width", "n_pvs")}
        if d.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(PARAMS, d)
        vk.append(v)
    assert z.verify(PARAMS, vk, [mu.NOPV] * 3, proof.tobytes()) == 0
