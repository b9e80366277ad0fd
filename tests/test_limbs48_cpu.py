"""CPU: the limb chips at 48 limbs -- a modulus above 2^256: the BLS12-381 base field of the reference's BATCH circuit
(crates/circuits/batch-circuit/openvm.toml:18-36: modular [p, r], fp2 Bls12_381Fp2, ecc Bls12_381G1Affine; OpenVM instantiates its
mod-builder chips with 32 or 48 limbs per configured modulus).  External parity: Python's integers, and the standard generators of G1
and G2, which satisfy their curve equations (checked here).
  * host arithmetic (what the executor runs) == Python's integers: modular mul / add / sub / div / is_eq, G1 doubling and addition, Fp2;
  * the modular chip: a Python twin of the columns (any limb count) satisfies the AIR the C++ builder emits for 48 limbs, the oracle
    proves chip + tables, the product's verifier accepts; a forged residue does not satisfy it;
  * widths of the chips and adapters for both limb counts;
  * the batch circuit's openvm.toml sections go through the executor's configuration reader; a guest that checks both curve equations and
    computes 3 G with 48-byte operands runs to the public values of the independent Python model (tests/rv32_model.py)."""
import json
import subprocess

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import modular_util as mu
import rv32_model as rv
import vm2_util as v2
from test_vm_cpu import (BATCH_CIRCUIT_CURVES, BATCH_CIRCUIT_MODULI, BLS12_381_G1, BLS12_381_G2X, BLS12_381_G2Y, BLS12_381_P, BLS12_381_R, batch_circuit_toml,
                         bls_data, bls_program)

P, R = BLS12_381_P, BLS12_381_R
PARAMS = (1, 0, 4, 3, 3)


def fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def test_the_generators_are_on_their_curves():
    gx, gy = BLS12_381_G1
    assert P.bit_length() == 381 and R.bit_length() == 255 and (gy * gy - gx ** 3 - 4) % P == 0
    x3 = fp2_mul(fp2_mul(BLS12_381_G2X, BLS12_381_G2X), BLS12_381_G2X)
    assert fp2_mul(BLS12_381_G2Y, BLS12_381_G2Y) == ((x3[0] + 4) % P, (x3[1] + 4) % P)


def test_host_arithmetic_is_pythons():
    rng = np.random.default_rng(48)
    big = lambda: int.from_bytes(rng.bytes(48), "little") % P  # noqa: E731
    cases = [(big(), big()) for _ in range(12)] + [(0, 0), (P - 1, P - 1), (1, P - 1), (P - 1, 1), (0, P - 1)]
    for a, b in cases:
        q, r = z.modmul_host(a, b, P)
        assert (q, r) == divmod(a * b, P)
        assert z.modular_host(1, a, b, P) == divmod(a + b, P)
        q, r = z.modular_host(2, a, b, P)
        assert r == (a - b) % P and a - b + q * P == r
        if b:
            assert z.modular_host(3, a, b, P)[1] == a * pow(b, -1, P) % P
        assert z.modular_host(4, a, b, P)[1] == int(a == b)
    top = (1 << 384) - 1                                          # unreduced operands: fine while the quotient fits 48 limbs
    assert z.modular_host(1, top, top, P) == divmod(2 * top, P) and z.modmul_host(top, 7, P) == divmod(7 * top, P)
    with pytest.raises(AssertionError):
        z.modmul_host(top, top, P)                                # q ~ 2^387: refused
    # the 255-bit scalar field keeps 32-limb operands
    assert z.limb_words(P) == 12 and z.limb_words(R) == 8
    assert z.modmul_host(R - 2, R - 3, R) == divmod((R - 2) * (R - 3), R)
    # G1: 2 G and 3 G by the textbook formulas
    gx, gy = BLS12_381_G1
    lam = 3 * gx * gx * pow(2 * gy, -1, P) % P
    x2 = (lam * lam - 2 * gx) % P
    y2 = (lam * (gx - x2) - gy) % P
    assert z.ec_host(1, P, 0, (gx, gy), (0, 0)) == (lam, x2, y2)
    lam3 = (gy - y2) * pow(gx - x2, -1, P) % P
    x3 = (lam3 * lam3 - x2 - gx) % P
    assert z.ec_host(0, P, 0, (x2, y2), (gx, gy)) == (lam3, x3, (lam3 * (x2 - x3) - y2) % P)
    assert (((lam3 * (x2 - x3) - y2) % P) ** 2 - x3 ** 3 - 4) % P == 0                 # 3 G is on the curve
    assert z.ec_host(0, P, 0, (gx, gy), (gx, gy)) is None                              # a chord needs two abscissae
    # Fp2
    x, y = BLS12_381_G2X, BLS12_381_G2Y
    assert z.fp2_host(0, P, x, y) == fp2_mul(x, y)
    assert z.fp2_host(1, P, x, y) == ((x[0] + y[0]) % P, (x[1] + y[1]) % P) and z.fp2_host(2, P, x, y) == ((x[0] - y[0]) % P, (x[1] - y[1]) % P)
    qt = z.fp2_host(3, P, x, y)
    assert fp2_mul(qt, y) == x and z.fp2_host(3, P, x, (0, 0)) is None


def test_chip_and_adapter_widths():
    for p, w_mod, w_fp2, w_ec in ((2**256 - 2**32 - 977, 325, 648, 772), (P, 485, 968, 1156)):
        nw = z.limb_words(p)
        assert z.modmul_air(p, 9, 6)[1] == w_mod and z.vm_modmul_air(p, 0, False)[1] == w_mod + 1 and z.vm_modmul_air(p, 0, True)[1] == 3 * nw + 11
        assert z.fp2_air(p, 9, 6)[1] == w_fp2 and z.vm_fp2_air(p, 0, False)[1] == w_fp2 + 1 and z.vm_fp2_air(p, 0, True)[1] == 6 * nw + 11
        assert z.ec_air(p, 0, 9, 6)[1] == w_ec and z.vm_ec_air(p, 0, 0, False)[1] == w_ec + 1 and z.vm_ec_air(p, 0, 0, True)[1] == 6 * nw + 11
    for prog in (z.modmul_air(P, 9, 6)[0], z.fp2_air(P, 9, 6)[0], z.ec_air(P, 0, 9, 6)[0]):
        assert air.quotient_chunks(prog) <= 2                  # degree <= 3, as at 32 limbs


def test_modular_chip_48_limbs_python_twin_satisfies_the_air(ora):
    rng = np.random.default_rng(7)
    big = lambda: int.from_bytes(rng.bytes(48), "little") % P  # noqa: E731
    top = (1 << 384) - 1                                          # operands need not be reduced (bytes, not residues): the quotient must fit 48 limbs
    rows = [(0, big(), big()) for _ in range(5)] + [(1, big(), big()), (1, P - 1, P - 1), (2, big(), big()), (2, 3, 5), (3, big(), big()), (3, 1, P - 1),
                                                   (4, 5, 5), (4, 5, 6), (0, P - 1, P - 1), (0, 0, 0), (1, top, top), (0, top, 7), (2, top, top - 3)]
    log_h = 5
    tr, bw, tup = mu.py_trace(rows, P, log_h)
    assert tr.shape == (485, 32)
    program, width = z.modmul_air(P, mu.BITWISE_BUS, mu.TUPLE_BUS)
    assert air.check_trace(program, tr, mu.NOPV, None) == []
    for row, (op, a, b) in enumerate(rows):   # the r columns (a division: the dividend there, the quotient in the a columns)
        r = int.from_bytes(bytes(tr[144:192, row].astype(np.uint8)), "little")
        if op == 3:
            assert r == a and int.from_bytes(bytes(tr[0:48, row].astype(np.uint8)), "little") == a * pow(b, -1, P) % P
        else:
            assert r == [a * b % P, (a + b) % P, (a - b) % P, None, (a - b) % P][op]
    inst = mu.instance(P, tr, bw, tup, log_h)
    proof = ora.stark_prove(PARAMS, inst)
    assert ora.stark_verify(PARAMS, inst, proof) == 0
    vk = []
    for d in inst:
        v = {k: d[k] for k in ("program", "log_height", "width", "n_pvs")}
        if d.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(PARAMS, d)
        vk.append(v)
    assert z.verify(PARAMS, vk, [mu.NOPV] * 3, proof.tobytes()) == 0
    bad = tr.copy()
    bad[144 + 2, 1] ^= 1                                          # another residue
    assert air.check_trace(program, bad, mu.NOPV, None) != []
    bad = tr.copy()
    bad[mu.cols(48)["MARK"] + 47, 0], bad[mu.cols(48)["MARK"] + 46, 0] = bad[mu.cols(48)["MARK"] + 46, 0], bad[mu.cols(48)["MARK"] + 47, 0]   # another limb marked
    assert air.check_trace(program, bad, mu.NOPV, None) != []


def test_batch_circuit_config_and_a_bls12_381_guest(tmp_path):
    words, data = bls_program(), bls_data()
    model = rv.run(words, b"", data=data, moduli=BATCH_CIRCUIT_MODULI, curves=((P, 0),), fp2=(P,))
    pv = [int.from_bytes(bytes(model["pvs"][4 * k:4 * k + 4]), "little") for k in range(8)]
    gx, gy = BLS12_381_G1
    lam = 3 * gx * gx * pow(2 * gy, -1, P) % P
    x2 = (lam * lam - 2 * gx) % P
    y2 = (lam * (gx - x2) - gy) % P
    lam = (gy - y2) * pow(gx - x2, -1, P) % P
    x3 = (lam * lam - x2 - gx) % P
    y3 = (lam * (x2 - x3) - y2) % P
    yy = fp2_mul(BLS12_381_G2Y, BLS12_381_G2Y)
    assert pv == [1, x3 & 0xFFFFFFFF, y3 & 0xFFFFFFFF, yy[0] & 0xFFFFFFFF, yy[0] & 0xFFFFFFFF, 0, (R - 2) * (R - 3) % R & 0xFFFFFFFF, gy * gy % P & 0xFFFFFFFF]
    (tmp_path / "exe.bin").write_bytes(rv.exe_bytes(words, data=data))
    (tmp_path / "stdin.bin").write_bytes(b"")
    (tmp_path / "batch.toml").write_text(batch_circuit_toml(PARAMS))
    r = subprocess.run([v2.CLI, "dump-segments", str(tmp_path / "exe.bin"), str(tmp_path / "stdin.bin"), str(tmp_path), "10", "5", "7", str(tmp_path / "batch.toml")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"]) and info["total_cycle"] == model["instret"]
    ids = np.fromfile(tmp_path / "air_ids.u32", dtype=np.uint32).tolist()
    # the base chips + keccak (2) + sha2 (2) + two moduli (4) + one curve (2) + one Fp2 field (2) + native (2) + castf (1): no section ignored
    assert len(ids) == 22 + 2 + 2 + 4 + 2 + 2 + 3
    segs = [tmp_path / ("seg-%d" % k) for k in range(info["segments"])]
    rec = lambda name: np.concatenate([np.fromfile(s / (name + ".u32"), dtype=np.uint32) for s in segs])  # noqa: E731
    # records of 12-word operands: op | a | b (25), op | x1 y1 x2 y2 | slope (61), op | a0 a1 | b0 b1 (49); adapter rows of 36 + 11 and 72 + 11 columns
    assert rec("mm_records_0").size == 5 * 25 and rec("mm_records_1").size == 17 and rec("ec_records_0").size == 2 * 61 and rec("fp2_records_0").size == 5 * 49
    assert rec("mmio_rows_0").size == 5 * 36 * 47 and rec("mmio_rows_1").size == 24 * 35 and rec("ecio_rows_0").size == 2 * 72 * 83 and rec("fp2io_rows_0").size == 5 * 72 * 83
    # the reference's own file, where its tree is present (not on the GPU box): the same chip set
    import os
    ref = "/root/reference/crates/circuits/batch-circuit/openvm.toml"
    if os.path.exists(ref):
        out = tmp_path / "ref"
        out.mkdir()
        r = subprocess.run([v2.CLI, "dump-segments", str(tmp_path / "exe.bin"), str(tmp_path / "stdin.bin"), str(out), "10", "5", "7", ref], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert np.fromfile(out / "air_ids.u32", dtype=np.uint32).tolist() == ids
        assert json.loads(r.stdout.strip().splitlines()[-1])["public_values"] == info["public_values"]


def test_ecc_and_fp2_chips_48_limbs_python_twins_satisfy_the_airs():
    """the twins of tests/ecc_util.py / tests/fp2_util.py (Python integers) at 48 limbs against the AIRs the C++ builders emit"""
    import ecc_util as eu
    import fp2_util as fu

    gx, gy = BLS12_381_G1
    lam = eu.slope_of(1, P, 0, (gx, gy), (0, 0))
    x2 = (lam * lam - 2 * gx) % P
    g2 = (x2, (lam * (gx - x2) - gy) % P)
    calls = [(1, (gx, gy), (0, 0), lam), (0, g2, (gx, gy), eu.slope_of(0, P, 0, g2, (gx, gy))), (1, g2, (5, 6), eu.slope_of(1, P, 0, g2, (0, 0)))]
    tr, bw, tup = eu.twin_trace(calls, P, 0, 2)
    assert tr.shape == (1156, 4)
    program, width = z.ec_air(P, 0, eu.BITWISE_BUS, eu.TUPLE_BUS)
    assert width == 1156 and air.check_trace(program, tr, eu.NOPV, None) == []
    bad = tr.copy()
    bad[5 * 48 + 1, 1] ^= 1                                       # another abscissa of 3 G
    assert air.check_trace(program, bad, eu.NOPV, None) != []
    x, y = BLS12_381_G2X, BLS12_381_G2Y
    qt = z.fp2_host(3, P, x, y)
    fcalls = [(0, x, y), (1, x, y), (2, x, y), (2, y, x), (3, qt, y), (0, (P - 1, P - 1), (P - 1, P - 1))]
    tr, bw, tup = fu.twin_trace(fcalls, P, 3)
    assert tr.shape == (968, 8)
    program, width = z.fp2_air(P, fu.BITWISE_BUS, fu.TUPLE_BUS)
    assert width == 968 and air.check_trace(program, tr, fu.NOPV, None) == []
    assert int.from_bytes(bytes(tr[4 * 48:5 * 48, 4].astype(np.uint8)), "little") == x[0]      # the division row's r columns hold the dividend
    bad = tr.copy()
    bad[4 * 48, 0] ^= 1
    assert air.check_trace(program, bad, fu.NOPV, None) != []
