"""GPU: the limb chips at 48 limbs on the device (BLS12-381: the reference's batch circuit, crates/circuits/batch-circuit/openvm.toml:18-36).
  * zkhip_modular_tracegen_x / zkhip_ec_tracegen_x / zkhip_fp2_tracegen_x with 12-word operands == the tests' Python twins cell for cell
    (485 / 1156 / 968 columns) and count for count in both lookup tables; the results are Python's integers; the HIP proof of each chip
    with its tables == the oracle's proof, byte for byte; a forged cell is refused;
  * ONE FLOW under the batch circuit's openvm.toml sections: a guest that checks the G1 and G2 curve equations and computes 3 G with the
    modular / ecc / fp2 intrinsics on 48-byte operands -> segments (32 chips per segment key) -> aggregation -> one root, verify-guest;
    the public values are the independent Python model's."""
import json
import subprocess

import numpy as np
import pytest
import torch

import zkvm_prover_amd as z

import ecc_util as eu
import fp2_util as fu
import modular_util as mu
import prover_mirror_util as pm
import rv32_model as rv
from test_vm_cpu import BATCH_CIRCUIT_MODULI, BLS12_381_G1, BLS12_381_G2X, BLS12_381_G2Y, BLS12_381_P, batch_circuit_toml, bls_data, bls_program

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
P = BLS12_381_P
N_STMT = 50


def tables(zk, sy):
    return torch.zeros(2 << 16, dtype=torch.int32, device=zk.device), torch.zeros(256 * sy, dtype=torch.int32, device=zk.device)


def prove_and_compare(zk, ora, inst, traces, forged_cell):
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [mu.NOPV] * 3
    proof = pk.prove(traces, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    bad = traces[0].clone()
    bad[forged_cell] ^= 1
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, pk.prove([bad] + traces[1:], pvs)) != 0
    pk.close()


def test_device_modular_trace_48_limbs(zk, ora):
    rng = np.random.default_rng(11)
    big = lambda: int.from_bytes(rng.bytes(48), "little") % P  # noqa: E731
    rows = [(0, big(), big()) for _ in range(6)] + [(1, big(), big()), (1, P - 1, P - 1), (2, big(), big()), (2, 3, 5), (3, big(), big()), (3, 1, P - 1), (4, 5, 5),
                                                   (4, 5, 6), (0, P - 1, P - 1), (0, 0, 0)]
    log_h = 4
    # a division's record holds the quotient in the a slot
    recs = np.array([[op] + eu.words(a * pow(b, -1, P) % P if op == 3 else a, 12) + eu.words(b, 12) for op, a, b in rows], dtype=np.uint32).reshape(-1)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
    d_bw, d_tup = tables(zk, mu.SY)
    d_tr = zk.modular_tracegen(P, d_recs, len(rows), log_h, d_bw, d_tup, mu.SX, mu.SY)
    got = zk.download(d_tr).reshape(485, -1)
    tr, bw, tup = mu.py_trace(rows, P, log_h)
    assert (got == tr).all()
    assert (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_tup) == tup).all()
    for row, (op, a, b) in enumerate(rows):
        if op == 0:
            assert bytes(got[144:192, row].astype(np.uint8)) == (a * b % P).to_bytes(48, "little")
    prove_and_compare(zk, ora, mu.instance(P, got, bw, tup, log_h), [d_tr, d_bw, d_tup], 144 * (1 << log_h) + 2)
    # a record whose quotient is not reduced is refused
    rec = np.array([3] + eu.words(P + 5, 12) + eu.words(3, 12), dtype=np.uint32)
    with pytest.raises(Exception):
        zk.modular_tracegen(P, torch.from_numpy(rec.view(np.int32)).to(zk.device), 1, 1, *tables(zk, mu.SY), mu.SX, mu.SY)


def test_device_ec_trace_48_limbs(zk, ora):
    g = BLS12_381_G1
    pts = [g]
    calls = []
    lam = eu.slope_of(1, P, 0, g, (0, 0))
    x2 = (lam * lam - 2 * g[0]) % P
    pts.append((x2, (lam * (g[0] - x2) - g[1]) % P))
    calls.append((1, g, (0, 0), lam))
    for _ in range(5):   # k G + G
        a, lam = pts[-1], eu.slope_of(0, P, 0, pts[-1], g)
        x3 = (lam * lam - a[0] - g[0]) % P
        pts.append((x3, (lam * (a[0] - x3) - a[1]) % P))
        calls.append((0, a, g, lam))
    calls.append((1, pts[3], (7, 9), eu.slope_of(1, P, 0, pts[3], (0, 0))))
    for q in pts:
        assert (q[1] * q[1] - q[0] ** 3 - 4) % P == 0
    log_h = 3
    recs = np.array([eu.record(*c, n=12) for c in calls], dtype=np.uint32).reshape(-1)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
    d_bw, d_tup = tables(zk, eu.SY)
    d_tr = zk.ec_tracegen(P, 0, d_recs, len(calls), log_h, d_bw, d_tup, eu.SX, eu.SY)
    got = zk.download(d_tr).reshape(1156, -1)
    tr, bw, tup = eu.twin_trace(calls, P, 0, log_h)
    assert (got == tr).all()
    assert (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_tup) == tup).all()
    for row in range(1, 6):   # the chord additions: (k + 1) G
        assert bytes(got[5 * 48:6 * 48, row].astype(np.uint8)) == pts[row + 1][0].to_bytes(48, "little")
        assert bytes(got[6 * 48:7 * 48, row].astype(np.uint8)) == pts[row + 1][1].to_bytes(48, "little")
    prove_and_compare(zk, ora, eu.instance(P, 0, got, bw, tup, log_h), [d_tr, d_bw, d_tup], 5 * 48 * (1 << log_h) + 2)
    bad = recs.copy()
    bad[1 + 4 * 12] ^= 1    # a slope that does not solve the tangent identity
    with pytest.raises(Exception):
        zk.ec_tracegen(P, 0, torch.from_numpy(bad.view(np.int32)).to(zk.device), len(calls), log_h, *tables(zk, eu.SY), eu.SX, eu.SY)


def test_device_fp2_trace_48_limbs(zk, ora):
    x, y = BLS12_381_G2X, BLS12_381_G2Y
    qt = z.fp2_host(3, P, x, y)
    calls = [(0, x, y), (0, y, y), (1, x, y), (2, x, y), (2, y, x), (3, qt, y), (0, (P - 1, P - 1), (P - 1, P - 1)), (1, (P - 1, 0), (1, 0))]
    log_h = 3
    recs = np.array([fu.record(*c, n=12) for c in calls], dtype=np.uint32).reshape(-1)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
    d_bw, d_tup = tables(zk, fu.SY)
    d_tr = zk.fp2_tracegen(P, d_recs, len(calls), log_h, d_bw, d_tup, fu.SX, fu.SY)
    got = zk.download(d_tr).reshape(968, -1)
    tr, bw, tup = fu.twin_trace(calls, P, log_h)
    assert (got == tr).all()
    assert (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_tup) == tup).all()
    yy = ((y[0] * y[0] - y[1] * y[1]) % P, 2 * y[0] * y[1] % P)
    assert bytes(got[4 * 48:5 * 48, 1].astype(np.uint8)) == yy[0].to_bytes(48, "little") and bytes(got[5 * 48:6 * 48, 1].astype(np.uint8)) == yy[1].to_bytes(48, "little")
    prove_and_compare(zk, ora, fu.instance(P, got, bw, tup, log_h), [d_tr, d_bw, d_tup], 4 * 48 * (1 << log_h) + 1)


def test_bls12_381_guest_under_the_batch_circuit_config_one_flow_one_proof(tmp_path):
    words, data = bls_program(), bls_data()
    model = rv.run(words, b"", data=data, moduli=BATCH_CIRCUIT_MODULI, curves=((P, 0),), fp2=(P,))
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(batch_circuit_toml(PARAMS))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "10"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified"] and out["total_cycles"] == model["instret"]
    assert out["chips_per_shape"][-1] == 22 + 2 + 2 + 4 + 2 + 2 + 3  # base + keccak + sha2 + two moduli + one curve + one Fp2 field + native (2) + castf (round 5)
    pv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    assert pv == bytes(model["pvs"])
    w = [int.from_bytes(pv[4 * k:4 * k + 4], "little") for k in range(8)]
    assert w[0] == 1 and w[5] == 0 and w[3] == w[4]                    # y^2 = x^3 + 4 on G1 (the equality chip's bit) and on G2 (both sides, their difference)
    assert w[7] == BLS12_381_G1[1] ** 2 % P & 0xFFFFFFFF
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0
    # the same guest under an app without the 381-bit modulus: the executor refuses the call
    (tmp_path / "small.toml").write_text(pm.TOML.format(*PARAMS))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "small.toml"), "10"], capture_output=True, text=True)
    assert r.returncode != 0 and "modular intrinsic" in r.stderr


def test_phantom_hints_guest_one_flow_one_proof(tmp_path):
    """Phantom instructions (square root / non-residue advice: tests/test_phantom_cpu.py) inside the one-statement flow: the circuit sees
    no-ops and the reads of the hint stream, the guest's own checks with the modular intrinsic (32- and 48-limb chips) are what is proven."""
    from test_vm_cpu import PHANTOM_MODULI, phantom_data, phantom_program

    words, data = phantom_program(), phantom_data()
    model = rv.run(words, b"", data=data, moduli=PHANTOM_MODULI)
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n[app_vm_config.modular]\nsupported_moduli = [\n" + ",\n".join('    "%d"' % m for m in PHANTOM_MODULI) + "\n]\n")
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "10"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified"] and out["total_cycles"] == model["instret"]
    pv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    assert pv == bytes(model["pvs"]) and [int.from_bytes(pv[4 * k:4 * k + 4], "little") for k in range(3)] == [3, 3, 4]
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0
