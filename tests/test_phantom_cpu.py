"""CPU: phantom instructions (OpenVM's algebra extension ModularPhantom::{HintNonQr, HintSqrt}; VERDICT round 3 "a phantom-hint
sub-executor").  A phantom instruction is a no-op to the circuit (here: a FENCE-coded word, fm = 0101) whose sub-executor leaves advice in
the hint stream; the guest CHECKS what it reads with the modular intrinsic -- r^2 = x for a square, r^2 = x z with the hinted non-residue
z otherwise.  The C++ executor's run == the independent Python model's (tests/rv32_model.py: its own Tonelli - Shanks), for the secp256k1
base field (p = 3 mod 4), its scalar field (n = 1 mod 4: the loop) and BLS12-381's base field (48-byte operands); the pairing
extension's final-exponentiation hint is refused by name."""
import json
import subprocess

import numpy as np

import rv32_model as rv
import vm2_util as v2
from test_vm_cpu import PHANTOM_MODULI, phantom_cases, phantom_data, phantom_program

A0, A7, S0 = 10, 17, 8


def dump(tmp_path, words, data, moduli):
    (tmp_path / "exe.bin").write_bytes(rv.exe_bytes(words, data=data))
    (tmp_path / "stdin.bin").write_bytes(b"")
    (tmp_path / "moduli.toml").write_text("[app_vm_config.modular]\nsupported_moduli = [\n" + ",\n".join('    "%d"' % m for m in moduli) + "\n]\n")
    return subprocess.run([v2.CLI, "dump-segments", str(tmp_path / "exe.bin"), str(tmp_path / "stdin.bin"), str(tmp_path), "10", "0", "0", str(tmp_path / "moduli.toml")],
                          capture_output=True, text=True)


def test_square_roots_and_non_residues_from_the_hint_stream(tmp_path):
    words, data = phantom_program(), phantom_data()
    model = rv.run(words, b"", data=data, moduli=PHANTOM_MODULI)
    pv = [int.from_bytes(bytes(model["pvs"][4 * k:4 * k + 4]), "little") for k in range(8)]
    for k, (mi, x) in enumerate(phantom_cases()):
        p = PHANTOM_MODULI[mi]
        assert pv[k] == (3 if pow(x, (p - 1) // 2, p) == 1 else 4)      # s | [r^2 = x] << 1 | [r^2 = x z] << 2: the guest's own check
    assert pv[:3] == [3, 3, 4]
    r = dump(tmp_path, words, data, PHANTOM_MODULI)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"]) and info["total_cycle"] == model["instret"]
    root0 = int.from_bytes(bytes(model["pvs"][28:32]), "little")
    gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    assert root0 in (gy & 0xFFFFFFFF, (PHANTOM_MODULI[0] - gy) & 0xFFFFFFFF)          # a root of Gy^2 is +- Gy (the published generator)
    # a phantom instruction changes nothing the circuit sees: the segment records hold it as the no-op `add x0, x0, 0`
    ids = np.fromfile(tmp_path / "air_ids.u32", dtype=np.uint32).tolist()
    assert len(ids) == 22 + 2 * len(PHANTOM_MODULI)


def test_the_pairing_hint_is_refused_by_name(tmp_path):
    words = rv.assemble(rv.li(S0, 0x00400000) + [("phantom", 2, S0), ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)])
    r = dump(tmp_path, words, bytes(64), PHANTOM_MODULI)
    assert r.returncode != 0 and "final-exponentiation hint is not built" in r.stderr
    # an ordinary FENCE stays a no-op
    words = rv.assemble([("fence",), ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)])
    r = dump(tmp_path, words, bytes(64), PHANTOM_MODULI)
    assert r.returncode == 0, r.stderr


def test_phantom_and_fence_words_are_proven_as_no_ops(tmp_path, monkeypatch):
    """The segment statements of a guest with phantom instructions and a FENCE (the two 32-limb cases: the CPU twins of the modular chip
    are 32-limb): every trace satisfies its AIR, every bus balances -- the frame chip hands the ALU chip `add x0, x0, 0` for each of those
    words.  (Round 4 found the executor recording nothing for a FENCE: a segment with one could not be proven.)"""
    import test_vm_cpu as t
    from zkvm_prover_amd import air

    cases = t.phantom_cases()[:2]
    monkeypatch.setattr(t, "PHANTOM_MODULI", t.PHANTOM_MODULI[:2])
    monkeypatch.setattr(t, "phantom_cases", lambda: cases)
    words, data = t.phantom_program(), t.phantom_data()
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 10, 0, 0, t.PHANTOM_MODULI)
    model = rv.run(words, b"", data=data, moduli=t.PHANTOM_MODULI)
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"])
    assert sum(int((s["alu_op"] == 0).sum()) for s in segs) >= 5        # two phantom pairs and the FENCE among the additions
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, 0x00200000, H)
        for a, d in enumerate(inst):
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], (k, a)
        assert v2.bus_imbalance(inst) == {}, k
