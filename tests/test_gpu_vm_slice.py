"""Execution -> trace hand-off -> proof (SURVEY.md 8(a) a4 + 8(f) f3): an RV32IM guest runs in the host executor
(include/zkhip_vm.hpp via vm_cli), its per-chip execution records go to the device trace generators -- program chip
(cached program + execution frequencies), execution frames, base ALU, less-than, multiplication, and the two lookup tables
they count into -- and the seven AIRs are proven in one proof from the device-resident traces.  Every trace equals the
oracle's trace from the same records, the proof bytes equal the oracle's, the host verifier accepts."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rv32_model as rv  # noqa: E402
from test_vm_cpu import mixed_program, fib_program, run_cli  # noqa: E402

import zkvm_prover_amd as z  # noqa: E402
from zkvm_prover_amd import air  # noqa: E402

pytestmark = pytest.mark.gpu
NOPV = np.zeros(0, np.uint32)
PARAMS = (1, 0, 12, 4, 4)
P = 2013265921


def program_table(words, log_program):
    """[9, 2^log_program] canonical: pc index, opcode, rd, funct3, rs1, rs2, funct7, and the instruction word split in two halves
    (rows beyond the program are zero: never executed)."""
    n = 1 << log_program
    t = np.zeros((9, n), np.uint32)
    w = np.array(words, dtype=np.uint32)
    k = len(w)
    t[0, :k] = np.arange(k)
    t[1, :k], t[2, :k], t[3, :k] = w & 0x7F, (w >> 7) & 31, (w >> 12) & 7
    t[4, :k], t[5, :k], t[6, :k] = (w >> 15) & 31, (w >> 20) & 31, w >> 25
    t[7, :k], t[8, :k] = w & 0xFFFF, w >> 16
    return t


def log2_ceil(n):
    return max(1, int(np.ceil(np.log2(max(n, 2)))))


@pytest.mark.parametrize("which,seed", [("mixed", 7), ("fib", 300)])
def test_guest_execution_to_proof(zk, ora, tmp_path, which, seed):
    words = mixed_program() if which == "mixed" else fib_program()
    r, js, rec = run_cli(tmp_path, words, int(seed).to_bytes(4, "little"))
    assert r.returncode == 0, r.stderr
    assert js["total_cycle"] == rv.run(words, int(seed).to_bytes(4, "little"))["instret"]
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(dev)  # noqa: E731
    lp, lf = log2_ceil(len(words)), log2_ceil(len(rec["pc_index"]))
    prog = program_table(words, lp)
    # ---- device trace generation from the records
    d_prog = zk.upload(prog.reshape(-1))
    d_idx = as_dev(rec["pc_index"])
    d_freq = zk.program_freq_tracegen(d_idx, lp)
    d_frames = zk.exec_frame_tracegen(d_idx, d_prog, 1 << lp, lf)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=dev)
    la, ll, lm = log2_ceil(len(rec["alu_op"])), log2_ceil(len(rec["lt_op"])), log2_ceil(len(rec["mul_b"]))
    d_alu = zk.rv32_alu_tracegen(as_dev(rec["alu_op"]), as_dev(rec["alu_b"]), as_dev(rec["alu_c"]), la, d_bw)
    d_lt = zk.rv32_lt_tracegen(as_dev(rec["lt_op"]), as_dev(rec["lt_b"]), as_dev(rec["lt_c"]), ll, d_bw)
    sx, sy = 256, 2048
    d_tab = torch.zeros(sx * sy, dtype=torch.int32, device=dev)
    d_mul = zk.rv32_mul_tracegen(as_dev(rec["mul_b"]), as_dev(rec["mul_c"]), lm, d_tab, sx, sy)
    # ---- the oracle's traces from the same records
    freq, _ = ora.program_freq_trace(rec["pc_index"], lp)
    frames, _ = ora.exec_frame_trace(rec["pc_index"], prog, lf)
    alu, xc, _ = ora.rv32_alu_trace(rec["alu_op"], rec["alu_b"], rec["alu_c"], la)
    lt, rc, _ = ora.rv32_lt_trace(rec["lt_op"], rec["lt_b"], rec["lt_c"], ll)
    mul, tc = ora.rv32_mul_trace(rec["mul_b"], rec["mul_c"], lm, sx, sy)
    bw = np.stack([rc, xc])
    assert (zk.download(d_freq) == freq).all() and (zk.download(d_frames).reshape(10, -1) == frames).all()
    assert (zk.download(d_alu).reshape(18, -1) == alu).all() and (zk.download(d_lt).reshape(18, -1) == lt).all()
    assert (zk.download(d_mul).reshape(13, -1) == mul).all()
    assert (zk.download(d_bw).reshape(2, -1) == bw).all() and (zk.download(d_tab) == tc).all()
    assert int(freq.astype(np.int64).sum()) == js["total_cycle"]
    # ---- one proof over the seven chips
    airs = [dict(program=air.program_air().program(), log_height=lp, width=10, n_pvs=0, trace=np.concatenate([prog, freq.reshape(1, -1)]), pvs=NOPV),
            dict(program=air.exec_frame_air().program(), log_height=lf, width=10, n_pvs=0, trace=frames, pvs=NOPV),
            dict(program=air.rv32_alu_core_air().program(), log_height=la, width=18, n_pvs=0, trace=alu, pvs=NOPV),
            dict(program=air.rv32_lt_core_air().program(), log_height=ll, width=18, n_pvs=0, trace=lt, pvs=NOPV),
            dict(program=air.rv32_mul_core_air().program(), log_height=lm, width=13, n_pvs=0, trace=mul, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8).program(), log_height=16, width=2, n_pvs=0, trace=bw, pvs=NOPV, prep=air.bitwise_lookup_prep(8)),
            dict(program=air.range_tuple_table_air(sx, sy).program(), log_height=19, width=1, n_pvs=0, trace=tc.reshape(1, -1), pvs=NOPV,
                 prep=air.range_tuple_prep(sx, sy))]
    pk = z.ProvingKey(zk, PARAMS, airs)
    pvs = [NOPV] * len(airs)
    proof = pk.prove([torch.cat([d_prog, d_freq]), d_frames, d_alu, d_lt, d_mul, d_bw, d_tab], pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def _memory_chips(zk, ora, rec, tamper=None):
    """Device + oracle traces of the memory access chip, the boundary chip and the 16-bit range table they send to."""
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(dev)  # noqa: E731
    acc = {k: rec["acc_" + k].copy() for k in ("as", "ptr", "prev_data", "prev_ts", "data", "ts", "is_read")}
    if tamper is not None:
        acc["data"][tamper] ^= 1          # a write that leaves another value than the next access of the cell consumes
    n, nb = len(acc["ts"]), len(rec["bnd_ptr"])
    la, lb = log2_ceil(n), log2_ceil(nb)
    order = ("as", "ptr", "prev_data", "prev_ts", "data", "ts", "is_read")
    d_acc = zk.memory_access_tracegen(*[as_dev(acc[k]) for k in order], la)
    exp_acc, bad = ora.memory_access_trace(*[acc[k] for k in order], la)
    assert bad == 0 and (zk.download(d_acc).reshape(10, -1) == exp_acc).all()
    d_bnd = zk.memory_boundary_tracegen(as_dev(rec["bnd_as"]), as_dev(rec["bnd_ptr"]), zk.upload(rec["bnd_initial"]), zk.upload(rec["bnd_final"]),
                                        as_dev(rec["bnd_final_ts"]), 2, 27, lb)
    exp_bnd, bad = ora.memory_boundary_trace(rec["bnd_as"], rec["bnd_ptr"], rec["bnd_initial"], rec["bnd_final"], rec["bnd_final_ts"], 2, 27, lb)
    assert bad == 0 and (zk.download(d_bnd).reshape(8, -1) == exp_bnd).all()
    # the range table's multiplicities, counted on the device from the requesting columns (valid rows only)
    NA, NB = 1 << la, 1 << lb
    d_rng = None
    for col in (8, 9, 4):
        d_rng = zk.range_counts_tracegen(d_acc[col * NA: col * NA + n], 16, t_counts=d_rng, accumulate=d_rng is not None)
    for col in (6, 7):
        d_rng = zk.range_counts_tracegen(d_bnd[col * NB: col * NB + nb], 16, t_counts=d_rng, accumulate=True)
    # 8 gap_hi of both chips (gaps below 2^29: a difference cannot wrap around p)
    d_rng = zk.range_counts_scaled_tracegen(d_acc[9 * NA: 9 * NA + n], 8, 16, d_rng)
    d_rng = zk.range_counts_scaled_tracegen(d_bnd[7 * NB: 7 * NB + nb], 8, 16, d_rng)
    cnt = np.zeros(1 << 16, np.int64)
    for v in (exp_acc[8][:n], exp_acc[9][:n], exp_acc[4][:n], exp_bnd[6][:nb], exp_bnd[7][:nb], 8 * exp_acc[9][:n], 8 * exp_bnd[7][:nb]):
        cnt += np.bincount(v.astype(np.int64), minlength=1 << 16)
    assert (zk.download(d_rng) == (cnt % P).astype(np.uint32)).all()
    airs = [dict(program=air.memory_access_air().program(), log_height=la, width=10, n_pvs=0, trace=exp_acc, pvs=NOPV),
            dict(program=air.memory_boundary_air().program(), log_height=lb, width=8, n_pvs=0, trace=exp_bnd, pvs=NOPV),
            dict(program=air.range_table_air(5).program(), log_height=16, width=1, n_pvs=0, trace=(cnt % P).astype(np.uint32).reshape(1, -1), pvs=NOPV,
                 prep=np.arange(1 << 16, dtype=np.uint32).reshape(1, -1))]
    return airs, [d_acc, d_bnd, d_rng]


@pytest.mark.parametrize("which,seed", [("mixed", 9), ("fib", 40)])
def test_memory_consistency_of_the_execution(zk, ora, tmp_path, which, seed):
    """The execution's memory log (register file and read-write memory as 16-bit cells) through the offline memory-checking
    argument: access chip + boundary chip + range table, traces generated on the device, balance the memory bus exactly because
    the log is a consistent history -- proof bytes == oracle, verifier accepts; one tampered write and the verifier refuses."""
    words = mixed_program() if which == "mixed" else fib_program()
    r, js, rec = run_cli(tmp_path, words, int(seed).to_bytes(4, "little"))
    assert r.returncode == 0, r.stderr
    assert air.check_trace(air.memory_access_air().program(), ora.memory_access_trace(
        *[rec["acc_" + k] for k in ("as", "ptr", "prev_data", "prev_ts", "data", "ts", "is_read")], log2_ceil(len(rec["acc_ts"])))[0], NOPV) == []
    airs, d_traces = _memory_chips(zk, ora, rec)
    pk = z.ProvingKey(zk, PARAMS, airs)
    pvs = [NOPV] * 3
    proof = pk.prove(d_traces, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    # a write whose value is not what the next access of that cell consumes: the buses no longer balance
    k = int(np.nonzero(rec["acc_is_read"] == 0)[0][3])
    bad_airs, bad_traces = _memory_chips(zk, ora, rec, tamper=k)
    bad_proof = pk.prove(bad_traces, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, bad_proof) != 0
    pk.close()


def test_prove_guest_in_cpp_end_to_end(zk, ora, tmp_path):
    """`prove_cli prove-guest`: the gen_proof_stark flow of the reference (mod.rs:342-413 -- execute, prove, encode, self-verify)
    from a guest image, all in C++ on the C ABI (include/zkhip_vm_prover.hpp): the nineteen-chip proof it writes equals, byte for byte,
    the oracle's proof over the oracle's traces of the same execution records; the JSON carries the cycle count and the public
    values like the reference's StarkProof."""
    import base64
    import json
    import subprocess

    from prover_mirror_util import CLI

    words = mixed_program()
    stdin = (77).to_bytes(4, "little")
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words))
    inp = tmp_path / "stdin.bin"
    inp.write_bytes(stdin)
    out = tmp_path / "proof.json"
    r = subprocess.run([CLI, "prove-guest", str(exe), str(inp), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, stdin)
    assert line["verified"] and line["total_cycles"] == model["instret"]
    sp = json.loads(out.read_text())
    unb64 = lambda s: base64.b64decode(s)   # noqa: E731  (fields are base64(bincode(Vec<u8>)): 8-byte length + bytes)
    proof = unb64(sp["proof"])[8:]
    assert unb64(sp["user_pvs_proof"])[8:] == model["pvs"] and sp["stat"]["total_cycles"] == model["instret"]
    heights = list(unb64(sp["baseline"])[8:])
    # the same proof from the Python side: records of the same run, the oracle's traces, the oracle's prover
    r2, js, rec = run_cli(tmp_path, words, stdin)
    assert r2.returncode == 0
    lp, lf = log2_ceil(len(words)), log2_ceil(len(rec["pc_index"]))
    la, ll, lm = log2_ceil(len(rec["alu_op"])), log2_ceil(len(rec["lt_op"])), log2_ceil(len(rec["mul_b"]))
    lacc, lbnd = log2_ceil(len(rec["acc_ts"])), log2_ceil(len(rec["bnd_ptr"]))
    lsh, lbe, lbl = log2_ceil(len(rec["shift_op"])), log2_ceil(len(rec["beq_op"])), log2_ceil(len(rec["blt_op"]))
    assert len(rec["blt_op"]) > 100 and set(rec["blt_op"].tolist()) == {0, 1, 2, 3} and set(rec["beq_op"].tolist()) == {0, 1}
    ljl, lau, ljr = log2_ceil(len(rec["jal_op"])), log2_ceil(len(rec["auipc_pc"])), log2_ceil(len(rec["jalr_pc"]))
    assert set(rec["jal_op"].tolist()) == {0, 1} and len(rec["auipc_pc"]) > 100 and len(rec["jalr_pc"]) > 100
    lmh = log2_ceil(len(rec["mulh_op"]))
    assert set(rec["mulh_op"].tolist()) == {0, 1, 2}
    lls = log2_ceil(len(rec["ls_case"]))
    assert len(set(rec["ls_case"].tolist())) >= 8
    ldv = log2_ceil(len(rec["div_op"]))
    assert set(rec["div_op"].tolist()) == {0, 1, 2, 3}
    assert heights == [lp, lf, la, ll, lm, lmh, ldv, lsh, lbe, lbl, ljl, lau, ljr, lls, 16, 19, lacc, lbnd, 16]
    prog = program_table(words, lp)
    freq, _ = ora.program_freq_trace(rec["pc_index"], lp)
    frames, _ = ora.exec_frame_trace(rec["pc_index"], prog, lf)
    alu, xc, _ = ora.rv32_alu_trace(rec["alu_op"], rec["alu_b"], rec["alu_c"], la)
    lt, rc, _ = ora.rv32_lt_trace(rec["lt_op"], rec["lt_b"], rec["lt_c"], ll)
    sx, sy = 256, 2048
    mul, tc = ora.rv32_mul_trace(rec["mul_b"], rec["mul_c"], lm, sx, sy)
    sh, rc, xc, _ = ora.rv32_shift_trace(rec["shift_op"], rec["shift_b"], rec["shift_c"], lsh, range_counts=rc, xor_counts=xc)
    mulh, tc, rc, _ = ora.rv32_mulh_trace(rec["mulh_op"], rec["mulh_b"], rec["mulh_c"], lmh, sx, sy, tuple_counts=tc, range_counts=rc)
    dv, tc, rc, _ = ora.rv32_divrem_trace(rec["div_op"], rec["div_b"], rec["div_c"], ldv, sx, sy, tuple_counts=tc, range_counts=rc)
    beq, _ = ora.rv32_branch_eq_trace(rec["beq_op"], rec["beq_a"], rec["beq_b"], rec["beq_imm"], lbe)
    blt, rc, _ = ora.rv32_branch_lt_trace(rec["blt_op"], rec["blt_a"], rec["blt_b"], rec["blt_imm"], lbl, range_counts=rc)
    jal, rc, _ = ora.rv32_jal_lui_trace(rec["jal_op"], rec["jal_pc"], rec["jal_imm"], ljl, range_counts=rc)
    auipc, rc, _ = ora.rv32_auipc_trace(rec["auipc_pc"], rec["auipc_imm"], lau, range_counts=rc)
    jalr, rc, _ = ora.rv32_jalr_trace(rec["jalr_pc"], rec["jalr_rs1"], rec["jalr_imm"], ljr, range_counts=rc)
    ls, rc, _ = ora.rv32_loadstore_trace(rec["ls_case"], rec["ls_read"], rec["ls_prev"], lls, range_counts=rc)
    order = ("as", "ptr", "prev_data", "prev_ts", "data", "ts", "is_read")
    acc, _ = ora.memory_access_trace(*[rec["acc_" + k] for k in order], lacc)
    bnd, _ = ora.memory_boundary_trace(rec["bnd_as"], rec["bnd_ptr"], rec["bnd_initial"], rec["bnd_final"], rec["bnd_final_ts"], 2, 27, lbnd)
    n, nb = len(rec["acc_ts"]), len(rec["bnd_ptr"])
    cnt = np.zeros(1 << 16, np.int64)
    for v in (acc[8][:n], acc[9][:n], acc[4][:n], bnd[6][:nb], bnd[7][:nb], 8 * acc[9][:n], 8 * bnd[7][:nb]):
        cnt += np.bincount(v.astype(np.int64), minlength=1 << 16)
    A = lambda program, lh, w, tr, prep=None: dict(program=program, log_height=lh, width=w, n_pvs=0, trace=tr, pvs=NOPV, **({"prep": prep} if prep is not None else {}))  # noqa: E731
    airs = [A(air.program_air().program(), lp, 10, np.concatenate([prog, freq.reshape(1, -1)])),
            A(air.exec_frame_air().program(), lf, 10, frames),
            A(air.rv32_alu_core_air().program(), la, 18, alu),
            A(air.rv32_lt_core_air().program(), ll, 18, lt),
            A(air.rv32_mul_core_air().program(), lm, 13, mul),
            A(air.rv32_mulh_core_air().program(), lmh, 21, mulh),
            A(air.rv32_divrem_core_air().program(), ldv, 41, dv),
            A(air.rv32_shift_core_air().program(), lsh, 32, sh),
            A(air.rv32_branch_eq_core_air().program(), lbe, 17, beq),
            A(air.rv32_branch_lt_core_air().program(), lbl, 23, blt),
            A(air.rv32_jal_lui_core_air().program(), ljl, 9, jal),
            A(air.rv32_auipc_core_air().program(), lau, 14, auipc),
            A(air.rv32_jalr_core_air().program(), ljr, 20, jalr),
            A(air.rv32_loadstore_core_air().program(), lls, 33, ls),
            A(air.bitwise_lookup_air(8).program(), 16, 2, np.stack([rc, xc]), air.bitwise_lookup_prep(8)),
            A(air.range_tuple_table_air(sx, sy).program(), 19, 1, tc.reshape(1, -1), air.range_tuple_prep(sx, sy)),
            A(air.memory_access_air().program(), lacc, 10, acc),
            A(air.memory_boundary_air().program(), lbnd, 8, bnd),
            A(air.range_table_air(5).program(), 16, 1, (cnt % P).astype(np.uint32).reshape(1, -1), np.arange(1 << 16, dtype=np.uint32).reshape(1, -1))]
    params = (1, 0, 100, 16, 16)
    assert proof == ora.stark_prove(params, airs).tobytes()


def test_continuation_segments_proven_over_lanes(tmp_path):
    """`prove_cli prove-guest ... <segment_instr> <inflight>`: the run is cut into segments, each proven on its own (self-verified)
    over `inflight` prover lanes on the GPU; the cycle counts add up and the last segment carries the public values."""
    import base64
    import json
    import subprocess

    from prover_mirror_util import CLI

    words = mixed_program()
    stdin = (99).to_bytes(4, "little")
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words))
    inp = tmp_path / "stdin.bin"
    inp.write_bytes(stdin)
    out = tmp_path / "segments"
    out.mkdir()
    toml = tmp_path / "openvm.toml"
    toml.write_text("[app_fri_params.fri_params]\nlog_blowup = 1\nlog_final_poly_len = 0\nnum_queries = 30\n"
                    "commit_proof_of_work_bits = 4\nquery_proof_of_work_bits = 8\n")
    r = subprocess.run([CLI, "prove-guest", str(exe), str(inp), str(out), str(toml), "0", "700", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, stdin)
    n_seg = (model["instret"] + 699) // 700
    assert line["verified"] and line["total_cycles"] == model["instret"] and line["segments"] == n_seg and line["inflight"] == 3
    cycles = 0
    for k in range(n_seg):
        sp = json.loads((out / ("segment-%d.json" % k)).read_text())
        cycles += sp["stat"]["total_cycles"]
        pv = base64.b64decode(sp["user_pvs_proof"])[8:]
        assert pv == (model["pvs"] if k == n_seg - 1 else b"")
        assert len(base64.b64decode(sp["proof"])) > 1000
    assert cycles == model["instret"]


def test_prove_guest_reports_guest_failures(tmp_path):
    """Errors of the execution step surface as the reference's Error::GenProof (mod.rs:318-319): a guest that exits non-zero, one
    whose public values stay zero, an image that is not an executable."""
    import subprocess

    from prover_mirror_util import CLI

    cases = {"exit3": (rv.assemble([("addi", 10, 0, 3), ("addi", 17, 0, 93), ("ecall",)]), "exited with code 3"),
             "zero_pvs": (rv.assemble([("addi", 10, 0, 0), ("addi", 17, 0, 93), ("ecall",)]), "public_values are all 0s")}
    for name, (words, msg) in cases.items():
        exe = tmp_path / (name + ".elf")
        exe.write_bytes(rv.elf_bytes(words))
        r = subprocess.run([CLI, "prove-guest", str(exe), "-", str(tmp_path / "out.json")], capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr and "kind 3" in r.stderr, r.stderr   # kind 3 = Error::GenProof
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"not an executable at all")
    r = subprocess.run([CLI, "prove-guest", str(junk), "-", str(tmp_path / "out.json")], capture_output=True, text=True)
    assert r.returncode != 0
