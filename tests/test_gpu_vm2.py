"""GPU: the one-statement VM flow (include/zkhip_vm_flow.hpp).
  * device trace generation of a segment == the CPU twins, cell for cell, for all 22 chips (24 with the keccak extension); the HIP proof == the oracle's, byte
    for byte; tampered records (a wrong instruction result, swapped operands) give proofs the verifier refuses;
  * `prove_cli prove-elf`: ELF + input bytes -> execution -> segment proofs -> aggregation tree -> ONE root proof, self-verified:
    the root statement starts at the guest's entry pc on the guest image's memory root, ends at pc = 0, and the public values open
    in the final root; segments that do not chain are refused by the aggregation circuit."""
import json
import subprocess

import numpy as np
import pytest
import torch

import zkvm_prover_amd as z

import prover_mirror_util as pm
import recursion_util as ru
import rv32_model as rv
import vm2_util as v2
from test_vm_cpu import (ALL_EXT_MSG, CHUNK_CIRCUIT_CURVES, CHUNK_CIRCUIT_MODULI, INT256_OPERANDS, all_extensions_data, all_extensions_program, chunk_circuit_toml, SECP256K1_GX, SECP256K1_GY, SECP256K1_N, SECP256K1_P, fib_program, int256_data, int256_program, keccak_data,
                         keccak_program, mixed_program, modmul_data, modmul_program, sha256_data, sha256_program, EC_CURVES, ec_data, ec_program, BN254_P, fp2_data, fp2_program)

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
PC_BASE = 0x00200000
N_STMT = 8 + 9 + 9 + 8 + 16   # a root's statement: app digest, start / end state (pc, memory root), accumulator, leaf / internal commitment


def device_traces(zk, rec, prog, H):
    """The segment's traces through the device generators, as SegmentProver::prove drives them."""
    dev = zk.device
    D = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint32).view(np.int32)).to(dev)  # noqa: E731
    T = [None] * v2.N_AIRS
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=dev)
    d_tup = torch.zeros(1 << 19, dtype=torch.int32, device=dev)
    d_prog = zk.upload(prog.reshape(-1))
    idx = D(rec["pc_index"])
    T[v2.A_PROGRAM] = zk.program_freq_tracegen(idx, H[v2.A_PROGRAM])
    T[v2.A_FRAME] = zk.vm_frame_tracegen(idx, D(rec["f_x"]), D(rec["f_y"]), D(rec["f_z"]), D(rec["f_rdprev"]), D(rec["f_pcinc"]), D(rec["f_pts1"]),
                                         D(rec["f_pts2"]), D(rec["f_pts3"]), d_prog, 1 << H[v2.A_PROGRAM], H[v2.A_FRAME])
    T[v2.A_ALU] = zk.rv32_alu_tracegen(D(rec["alu_op"]), D(rec["alu_b"]), D(rec["alu_c"]), H[v2.A_ALU], d_bw)
    T[v2.A_LT] = zk.rv32_lt_tracegen(D(rec["lt_op"]), D(rec["lt_b"]), D(rec["lt_c"]), H[v2.A_LT], d_bw)
    T[v2.A_MUL] = zk.rv32_mul_tracegen(D(rec["mul_b"]), D(rec["mul_c"]), H[v2.A_MUL], d_tup, v2.SX, v2.SY)
    T[v2.A_MULH] = zk.rv32_mulh_tracegen(D(rec["mulh_op"]), D(rec["mulh_b"]), D(rec["mulh_c"]), H[v2.A_MULH], d_tup, d_bw, v2.SX, v2.SY)
    T[v2.A_DIVREM] = zk.rv32_divrem_tracegen(D(rec["div_op"]), D(rec["div_b"]), D(rec["div_c"]), H[v2.A_DIVREM], d_tup, d_bw, v2.SX, v2.SY)
    T[v2.A_SHIFT] = zk.rv32_shift_tracegen(D(rec["shift_op"]), D(rec["shift_b"]), D(rec["shift_c"]), H[v2.A_SHIFT], d_bw)
    T[v2.A_BEQ] = zk.rv32_branch_eq_tracegen(D(rec["beq_op"]), D(rec["beq_a"]), D(rec["beq_b"]), D(rec["beq_imm"]), H[v2.A_BEQ])
    T[v2.A_BLT] = zk.rv32_branch_lt_tracegen(D(rec["blt_op"]), D(rec["blt_a"]), D(rec["blt_b"]), D(rec["blt_imm"]), H[v2.A_BLT], d_bw)
    T[v2.A_JAL_LUI] = zk.rv32_jal_lui_tracegen(D(rec["jal_op"]), D(rec["jal_pc"]), D(rec["jal_imm"]), H[v2.A_JAL_LUI], d_bw)
    T[v2.A_AUIPC] = zk.rv32_auipc_tracegen(D(rec["auipc_pc"]), D(rec["auipc_imm"]), H[v2.A_AUIPC], d_bw)
    T[v2.A_JALR] = zk.rv32_jalr_tracegen(D(rec["jalr_pc"]), D(rec["jalr_rs1"]), D(rec["jalr_imm"]), H[v2.A_JALR], d_bw)
    T[v2.A_LS] = zk.vm_loadstore_tracegen(D(rec["ls_case"]), D(rec["ls_read"]), D(rec["ls_prev"]), D(rec["ls_ts"]), D(rec["ls_base"]), D(rec["ls_imm"]),
                                          D(rec["ls_pts"]), H[v2.A_LS], d_bw)
    n_ec, n_leaf, n_mk = len(rec["ecall_rows"]) // v2.ECALL_WIDTH, len(rec["leaf_rows"]) // v2.LEAF_WIDTH, len(rec["merkle_rows"]) // v2.MERKLE_WIDTH
    leaf_pad = np.zeros(v2.LEAF_WIDTH, np.uint32)
    leaf_pad[0] = 1
    T[v2.A_ECALL] = zk.rows_tracegen(D(rec["ecall_rows"]), n_ec, v2.ECALL_WIDTH, H[v2.A_ECALL])
    T[v2.A_LEAF] = zk.rows_tracegen(D(rec["leaf_rows"]), n_leaf, v2.LEAF_WIDTH, H[v2.A_LEAF], leaf_pad)
    T[v2.A_MERKLE] = zk.rows_tracegen(D(rec["merkle_rows"]), n_mk, v2.MERKLE_WIDTH, H[v2.A_MERKLE])
    ts_end = int(rec["meta"][2])
    T[v2.A_CONNECTOR] = zk.rows_tracegen(D(np.array([ts_end & 0xFFFF, ts_end >> 16], np.uint32)), 1, 2, 0)
    p2in = zk.upload(rec["p2_inputs"])
    T[v2.A_POSEIDON2] = zk.vm_poseidon2_tracegen(p2in, len(rec["p2_inputs"]) // 16, H[v2.A_POSEIDON2])
    if v2.A_SHA256 in H.ids:
        n_sh = len(rec["sha_ts"])
        T[v2.A_SHA256] = zk.vm_sha256_tracegen(D(rec["sha_blocks"]) if n_sh else None, D(rec["sha_ts"]) if n_sh else None, n_sh, H[v2.A_SHA256])
        T[v2.A_SHA256_IO] = zk.rows_tracegen(D(rec["shaio_rows"]), len(rec["shaio_rows"]) // v2.SHA_IO_WIDTH, v2.SHA_IO_WIDTH, H[v2.A_SHA256_IO])
    if v2.A_KECCAK in H.ids:
        n_kk = len(rec["kk_ts"])
        T[v2.A_KECCAK] = zk.vm_keccak_tracegen(D(rec["kk_states"]) if n_kk else None, D(rec["kk_ts"]) if n_kk else None, n_kk, H[v2.A_KECCAK])
        T[v2.A_KECCAK_IO] = zk.rows_tracegen(D(rec["kio_rows"]), len(rec["kio_rows"]) // v2.KECCAK_IO_WIDTH, v2.KECCAK_IO_WIDTH, H[v2.A_KECCAK_IO])
    if v2.A_INT256 in H.ids:   # the 256-bit ALU counts its own lookups into the XOR column
        n_i = len(rec["i256_ts"])
        T[v2.A_INT256] = zk.vm_int256_tracegen(D(rec["i256_records"]) if n_i else None, D(rec["i256_ts"]) if n_i else None, n_i, H[v2.A_INT256], d_bw)
        T[v2.A_INT256_IO] = zk.rows_tracegen(D(rec["i256io_rows"]), len(rec["i256io_rows"]) // v2.INT256_IO_WIDTH, v2.INT256_IO_WIDTH, H[v2.A_INT256_IO])
        n_m = len(rec["mul256_ts"])
        T[v2.A_MUL256] = zk.vm_mul256_tracegen(D(rec["mul256_records"]) if n_m else None, D(rec["mul256_ts"]) if n_m else None, n_m, H[v2.A_MUL256], d_bw, d_tup,
                                               v2.SX, v2.SY)
        n_c = len(rec["cmp256_ts"])
        T[v2.A_CMP256] = zk.int256_cmp_tracegen(D(rec["cmp256_records"]) if n_c else None, n_c, H[v2.A_CMP256], d_bw,
                                                t_ts=D(rec["cmp256_ts"]) if n_c else torch.zeros(1, dtype=torch.int32, device=dev))
        n_s = len(rec["sh256_ts"])
        T[v2.A_SHIFT256] = zk.int256_shift_tracegen(D(rec["sh256_records"]) if n_s else None, n_s, H[v2.A_SHIFT256], d_bw,
                                                    t_ts=D(rec["sh256_ts"]) if n_s else torch.zeros(1, dtype=torch.int32, device=dev))
    for i, p_ in enumerate(H.moduli):   # the multiplication chips count their own lookups into the two tables
        n_mm = len(rec["mm_ts_%d" % i])
        T[v2.A_MODMUL(i)] = zk.vm_modmul_tracegen(p_, D(rec["mm_records_%d" % i]) if n_mm else None, D(rec["mm_ts_%d" % i]) if n_mm else None, n_mm,
                                                  H[v2.A_MODMUL(i)], d_bw, d_tup, v2.SX, v2.SY)
        T[v2.A_MODMUL_IO(i)] = zk.rows_tracegen(D(rec["mmio_rows_%d" % i]), len(rec["mmio_rows_%d" % i]) // v2.MODMUL_IO_WIDTH, v2.MODMUL_IO_WIDTH,
                                                H[v2.A_MODMUL_IO(i)])
    for i, (p_, a_) in enumerate(H.curves):   # the point chips count their own lookups into the two tables
        n_ec = len(rec["ec_ts_%d" % i])
        T[v2.A_EC(i)] = zk.ec_tracegen(p_, a_, D(rec["ec_records_%d" % i]) if n_ec else None, n_ec, H[v2.A_EC(i)], d_bw, d_tup, v2.SX, v2.SY,
                                       t_ts=D(rec["ec_ts_%d" % i]) if n_ec else torch.zeros(1, dtype=torch.int32, device=dev))
        T[v2.A_EC_IO(i)] = zk.rows_tracegen(D(rec["ecio_rows_%d" % i]), len(rec["ecio_rows_%d" % i]) // v2.EC_IO_WIDTH, v2.EC_IO_WIDTH, H[v2.A_EC_IO(i)])
    for i, p_ in enumerate(H.fp2):   # the Fp2 chips count their own lookups into the two tables
        n_f = len(rec["fp2_ts_%d" % i])
        T[v2.A_FP2(i)] = zk.fp2_tracegen(p_, D(rec["fp2_records_%d" % i]) if n_f else None, n_f, H[v2.A_FP2(i)], d_bw, d_tup, v2.SX, v2.SY,
                                         t_ts=D(rec["fp2_ts_%d" % i]) if n_f else torch.zeros(1, dtype=torch.int32, device=dev))
        T[v2.A_FP2_IO(i)] = zk.rows_tracegen(D(rec["fp2io_rows_%d" % i]), len(rec["fp2io_rows_%d" % i]) // v2.EC_IO_WIDTH, v2.EC_IO_WIDTH, H[v2.A_FP2_IO(i)])
    n_nat, n_next, n_cf = len(rec["nat_records"]) // 9, len(rec["next_records"]) // 27, len(rec["castf_records"]) // 6
    if v2.A_NATIVE_ARITH in H.ids:   # native chips: ONE row per call, made on the device from the call's record
        T[v2.A_NATIVE_ARITH] = zk.vm_native_tracegen("arith", D(rec["nat_records"]) if n_nat else None, n_nat, H[v2.A_NATIVE_ARITH])
        T[v2.A_NATIVE_EXT] = zk.vm_native_tracegen("ext", D(rec["next_records"]) if n_next else None, n_next, H[v2.A_NATIVE_EXT])
    if v2.A_CASTF in H.ids:
        T[v2.A_CASTF] = zk.vm_native_tracegen("castf", D(rec["castf_records"]) if n_cf else None, n_cf, H[v2.A_CASTF])
    n, nls = len(rec["f_x"]), len(rec["ls_case"])
    zeros = torch.zeros(max(n, nls, 64), dtype=torch.int32, device=dev)
    NF, NL = 1 << H[v2.A_FRAME], 1 << H[v2.A_LS]
    for q in (18, 20, 22, 24, 26, 28):
        zk.bitwise_lookup_tracegen(T[v2.A_FRAME][q * NF:q * NF + n], T[v2.A_FRAME][(q + 1) * NF:(q + 1) * NF + n], zeros[:n], 8, d_bw, accumulate=True)
    for q in (0, 2, 4, 6):
        if nls:
            zk.bitwise_lookup_tracegen(T[v2.A_LS][q * NL:q * NL + nls], T[v2.A_LS][(q + 1) * NL:(q + 1) * NL + nls], zeros[:nls], 8, d_bw, accumulate=True)
    d_rng = None
    for q in (35, 38, 41):   # the adapters' timestamp gaps: gap_lo in the range table, (0, gap_hi) in the range-tuple table
        d_rng = zk.range_counts_tracegen(T[v2.A_FRAME][q * NF:q * NF + n], 16, t_counts=d_rng, accumulate=d_rng is not None)
        zk.range_tuple_counts_tracegen(zeros[:n], T[v2.A_FRAME][(q + 1) * NF:(q + 1) * NF + n], v2.SX, v2.SY, t_counts=d_tup, accumulate=True)
    if nls:
        d_rng = zk.range_counts_tracegen(T[v2.A_LS][46 * NL:46 * NL + nls], 16, t_counts=d_rng, accumulate=True)
        zk.range_tuple_counts_tracegen(zeros[:nls], T[v2.A_LS][47 * NL:47 * NL + nls], v2.SX, v2.SY, t_counts=d_tup, accumulate=True)
    for q, s in ((40, 1), (41, 4), (44, 4), (44, 1), (41, 1)):
        col = T[v2.A_LS][q * NL:q * NL + nls]
        if not nls:
            continue
        d_rng = zk.range_counts_tracegen(col, 16, t_counts=d_rng, accumulate=True) if s == 1 else zk.range_counts_scaled_tracegen(col, s, 16, d_rng)
    # the native chips' lookups, counted from their trace columns as the flow does (include/zkhip_vm_flow.hpp)
    def col(a, q, cnt):
        N_ = 1 << H[a]
        return T[a][q * N_:q * N_ + cnt]

    def access(a, first, cnt):
        nonlocal d_rng
        if cnt:
            d_rng = zk.range_counts_tracegen(col(a, first + 1, cnt), 16, t_counts=d_rng, accumulate=True)
            zk.range_tuple_counts_tracegen(zeros[:cnt], col(a, first + 2, cnt), v2.SX, v2.SY, t_counts=d_tup, accumulate=True)

    if v2.A_NATIVE_ARITH in H.ids:
        for q in (6, 7, 8):
            if n_nat:
                d_rng = zk.range_counts_tracegen(col(v2.A_NATIVE_ARITH, q, n_nat), 16, t_counts=d_rng, accumulate=True)
        for q in (18, 21, 24):
            access(v2.A_NATIVE_ARITH, q, n_nat)
        for q in [18 + 5 * i + k for i in range(4) for k in range(3)]:
            if n_next:
                d_rng = zk.range_counts_tracegen(col(v2.A_NATIVE_EXT, q, n_next), 16, t_counts=d_rng, accumulate=True)
        for k in range(12):
            access(v2.A_NATIVE_EXT, 54 + 3 * k, n_next)
    if v2.A_CASTF in H.ids and n_cf:
        for qx, qy in ((2, 3), (4, 5)):
            zk.bitwise_lookup_tracegen(col(v2.A_CASTF, qx, n_cf), col(v2.A_CASTF, qy, n_cf), zeros[:n_cf], 8, d_bw, accumulate=True)
        zk.bitwise_lookup_tracegen(col(v2.A_CASTF, 6, n_cf), zeros[:n_cf], zeros[:n_cf], 8, d_bw, accumulate=True)
        access(v2.A_CASTF, 9, n_cf), access(v2.A_CASTF, 12, n_cf)
    misc = [ts_end & 0xFFFF, 8 * (ts_end >> 16), ts_end >> 16]
    tup_y, bw_x, bw_y = [], [], []
    for row in rec["ecall_rows"].reshape(-1, v2.ECALL_WIDTH):
        if row[15]:
            misc += [int(row[17]) * 8192, int(row[17]), int(row[22]), int(row[25])]
            tup_y += [int(row[23]), int(row[26])]
        if row[16]:
            bw_x += [int(row[10]), int(row[12])]
            bw_y += [int(row[11]), int(row[13])]
        if row[30] or row[31] or row[32] or row[33] or row[34] or row[35]:
            misc += [int(row[22])]
            tup_y += [int(row[23])]
        if row[37]:                                          # a 256-bit branch: the a2 read's gap, the offset's sign split
            misc += [int(row[43]), (int(row[40]) - 32768 * int(row[41])) * 2]
            tup_y += [int(row[44])]
        if row[27] or row[29] or row[30] or row[31] or row[32] or row[33] or row[34] or row[35] or row[36]:
            misc += [int(row[28]) * 1024, int(row[28]), int(row[9]) * 1024]
    for row in rec["kio_rows"].reshape(-1, v2.KECCAK_IO_WIDTH):
        misc += [int(row[36]), int(row[39])]
        tup_y += [int(row[37]), int(row[40])]
    for row in rec["shaio_rows"].reshape(-1, v2.SHA_IO_WIDTH):
        misc += [int(row[31])]
        tup_y += [int(row[32])]
    for row in rec["i256io_rows"].reshape(-1, v2.INT256_IO_WIDTH):
        misc += [int(row[31])]
        tup_y += [int(row[32])]
    for i in range(len(H.moduli)):
        for row in rec["mmio_rows_%d" % i].reshape(-1, v2.MODMUL_IO_WIDTH):
            misc += [int(row[31])]
            tup_y += [int(row[32])]
    for i in range(len(H.curves)):
        for row in rec["ecio_rows_%d" % i].reshape(-1, v2.EC_IO_WIDTH):
            misc += [int(row[55])]
            tup_y += [int(row[56])]
    for i in range(len(H.fp2)):
        for row in rec["fp2io_rows_%d" % i].reshape(-1, v2.EC_IO_WIDTH):
            misc += [int(row[55])]
            tup_y += [int(row[56])]
    for row in rec["leaf_rows"].reshape(-1, v2.LEAF_WIDTH):
        misc += [int(row[39]), int(row[40]) * 16, int(row[40]), int(row[41]), int(row[42]) * 64, int(row[42])]
    d_rng = zk.range_counts_tracegen(zk.upload(np.array(misc, np.uint32)), 16, t_counts=d_rng, accumulate=True)
    if tup_y:
        # (a zero column of the lists' own length: at tiny frames the adapters' rows outnumber the frame's)
        zk.range_tuple_counts_tracegen(torch.zeros(len(tup_y), dtype=torch.int32, device=dev), zk.upload(np.array(tup_y, np.uint32)), v2.SX, v2.SY, t_counts=d_tup, accumulate=True)
    if bw_x:
        zk.bitwise_lookup_tracegen(zk.upload(np.array(bw_x, np.uint32)), zk.upload(np.array(bw_y, np.uint32)), torch.zeros(len(bw_x), dtype=torch.int32, device=dev), 8, d_bw, accumulate=True)
    T[v2.A_BITWISE], T[v2.A_RANGE_TUPLE], T[v2.A_RANGE] = d_bw, d_tup, d_rng
    return [T[a] for a in H.ids]   # proof order


@pytest.fixture(scope="module")
def mixed(tmp_path_factory):
    words = mixed_program()
    info, heights, segs, image_root, pv_open = v2.dump_segments(tmp_path_factory.mktemp("mixed"), rv.exe_bytes(words), (7).to_bytes(4, "little"), 9)
    return dict(words=words, heights=heights, segs=segs)


def test_device_traces_equal_the_twins_and_the_proof_equals_the_oracles(zk, ora, mixed):
    rec, H = mixed["segs"][3], mixed["heights"]
    inst = v2.segment_instance(rec, mixed["words"], PC_BASE, H)
    prog = v2.program_table(mixed["words"], PC_BASE, H[0])
    T = device_traces(zk, rec, prog, H)
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "chip %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    # tampered records: the device generates the traces all the same; their proofs are refused
    cls = prog[1][rec["pc_index"]]
    alu_rows = np.nonzero(cls == 0)[0]

    def wrong_result(r):
        r["f_z"][int(alu_rows[3])] ^= 4

    def swapped(r):
        k = next(int(q) for q in alu_rows if prog[2][rec["pc_index"][q]] == 1 and rec["f_x"][q] != rec["f_y"][q])
        r["f_x"][k], r["f_y"][k] = r["f_y"][k], r["f_x"][k]

    for edit in (wrong_result, swapped):
        bad = {n: v.copy() for n, v in rec.items()}
        edit(bad)
        proof = pk.prove(device_traces(zk, bad, prog, H), pvs)
        assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) != 0
    pk.close()


@pytest.mark.parametrize("which,seed", [("mixed", 21), ("fib", 500)])
def test_prove_elf_one_flow_one_proof(tmp_path, which, seed):
    words = mixed_program() if which == "mixed" else fib_program()
    stdin = int(seed).to_bytes(4, "little")
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words))
    (tmp_path / "stdin.bin").write_bytes(stdin)
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), str(tmp_path / "stdin.bin"), str(tmp_path), str(tmp_path / "openvm.toml"), "9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, stdin)
    assert info["verified"] and info["total_cycles"] == model["instret"] and info["segments"] >= 3 and info["levels"] >= 2
    root = json.loads((tmp_path / "root.json").read_text())
    upv = pm.un_b64_bincode(root["user_pvs_proof"])
    stmt = np.frombuffer(upv[:4 * N_STMT], dtype=np.uint32)
    assert upv[4 * N_STMT:4 * N_STMT + 32] == bytes(model["pvs"]) and len(upv) == 4 * N_STMT + 32 + 4 * 2 * 8 * 28
    assert int(stmt[8]) == PC_BASE and int(stmt[17]) == 0                                           # from the entry point to the exit
    cfg = str(tmp_path / "openvm.toml")
    # the root proof verifies under the root verifying key alone; the whole statement about this guest with verify-guest
    assert pm.run_cli("verify", str(tmp_path / "root.vk"), cfg, str(tmp_path / "root.json")).returncode == 0
    r = pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), cfg, str(tmp_path / "root.json"))
    assert r.returncode == 0 and bytes(model["pvs"]).hex() in r.stdout, r.stderr

    def variant(edit, name):
        b = bytearray(upv)
        edit(b)
        sp = dict(root)
        sp["user_pvs_proof"] = pm.b64_bincode(bytes(b))
        (tmp_path / name).write_text(json.dumps(sp))
        return str(tmp_path / name)

    lie = variant(lambda b: b.__setitem__(4 * 20, b[4 * 20] ^ 1), "lie_root.json")                    # another final memory root
    assert pm.run_cli("verify", str(tmp_path / "root.vk"), cfg, lie).returncode != 0
    lie = variant(lambda b: b.__setitem__(4 * N_STMT, b[4 * N_STMT] ^ 1), "lie_pv.json")                      # other public values
    assert pm.run_cli("verify", str(tmp_path / "root.vk"), cfg, lie).returncode == 0                  # (the node proof does not see them)
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), cfg, lie).returncode != 0  # the openings do
    # another initial memory image: the statement starts elsewhere (the PROGRAM is pinned by root.vk: the segment key holds the
    # commitment of the decoded program table, and every circuit above holds its child's key)
    other = tmp_path / "other.elf"
    other.write_bytes(rv.elf_bytes(words, data=b"\x01\x02\x03\x04"))
    assert pm.run_cli("verify-guest", str(other), str(tmp_path / "root.vk"), cfg, str(tmp_path / "root.json")).returncode != 0


def test_one_aggregation_key_for_every_depth(tmp_path):
    """ONE aggregation key (crates/prover/src/prover/mod.rs:147-170, crates/verifier/src/verifier.rs:96-111): the same guest run on a short
    and on a long input gives trees of different depth whose roots verify under ONE key -- root.vk is byte-identical, each root verifies
    under the other run's key; a key with another leaf commitment (another app) or another app digest refuses both."""
    words = fib_program()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words))
    cfg = tmp_path / "openvm.toml"
    cfg.write_text(pm.TOML.format(*PARAMS))
    infos = {}
    for name, n in (("short", 100), ("long", 3000)):
        d = tmp_path / name
        d.mkdir()
        (d / "stdin.bin").write_bytes(int(n).to_bytes(4, "little"))
        r = subprocess.run([pm.CLI, "prove-elf", str(exe), str(d / "stdin.bin"), str(d), str(cfg), "9"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        infos[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert infos["short"]["levels"] == 2 and infos["short"]["segments"] <= 4      # one leaf node wrapped by one internal node
    assert infos["long"]["levels"] >= 4 and infos["long"]["segments"] > 27
    key = (tmp_path / "short" / "root.vk").read_bytes()
    assert key == (tmp_path / "long" / "root.vk").read_bytes()
    for vk_of, root_of in (("short", "long"), ("long", "short")):
        vk, root = str(tmp_path / vk_of / "root.vk"), str(tmp_path / root_of / "root.json")
        assert pm.run_cli("verify", vk, str(cfg), root).returncode == 0
        assert pm.run_cli("verify-guest", str(exe), vk, str(cfg), root).returncode == 0
    # the key's trailer: [magic | leaf commitment (8) | app digest (8)]; another leaf commitment / app digest: refused
    for word in (1, 9):
        bad = bytearray(key)
        bad[len(key) - 4 * 17 + 4 * word] ^= 1
        (tmp_path / "bad.vk").write_bytes(bytes(bad))
        for name in ("short", "long"):
            r = pm.run_cli("verify", str(tmp_path / "bad.vk"), str(cfg), str(tmp_path / name / "root.json"))
            assert r.returncode != 0 and ("leaf circuit" in r.stderr or "this app" in r.stderr), r.stderr
    # another guest under the same configuration: another leaf circuit, hence another key
    other = tmp_path / "mixed.elf"
    other.write_bytes(rv.elf_bytes(mixed_program()))
    d = tmp_path / "mixed"
    d.mkdir()
    (d / "stdin.bin").write_bytes(int(21).to_bytes(4, "little"))
    r = subprocess.run([pm.CLI, "prove-elf", str(other), str(d / "stdin.bin"), str(d), str(cfg), "9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (d / "root.vk").read_bytes() != key
    assert pm.run_cli("verify", str(d / "root.vk"), str(cfg), str(tmp_path / "long" / "root.json")).returncode != 0


def test_keccak_guest_device_traces_and_one_proof(zk, ora, tmp_path):
    """A guest hashing with the keccak intrinsic: the segment with the calls on the device (24 chips) == the CPU twins, proof bytes ==
    the oracle's; then the whole flow from the ELF (openvm.toml enables the keccak extension): the root's public values carry hashlib's
    SHA3-256 digest."""
    import hashlib

    msg = b"one block of a message for the keccak intrinsic"
    words, data = keccak_program(2), keccak_data(msg)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 6)
    k = next(i for i, s in enumerate(segs) if len(s["kk_ts"]))
    inst = v2.segment_instance(segs[k], words, PC_BASE, H)
    prog = v2.program_table(words, PC_BASE, H[0])
    T = device_traces(zk, segs[k], prog, H)
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "chip %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    pk.close()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n[app_vm_config.keccak]\n")
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, b"", data=data)
    assert out["verified"] and out["total_cycles"] == model["instret"]
    root = json.loads((tmp_path / "root.json").read_text())
    upv = pm.un_b64_bincode(root["user_pvs_proof"])
    assert upv[4 * N_STMT:4 * N_STMT + 28] == hashlib.sha3_256(msg).digest()[:28] and upv[4 * N_STMT:4 * N_STMT + 32] == bytes(model["pvs"])
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0


def test_sha256_guest_device_traces_and_one_proof(zk, ora, tmp_path):
    """A guest hashing a three-block message with the sha256 intrinsic under an app with both hash extensions (26 chips): device traces
    == CPU twins, proof bytes == the oracle's; the whole flow from the ELF: the root's public values carry hashlib's SHA-256 digest."""
    import hashlib

    msg = bytes(range(150))
    data, n_blocks = sha256_data(msg)
    words = sha256_program(n_blocks)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 5, 8)
    k = next(i for i, s in enumerate(segs) if len(s["sha_ts"]))
    inst = v2.segment_instance(segs[k], words, PC_BASE, H)
    T = device_traces(zk, segs[k], v2.program_table(words, PC_BASE, H[0]), H)
    assert len(inst) == len(T) == 26
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "position %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    pk.close()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n[app_vm_config.sha256]\n")
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, b"", data=data)
    assert out["verified"] and out["total_cycles"] == model["instret"]
    upv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])
    pv = upv[4 * N_STMT:4 * N_STMT + 32]
    assert b"".join(int.from_bytes(pv[4 * k:4 * k + 4], "little").to_bytes(4, "big") for k in range(8)) == hashlib.sha256(msg).digest()
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0
    # the same guest under an app without the extension has no proof
    (tmp_path / "plain.toml").write_text(pm.TOML.format(*PARAMS))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "plain.toml"), "8"], capture_output=True, text=True)
    assert r.returncode != 0 and "does not enable the sha256 extension" in r.stderr


def test_modmul_guest_device_traces_and_one_proof(zk, ora, tmp_path):
    """A guest evaluating the secp256k1 curve equation at the published generator with the modmul intrinsic, two moduli (26 chips):
    device traces == CPU twins, proof bytes == the oracle's; the whole flow from the ELF with the reference's `supported_moduli` syntax:
    the root's public values carry y^2 and x^3 (they differ by 7 modulo p)."""
    moduli = (SECP256K1_P, SECP256K1_N)
    words, data = modmul_program(), modmul_data()
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 0, moduli)
    k = next(i for i, s in enumerate(segs) if len(s["mm_ts_0"]))
    inst = v2.segment_instance(segs[k], words, PC_BASE, H)
    T = device_traces(zk, segs[k], v2.program_table(words, PC_BASE, H[0]), H)
    assert len(inst) == len(T) == 26
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "position %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    pk.close()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n[app_vm_config.modular]\nsupported_moduli = [\n    \"%d\",\n    \"%d\"\n]\n" % moduli)
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, b"", data=data, moduli=moduli)
    assert out["verified"] and out["total_cycles"] == model["instret"]
    pv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    assert pv == bytes(model["pvs"])
    y2_low, x3_low = int.from_bytes(pv[:16], "little"), int.from_bytes(pv[16:28], "little")
    assert y2_low == SECP256K1_GY ** 2 % SECP256K1_P % (1 << 128) and x3_low == SECP256K1_GX ** 3 % SECP256K1_P % (1 << 96)
    assert (y2_low - x3_low - 7) % (1 << 96) == 0                        # y^2 = x^3 + 7 (no wrap in the low words here)
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0


def test_ecc_guest_device_traces_and_one_proof(zk, ora, tmp_path):
    """A guest computing 3 G = 2 G + G on secp256k1 and 2 G on bn254 with the ecc intrinsic, two curves (26 chips): device traces == CPU
    twins, proof bytes == the oracle's; the whole flow from the ELF with the reference's `[[app_vm_config.ecc.supported_curves]]` syntax:
    the root's public values carry the published 3 G."""
    words, data = ec_program(), ec_data()
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 0, (), False, EC_CURVES)
    k = next(i for i, s in enumerate(segs) if len(s["ec_ts_0"]))
    inst = v2.segment_instance(segs[k], words, PC_BASE, H)
    T = device_traces(zk, segs[k], v2.program_table(words, PC_BASE, H[0]), H)
    assert len(inst) == len(T) == 26
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "position %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    pk.close()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n" + v2.curves_toml(EC_CURVES))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, b"", data=data, curves=EC_CURVES)
    assert out["verified"] and out["total_cycles"] == model["instret"]
    pv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    assert pv == bytes(model["pvs"])
    assert pv[:16] == (0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9).to_bytes(32, "little")[:16]   # 3 G of secp256k1 (published)
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0


def test_fp2_guest_device_traces_and_one_proof(zk, ora, tmp_path):
    """A guest evaluating bn254's twist equation at the published G2 generator with the fp2 intrinsic (24 chips): device traces == CPU
    twins, proof bytes == the oracle's; the whole flow from the ELF with the reference's `[app_vm_config.fp2]` syntax: both sides of the
    equation in the root's public values agree."""
    words, data = fp2_program(), fp2_data()
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 0, (), False, (), (BN254_P,))
    k = next(i for i, s in enumerate(segs) if len(s["fp2_ts_0"]))
    inst = v2.segment_instance(segs[k], words, PC_BASE, H)
    T = device_traces(zk, segs[k], v2.program_table(words, PC_BASE, H[0]), H)
    assert len(inst) == len(T) == 24
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "position %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    pk.close()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n" + v2.fp2_toml((BN254_P,)))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, b"", data=data, fp2=(BN254_P,))
    assert out["verified"] and out["total_cycles"] == model["instret"]
    pv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    assert pv == bytes(model["pvs"]) and pv[:16] == pv[16:]                # y^2 == x^3 + 3 / (9 + u) at EIP-197's G2 generator
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0


def test_native_and_castf_guest_device_traces_and_one_proof(zk, ora, tmp_path):
    """A guest running BabyBear arithmetic, its quartic extension and casts through the native intrinsics (`[app_vm_config.native]` +
    `[app_vm_config.castf]`, the sections of the reference's batch and bundle circuits; 25 chips): device traces == CPU twins, proof bytes ==
    the oracle's; the whole flow from the ELF at the reference's parameters' syntax: the root's public values are Python's."""
    from test_vm_cpu import native_data, native_program

    words, data = native_program(), native_data()
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, sections=("native", "castf"))
    k = next(i for i, s in enumerate(segs) if len(s["nat_records"]))
    inst = v2.segment_instance(segs[k], words, PC_BASE, H)
    T = device_traces(zk, segs[k], v2.program_table(words, PC_BASE, H[0]), H)
    assert len(inst) == len(T) == 25
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "position %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    pk.close()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n[app_vm_config.castf]\n\n[app_vm_config.native]\n")
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, b"", data=data)
    assert out["verified"] and out["total_cycles"] == model["instret"]
    pv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    assert pv == bytes(model["pvs"])
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0
    # the same guest under an app without the sections has no proof
    (tmp_path / "plain.toml").write_text(pm.TOML.format(*PARAMS))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "plain.toml"), "8"], capture_output=True, text=True)
    assert r.returncode != 0 and "native extension" in r.stderr


def test_int256_guest_device_traces_and_one_proof(zk, ora, tmp_path):
    """A guest running the six 256-bit opcodes through the int256 intrinsic (bigint extension, 27 chips): device traces == CPU twins,
    proof bytes == the oracle's; the whole flow from the ELF with `[app_vm_config.bigint]`: the root's public values are Python's."""
    words, data = int256_program(), int256_data()
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 0, (), True)
    k = next(i for i, s in enumerate(segs) if len(s["i256_ts"]))
    inst = v2.segment_instance(segs[k], words, PC_BASE, H)
    T = device_traces(zk, segs[k], v2.program_table(words, PC_BASE, H[0]), H)
    assert len(inst) == len(T) == 27
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "position %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    pk.close()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n[app_vm_config.bigint]\nrange_tuple_checker_sizes = [256, 8192]\n")
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, b"", data=data)
    assert out["verified"] and out["total_cycles"] == model["instret"]
    pv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    M = 1 << 256
    want = [[(b + c) % M, (b - c) % M, b ^ c, b | c, b & c, (b * c) % M][op] for op, (b, c) in enumerate(INT256_OPERANDS)]
    assert pv == bytes(model["pvs"]) and [int.from_bytes(pv[4 * k:4 * k + 4], "little") for k in range(6)] == [w & 0xFFFFFFFF for w in want]
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0


def test_256_bit_branch_guest_device_traces_and_one_proof(zk, ora, tmp_path):
    """The bigint extension's 256-bit branches (int256 opcodes 12 .. 17; crates/circuits/chunk-circuit/openvm.toml:17-18): a guest that takes and
    does not take every one and closes a loop with a backward bne256 -- device traces == CPU twins in every segment, proof bytes == the
    oracle's, and the whole flow from the ELF: the root's public values are the Python model's."""
    from test_vm_cpu import BRANCH256_LOOP, branch256_data, branch256_program

    words, data = branch256_program(), branch256_data()
    model = rv.run(words, b"", data=data)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 6, 0, 0, (), True)
    assert len(segs) >= 2 and sum(len(s["cmp256_ts"]) for s in segs) == 12 + BRANCH256_LOOP
    proven = 0
    for k, rec in enumerate(segs):
        if not len(rec["cmp256_ts"]):
            continue
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        T = device_traces(zk, rec, v2.program_table(words, PC_BASE, H[0]), H)
        assert len(inst) == len(T) == 27
        for a, d in enumerate(inst):
            assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "segment %d position %d" % (k, a)
        if proven < 2:
            pk = z.ProvingKey(zk, PARAMS, inst)
            pvs = [d["pvs"] for d in inst]
            proof = pk.prove(T, pvs)
            assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
            assert proof == ora.stark_prove(PARAMS, inst).tobytes()
            pk.close()
            proven += 1
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS) + "\n[app_vm_config.bigint]\nrange_tuple_checker_sizes = [256, 8192]\n")
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "6"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified"] and out["total_cycles"] == model["instret"] and out["segments"] >= 2 and out["segments_retried"] == 0
    pv = pm.un_b64_bincode(json.loads((tmp_path / "root.json").read_text())["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    assert pv == bytes(model["pvs"]) and [int.from_bytes(pv[4 * k:4 * k + 4], "little") for k in range(3)] == [0b1111110, BRANCH256_LOOP, 1]
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0


def test_the_chunk_circuit_configuration_end_to_end(tmp_path):
    """`prove-elf` under an openvm.toml with the sections of the reference's chunk circuit in the reference's syntax (keccak, sha2, bigint,
    modular with its six moduli, fp2, ecc with its three curves: 51 chips per segment) for a guest that uses all five intrinsics: ONE root proof whose public values
    carry hashlib's SHA3-256 / SHA-256 words and Python's field product and 256-bit difference."""
    import hashlib

    words, data = all_extensions_program(True), all_extensions_data()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=data))
    (tmp_path / "openvm.toml").write_text(chunk_circuit_toml(PARAMS))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), "-", str(tmp_path), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    model = rv.run(words, b"", data=data, moduli=CHUNK_CIRCUIT_MODULI, curves=tuple((c[1], c[3]) for c in CHUNK_CIRCUIT_CURVES))
    assert out["verified"] and out["total_cycles"] == model["instret"]
    root = json.loads((tmp_path / "root.json").read_text())
    pv = pm.un_b64_bincode(root["user_pvs_proof"])[4 * N_STMT:4 * N_STMT + 32]
    assert pv == bytes(model["pvs"]) and pv[:8] == hashlib.sha3_256(ALL_EXT_MSG).digest()[:8]
    assert b"".join(pv[8 + 4 * k:12 + 4 * k][::-1] for k in range(2)) == hashlib.sha256(ALL_EXT_MSG).digest()[:8]
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0


def test_segments_carry_only_the_chips_they_use(zk, ora, tmp_path):
    """PER-PROOF CHIP PRESENCE (the reference's engine proves only the chips a segment used: the chunk circuit's 42 AIRs, AGENTS.md:183-185).
    Under an app with the keccak extension a segment WITHOUT keccak calls is proven under the base shape (22 chips): device traces ==
    twins, proof == the oracle's proof of the 22-chip instance; the segment WITH the calls needs the 24-chip shape -- proven under the base
    shape (the chip that receives its requests hidden) its bus does not balance and the verifier refuses.  Then `prove-elf` under the
    reference's chunk-circuit configuration (51 chips): a Fibonacci guest's segments all carry 22 chips, a keccak guest's segments 22 or
    26 (base + both hash intrinsics) and none 51; every root verifies under the app's ONE aggregation key."""
    msg = b"one block of a message for the keccak intrinsic"
    words, data = keccak_program(2), keccak_data(msg)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 6)
    assert v2.A_KECCAK in H.ids and len(H.ids) == 24
    base = v2.Heights(H)
    base.ids = tuple(i for i in H.ids if i < v2.A_KECCAK)
    assert len(base.ids) == 22
    prog = v2.program_table(words, PC_BASE, H[0])
    plain = next(i for i, s in enumerate(segs) if not len(s["kk_ts"]))
    inst = v2.segment_instance(segs[plain], words, PC_BASE, base)
    assert len(inst) == 22
    T = device_traces(zk, segs[plain], prog, base)
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "chip %d" % a
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    # the same segment under the full set: another proof (24 chips), also the oracle's
    full_inst = v2.segment_instance(segs[plain], words, PC_BASE, H)
    fpk = z.ProvingKey(zk, PARAMS, full_inst)
    fproof = fpk.prove(device_traces(zk, segs[plain], prog, H), [d["pvs"] for d in full_inst])
    assert fproof == ora.stark_prove(PARAMS, full_inst).tobytes() and len(fproof) > len(proof)
    fpk.close()
    # a segment that CALLS keccak, with the chip hidden: pending traffic on the request bus -> refused
    hashed = next(i for i, s in enumerate(segs) if len(s["kk_ts"]))
    hidden = v2.segment_instance(segs[hashed], words, PC_BASE, base)
    hproof = pk.prove(device_traces(zk, segs[hashed], prog, base), [d["pvs"] for d in hidden])
    assert z.verify(PARAMS, pk.verifying_airs(), [d["pvs"] for d in hidden], hproof) != 0
    assert ora.stark_verify(PARAMS, hidden, ora.stark_prove(PARAMS, hidden)) != 0
    pk.close()
    # ---- the flow under the reference's chunk-circuit configuration ----
    (tmp_path / "openvm.toml").write_text(chunk_circuit_toml(PARAMS))
    runs = {}
    for name, w, dat, stdin in (("fib", fib_program(), b"", (400).to_bytes(4, "little")), ("keccak", words, data, b"")):
        d = tmp_path / name
        d.mkdir()
        exe = d / "guest.elf"
        exe.write_bytes(rv.elf_bytes(w, data=dat))
        (d / "stdin.bin").write_bytes(stdin)
        r = subprocess.run([pm.CLI, "prove-elf", str(exe), str(d / "stdin.bin"), str(d), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        runs[name] = json.loads(r.stdout.strip().splitlines()[-1])
        assert pm.run_cli("verify-guest", str(exe), str(d / "root.vk"), str(tmp_path / "openvm.toml"), str(d / "root.json")).returncode == 0
    # shape 0 = the base chips with a SMALL memory system (a segment that stays in its registers), 1 = the base chips, 2 = + the hash
    # intrinsics, 3 = everything the configuration lists
    assert runs["fib"]["chips_per_shape"] == [22, 22, 26, 51] == runs["keccak"]["chips_per_shape"]
    fs = runs["fib"]["segments_per_shape"]
    assert fs[2:] == [0, 0] and fs[0] >= 1 and fs[0] + fs[1] == runs["fib"]["segments"] >= 3      # Fibonacci never leaves its registers
    ks = runs["keccak"]["segments_per_shape"]
    assert ks[2] >= 1 and ks[3] == 0 and sum(ks) == runs["keccak"]["segments"]     # a segment that calls only keccak carries <= 26 chips
    # ZKHIP_ONE_SHAPE=1: every segment under the full set, as round 3 (another key: one leaf circuit)
    import os
    d = tmp_path / "one"
    d.mkdir()
    (d / "stdin.bin").write_bytes((400).to_bytes(4, "little"))
    r = subprocess.run([pm.CLI, "prove-elf", str(tmp_path / "fib" / "guest.elf"), str(d / "stdin.bin"), str(d), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True,
                       env=dict(os.environ, ZKHIP_ONE_SHAPE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    one = json.loads(r.stdout.strip().splitlines()[-1])
    assert one["chips_per_shape"] == [51] and (d / "root.vk").read_bytes() != (tmp_path / "fib" / "root.vk").read_bytes()
    # ZKHIP_NO_LEAN_SHAPE=1: round 4's first set of shapes (another key again: three leaf circuits instead of four)
    d = tmp_path / "nolean"
    d.mkdir()
    (d / "stdin.bin").write_bytes((400).to_bytes(4, "little"))
    r = subprocess.run([pm.CLI, "prove-elf", str(tmp_path / "fib" / "guest.elf"), str(d / "stdin.bin"), str(d), str(tmp_path / "openvm.toml"), "8"], capture_output=True, text=True,
                       env=dict(os.environ, ZKHIP_NO_LEAN_SHAPE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    nl = json.loads(r.stdout.strip().splitlines()[-1])
    assert nl["chips_per_shape"] == [22, 26, 51] and nl["segments_per_shape"][0] == nl["segments"] and (d / "root.vk").read_bytes() != (tmp_path / "fib" / "root.vk").read_bytes()


def test_one_task_over_a_device_list(tmp_path):
    """the segments of ONE task spread over the GPUs of a node, and the aggregation tree's nodes go to whichever device
    is free.  ZKHIP_DEVICES=0,0,0 builds three device slots -- all mapped onto this box's one GPU -- each with its own copy of the segment
    and node keys, its own lanes and its own circuit forks: under the tree's FIXED grouping the root proof is byte-identical to the
    one-device run, the tree's nodes are spread over the slots, the key is the same.  The default (the greedy fold, three node pipelines
    per device: the tree's shape follows the arrival times) states the same about the same run under the same key -- the root proof's
    bytes and the statement's accumulator over the children depend on the shape; the app, the states, the commitments and the key do not.  A device that does not exist is an error, not a crash."""
    import os

    words = fib_program()
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words))
    cfg = tmp_path / "openvm.toml"
    cfg.write_text(pm.TOML.format(*PARAMS))
    out = {}
    fixed = {"ZKHIP_TREE_BALANCED": "1", "ZKHIP_AGG_SLOTS": "1"}
    for name, env in (("one", fixed), ("three", dict(fixed, ZKHIP_DEVICES="0,0,0", ZKHIP_LANES="1")), ("fold", {}), ("fold3", {"ZKHIP_DEVICES": "0,0,0"})):
        d = tmp_path / name
        d.mkdir()
        (d / "stdin.bin").write_bytes((3000).to_bytes(4, "little"))
        r = subprocess.run([pm.CLI, "prove-elf", str(exe), str(d / "stdin.bin"), str(d), str(cfg), "9"], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-3000:]
        out[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["one"]["segments"] == out["three"]["segments"] == out["fold"]["segments"] > 27
    assert len(out["one"]["tree_nodes_per_device_slot"]) == 1 and len(out["three"]["tree_nodes_per_device_slot"]) == 3
    assert sum(out["three"]["tree_nodes_per_device_slot"]) == sum(out["one"]["tree_nodes_per_device_slot"]) and min(out["three"]["tree_nodes_per_device_slot"]) >= 1
    a, b = json.loads((tmp_path / "one" / "root.json").read_text()), json.loads((tmp_path / "three" / "root.json").read_text())
    assert a["proof"] == b["proof"] and a["user_pvs_proof"] == b["user_pvs_proof"]
    assert (tmp_path / "one" / "root.vk").read_bytes() == (tmp_path / "three" / "root.vk").read_bytes()
    # the greedy fold: as many internal nodes as the fixed grouping needs at most, three (resp. nine) pipelines, the same statement and key
    for name, n_slots in (("fold", 3), ("fold3", 9)):
        assert len(out[name]["tree_nodes_per_device_slot"]) == n_slots
        assert sum(out[name]["tree_nodes_per_device_slot"]) <= sum(out["one"]["tree_nodes_per_device_slot"])
        f = json.loads((tmp_path / name / "root.json").read_text())
        # [app (8) | pc, memory root | pc, memory root | accumulator (8) | leaf (8) | internal (8)] + the public values and their openings:
        # all the same but the accumulator -- a hash over the children's payloads along the tree, which has another shape
        fu, au = pm.un_b64_bincode(f["user_pvs_proof"]), pm.un_b64_bincode(a["user_pvs_proof"])
        assert len(fu) == len(au) and fu[:4 * 26] == au[:4 * 26] and fu[4 * 34:] == au[4 * 34:]
        assert (tmp_path / name / "root.vk").read_bytes() == (tmp_path / "one" / "root.vk").read_bytes()
        assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "one" / "root.vk"), str(cfg), str(tmp_path / name / "root.json")).returncode == 0
    d = tmp_path / "bad"
    d.mkdir()
    (d / "stdin.bin").write_bytes((100).to_bytes(4, "little"))
    r = subprocess.run([pm.CLI, "prove-elf", str(exe), str(d / "stdin.bin"), str(d), str(cfg), "9"], capture_output=True, text=True, env=dict(os.environ, ZKHIP_DEVICES="0,63"))
    assert r.returncode == 1 and "device" in r.stderr


def test_gen_proof_universal_from_an_elf_and_witness_bytes(tmp_path):
    """Prover::gen_proof_universal (mod.rs:287-309) over the one-statement flow: the task's serialized witnesses reach the guest
    through ProvingTask::build_guest_input (length-framed items), the guest folds them into its public values, ONE proof comes out."""
    import struct
    from test_vm_cpu import A0, A1, A7, S0, S1, T0, T3

    p = [("addi", S0, 0, 0), ("addi", S1, 0, 0),
         ("label", "item"), ("slti", T3, S1, 3), ("beq", T3, 0, "out"),
         ("addi", A7, 0, 2), ("ecall",), ("add", S0, S0, A0), ("addi", T0, A0, 3), ("srli", T0, T0, 2),
         ("label", "w"), ("beq", T0, 0, "next"), ("ecall",), ("add", S0, S0, A0), ("addi", T0, T0, -1), ("jal", 0, "w"),
         ("label", "next"), ("addi", S1, S1, 1), ("jal", 0, "item"),
         ("label", "out"), ("add", A0, S0, 0), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
         ("addi", A0, S1, 0), ("addi", A1, 0, 1), ("ecall",), ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    words = rv.assemble(p)
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words))
    wit = [bytes(range(1, 6)), b"", bytes(range(200, 216))]
    with open(tmp_path / "task.bin", "wb") as f:
        f.write(struct.pack("<I", 7) + b"chunk-7" + struct.pack("<I", len(wit)))
        for w in wit:
            f.write(struct.pack("<Q", len(w)) + w)
    (tmp_path / "openvm.toml").write_text(pm.TOML.format(*PARAMS))
    r = pm.run_cli("prove-task", str(exe), str(tmp_path / "task.bin"), str(tmp_path), str(tmp_path / "openvm.toml"), "8")
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    stream = b"".join(struct.pack("<I", len(w)) + w + bytes(-len(w) % 4) for w in wit)
    model = rv.run(words, stream)
    assert info["identifier"] == "chunk-7" and info["verified"] and info["total_cycles"] == model["instret"]
    root = json.loads((tmp_path / "root.json").read_text())
    assert set(root) == {"proof", "user_pvs_proof", "baseline", "deferral_merkle_proofs", "stat"}
    assert root["stat"]["total_cycles"] == model["instret"]
    upv = pm.un_b64_bincode(root["user_pvs_proof"])
    assert upv[4 * N_STMT:4 * N_STMT + 32] == bytes(model["pvs"])
    total = sum(int.from_bytes(stream[i:i + 4], "little") for i in range(0, len(stream), 4)) & 0xFFFFFFFF
    assert int.from_bytes(upv[4 * N_STMT:4 * N_STMT + 4], "little") == total and upv[4 * N_STMT + 4] == 3
    assert pm.run_cli("verify-guest", str(exe), str(tmp_path / "root.vk"), str(tmp_path / "openvm.toml"), str(tmp_path / "root.json")).returncode == 0
    # a guest that fails (exit code 1) has no proof; neither has one whose public values stay zero
    bad = rv.assemble([("addi", A0, 0, 1), ("addi", A7, 0, 93), ("ecall",)])
    exe.write_bytes(rv.elf_bytes(bad))
    r = pm.run_cli("prove-task", str(exe), str(tmp_path / "task.bin"), str(tmp_path), str(tmp_path / "openvm.toml"), "8")
    assert r.returncode == 1 and "exited with code 1" in r.stderr
    silent = rv.assemble([("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)])
    exe.write_bytes(rv.elf_bytes(silent))
    r = pm.run_cli("prove-task", str(exe), str(tmp_path / "task.bin"), str(tmp_path), str(tmp_path / "openvm.toml"), "8")
    assert r.returncode == 1 and "public_values are all 0s" in r.stderr


def test_segments_that_do_not_chain_are_refused_in_circuit(zk, mixed):
    """The aggregation circuit chains (pc, memory root): segment k followed by segment k + 2 has no witness."""
    H, words = mixed["heights"], mixed["words"]
    insts = [v2.segment_instance(mixed["segs"][k], words, PC_BASE, H) for k in (1, 2, 3)]
    prog = v2.program_table(words, PC_BASE, H[0])
    pk = z.ProvingKey(zk, PARAMS, insts[0])
    proofs, pvs = [], []
    for k, inst in zip((1, 2, 3), insts):
        pv = [d["pvs"] for d in inst]
        proofs.append(pk.prove(device_traces(zk, mixed["segs"][k], prog, H), pv))
        pvs.append(pv)
    stmt = dict(start=[(v2.A_CONNECTOR, 0)] + [(v2.A_MERKLE, k) for k in range(8)], end=[(v2.A_CONNECTOR, 1)] + [(v2.A_MERKLE, 8 + k) for k in range(8)])
    rc = z.RecursionCircuit(PARAMS, pk.verifying_airs(), 4, stmt=stmt)
    st, npv = rc.witness(proofs, pvs)
    assert st == 0, rc.last_error()
    assert int(npv[8]) == int(mixed["segs"][1]["meta"][0]) and int(npv[17]) == int(mixed["segs"][3]["meta"][1])
    assert npv[9:17].tolist() == mixed["segs"][1]["meta"][4:12].tolist() and npv[18:26].tolist() == mixed["segs"][3]["meta"][12:20].tolist()
    assert rc.witness([proofs[0], proofs[2]], [pvs[0], pvs[2]])[0] == -7
    assert rc.witness([proofs[1], proofs[0]], [pvs[1], pvs[0]])[0] == -7
    pk.close()
