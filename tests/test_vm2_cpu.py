"""CPU: the one-statement VM circuit (include/zkhip_vm_circuit.hpp) -- adapters, execution bus, connector, persistent memory -- on
the segmenting executor's records (include/zkhip_vm_exec.hpp, `prove_cli dump-segments`):
  * the run agrees with the independent Python interpreter; segments chain by (pc, memory root); the first root is the guest image's,
    the last pc is 0, the public values open in the final root;
  * every one of the 22 traces (24 with the keccak extension) (CPU twins of the device generators, tests/vm2_util.py) satisfies its AIR, and every bus balances
    exactly: program, execution, operand, access, memory, merkle, hash and the three lookup buses;
  * the oracle proves a whole segment and both verifiers accept;
  * a tampered instruction result, swapped operands, a forged memory value, a skipped instruction, a wrong next pc: some bus no
    longer balances (what makes the proof fail);
  * the decode table against an independent decoder."""
import os

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import recursion_util as ru
import rv32_model as rv
import vm2_util as v2
from test_vm_cpu import (BN254_G2X, BN254_G2Y, BN254_P, fp2_data, fp2_program, INT256_SHIFT_OPERANDS, ALL_EXT_MSG, CHUNK_CIRCUIT_CURVES, EC_CURVES, ec_data, ec_program, CHUNK_CIRCUIT_MODULI, INT256_OPERANDS, all_extensions_data, all_extensions_program, SECP256K1_GX, SECP256K1_GY, SECP256K1_N, SECP256K1_P, fib_program, int256_data, int256_program, keccak_data,
                         keccak_program, mixed_program, modmul_data, modmul_program, sha256_data, sha256_program)

PARAMS = (1, 0, 4, 3, 3)
PC_BASE = 0x00200000


@pytest.fixture(scope="module")
def mixed(tmp_path_factory):
    words = mixed_program()
    stdin = (7).to_bytes(4, "little")
    info, heights, segs, image_root, pv_open = v2.dump_segments(tmp_path_factory.mktemp("mixed"), rv.exe_bytes(words), stdin, 9)
    return dict(words=words, stdin=stdin, info=info, heights=heights, segs=segs, image_root=image_root, pv_open=pv_open)


def test_segments_chain_and_agree_with_the_model(mixed):
    model = rv.run(mixed["words"], mixed["stdin"])
    info, segs = mixed["info"], mixed["segs"]
    assert info["total_cycle"] == model["instret"] and bytes.fromhex(info["public_values"]) == bytes(model["pvs"])
    assert len(segs) > 5 and sum(int(s["meta"][3]) for s in segs) == model["instret"]
    meta = [s["meta"] for s in segs]
    assert int(meta[0][0]) == PC_BASE and meta[0][4:12].tolist() == mixed["image_root"].tolist()
    for a, b in zip(meta, meta[1:]):
        assert int(a[1]) == int(b[0]) != 0 and a[12:20].tolist() == b[4:12].tolist()
    assert int(meta[-1][1]) == 0                       # exited
    for m, s in zip(meta, segs):
        assert int(m[2]) == 1 + 16 * int(m[3]) and s["f_x"].size == int(m[3])
    # the public values open in the final root: two blocks of address space 3, 28 siblings each
    pv = bytes(model["pvs"])
    final_root = meta[-1][12:20]
    for blk in range(2):
        cells = [pv[16 * blk + 2 * j] | (pv[16 * blk + 2 * j + 1] << 8) for j in range(8)]
        cur = ru.sponge(cells)
        idx = (3 << 26) | blk
        sib = mixed["pv_open"][blk * 8 * 28:(blk + 1) * 8 * 28].reshape(28, 8)
        for lvl in range(28):
            cur = ru.compress(sib[lvl], cur) if idx & 1 else ru.compress(cur, sib[lvl])
            idx >>= 1
        assert cur.tolist() == final_root.tolist()


@pytest.mark.parametrize("k", [0, 7, -1])
def test_every_trace_satisfies_its_air_and_every_bus_balances(mixed, k):
    inst = v2.segment_instance(mixed["segs"][k], mixed["words"], PC_BASE, mixed["heights"])
    assert len(inst) == 22
    for a, d in enumerate(inst):
        assert air.quotient_chunks(d["program"]) <= 2, a                     # degree <= 3: blow-up 2 suffices
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], a
    assert v2.bus_imbalance(inst) == {}


def test_a_segment_proves_and_verifies(ora, mixed):
    inst = v2.segment_instance(mixed["segs"][2], mixed["words"], PC_BASE, mixed["heights"])
    proof = ora.stark_prove(PARAMS, inst)
    assert ora.stark_verify(PARAMS, inst, proof) == 0
    vk = ru.verifying(PARAMS, inst)
    pvs = [d["pvs"] for d in inst]
    assert z.verify(PARAMS, vk, pvs, proof.tobytes()) == 0
    # the statement is bound: another pc_end, another final root
    for a, i in ((v2.A_CONNECTOR, 1), (v2.A_MERKLE, 9)):
        bad = [p.copy() for p in pvs]
        bad[a][i] = (int(bad[a][i]) + 1) % ora.P
        assert z.verify(PARAMS, vk, bad, proof.tobytes()) != 0


def _tampered(mixed, k, edit):
    rec = {n: v.copy() for n, v in mixed["segs"][k].items()}
    edit(rec)
    inst = v2.segment_instance(rec, mixed["words"], PC_BASE, mixed["heights"])
    local = [a for a, d in enumerate(inst) if air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep"))]
    return local, v2.bus_imbalance(inst)


def test_tampering_breaks_a_bus(mixed):
    seg = mixed["segs"][4]
    prog = v2.program_table(mixed["words"], PC_BASE, mixed["heights"][0])
    cls = prog[1][seg["pc_index"]]
    alu_rows = np.nonzero(cls == 0)[0]
    sub_row = next(int(r) for r in alu_rows if prog[2][seg["pc_index"][r]] == 1 and seg["f_x"][r] != seg["f_y"][r])

    # 1. a wrong instruction result in the frame (the core still holds the true one): operand + memory bus
    def wrong_result(rec):
        rec["f_z"][int(alu_rows[3])] ^= 4
    assert _tampered(mixed, 4, wrong_result)[1].keys() >= {3, 1}

    # 2. swapped operands of a SUB: the registers say otherwise
    def swapped(rec):
        r = sub_row
        rec["f_x"][r], rec["f_y"][r] = rec["f_y"][r], rec["f_x"][r]
    local, bad = _tampered(mixed, 4, swapped)
    assert 3 in bad and 1 in bad

    # 3. the core is made consistent with a forged operand (b + 1): the core row is valid, but no register holds that value
    def forged_operand(rec):
        j = 2
        rec["alu_b"][j] = (int(rec["alu_b"][j]) + 1) & 0xFFFFFFFF
    local, bad = _tampered(mixed, 4, forged_operand)
    assert local == [] and 3 in bad

    # 4. a skipped instruction: the execution-bus chain has a hole
    def skipped(rec):
        for n in ("pc_index", "f_x", "f_y", "f_z", "f_rdprev", "f_pcinc", "f_pts1", "f_pts2", "f_pts3"):
            rec[n] = np.delete(rec[n], 10)
    assert 2 in _tampered(mixed, 4, skipped)[1]

    # 5. a branch that goes elsewhere
    def wrong_pc(rec):
        rec["f_pcinc"][20] = (int(rec["f_pcinc"][20]) + 4) % v2.P
    assert {2, 3} <= _tampered(mixed, 4, wrong_pc)[1].keys()

    # 6. a register that changes between two accesses (the value a write claims to replace), and an access that claims an older
    #    predecessor than the word's last access: the memory bus no longer is a history
    writes = np.nonzero(prog[12][seg["pc_index"]] == 1)[0]
    def forged_memory(rec):
        rec["f_rdprev"][int(writes[40])] ^= 1
    assert 1 in _tampered(mixed, 4, forged_memory)[1]
    reads = np.nonzero((prog[9][seg["pc_index"]] == 1) & (seg["f_pts1"] > 16))[0]
    def forged_history(rec):
        rec["f_pts1"][int(reads[5])] -= 16
    local, bad = _tampered(mixed, 4, forged_history)
    assert local == [] and 1 in bad

    # 7. another initial memory root than the blocks hash to
    def wrong_root(rec):
        rec["meta"][5] = (int(rec["meta"][5]) + 1) % v2.P
    local, bad = _tampered(mixed, 4, wrong_root)
    assert v2.A_MERKLE in local


def test_keccak_intrinsic_is_part_of_the_statement(tmp_path):
    """A guest that computes SHA3-256 with the keccak intrinsic: the revealed digest is hashlib's (FIPS 202 -- parity anchored outside
    this repository), the calls' 50 memory words go through the keccak adapter, the permutation through the Keccak-f chip: all traces
    satisfy their AIRs, every bus balances, and a forged output lane breaks the lane bus."""
    import hashlib

    msg = b"one block of a message for the keccak intrinsic"
    words, data = keccak_program(2), keccak_data(msg)
    model = rv.run(words, b"", data=data)
    digest = hashlib.sha3_256(msg).digest()
    assert bytes(model["pvs"])[:28] == digest[:28]                       # (word 7 is folded with the second permutation's output)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 6)   # room for two calls per segment
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"]) and info["total_cycle"] == model["instret"]
    assert sum(len(s["kk_ts"]) for s in segs) == 2 and len(H.ids) == 24
    with pytest.raises(AssertionError, match="does not enable the keccak extension"):   # the same guest under an app without it
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7)
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for a, d in enumerate(inst):
            assert air.quotient_chunks(d["program"]) <= 2, a
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], (k, a)
        assert v2.bus_imbalance(inst) == {}
    k = next(i for i, s in enumerate(segs) if len(s["kk_ts"]))
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["kio_rows"].reshape(-1, v2.KECCAK_IO_WIDTH)[3, 31] ^= 1           # an output limb of lane 3 that the permutation does not produce
    inst = v2.segment_instance(rec, words, PC_BASE, H)
    bad = v2.bus_imbalance(inst)
    assert 13 in bad and 1 in bad                                         # the lane bus and (the next reader of that word) the memory bus
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["kk_states"][5] ^= 1                                              # the chip permutes another state than the memory holds
    assert 13 in v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))


def test_sha256_intrinsic_is_part_of_the_statement(tmp_path):
    """A guest that hashes a three-block message with the sha256 intrinsic under an app with BOTH hash extensions (26 chips): the
    revealed digest is hashlib's (FIPS 180-4), every trace satisfies its AIR, every bus balances; forged words break the buses."""
    import hashlib

    msg = bytes(range(150))
    data, n_blocks = sha256_data(msg)
    words = sha256_program(n_blocks)
    model = rv.run(words, b"", data=data)
    digest = hashlib.sha256(msg).digest()
    assert b"".join(int.from_bytes(bytes(model["pvs"])[4 * k:4 * k + 4], "little").to_bytes(4, "big") for k in range(8)) == digest
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 5, 8)
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"]) and info["total_cycle"] == model["instret"]
    assert sum(len(s["sha_ts"]) for s in segs) == n_blocks == 3 and len(H.ids) == 26
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d in inst:
            assert air.quotient_chunks(d["program"]) <= 2
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], k
        assert v2.bus_imbalance(inst) == {}
    k = next(i for i, s in enumerate(segs) if len(s["sha_ts"]))
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["shaio_rows"].reshape(-1, v2.SHA_IO_WIDTH)[2, 28] ^= 1            # a state word written that the compression does not produce
    bad = v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    assert 16 in bad and 1 in bad                                         # state bus, memory bus
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["sha_blocks"][8 + 3] ^= 1                                         # the chip compresses another message word than the memory holds
    assert 15 in v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    # sha256 alone (24 chips: the extension chips follow the base chips in the order of their ids)
    info, H2, segs2, _, _ = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 8)
    assert H2.ids[-2:] == (v2.A_SHA256, v2.A_SHA256_IO) and len(H2.ids) == 24
    inst = v2.segment_instance(segs2[0], words, PC_BASE, H2)
    assert all(air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [] for d in inst) and v2.bus_imbalance(inst) == {}


def test_modmul_intrinsic_is_part_of_the_statement(tmp_path):
    """A guest that evaluates both sides of the secp256k1 curve equation at the published generator with the modmul intrinsic, under an app
    with two moduli (26 chips: a multiplication chip and an adapter per modulus): the revealed y^2 and x^3 are Python's and differ by 7
    modulo p; every trace satisfies its AIR, every bus balances; a forged product breaks the word bus."""
    moduli = (SECP256K1_P, SECP256K1_N)
    words, data = modmul_program(), modmul_data()
    model = rv.run(words, b"", data=data, moduli=moduli)
    pv = bytes(model["pvs"])
    y2, x3 = SECP256K1_GY ** 2 % SECP256K1_P, SECP256K1_GX ** 3 % SECP256K1_P
    assert (y2 - x3 - 7) % SECP256K1_P == 0
    other = (SECP256K1_N - 2) * (SECP256K1_N - 3) % SECP256K1_N
    assert pv[:16] == y2.to_bytes(32, "little")[:16] and pv[16:28] == x3.to_bytes(32, "little")[:12]
    assert int.from_bytes(pv[28:32], "little") == (x3 >> 96 & 0xFFFFFFFF) ^ (other & 0xFFFFFFFF)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 0, moduli)
    assert bytes.fromhex(info["public_values"]) == pv and info["total_cycle"] == model["instret"]
    assert len(H.ids) == 26 and sum(len(s["mm_ts_0"]) for s in segs) == 8 and sum(len(s["mm_ts_1"]) for s in segs) == 1
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d in inst:
            assert air.quotient_chunks(d["program"]) <= 2
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], k
        assert v2.bus_imbalance(inst) == {}
    k = next(i for i, s in enumerate(segs) if len(s["mm_ts_0"]))
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["mmio_rows_0"].reshape(-1, v2.MODMUL_IO_WIDTH)[17, 28] ^= 1        # a result word that is not the product's
    bad = v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    assert 18 in bad and 1 in bad                                         # modulus 0's word bus, memory bus
    # the division row relabelled as a multiplication in the chip: it is a product, but the adapter's words carry opcode 3 and the
    # operands in the other order
    kd = next(i for i, s in enumerate(segs) if 3 in s["mm_records_0"].reshape(-1, 17)[:, 0])
    inst = v2.segment_instance(segs[kd], words, PC_BASE, H)
    pos = H.ids.index(v2.A_MODMUL(0))
    row = int(np.nonzero(segs[kd]["mm_records_0"].reshape(-1, 17)[:, 0] == 3)[0][0])
    inst[pos]["trace"] = inst[pos]["trace"].copy()
    inst[pos]["trace"][288:322, row] = 0
    assert air.check_trace(inst[pos]["program"], inst[pos]["trace"], inst[pos]["pvs"]) == [] and 18 in v2.bus_imbalance(inst)
    # without the extension the guest has no proof
    with pytest.raises(AssertionError, match="lists 0 moduli"):
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7)


def test_int256_intrinsic_is_part_of_the_statement(tmp_path):
    """A guest that runs the six 256-bit opcodes through the int256 intrinsic under an app with the bigint extension (25 chips): the
    revealed words are Python's; every trace satisfies its AIR, every bus balances; a forged result word, a comparison answered the
    other way, or a call whose adapter claims another opcode than the ecall passed, breaks a bus."""
    words, data = int256_program(), int256_data()
    model = rv.run(words, b"", data=data)
    pv = bytes(model["pvs"])
    M = 1 << 256
    want = [[(b + c) % M, (b - c) % M, b ^ c, b | c, b & c, (b * c) % M][op] for op, (b, c) in enumerate(INT256_OPERANDS)]
    assert [int.from_bytes(pv[4 * k:4 * k + 4], "little") for k in range(6)] == [w & 0xFFFFFFFF for w in want]
    M_ = 1 << 256
    (sb, sc), (rb, rc), (ab, ac) = INT256_SHIFT_OPERANDS
    folded = ((sb << sc) % M_ & 0xFFFFFFFF) ^ ((rb >> rc) >> 224) ^ ((((ab - M_) >> (ac % 256)) % M_) >> 224)
    assert [int.from_bytes(pv[24 + 4 * j:28 + 4 * j], "little") for j in range(2)] == [(want[1] >> 224) ^ folded, (want[5] >> 224) ^ 0b110]   # sltu 0, slt 1, eq 1
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 0, (), True)
    assert bytes.fromhex(info["public_values"]) == pv and info["total_cycle"] == model["instret"]
    assert len(H.ids) == 27 and H.ids[-5:] == (v2.A_INT256, v2.A_INT256_IO, v2.A_MUL256, v2.A_CMP256, v2.A_SHIFT256)
    assert sum(len(s["i256_ts"]) for s in segs) == 5 and sum(len(s["mul256_ts"]) for s in segs) == 1 and sum(len(s["cmp256_ts"]) for s in segs) == 3 and sum(len(s["sh256_ts"]) for s in segs) == 3
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d in inst:
            assert air.quotient_chunks(d["program"]) <= 2
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], k
        assert v2.bus_imbalance(inst) == {}
    k = next(i for i, s in enumerate(segs) if len(s["i256_ts"]))
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["i256io_rows"].reshape(-1, v2.INT256_IO_WIDTH)[20, 28] ^= 1        # a result word that is not the ALU's
    bad = v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    assert 31 in bad and 1 in bad                                         # the word bus, the memory bus
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["i256io_rows"].reshape(-1, v2.INT256_IO_WIDTH)[:24, 34] = 4        # the adapter of the first call claims opcode AND
    bad = v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    assert 30 in bad and 31 in bad                                        # the request bus (ecall passed another opcode), the word bus
    kc = next(i for i, s in enumerate(segs) if len(s["cmp256_ts"]))
    inst = v2.segment_instance(segs[kc], words, PC_BASE, H)
    pos = H.ids.index(v2.A_CMP256)
    inst[pos]["trace"] = inst[pos]["trace"].copy()
    inst[pos]["trace"][64, 0] ^= 1                                        # "2^256 - 5 < 77" answered yes
    assert air.check_trace(inst[pos]["program"], inst[pos]["trace"], inst[pos]["pvs"]) != [] or v2.bus_imbalance(inst) != {}
    with pytest.raises(AssertionError, match="does not enable the bigint extension"):
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7)


def test_ecc_intrinsic_is_part_of_the_statement(tmp_path):
    """A guest that computes 3 G = 2 G + G on secp256k1 and 2 G on bn254 with the ecc intrinsic, under an app with two curves (26 chips: a
    point chip and an adapter per curve): the revealed coordinates are the published 3 G of secp256k1 (and Python's); every trace
    satisfies its AIR, every bus balances; a forged result word breaks the word bus; an addition of a point to itself has no execution."""
    words, data = ec_program(), ec_data()
    model = rv.run(words, b"", data=data, curves=EC_CURVES)
    pv = bytes(model["pvs"])
    x3g, y3g = 0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9, 0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672
    bn2g = 1368015179489954701390400359078579693043519447331113978918064868415326638035
    assert pv[:16] == x3g.to_bytes(32, "little")[:16] and pv[16:28] == y3g.to_bytes(32, "little")[:12]
    assert int.from_bytes(pv[28:32], "little") == (y3g >> 96 & 0xFFFFFFFF) ^ (bn2g & 0xFFFFFFFF)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 0, (), False, EC_CURVES)
    assert bytes.fromhex(info["public_values"]) == pv and info["total_cycle"] == model["instret"]
    assert len(H.ids) == 26 and sum(len(s["ec_ts_0"]) for s in segs) == 2 and sum(len(s["ec_ts_1"]) for s in segs) == 1
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d in inst:
            assert air.quotient_chunks(d["program"]) <= 2
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], k
        assert v2.bus_imbalance(inst) == {}
    k = next(i for i, s in enumerate(segs) if len(s["ec_ts_0"]))
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["ecio_rows_0"].reshape(-1, v2.EC_IO_WIDTH)[33, 52] ^= 1            # a result word that is not the chip's
    bad = v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    assert 33 in bad and 1 in bad                                         # curve 0's word bus, memory bus
    with pytest.raises(AssertionError, match="lists 0 curves"):
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7)
    # G + G through the chord operation: no slope, no execution
    g = SECP256K1_GX.to_bytes(32, "little") + SECP256K1_GY.to_bytes(32, "little")
    same = rv.assemble(rv.li(8, 0x00400000) + [("addi", 10, 8, 0), ("addi", 11, 0, 0), ("addi", 17, 0, 7), ("ecall",), ("addi", 10, 0, 0), ("addi", 17, 0, 93), ("ecall",)])
    with pytest.raises(AssertionError, match="no slope"):
        v2.dump_segments(tmp_path, rv.exe_bytes(same, data=g + g + bytes(64)), b"", 7, 0, 0, (), False, EC_CURVES)


def test_fp2_intrinsic_is_part_of_the_statement(tmp_path):
    """A guest that evaluates bn254's twist equation y^2 = x^3 + 3 / (9 + u) at the published G2 generator with the fp2 intrinsic
    (multiplications, a division, an addition, a subtraction), under an app with the fp2 extension (24 chips): the revealed components
    are Python's and both sides agree; every trace satisfies its AIR, every bus balances; a forged result word breaks the word bus."""
    P2 = BN254_P
    words, data = fp2_program(), fp2_data()
    model = rv.run(words, b"", data=data, fp2=(P2,))
    pv = bytes(model["pvs"])
    mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P2, (a[0] * b[1] + a[1] * b[0]) % P2)  # noqa: E731
    y2 = mul(BN254_G2Y, BN254_G2Y)
    low = lambda v: int(v % (1 << 64)).to_bytes(8, "little")  # noqa: E731
    assert pv[:16] == low(y2[0]) + low(y2[1]) and pv[16:32] == pv[:16]      # x^3 + b' = y^2 (and the difference folded in is zero)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 0, 0, (), False, (), (P2,))
    assert bytes.fromhex(info["public_values"]) == pv and info["total_cycle"] == model["instret"]
    assert len(H.ids) == 24 and sum(len(s["fp2_ts_0"]) for s in segs) == 6
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d in inst:
            assert air.quotient_chunks(d["program"]) <= 2
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], k
        assert v2.bus_imbalance(inst) == {}
    k = next(i for i, s in enumerate(segs) if len(s["fp2_ts_0"]))
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["fp2io_rows_0"].reshape(-1, v2.EC_IO_WIDTH)[33, 52] ^= 1           # a result word that is not the chip's
    bad = v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    assert 38 in bad and 1 in bad                                         # field 0's word bus, memory bus
    with pytest.raises(AssertionError, match="lists 0 fp2 fields"):
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7)


def test_all_extensions_of_the_chunk_circuit_in_one_statement(tmp_path):
    """The chip set the reference's chunk-circuit configuration asks for (keccak, sha2, bigint, six moduli, one fp2 field, three curves: 51 chips) and a
    guest that uses all five intrinsics: SHA3-256 and SHA-256 of one message are hashlib's, the secp256k1 field product, the 256-bit
    difference and the doubled generator Python's; every trace satisfies its AIR and every bus balances."""
    import hashlib

    curves = tuple((c[1], c[3]) for c in CHUNK_CIRCUIT_CURVES)
    words, data = all_extensions_program(True), all_extensions_data()
    model = rv.run(words, b"", data=data, moduli=CHUNK_CIRCUIT_MODULI, curves=curves)
    pv = bytes(model["pvs"])
    assert pv[:8] == hashlib.sha3_256(ALL_EXT_MSG).digest()[:8]
    assert b"".join(pv[8 + 4 * k:12 + 4 * k][::-1] for k in range(2)) == hashlib.sha256(ALL_EXT_MSG).digest()[:8]
    two_g_x = 0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5   # 2 G of secp256k1 (published)
    assert int.from_bytes(pv[28:32], "little") == (((5 - 7) % (1 << 256)) >> 224) ^ (two_g_x & 0xFFFFFFFF)
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, 5, 7, CHUNK_CIRCUIT_MODULI, True, curves, (CHUNK_CIRCUIT_MODULI[0],))
    assert bytes.fromhex(info["public_values"]) == pv and info["total_cycle"] == model["instret"] and len(H.ids) == 51
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d in inst:
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], k
        assert v2.bus_imbalance(inst) == {}


def test_the_reference_config_file_itself_gives_the_same_chip_set(tmp_path):
    """crates/circuits/chunk-circuit/openvm.toml as it lies in the reference tree (read here only; skipped where the tree is absent, e.g. on
    the GPU box) goes through the executor's configuration reader: the chip ids, the heights and the run of the all-extensions guest are
    those of this repository's restatement of that file (test_vm_cpu.chunk_circuit_toml) -- 51 chips."""
    import subprocess
    from test_vm_cpu import chunk_circuit_toml

    ref = "/root/reference/crates/circuits/chunk-circuit/openvm.toml"
    if not os.path.exists(ref):
        pytest.skip("no reference tree here")
    words, data = all_extensions_program(True), all_extensions_data()
    (tmp_path / "exe.bin").write_bytes(rv.exe_bytes(words, data=data))
    (tmp_path / "stdin.bin").write_bytes(b"")
    (tmp_path / "mine.toml").write_text(chunk_circuit_toml(PARAMS))
    seen = []
    for cfg in (ref, str(tmp_path / "mine.toml")):
        out = tmp_path / ("out_%d" % len(seen))
        out.mkdir()
        r = subprocess.run([v2.CLI, "dump-segments", str(tmp_path / "exe.bin"), str(tmp_path / "stdin.bin"), str(out), "7", "5", "7", cfg], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        seen.append((np.fromfile(out / "air_ids.u32", dtype=np.uint32).tolist(), np.fromfile(out / "heights.u32", dtype=np.uint32).tolist(), r.stdout.strip().splitlines()[-1]))
    assert seen[0] == seen[1] and len(seen[0][0]) == 51


def test_decode_table_against_an_independent_decoder(mixed):
    """cls / op / registers / immediates of every instruction of both test programs, re-derived from the RISC-V encoding here."""
    for words in (mixed["words"], fib_program()):
        t = v2.program_table(words, PC_BASE, max(1, int(np.ceil(np.log2(len(words))))))
        for k, w in enumerate(words):
            f = t[:, k].tolist()
            opc, rd, f3, rs1, rs2, f7 = w & 0x7F, (w >> 7) & 31, (w >> 12) & 7, (w >> 15) & 31, (w >> 20) & 31, w >> 25
            assert f[0] == PC_BASE + 4 * k
            sext = lambda v, bits: v - (1 << bits) if v >> (bits - 1) else v   # noqa: E731
            if opc == 0x33:
                assert f[3:6] == [rd, rs1, rs2] and f[9:12] == [1, 1, 0] and f[12] == (rd != 0)
                exp = {(0, 0): (0, 0), (0x20, 0): (0, 1), (0, 4): (0, 2), (0, 6): (0, 3), (0, 7): (0, 4), (0, 2): (1, 0), (0, 3): (1, 1), (0, 1): (5, 0),
                       (0, 5): (5, 1), (0x20, 5): (5, 2), (1, 0): (2, 0), (1, 1): (3, 0), (1, 2): (3, 1), (1, 3): (3, 2), (1, 4): (4, 0), (1, 5): (4, 1),
                       (1, 6): (4, 2), (1, 7): (4, 3)}[(f7, f3)]
                assert tuple(f[1:3]) == exp
            elif opc == 0x13:
                imm = sext(w >> 20, 12) & 0xFFFFFFFF
                if f3 in (1, 5):
                    assert f[1] == 5 and f[6:8] == [rs2, 0] and f[16] == 1
                else:
                    assert f[6:8] == [imm & 0xFFFF, imm >> 16]
                assert f[9:12] == [1, 0, 1]
            elif opc == 0x63:
                off = sext(((w >> 31) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 63) << 5) | (((w >> 8) & 15) << 1), 13)
                assert f[8] == off % v2.P and f[9:13] == [1, 1, 0, 0] and f[1] == (6 if f3 < 2 else 7)
            elif opc in (0x03, 0x23):
                assert f[1] == 11 and f[15] == 1
            elif opc == 0x73:
                assert f[1] == 12 and f[3:6] == [10, 17, 10]


def test_native_and_castf_intrinsics_are_part_of_the_statement(tmp_path):
    """A guest that runs BabyBear arithmetic, its quartic extension and the cast to bytes through the native intrinsics under an app with
    `[app_vm_config.native]` and `[app_vm_config.castf]` (the sections of the reference's batch and bundle circuits; 25 chips): the revealed
    words are Python's integers'; every trace satisfies its AIR, every bus balances; a forged result, a non-canonical result word and a
    cast of a value that is not below 2^30 are refused; an app without the sections refuses the guest."""
    from test_vm_cpu import BABYBEAR, CASTF_VALUES, NATIVE_EXT_OPERANDS, NATIVE_OPERANDS, NATIVE_OPS, native_data, native_program

    words, data = native_program(), native_data()
    model = rv.run(words, b"", data=data)
    pv = [int.from_bytes(bytes(model["pvs"])[4 * k:4 * k + 4], "little") for k in range(8)]
    F = BABYBEAR
    want = [[(b + c) % F, (b - c) % F, b * c % F, b * pow(c, -1, F) % F][op] for op, (b, c) in zip(NATIVE_OPS, NATIVE_OPERANDS)]
    assert pv[:4] == want[:4] and pv[4] == want[4] ^ want[5] and want[4] == 0 and want[5] == (0x80000005 % F) * (0xFFFFFFFF % F) % F
    x, y = NATIVE_EXT_OPERANDS[2]
    assert pv[5] == (x[0] * y[0] + 11 * (x[1] * y[3] + x[2] * y[2] + x[3] * y[1])) % F       # the product's constant coefficient
    xq, yq = NATIVE_EXT_OPERANDS[3]
    assert pv[6] == 1 and xq == yq                                                             # x / x = 1
    assert pv[7] == CASTF_VALUES[0] ^ CASTF_VALUES[1]
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, sections=("native", "castf"))
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"]) and info["total_cycle"] == model["instret"]
    assert len(H.ids) == 25 and H.ids[-3:] == (v2.A_NATIVE_ARITH, v2.A_NATIVE_EXT, v2.A_CASTF)
    assert sum(len(s["nat_records"]) for s in segs) == 9 * 6 and sum(len(s["next_records"]) for s in segs) == 27 * 5 and sum(len(s["castf_records"]) for s in segs) == 6 * 2
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d in inst:
            assert air.quotient_chunks(d["program"]) <= 2
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], k
        assert v2.bus_imbalance(inst) == {}
    k = next(i for i, s in enumerate(segs) if len(s["nat_records"]))
    inst = v2.segment_instance(segs[k], words, PC_BASE, H)
    pos = H.ids.index(v2.A_NATIVE_ARITH)
    # a forged result: the chip's own constraints refuse it
    forged = {n: (v.copy() if hasattr(v, "copy") else v) for n, v in inst[pos].items()}
    forged["trace"] = inst[pos]["trace"].copy()
    forged["trace"][6, 0] = (int(forged["trace"][6, 0]) + 1) % v2.P
    assert air.check_trace(forged["program"], forged["trace"], forged["pvs"], None) != []
    # the same residue written non-canonically (r + p, split into halves): the canonical-word constraints refuse it
    r0 = int(inst[pos]["trace"][6, 0]) + (int(inst[pos]["trace"][7, 0]) << 16)
    if r0 + F < 1 << 32:
        nc = inst[pos]["trace"].copy()
        nc[6, 0], nc[7, 0] = (r0 + F) & 0xFFFF, (r0 + F) >> 16
        assert air.check_trace(inst[pos]["program"], nc, inst[pos]["pvs"], None) != []
    # a record the executor did not write (another operand): the memory bus no longer balances
    rec = {n: v.copy() for n, v in segs[k].items()}
    rec["nat_records"][1] ^= 1
    bad = v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    assert 1 in bad
    # an app without the sections has no such intrinsic
    with pytest.raises(AssertionError, match="does not enable the native extension"):
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7)
    with pytest.raises(AssertionError, match="does not enable the castf extension"):
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7, sections=("native",))
    # a cast of 2^30 is no cast
    import struct
    big = data[:-16] + struct.pack("<2I", 1 << 30, 0) + data[-8:]
    with pytest.raises(AssertionError, match="does not lie below 2\\^30"):
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=big), b"", 7, sections=("native", "castf"))


def test_the_mixed_chunk_like_guest_runs_the_same_in_the_executor_and_the_model(tmp_path):
    """tools/guest_bench2.py's `mixed` guest (tests/test_vm_cpu.py mixed_chunk_program: register loops, strided loads, Keccak-f, SHA-256,
    secp256k1 additions and doublings, modular and 256-bit arithmetic in phases) under the reference's chunk-circuit configuration: the C++
    segmenting executor and the independent Python model agree on the instruction count and on all eight revealed words; the point sums
    are Python's (Q = (2 + k) G, D = 2^k G after k full iterations); segments chain; a segment of every phase satisfies its AIRs and
    balances every bus."""
    from test_vm_cpu import MIXED_PHASE_ITERATIONS, mixed_chunk_data, mixed_chunk_program

    curves = tuple((c[1], c[3]) for c in CHUNK_CIRCUIT_CURVES)
    words, data = mixed_chunk_program(), mixed_chunk_data()
    n = 3 * MIXED_PHASE_ITERATIONS + 5                       # plain, hash, hash, then five full iterations
    stdin = n.to_bytes(4, "little")
    model = rv.run(words, stdin, data=data, moduli=CHUNK_CIRCUIT_MODULI, curves=curves)
    pv = [int.from_bytes(bytes(model["pvs"])[4 * k:4 * k + 4], "little") for k in range(8)]
    # secp256k1 with Python integers: Q = (2 + 5) G, D = 2^5 G
    P_ = SECP256K1_P

    def add(p1, p2):
        if p1 == p2:
            lam = 3 * p1[0] * p1[0] * pow(2 * p1[1], -1, P_) % P_
        else:
            lam = (p2[1] - p1[1]) * pow(p2[0] - p1[0], -1, P_) % P_
        x3 = (lam * lam - p1[0] - p2[0]) % P_
        return x3, (lam * (p1[0] - x3) - p1[1]) % P_

    G = (SECP256K1_GX, SECP256K1_GY)
    q = add(G, G)
    for _ in range(5):
        q = add(q, G)
    d = G
    for _ in range(5):
        d = add(d, d)
    assert pv[3] == q[0] & 0xFFFFFFFF and pv[4] == d[1] & 0xFFFFFFFF
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), stdin, 12, 5, 7, CHUNK_CIRCUIT_MODULI, True, curves, (CHUNK_CIRCUIT_MODULI[0],))
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"]) and info["total_cycle"] == model["instret"] and len(H.ids) == 51
    meta = [s_["meta"] for s_ in segs]
    for a, b in zip(meta, meta[1:]):
        assert int(a[1]) == int(b[0]) != 0 and a[12:20].tolist() == b[4:12].tolist()
    assert int(meta[-1][1]) == 0
    plain = next(k for k, s_ in enumerate(segs) if not len(s_["kk_ts"]) and not len(s_["ec_ts_0"]))
    hashy = next(k for k, s_ in enumerate(segs) if len(s_["kk_ts"]) and len(s_["sha_ts"]) and not len(s_["ec_ts_0"]))
    full = next(k for k, s_ in enumerate(segs) if len(s_["ec_ts_0"]) and len(s_["mm_ts_2"]) and len(s_["i256_ts"]) and len(s_["mul256_ts"]))
    for k in (plain, hashy, full):
        inst = v2.segment_instance(segs[k], words, PC_BASE, H)
        for d_ in inst:
            assert air.check_trace(d_["program"], d_["trace"], d_["pvs"], d_.get("prep")) == [], k
        assert v2.bus_imbalance(inst) == {}


@pytest.mark.parametrize("seed", [1, 2])
def test_random_native_calls_agree_with_python_and_satisfy_the_chips(tmp_path, seed):
    """Seeded random operands (any 32-bit words: the chips read them modulo p), every opcode of the field and the extension chip, casts of
    random values below 2^30: the executor's memory after the calls is the Python model's (the guest XOR-folds every result word into its
    public values), every trace satisfies its AIR and every bus balances."""
    import random
    import struct

    from test_vm_cpu import A0, A1, A7, S0, T0, T1

    rnd = random.Random(seed)
    F = 2013265921
    n_a, n_e, n_c = 24, 12, 6
    arith = [(rnd.randrange(4), rnd.choice([rnd.randrange(1 << 32), rnd.randrange(F), F - 1, 0, 1]), rnd.choice([rnd.randrange(1, 1 << 32), F - 1, 1, 2])) for _ in range(n_a)]
    arith = [(op, b, c if (op != 3 or c % F) else 7) for op, b, c in arith]                         # no division by zero
    ext = [(rnd.randrange(4), [rnd.randrange(1 << 32) for _ in range(4)], [rnd.randrange(1, F) for _ in range(4)]) for _ in range(n_e)]
    casts = [rnd.randrange(1 << 30) for _ in range(n_c)]
    data = b"".join(struct.pack("<3I", b, c, 0) for _, b, c in arith) + b"".join(struct.pack("<12I", *x, *y, 0, 0, 0, 0) for _, x, y in ext) + b"".join(struct.pack("<2I", v, 0) for v in casts)
    data += bytes(64)
    p = rv.li(S0, 0x00400000)
    e0, c0 = 12 * n_a, 12 * n_a + 48 * n_e
    fold0 = c0 + 8 * n_c
    for k, (op, _, _) in enumerate(arith):
        p += [("addi", A0, S0, 12 * k), ("addi", A1, 0, op), ("addi", A7, 0, 9), ("ecall",), ("lw", T0, S0, 12 * k + 8), ("lw", T1, S0, fold0 + 4 * (k % 8)), ("xor", T1, T1, T0), ("sw", T1, S0, fold0 + 4 * (k % 8))]
    for k, (op, _, _) in enumerate(ext):
        p += [("addi", A0, S0, e0 + 48 * k), ("addi", A1, 0, op), ("addi", A7, 0, 10), ("ecall",)]
        for q in range(4):
            p += [("lw", T0, S0, e0 + 48 * k + 32 + 4 * q), ("lw", T1, S0, fold0 + 4 * ((k + q) % 8)), ("xor", T1, T1, T0), ("sw", T1, S0, fold0 + 4 * ((k + q) % 8))]
    for k in range(n_c):
        p += [("addi", A0, S0, c0 + 8 * k), ("addi", A7, 0, 11), ("ecall",), ("lw", T0, S0, c0 + 8 * k + 4), ("lw", T1, S0, fold0 + 4 * (k % 8)), ("xor", T1, T1, T0), ("sw", T1, S0, fold0 + 4 * (k % 8))]
    for k in range(8):
        p += [("lw", A0, S0, fold0 + 4 * k), ("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    words = rv.assemble(p)
    model = rv.run(words, b"", data=data)
    # the model's results against plain modular arithmetic (the field chip's four operations)
    fold = [0] * 8
    for k, (op, b, c) in enumerate(arith):
        b_, c_ = b % F, c % F
        fold[k % 8] ^= [(b_ + c_) % F, (b_ - c_) % F, b_ * c_ % F, b_ * pow(c_, -1, F) % F][op]
    for k, (op, x, y) in enumerate(ext):
        x_, y_ = [v % F for v in x], [v % F for v in y]
        z = [(u + v) % F for u, v in zip(x_, y_)] if op == 0 else [(u - v) % F for u, v in zip(x_, y_)] if op == 1 else rv.ext4_mul(x_, y_) if op == 2 else rv.ext4_mul(x_, rv.ext4_pow(y_, F ** 4 - 2))
        if op == 3:
            assert rv.ext4_mul(z, y_) == x_                                                          # a quotient is what multiplies back
        for q in range(4):
            fold[(k + q) % 8] ^= z[q]
    for k, v in enumerate(casts):
        fold[k % 8] ^= v
    assert [int.from_bytes(bytes(model["pvs"])[4 * k:4 * k + 4], "little") for k in range(8)] == fold
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 9, sections=("native", "castf"))
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"]) and info["total_cycle"] == model["instret"]
    for rec in segs:
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d_ in inst:
            assert air.check_trace(d_["program"], d_["trace"], d_["pvs"], d_.get("prep")) == []
        assert v2.bus_imbalance(inst) == {}


def test_a_segment_that_touches_two_thousand_blocks(tmp_path):
    """A guest that stores one word into each of 1 900 blocks and reads every other one back: a segment whose memory chips carry thousands
    of rows -- the executor closes it with batched hashes (sixteen permutations at a time) and rows built on several threads
    (include/zkhip_vm_exec.hpp close_memory: those paths need >= 512 blocks).  The leaf, merkle and Poseidon2 traces satisfy their AIRs,
    every bus balances (the hash bus: every request of the memory chips is a row of the Poseidon2 chip), roots chain, the run is the model's."""
    A0, A1, A7, T0, T1, T2, T3 = 10, 11, 17, 5, 6, 7, 28
    n_blocks = 1900
    p = rv.li(T0, 0x00400000) + rv.li(T3, 0x00400000 + 16 * n_blocks)
    p += [("addi", T1, 0, 5),
          ("label", "fill"), ("sw", T1, T0, 4), ("addi", T1, T1, 7), ("addi", T0, T0, 16), ("bne", T0, T3, "fill")]
    p += rv.li(T0, 0x00400000) + [("addi", T2, 0, 0),
          ("label", "sum"), ("lw", T1, T0, 4), ("add", T2, T2, T1), ("addi", T0, T0, 32), ("blt", T0, T3, "sum"),
          ("add", A0, T2, 0), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
          ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    words = rv.assemble(p)
    model = rv.run(words, b"")
    expect = sum(5 + 7 * i for i in range(0, n_blocks, 2)) & 0xFFFFFFFF
    assert int.from_bytes(bytes(model["pvs"])[:4], "little") == expect
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words), b"", 15)
    assert bytes.fromhex(info["public_values"]) == bytes(model["pvs"]) and info["total_cycle"] == model["instret"]
    assert len(segs) == 1 and segs[0]["leaf_rows"].size // v2.LEAF_WIDTH >= n_blocks
    meta = segs[0]["meta"]
    assert meta[4:12].tolist() == image_root.tolist() and int(meta[1]) == 0
    inst = v2.segment_instance(segs[0], words, PC_BASE, H)
    for a, d in enumerate(inst):
        assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], a
    assert v2.bus_imbalance(inst) == {}
    # the public values open in the final root (the tree the executor keeps is what the rows state)
    pv = bytes(model["pvs"])
    for blk in range(2):
        cur = ru.sponge([pv[16 * blk + 2 * j] | (pv[16 * blk + 2 * j + 1] << 8) for j in range(8)])
        idx = (3 << 26) | blk
        sib = pv_open[blk * 8 * 28:(blk + 1) * 8 * 28].reshape(28, 8)
        for lvl in range(28):
            cur = ru.compress(sib[lvl], cur) if idx & 1 else ru.compress(cur, sib[lvl])
            idx >>= 1
        assert cur.tolist() == meta[12:20].tolist()


def test_256_bit_branches_are_part_of_the_statement(tmp_path):
    """The bigint extension's 256-bit branches (crates/circuits/chunk-circuit/openvm.toml:17-18; OpenVM's Rv32BranchEqual256 /
    Rv32BranchLessThan256): int256 opcodes 12 beq, 13 bne, 14 bltu, 15 blt, 16 bgeu, 17 bge -- the comparison chip decides, the ecall chip's
    row steps the pc by 4 or by a2.  A guest that takes and does not take every one of them and closes a loop with a BACKWARD bne256: the
    revealed words are the Python model's (mask of the branches not taken, the loop's count); every trace satisfies its AIR, every bus
    balances, at frames small enough that branches fall on both sides of segment cuts; a decision answered the other way, a forged offset
    and a comparison chip that disagrees with the ecall chip are refused."""
    from test_vm_cpu import BRANCH256_LOOP, branch256_data, branch256_program

    words, data = branch256_program(), branch256_data()
    model = rv.run(words, b"", data=data)
    pv = bytes(model["pvs"])
    assert [int.from_bytes(pv[4 * k:4 * k + 4], "little") for k in range(3)] == [0b1111110, BRANCH256_LOOP, 1]
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 6, 0, 0, (), True)
    assert bytes.fromhex(info["public_values"]) == pv and info["total_cycle"] == model["instret"] and len(segs) >= 2
    assert sum(len(s["cmp256_ts"]) for s in segs) == 12 + BRANCH256_LOOP
    n_br = 0
    for k, rec in enumerate(segs):
        inst = v2.segment_instance(rec, words, PC_BASE, H)
        for d in inst:
            assert air.quotient_chunks(d["program"]) <= 2
            assert air.check_trace(d["program"], d["trace"], d["pvs"], d.get("prep")) == [], (k, d.get("name"))
        assert v2.bus_imbalance(inst) == {}, k
        n_br += int(rec["ecall_rows"].reshape(-1, v2.ECALL_WIDTH)[:, 37].sum())
    assert n_br == 12 + BRANCH256_LOOP
    # forgeries, on a segment that holds a branch
    kb = next(i for i, s in enumerate(segs) if s["ecall_rows"].reshape(-1, v2.ECALL_WIDTH)[:, 37].any())
    rows = segs[kb]["ecall_rows"].reshape(-1, v2.ECALL_WIDTH)
    rb = int(np.nonzero(rows[:, 37])[0][0])
    pos_e, pos_c = H.ids.index(v2.A_ECALL), H.ids.index(v2.A_CMP256)
    # (i) the ecall chip claims the other decision: its own pc_inc constraint or the branch bus refuses
    rec = {n: v.copy() for n, v in segs[kb].items()}
    rec["ecall_rows"].reshape(-1, v2.ECALL_WIDTH)[rb, 38] ^= 1
    inst = v2.segment_instance(rec, words, PC_BASE, H)
    assert air.check_trace(inst[pos_e]["program"], inst[pos_e]["trace"], inst[pos_e]["pvs"]) != [] or 43 in v2.bus_imbalance(inst)
    # (ii) ... and adjusts pc_inc to match: the branch bus (the comparison chip's word against the ecall chip's) is out of balance
    rec = {n: v.copy() for n, v in segs[kb].items()}
    er = rec["ecall_rows"].reshape(-1, v2.ECALL_WIDTH)
    off = int(er[rb, 39]) + (int(er[rb, 40]) << 16) - (int(er[rb, 41]) << 32)
    er[rb, 38] ^= 1
    er[rb, 20] = (off if er[rb, 38] else 4) % 2013265921
    bad = v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    assert 43 in bad                                                      # (and the execution bus: the next instruction sits elsewhere)
    # (iii) the comparison chip's row says the other thing: its own constraint taken = out XOR neg
    inst = v2.segment_instance(segs[kb], words, PC_BASE, H)
    tr = inst[pos_c]["trace"] = inst[pos_c]["trace"].copy()
    r0 = int(np.nonzero(tr[104])[0][0])
    tr[106, r0] ^= 1
    assert air.check_trace(inst[pos_c]["program"], tr, inst[pos_c]["pvs"]) != []
    # (iv) a forged offset: the memory bus (the a2 read) is out of balance
    rec = {n: v.copy() for n, v in segs[kb].items()}
    rec["ecall_rows"].reshape(-1, v2.ECALL_WIDTH)[rb, 39] ^= 4
    assert 1 in v2.bus_imbalance(v2.segment_instance(rec, words, PC_BASE, H))
    with pytest.raises(AssertionError, match="does not enable the bigint extension"):
        v2.dump_segments(tmp_path, rv.exe_bytes(words, data=data), b"", 7)
