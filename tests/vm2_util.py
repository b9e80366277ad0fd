"""CPU twins of the one-statement VM circuit's trace generators (include/zkhip_vm_circuit.hpp, zkvm-prover_amd/csrc/vm_chips.hip):
the segment records the C++ executor dumps (`prove_cli dump-segments`) -> the 24 traces, in numpy and through the oracle's
existing core-chip generators.  TEST INFRASTRUCTURE.  The AIR programs themselves come from the library (zkhip_vm_air): they have
ONE definition, the C++ one."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_lib as ora
import zkvm_prover_amd as z
from zkvm_prover_amd import _binding as zb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "zkvm-prover_amd", "prove_cli")
P = ora.P
NOPV = np.zeros(0, np.uint32)
(A_PROGRAM, A_FRAME, A_ALU, A_LT, A_MUL, A_MULH, A_DIVREM, A_SHIFT, A_BEQ, A_BLT, A_JAL_LUI, A_AUIPC, A_JALR, A_LS, A_ECALL, A_LEAF,
 A_MERKLE, A_POSEIDON2, A_CONNECTOR, A_BITWISE, A_RANGE_TUPLE, A_RANGE, A_KECCAK, A_KECCAK_IO, A_SHA256, A_SHA256_IO, A_INT256, A_INT256_IO, A_MUL256,
 A_CMP256, A_SHIFT256, A_NATIVE_ARITH, A_NATIVE_EXT, A_CASTF, N_STATIC_AIRS) = range(35)
MAX_MODULI, MAX_CURVES, MAX_FP2 = 8, 4, 2
# the modular extension: chips N_STATIC_AIRS + 2 i (multiplication), + 2 i + 1 (adapter) of modulus i; the ecc extension's follow
N_AIRS = N_STATIC_AIRS + 2 * MAX_MODULI + 2 * MAX_CURVES + 2 * MAX_FP2
PROGRAM_FIELDS, FRAME_WIDTH, LS_WIDTH, ECALL_WIDTH, LEAF_WIDTH, MERKLE_WIDTH, KECCAK_IO_WIDTH, KECCAK_WIDTH = 17, 43, 48, 45, 43, 54, 42, 2634
NATIVE_ARITH_WIDTH, NATIVE_EXT_WIDTH, CASTF_WIDTH, P_HI = 27, 90, 16, 0x7800
SHA_IO_WIDTH, SHA_WIDTH = 34, 434
MODMUL_IO_WIDTH, MODMUL_WIDTH = 35, 326
INT256_IO_WIDTH, INT256_WIDTH, MUL256_WIDTH, CMP256_WIDTH, SHIFT256_WIDTH = 35, 102, 162, 108, 190
EC_IO_WIDTH, EC_WIDTH, FP2_WIDTH = 59, 773, 649
TS_STEP, GAP_HI_BITS = 16, 11
SX, SY = 256, 2048


def vm_airs():
    """[(program, width, n_pvs, prep_width)] of the segment's AIR set, from the library."""
    lib = z.load_library()
    lib.zkhip_vm_n_airs.restype = C.c_size_t
    lib.zkhip_vm_air.restype = C.c_int
    lib.zkhip_vm_air.argtypes = [C.c_uint, C.POINTER(zb._Air), C.POINTER(C.c_size_t)]
    out = []
    for i in range(lib.zkhip_vm_n_airs()):
        a, pw = zb._Air(), C.c_size_t()
        assert lib.zkhip_vm_air(i, C.byref(a), C.byref(pw)) == 0
        out.append((np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width), int(a.n_pvs), int(pw.value)))
    return out


def program_table(words, pc_base, log_program):
    lib = z.load_library()
    lib.zkhip_vm_program_table.restype = C.c_int
    lib.zkhip_vm_program_table.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_uint32, C.c_uint, C.POINTER(C.c_uint32)]
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.zeros((PROGRAM_FIELDS, 1 << log_program), np.uint32)
    assert lib.zkhip_vm_program_table(zb._u32p(w), w.size, pc_base, log_program, zb._u32p(out)) == 0
    return out


class Heights(list):
    """heights[AirId] of the app's chips (None for an absent extension chip); .ids = the chips in proof order; .moduli = the modular
    extension's moduli (integers); .curves = the ecc extension's (modulus, a) pairs"""
    ids = ()
    moduli = ()
    curves = ()
    fp2 = ()


def A_MODMUL(i):
    return N_STATIC_AIRS + 2 * i


def A_MODMUL_IO(i):
    return N_STATIC_AIRS + 2 * i + 1


def A_EC(i):
    return N_STATIC_AIRS + 2 * MAX_MODULI + 2 * i


def A_EC_IO(i):
    return N_STATIC_AIRS + 2 * MAX_MODULI + 2 * i + 1


def A_FP2(i):
    return N_STATIC_AIRS + 2 * MAX_MODULI + 2 * MAX_CURVES + 2 * i


def A_FP2_IO(i):
    return N_STATIC_AIRS + 2 * MAX_MODULI + 2 * MAX_CURVES + 2 * i + 1


def fp2_toml(fp2):
    return "[app_vm_config.fp2]\nsupported_moduli = [\n" + ",\n".join('    ["Field%d","%d"]' % (i, p) for i, p in enumerate(fp2)) + "\n]\n\n" if fp2 else ""


def curves_toml(curves):
    return "".join('[[app_vm_config.ecc.supported_curves]]\nstruct_name = "Curve%d"\nmodulus = "%d"\nscalar = "1"\na = "%d"\nb = "0"\n\n' % (i, p, a)
                   for i, (p, a) in enumerate(curves))


def dump_segments(tmp, exe_bytes, stdin, log_frame, log_keccak=0, log_sha256=0, moduli=(), bigint=False, curves=(), fp2=(), sections=()):
    """Runs the C++ segmenting executor; returns (info json, heights, [segment record dicts], image root, pv openings)."""
    import json

    d = str(tmp)
    open(os.path.join(d, "exe.bin"), "wb").write(exe_bytes)
    open(os.path.join(d, "stdin.bin"), "wb").write(stdin)
    cmd = [CLI, "dump-segments", os.path.join(d, "exe.bin"), os.path.join(d, "stdin.bin"), d, str(log_frame), str(log_keccak), str(log_sha256)]
    if moduli or bigint or curves or fp2 or sections:   # sections: further `[app_vm_config.<name>]` lines, e.g. ("native", "castf")
        toml = "[app_vm_config.bigint]\n" if bigint else ""
        toml += "".join("[app_vm_config.%s]\n" % name for name in sections)
        toml += curves_toml(curves) + fp2_toml(fp2)
        if moduli:
            toml += "[app_vm_config.modular]\nsupported_moduli = [\n" + ",\n".join('    "%d"' % m for m in moduli) + "\n]\n"
        open(os.path.join(d, "moduli.toml"), "w").write(toml)
        cmd.append(os.path.join(d, "moduli.toml"))
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    by_pos = np.fromfile(os.path.join(d, "heights.u32"), dtype=np.uint32).tolist()
    ids = np.fromfile(os.path.join(d, "air_ids.u32"), dtype=np.uint32).tolist()
    heights = Heights([None] * N_AIRS)
    heights.ids, heights.moduli, heights.curves, heights.fp2 = tuple(ids), tuple(moduli), tuple(curves), tuple(fp2)
    for i, h in zip(ids, by_pos):
        heights[i] = h
    segs = []
    for k in range(info["segments"]):
        sd = os.path.join(d, "seg-%d" % k)
        segs.append({f[:-4]: np.fromfile(os.path.join(sd, f), dtype=np.uint32) for f in os.listdir(sd) if f.endswith(".u32")})
    extra = np.fromfile(os.path.join(d, "image_root_and_pv_openings.u32"), dtype=np.uint32)
    return info, heights, segs, extra[:8], extra[8:]


def _bytes4(v):
    v = np.asarray(v, dtype=np.uint32)
    return [(v >> (8 * i)) & 255 for i in range(4)]


def frame_trace(rec, prog, log_height):
    n, N = len(rec["pc_index"]), 1 << log_height
    t = np.zeros((FRAME_WIDTH, N), np.uint32)
    k = rec["pc_index"].astype(np.int64)
    t[0, :n] = prog[0][k]
    t[1, :n] = 1 + TS_STEP * np.arange(n)
    for q in range(1, PROGRAM_FIELDS):
        t[1 + q, :n] = prog[q][k]
    for base, name in ((18, "f_x"), (22, "f_y"), (26, "f_z")):
        for i, b in enumerate(_bytes4(rec[name])):
            t[base + i, :n] = b
    t[30, :n], t[31, :n] = rec["f_rdprev"] & 0xFFFF, rec["f_rdprev"] >> 16
    t[32, :n], t[33, :n] = rec["f_pcinc"], 1
    # register adapter: (prev_ts, gap_lo, gap_hi) of the rs1 / rs2 / rd accesses at slots 0, 2, 12; zero where the access is skipped
    ts = t[1, :n].astype(np.int64)
    for a, (flag_col, slot, name) in enumerate(((10, 0, "f_pts1"), (11, 2, "f_pts2"), (13, 12, "f_pts3"))):
        used = t[flag_col, :n].astype(np.int64)
        prev = rec[name].astype(np.int64)
        gap = ((ts + slot - prev - 1) * used) & ((1 << (16 + GAP_HI_BITS)) - 1)   # (an honest log never wraps; tampered ones are tests)
        t[34 + 3 * a, :n], t[35 + 3 * a, :n], t[36 + 3 * a, :n] = prev * used, gap & 0xFFFF, gap >> 16
    return t


def loadstore_trace(rec, log_height, rc):
    core, rc, bad = ora.rv32_loadstore_trace(rec["ls_case"], rec["ls_read"], rec["ls_prev"], log_height, range_counts=rc)
    assert bad == 0
    n, N = len(rec["ls_case"]), 1 << log_height
    t = np.zeros((LS_WIDTH, N), np.uint32)
    t[:33] = core
    base, imm = rec["ls_base"].astype(np.int64), rec["ls_imm"].astype(np.int64)
    lo = (base & 0xFFFF) + (imm & 0xFFFF)
    hi = (base >> 16) + (imm >> 16) + (lo >> 16)
    t[33, :n] = rec["ls_ts"]
    for i, b in enumerate(_bytes4(rec["ls_base"])):
        t[34 + i, :n] = b
    t[38, :n], t[39, :n] = imm & 0xFFFF, imm >> 16
    t[40, :n], t[41, :n], t[42, :n], t[43, :n] = lo & 0xFFFF, hi & 0xFFFF, lo >> 16, hi >> 16
    t[44, :n] = (lo & 0xFFFF) >> 2
    prev = rec["ls_pts"].astype(np.int64)
    gap = (rec["ls_ts"].astype(np.int64) + 4 - prev - 1) & ((1 << (16 + GAP_HI_BITS)) - 1)
    t[45, :n], t[46, :n], t[47, :n] = prev, gap & 0xFFFF, gap >> 16
    return t, rc


def rows_trace(rows, width, log_height, pad=None):
    rows = np.asarray(rows, dtype=np.uint32).reshape(-1, width)
    t = np.zeros((width, 1 << log_height), np.uint32)
    if pad is not None:
        t[:] = np.asarray(pad, dtype=np.uint32).reshape(-1, 1)
    t[:, :len(rows)] = rows.T
    return t


def keccak_traces(rec, H):
    """keccak calls: the oracle's Keccak-f trace + the call's timestamp on its rows; the adapter's rows as the executor wrote them"""
    n_kk = len(rec["kk_ts"])
    kk = np.zeros((KECCAK_WIDTH, 1 << H[A_KECCAK]), np.uint32)
    lib = ora.lib()
    lib.ora_keccak_f_trace.restype = None
    lib.ora_keccak_f_trace.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32)]
    states = np.ascontiguousarray(rec["kk_states"], dtype=np.uint32).view(np.uint64).reshape(-1, 25) if n_kk else np.zeros((0, 25), np.uint64)
    body = np.zeros((KECCAK_WIDTH - 1, 1 << H[A_KECCAK]), np.uint32)
    lib.ora_keccak_f_trace(states.ctypes.data_as(C.POINTER(C.c_uint64)), n_kk, H[A_KECCAK], ora.p32(body))
    kk[:KECCAK_WIDTH - 1] = body
    kk[KECCAK_WIDTH - 1, :24 * n_kk] = np.repeat(rec["kk_ts"], 24)
    return kk, rows_trace(rec["kio_rows"], KECCAK_IO_WIDTH, H[A_KECCAK_IO])


def sha256_vm_prep(log_height):
    """the VM SHA-256 chip's nine preprocessed columns (K_t limbs, round / final / first / schedule / input / digest gates, round index),
    written here from the chip's description with K_t derived in tests/rv32_model.py"""
    import rv32_model as rv

    k = rv.sha256_constants()
    n = 1 << log_height
    p = np.zeros((9, n), np.uint32)
    for b in range(n // 65):
        for t in range(65):
            row = 65 * b + t
            if t < 64:
                p[0, row], p[1, row] = k[t] & 0xFFFF, k[t] >> 16
            p[2, row], p[3, row], p[4, row], p[5, row] = t < 63, t == 63, t == 0, 15 <= t < 63
            p[6, row], p[7, row], p[8, row] = t < 16, t == 64, t
    return p


def int256_traces(rec, H):
    """int256 calls: the oracle's ALU-chip trace (and its XOR-column counts) + the call's timestamp; the adapter's rows"""
    n = len(rec["i256_ts"])
    l = ora.lib()
    l.ora_int256_alu_trace.restype = C.c_size_t
    l.ora_int256_alu_trace.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lh = H[A_INT256]
    body = np.zeros((INT256_WIDTH - 1, 1 << lh), np.uint32)
    xcnt = np.zeros(1 << 16, np.uint32)
    recs = np.ascontiguousarray(rec["i256_records"], dtype=np.uint32)
    assert l.ora_int256_alu_trace(ora.p32(recs) if n else None, n, lh, ora.p32(body), ora.p32(xcnt)) == 0
    tr = np.zeros((INT256_WIDTH, 1 << lh), np.uint32)
    tr[:INT256_WIDTH - 1] = body
    tr[INT256_WIDTH - 1, :n] = rec["i256_ts"]
    return tr, rows_trace(rec["i256io_rows"], INT256_IO_WIDTH, H[A_INT256_IO]), xcnt.astype(np.int64)


def mul256_traces(rec, H):
    """int256 calls with opcode 5: the oracle's multiplication-chip trace (and its lookup counts) + the call's timestamp"""
    n = len(rec["mul256_ts"])
    l = ora.lib()
    l.ora_mul256_trace.restype = C.c_size_t
    l.ora_mul256_trace.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32), C.c_uint32]
    lh = H[A_MUL256]
    body = np.zeros((MUL256_WIDTH - 1, 1 << lh), np.uint32)
    bw, tup = np.zeros(1 << 16, np.uint32), np.zeros(SX * SY, np.uint32)
    recs = np.ascontiguousarray(rec["mul256_records"], dtype=np.uint32)
    assert l.ora_mul256_trace(ora.p32(recs) if n else None, 17, 1, n, lh, ora.p32(body), ora.p32(bw), ora.p32(tup), SY) == 0
    tr = np.zeros((MUL256_WIDTH, 1 << lh), np.uint32)
    tr[:MUL256_WIDTH - 1] = body
    tr[MUL256_WIDTH - 1, :n] = rec["mul256_ts"]
    return tr, bw.astype(np.int64), tup.astype(np.int64)


def ec_traces(rec, H, i, curve):
    """ecc calls of curve i: the tests' twin of the chip's trace (and its lookup counts) + the call's timestamp; the adapter's rows"""
    import ecc_util as eu

    p, a = curve
    recs = np.ascontiguousarray(rec["ec_records_%d" % i], dtype=np.uint32).reshape(-1, eu.RECORD_WORDS)
    val = lambda w: sum(int(x) << (32 * k) for k, x in enumerate(w))  # noqa: E731
    calls = [(int(r[0]), (val(r[1:9]), val(r[9:17])), (val(r[17:25]), val(r[25:33])), val(r[33:41])) for r in recs]
    lh = H[A_EC(i)]
    body, bw, tup = eu.twin_trace(calls, p, a, lh)
    tr = np.zeros((EC_WIDTH, 1 << lh), np.uint32)
    tr[:EC_WIDTH - 1] = body
    tr[EC_WIDTH - 1, :len(calls)] = rec["ec_ts_%d" % i]
    return tr, rows_trace(rec["ecio_rows_%d" % i], EC_IO_WIDTH, H[A_EC_IO(i)]), bw.astype(np.int64), tup.astype(np.int64)


def cmp256_traces(rec, H):
    """int256 calls with opcodes 6..8 and the 256-bit branches 12..17: the tests' twin of the comparison chip's trace (and its lookup counts) +
    the call's timestamp + the branch columns (is_br | neg | taken | the opcode the adapter announces)"""
    import int256_util as iu

    recs = np.ascontiguousarray(rec["cmp256_records"], dtype=np.uint32).reshape(-1, 17)
    val = lambda w: sum(int(x) << (32 * k) for k, x in enumerate(w))  # noqa: E731
    lh = H[A_CMP256]
    core = {12: 8, 13: 8, 14: 6, 16: 6, 15: 7, 17: 7}     # the comparison a branch opcode rests on: eq / sltu / slt
    body, bw = iu.cmp_twin_trace([(core.get(int(r[0]), int(r[0])), val(r[1:9]), val(r[9:17])) for r in recs], lh)
    n_core = body.shape[0]                                  # 103: the stand-alone chip's columns
    tr = np.zeros((CMP256_WIDTH, 1 << lh), np.uint32)
    tr[:n_core] = body
    tr[n_core, :len(recs)] = rec["cmp256_ts"]
    for row, r in enumerate(recs):
        op = int(r[0])
        is_br, neg = int(op >= 12), int(op in (13, 16, 17))
        out = int(body[65:97, row].sum() == 0) if core.get(op, op) == 8 else int(body[64, row])     # eq: no marker; less-than: t
        tr[n_core + 1, row], tr[n_core + 2, row], tr[n_core + 3, row], tr[n_core + 4, row] = is_br, neg, is_br & (out ^ neg), op
    return tr, bw.astype(np.int64)


def shift256_traces(rec, H):
    """int256 calls with opcodes 9..11: the tests' twin of the shift chip's trace (and its lookup counts) + the call's timestamp"""
    import int256_util as iu

    recs = np.ascontiguousarray(rec["sh256_records"], dtype=np.uint32).reshape(-1, 17)
    val = lambda w: sum(int(x) << (32 * k) for k, x in enumerate(w))  # noqa: E731
    lh = H[A_SHIFT256]
    body, bw, xc = iu.shift_twin_trace([(int(r[0]), val(r[1:9]), val(r[9:17])) for r in recs], lh)
    tr = np.zeros((SHIFT256_WIDTH, 1 << lh), np.uint32)
    tr[:SHIFT256_WIDTH - 1] = body
    tr[SHIFT256_WIDTH - 1, :len(recs)] = rec["sh256_ts"]
    return tr, bw.astype(np.int64), xc.astype(np.int64)


def fp2_traces(rec, H, i, p):
    """fp2 calls of field i: the tests' twin of the chip's trace (and its lookup counts) + the call's timestamp; the adapter's rows"""
    import fp2_util as fu

    recs = np.ascontiguousarray(rec["fp2_records_%d" % i], dtype=np.uint32).reshape(-1, fu.RECORD_WORDS)
    val = lambda w: sum(int(x) << (32 * k) for k, x in enumerate(w))  # noqa: E731
    calls = [(int(r[0]), (val(r[1:9]), val(r[9:17])), (val(r[17:25]), val(r[25:33]))) for r in recs]
    lh = H[A_FP2(i)]
    body, bw, tup = fu.twin_trace(calls, p, lh)
    tr = np.zeros((FP2_WIDTH, 1 << lh), np.uint32)
    tr[:FP2_WIDTH - 1] = body
    tr[FP2_WIDTH - 1, :len(calls)] = rec["fp2_ts_%d" % i]
    return tr, rows_trace(rec["fp2io_rows_%d" % i], EC_IO_WIDTH, H[A_FP2_IO(i)]), bw.astype(np.int64), tup.astype(np.int64)


def modmul_traces(rec, H, i, p):
    """modmul calls of modulus i: the oracle's chip trace (and its lookup counts) + the call's timestamp; the adapter's rows"""
    n = len(rec["mm_ts_%d" % i])
    l = ora.lib()
    l.ora_modular_trace.restype = C.c_size_t
    l.ora_modular_trace.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint8), C.c_uint, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]
    lh = H[A_MODMUL(i)]
    body = np.zeros((MODMUL_WIDTH - 1, 1 << lh), np.uint32)
    bw, tup = np.zeros(1 << 16, np.uint32), np.zeros(SX * SY, np.uint32)
    all_words = np.ascontiguousarray(rec["mm_records_%d" % i], dtype=np.uint32).reshape(-1, 17)   # op | a | b
    ops = np.ascontiguousarray(all_words[:, 0])
    recs = np.ascontiguousarray(all_words[:, 1:]).view(np.uint8)
    pb = np.frombuffer(int(p).to_bytes(32, "little"), dtype=np.uint8).copy()
    bad = l.ora_modular_trace(recs.ctypes.data_as(C.POINTER(C.c_uint8)) if n else None, ora.p32(ops) if n else None, n, pb.ctypes.data_as(C.POINTER(C.c_uint8)), lh,
                              ora.p32(body), ora.p32(bw), ora.p32(tup), SY)
    assert bad == 0
    tr = np.zeros((MODMUL_WIDTH, 1 << lh), np.uint32)
    tr[:MODMUL_WIDTH - 1] = body
    tr[MODMUL_WIDTH - 1, :n] = rec["mm_ts_%d" % i]
    return tr, rows_trace(rec["mmio_rows_%d" % i], MODMUL_IO_WIDTH, H[A_MODMUL_IO(i)]), bw.astype(np.int64), tup.astype(np.int64)


def sha256_traces(rec, H):
    """sha256 calls: the oracle's compression-chip trace + the call's timestamp on its rows; the adapter's rows as the executor wrote them"""
    n = len(rec["sha_ts"])
    tr = np.zeros((SHA_WIDTH, 1 << H[A_SHA256]), np.uint32)
    lib = ora.lib()
    lib.ora_sha256_trace.restype = None
    lib.ora_sha256_trace.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32)]
    body = np.zeros((SHA_WIDTH - 1, 1 << H[A_SHA256]), np.uint32)
    blocks = np.ascontiguousarray(rec["sha_blocks"], dtype=np.uint32)
    lib.ora_sha256_trace(ora.p32(blocks) if n else None, n, H[A_SHA256], ora.p32(body))
    tr[:SHA_WIDTH - 1] = body
    tr[SHA_WIDTH - 1, :65 * n] = np.repeat(rec["sha_ts"], 65)
    return tr, rows_trace(rec["shaio_rows"], SHA_IO_WIDTH, H[A_SHA256_IO])


# ---- the native field / extension / castf chips (include/zkhip_vm_circuit.hpp native_arith_air, native_ext_air, castf_vm_air): the rows the
# ---- device makes from the executor's call records, restated with Python integers ------------------------------------------------
def _canonical_cols(v):
    lo, hi = v & 0xFFFF, v >> 16
    gap = P_HI - hi
    assert v < P and gap >= 0
    return [lo, hi, gap, 1 if gap == 0 else 0, 0 if gap == 0 else pow(gap, -1, P)]


def _access_cols(prev, at):
    gap = at - prev - 1
    assert 0 <= gap < 1 << (16 + GAP_HI_BITS)
    return [prev % P, gap & 0xFFFF, gap >> 16]


def _ext4_mul(a, b):
    t = [0] * 7
    for i in range(4):
        for j in range(4):
            t[i + j] += a[i] * b[j]
    return [(t[k] + 11 * (t[k + 4] if k < 3 else 0)) % P for k in range(4)]


def _ext4_inv(a):
    r, e, b = [1, 0, 0, 0], P ** 4 - 2, list(a)
    while e:
        if e & 1:
            r = _ext4_mul(r, b)
        b = _ext4_mul(b, b)
        e >>= 1
    return r


def native_arith_trace(rec, H):
    recs = rec["nat_records"].reshape(-1, 9).astype(np.int64).tolist()
    tr = np.zeros((NATIVE_ARITH_WIDTH, 1 << H[A_NATIVE_ARITH]), np.uint32)
    for r, (op, bw, cw, prev, base, ts, *pts) in enumerate(recs):
        b_, c_ = bw % P, cw % P
        inv = pow(c_, -1, P) if op == 3 else 0
        a_ = [(b_ + c_) % P, (b_ - c_) % P, b_ * c_ % P, b_ * inv % P][op]
        flags = [1 if op == k else 0 for k in range(4)]
        row = [ts, base, bw & 0xFFFF, bw >> 16, cw & 0xFFFF, cw >> 16] + _canonical_cols(a_) + [prev & 0xFFFF, prev >> 16] + flags + [inv]
        for k in range(3):
            row += _access_cols(pts[k], ts + 5)
        tr[:, r] = row
    return tr


def native_ext_trace(rec, H):
    recs = rec["next_records"].reshape(-1, 27).astype(np.int64).tolist()
    tr = np.zeros((NATIVE_EXT_WIDTH, 1 << H[A_NATIVE_EXT]), np.uint32)
    for r, w in enumerate(recs):
        op, xw, yw, zp, base, ts, pts = w[0], w[1:5], w[5:9], w[9:13], w[13], w[14], w[15:27]
        x, y = [v % P for v in xw], [v % P for v in yw]
        inv = _ext4_inv(y) if op == 3 else [0, 0, 0, 0]
        z = [(u + v) % P for u, v in zip(x, y)] if op == 0 else [(u - v) % P for u, v in zip(x, y)] if op == 1 else _ext4_mul(x, y) if op == 2 else _ext4_mul(x, inv)
        row = [ts, base]
        for v in xw:
            row += [v & 0xFFFF, v >> 16]
        for v in yw:
            row += [v & 0xFFFF, v >> 16]
        for v in z:
            row += _canonical_cols(v)
        for v in zp:
            row += [v & 0xFFFF, v >> 16]
        row += [1 if op == k else 0 for k in range(4)] + inv
        for k in range(12):
            row += _access_cols(pts[k], ts + 5)
        tr[:, r] = row
    return tr


def castf_trace(rec, H):
    recs = rec["castf_records"].reshape(-1, 6).astype(np.int64).tolist()
    tr = np.zeros((CASTF_WIDTH, 1 << H[A_CASTF]), np.uint32)
    for r, (x, prev, base, ts, p0, p1) in enumerate(recs):
        assert x < 1 << 30
        limbs = [(x >> (8 * i)) & 255 for i in range(4)]
        tr[:, r] = [ts, base] + limbs + [4 * limbs[3], prev & 0xFFFF, prev >> 16] + _access_cols(p0, ts + 5) + _access_cols(p1, ts + 5) + [1]
    return tr


def segment_instance(rec, words, pc_base, heights):
    """The 24 AIR dicts (program, shapes, trace, pvs[, prep]) of one segment, traces from the CPU twins."""
    H = heights
    all_shapes = vm_airs()
    prog = program_table(words, pc_base, H[A_PROGRAM])
    tr = [None] * N_AIRS
    freq, bad = ora.program_freq_trace(rec["pc_index"], H[A_PROGRAM])
    assert bad == 0
    tr[A_PROGRAM] = freq.reshape(1, -1)
    tr[A_FRAME] = frame_trace(rec, prog, H[A_FRAME])
    tr[A_ALU], xc, _ = ora.rv32_alu_trace(rec["alu_op"], rec["alu_b"], rec["alu_c"], H[A_ALU])
    tr[A_LT], rc, _ = ora.rv32_lt_trace(rec["lt_op"], rec["lt_b"], rec["lt_c"], H[A_LT])
    tr[A_MUL], tc = ora.rv32_mul_trace(rec["mul_b"], rec["mul_c"], H[A_MUL], SX, SY)
    tr[A_SHIFT], rc, xc, _ = ora.rv32_shift_trace(rec["shift_op"], rec["shift_b"], rec["shift_c"], H[A_SHIFT], range_counts=rc, xor_counts=xc)
    tr[A_MULH], tc, rc, _ = ora.rv32_mulh_trace(rec["mulh_op"], rec["mulh_b"], rec["mulh_c"], H[A_MULH], SX, SY, tuple_counts=tc, range_counts=rc)
    tr[A_DIVREM], tc, rc, _ = ora.rv32_divrem_trace(rec["div_op"], rec["div_b"], rec["div_c"], H[A_DIVREM], SX, SY, tuple_counts=tc, range_counts=rc)
    tr[A_BEQ], _ = ora.rv32_branch_eq_trace(rec["beq_op"], rec["beq_a"], rec["beq_b"], rec["beq_imm"], H[A_BEQ])
    tr[A_BLT], rc, _ = ora.rv32_branch_lt_trace(rec["blt_op"], rec["blt_a"], rec["blt_b"], rec["blt_imm"], H[A_BLT], range_counts=rc)
    tr[A_JAL_LUI], rc, _ = ora.rv32_jal_lui_trace(rec["jal_op"], rec["jal_pc"], rec["jal_imm"], H[A_JAL_LUI], range_counts=rc)
    tr[A_AUIPC], rc, _ = ora.rv32_auipc_trace(rec["auipc_pc"], rec["auipc_imm"], H[A_AUIPC], range_counts=rc)
    tr[A_JALR], rc, _ = ora.rv32_jalr_trace(rec["jalr_pc"], rec["jalr_rs1"], rec["jalr_imm"], H[A_JALR], range_counts=rc)
    tr[A_LS], rc = loadstore_trace(rec, H[A_LS], rc)
    tr[A_ECALL] = rows_trace(rec["ecall_rows"], ECALL_WIDTH, H[A_ECALL])
    leaf_pad = np.zeros(LEAF_WIDTH, np.uint32)
    leaf_pad[0] = 1
    tr[A_LEAF] = rows_trace(rec["leaf_rows"], LEAF_WIDTH, H[A_LEAF], leaf_pad)
    tr[A_MERKLE] = rows_trace(rec["merkle_rows"], MERKLE_WIDTH, H[A_MERKLE])
    p2in = rec["p2_inputs"].reshape(-1, 16)
    p2 = np.zeros((299, 1 << H[A_POSEIDON2]), np.uint32)
    p2[:298] = ora.poseidon2_air_trace(p2in, H[A_POSEIDON2])
    p2[298, :len(p2in)] = 1
    tr[A_POSEIDON2] = p2
    if A_KECCAK in H.ids:
        tr[A_KECCAK], tr[A_KECCAK_IO] = keccak_traces(rec, H)
    if A_SHA256 in H.ids:
        tr[A_SHA256], tr[A_SHA256_IO] = sha256_traces(rec, H)
    i256_xor = None
    mul_counts = None
    if A_INT256 in H.ids:
        tr[A_INT256], tr[A_INT256_IO], i256_xor = int256_traces(rec, H)
        tr[A_MUL256], mul_bw, mul_tup = mul256_traces(rec, H)
        tr[A_CMP256], cmp_bw = cmp256_traces(rec, H)
        tr[A_SHIFT256], sh_bw, sh_xor = shift256_traces(rec, H)
        i256_xor = i256_xor + sh_xor
        mul_counts = (mul_bw + cmp_bw + sh_bw, mul_tup)
    mm_counts = []
    for i, p_ in enumerate(H.moduli):
        tr[A_MODMUL(i)], tr[A_MODMUL_IO(i)], bw_i, tup_i = modmul_traces(rec, H, i, p_)
        mm_counts.append((bw_i, tup_i))
    for i, cv in enumerate(H.curves):
        tr[A_EC(i)], tr[A_EC_IO(i)], bw_i, tup_i = ec_traces(rec, H, i, cv)
        mm_counts.append((bw_i, tup_i))
    for i, p_ in enumerate(H.fp2):
        tr[A_FP2(i)], tr[A_FP2_IO(i)], bw_i, tup_i = fp2_traces(rec, H, i, p_)
        mm_counts.append((bw_i, tup_i))
    if A_NATIVE_ARITH in H.ids:
        tr[A_NATIVE_ARITH], tr[A_NATIVE_EXT] = native_arith_trace(rec, H), native_ext_trace(rec, H)
    if A_CASTF in H.ids:
        tr[A_CASTF] = castf_trace(rec, H)
    meta = rec["meta"]
    pc_start, pc_end, ts_end, n_instr = (int(x) for x in meta[:4])
    tr[A_CONNECTOR] = np.array([[ts_end & 0xFFFF], [ts_end >> 16]], np.uint32)
    # lookup multiplicities of the new chips
    rc = rc.astype(np.int64)
    tc = tc.astype(np.int64)
    for bw_i, tup_i in mm_counts:   # the multiplication chips' own lookups (byte pairs, carries)
        rc += bw_i
        tc += tup_i
    if mul_counts is not None:     # the 256-bit multiplication chip's byte pairs and carries
        rc += mul_counts[0]
        tc += mul_counts[1]
    if i256_xor is not None:       # the 256-bit ALU's lookups go to the XOR column
        xc = ((xc.astype(np.int64) + i256_xor) % P).astype(np.uint32)
    n, nls = n_instr, len(rec["ls_case"])
    fr, ls = tr[A_FRAME].astype(np.int64), tr[A_LS].astype(np.int64)
    for q in (18, 20, 22, 24, 26, 28):
        rc += np.bincount(fr[q][:n] * 256 + fr[q + 1][:n], minlength=1 << 16)
    for q in (0, 2, 4, 6):
        rc += np.bincount(ls[q][:nls] * 256 + ls[q + 1][:nls], minlength=1 << 16)
    cnt = np.zeros(1 << 16, np.int64)
    for q in (35, 38, 41):                              # timestamp gaps: gap_lo in the range table, (0, gap_hi) in the range-tuple table
        cnt += np.bincount(fr[q][:n], minlength=1 << 16)
        tc += np.bincount(fr[q + 1][:n], minlength=SX * SY)
    cnt += np.bincount(ls[46][:nls], minlength=1 << 16)
    tc += np.bincount(ls[47][:nls], minlength=SX * SY)
    for q, s in ((40, 1), (41, 4), (44, 4), (44, 1), (41, 1)):
        cnt += np.bincount(ls[q][:nls] * s, minlength=1 << 16)
    misc = [ts_end & 0xFFFF, 8 * (ts_end >> 16), ts_end >> 16]
    for row in rec["ecall_rows"].reshape(-1, ECALL_WIDTH):
        if row[15]:
            misc += [int(row[17]) * 8192, int(row[17]), int(row[22]), int(row[25])]
            tc[int(row[23])] += 1
            tc[int(row[26])] += 1
        if row[16]:
            rc[int(row[10]) * 256 + int(row[11])] += 1
            rc[int(row[12]) * 256 + int(row[13])] += 1
        if row[30] or row[31] or row[32] or row[33] or row[34] or row[35]:
            misc += [int(row[22])]
            tc[int(row[23])] += 1
        if row[37]:                                          # a 256-bit branch: the a2 read's gap, the offset's sign split
            misc += [int(row[43]), (int(row[40]) - 32768 * int(row[41])) * 2]
            tc[int(row[44])] += 1
        if row[27] or row[29] or row[30] or row[31] or row[32] or row[33] or row[34] or row[35] or row[36]:
            misc += [int(row[28]) * 1024, int(row[28]), int(row[9]) * 1024]
    for row in rec["kio_rows"].reshape(-1, KECCAK_IO_WIDTH):
        misc += [int(row[36]), int(row[39])]
        tc[int(row[37])] += 1
        tc[int(row[40])] += 1
    for row in rec["shaio_rows"].reshape(-1, SHA_IO_WIDTH):
        misc += [int(row[31])]
        tc[int(row[32])] += 1
    for row in rec["i256io_rows"].reshape(-1, INT256_IO_WIDTH):
        misc += [int(row[31])]
        tc[int(row[32])] += 1
    for i in range(len(H.moduli)):
        for row in rec["mmio_rows_%d" % i].reshape(-1, MODMUL_IO_WIDTH):
            misc += [int(row[31])]
            tc[int(row[32])] += 1
    for i in range(len(H.curves)):
        for row in rec["ecio_rows_%d" % i].reshape(-1, EC_IO_WIDTH):
            misc += [int(row[55])]
            tc[int(row[56])] += 1
    for i in range(len(H.fp2)):
        for row in rec["fp2io_rows_%d" % i].reshape(-1, EC_IO_WIDTH):
            misc += [int(row[55])]
            tc[int(row[56])] += 1
    # the native chips: (lo, hi, hi_gap) of every canonical result word, every access's gap_lo / (0, gap_hi); castf's limbs as byte pairs
    if A_NATIVE_ARITH in H.ids:
        t_ = tr[A_NATIVE_ARITH].astype(np.int64)[:, :len(rec["nat_records"]) // 9]
        for q in (6, 7, 8, 19, 22, 25):
            cnt += np.bincount(t_[q], minlength=1 << 16)
        for q in (20, 23, 26):
            tc += np.bincount(t_[q], minlength=SX * SY)
        t_ = tr[A_NATIVE_EXT].astype(np.int64)[:, :len(rec["next_records"]) // 27]
        for q in [18 + 5 * i + k for i in range(4) for k in range(3)] + [55 + 3 * k for k in range(12)]:
            cnt += np.bincount(t_[q], minlength=1 << 16)
        for q in [56 + 3 * k for k in range(12)]:
            tc += np.bincount(t_[q], minlength=SX * SY)
    if A_CASTF in H.ids:
        t_ = tr[A_CASTF].astype(np.int64)[:, :len(rec["castf_records"]) // 6]
        for qx, qy in ((2, 3), (4, 5)):
            rc += np.bincount(t_[qx] * 256 + t_[qy], minlength=1 << 16)
        rc += np.bincount(t_[6] * 256, minlength=1 << 16)
        for q in (10, 13):
            cnt += np.bincount(t_[q], minlength=1 << 16)
        for q in (11, 14):
            tc += np.bincount(t_[q], minlength=SX * SY)
    for row in rec["leaf_rows"].reshape(-1, LEAF_WIDTH):
        misc += [int(row[39]), int(row[40]) * 16, int(row[40]), int(row[41]), int(row[42]) * 64, int(row[42])]
    cnt += np.bincount(np.array(misc, dtype=np.int64), minlength=1 << 16)
    tc = (tc % P).astype(np.uint32)
    tr[A_BITWISE] = np.stack([(rc % P).astype(np.uint32), xc])
    tr[A_RANGE_TUPLE] = tc.reshape(1, -1)
    tr[A_RANGE] = (cnt % P).astype(np.uint32).reshape(1, -1)
    from zkvm_prover_amd import air

    preps = {A_PROGRAM: prog, A_BITWISE: air.bitwise_lookup_prep(8), A_RANGE_TUPLE: air.range_tuple_prep(SX, SY),
             A_RANGE: np.arange(1 << 16, dtype=np.uint32).reshape(1, -1)}
    if A_SHA256 in H.ids:
        preps[A_SHA256] = sha256_vm_prep(H[A_SHA256])
    pvs = {A_MERKLE: np.concatenate([meta[4:12], meta[12:20]]).astype(np.uint32), A_CONNECTOR: np.array([pc_start, pc_end], np.uint32)}
    out = []
    for a in H.ids:   # proof order
        if a >= A_FP2(0):
            i = (a - A_FP2(0)) // 2
            program, width = z.vm_fp2_air(H.fp2[i], i, (a - A_FP2(0)) & 1)
            n_pvs, pw = 0, 0
        elif a >= A_EC(0):
            i = (a - A_EC(0)) // 2
            program, width = z.vm_ec_air(H.curves[i][0], H.curves[i][1], i, (a - A_EC(0)) & 1)
            n_pvs, pw = 0, 0
        elif a >= N_STATIC_AIRS:
            i = (a - N_STATIC_AIRS) // 2
            program, width = z.vm_modmul_air(H.moduli[i], i, (a - N_STATIC_AIRS) & 1)
            n_pvs, pw = 0, 0
        else:
            program, width, n_pvs, pw = all_shapes[a]
        d = dict(program=program, log_height=H[a], width=width, n_pvs=n_pvs, trace=tr[a], pvs=pvs.get(a, NOPV))
        assert tr[a].shape == (width, 1 << H[a]), (a, tr[a].shape, width, H[a])
        if pw:
            d["prep"] = preps[a]
        out.append(d)
    return out


# ---- exact bus accounting (what the LogUp argument enforces cryptographically): per bus, the multiset of sent messages equals the
# ---- multiset of received ones -------------------------------------------------------------------------------------------------
def _eval_nodes(program, trace, pvs, prep):
    from zkvm_prover_amd import air

    w = [int(x) for x in program]
    n_nodes, n_cons = w[1], w[2]
    nodes = [tuple(w[4 + 3 * i: 7 + 3 * i]) for i in range(n_nodes)]
    n = trace.shape[1]
    t = trace.astype(np.int64)
    pt = None if prep is None else np.asarray(prep).astype(np.int64)
    vals = [None] * n_nodes
    for i, (op, a, b) in enumerate(nodes):
        if op == air.OP_VAR:
            vals[i] = np.roll(t[a], -1) if b else t[a]
        elif op == air.OP_PREP:
            vals[i] = np.roll(pt[a], -1) if b else pt[a]
        elif op == air.OP_PUB:
            vals[i] = np.full(n, int(pvs[a]), dtype=np.int64)
        elif op == air.OP_CONST:
            vals[i] = np.full(n, a, dtype=np.int64)
        elif op in (air.OP_ADD, air.OP_SUB, air.OP_MUL):
            x, y = vals[a], vals[b]
            if x is None or y is None:
                continue
            vals[i] = (x + y) % P if op == air.OP_ADD else (x - y) % P if op == air.OP_SUB else (x * y) % P
        elif op == air.OP_NEG:
            vals[i] = None if vals[a] is None else (-vals[a]) % P
    pos = 4 + 3 * n_nodes + n_cons
    if pos + 2 <= len(w) and w[pos] == air.PREP_MAGIC:
        pos += 2
    if pos + 2 <= len(w) and w[pos] == air.CACHED_MAGIC:
        pos += 2
    ints = []
    if pos < len(w):
        assert w[pos] == air.LOGUP_MAGIC
        n_int = w[pos + 1]
        pos += 2
        for _ in range(n_int):
            bus, sign, count, nf = w[pos:pos + 4]
            fields = w[pos + 4:pos + 4 + nf]
            pos += 4 + nf + 1
            ints.append((bus, sign, count, fields))
    return vals, ints


def bus_imbalance(instance):
    """{bus: number of distinct messages whose sends and receives do not cancel (mod p)}; {} = every bus balances."""
    per_bus = {}
    for d in instance:
        vals, ints = _eval_nodes(d["program"], d["trace"], d["pvs"], d.get("prep"))
        for bus, sign, count, fields in ints:
            c = vals[count]
            nz = np.nonzero(c)[0]
            if not len(nz):
                continue
            msg = np.stack([vals[f][nz] for f in fields], axis=1)
            cnt = c[nz] if sign == 0 else (-c[nz]) % P
            per_bus.setdefault(bus, []).append((msg, cnt))
    bad = {}
    for bus, parts in per_bus.items():
        width = max(m.shape[1] for m, _ in parts)
        assert all(m.shape[1] == width for m, _ in parts), "bus %d carries messages of different lengths" % bus
        msg = np.concatenate([m for m, _ in parts])
        cnt = np.concatenate([c for _, c in parts])
        uniq, inv = np.unique(msg, axis=0, return_inverse=True)
        tot = np.zeros(len(uniq), dtype=np.int64)
        np.add.at(tot, inv.reshape(-1), cnt)
        n_bad = int(np.count_nonzero(tot % P))
        if n_bad:
            bad[bus] = n_bad
    return bad
