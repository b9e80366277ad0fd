"""ctypes wrapper of oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE: imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

def effective_cpus():
    """CPUs this process may actually use: min(affinity, cgroup CPU quota).  A container can SEE every host thread
    (nproc = 256 on the GPU box) while its cgroup grants far fewer (cpu.max = 16 CPUs there): OpenMP's default of one
    thread per visible CPU then spends its time being throttled (measured: 2^19-row proof 0.52 s with 16 threads, 58 s
    with 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


# must happen before libgomp starts (first dlopen of an OpenMP library in this process)
os.environ.setdefault("OMP_NUM_THREADS", str(effective_cpus()))

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORA_DIR = os.path.join(_ROOT, "oracle")
_LIB = None
P = 2013265921


def range_counts(values, log_table, counts=None):
    """(counts, n_out_of_range): multiplicity column of a range table for the canonical values."""
    values = np.ascontiguousarray(values, dtype=np.uint32).reshape(-1)
    acc = counts is not None
    counts = np.ascontiguousarray(counts, dtype=np.uint32).copy() if acc else np.zeros(1 << log_table, np.uint32)
    bad = lib().ora_range_counts(p32(values), len(values), log_table, p32(counts), 1 if acc else 0)
    return counts, int(bad)


def range_tuple_counts(xs, ys, size_x, size_y, counts=None):
    xs, ys = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (xs, ys))
    acc = counts is not None
    counts = np.ascontiguousarray(counts, dtype=np.uint32).copy() if acc else np.zeros(size_x * size_y, np.uint32)
    l = lib()
    l.ora_range_tuple_counts.restype = C.c_size_t
    l.ora_range_tuple_counts.argtypes = [u32p_t(), u32p_t(), C.c_size_t, C.c_uint32, C.c_uint32, u32p_t(), C.c_int]
    bad = l.ora_range_tuple_counts(p32(xs), p32(ys), len(xs), size_x, size_y, p32(counts), 1 if acc else 0)
    return counts, int(bad)


def bitwise_lookup_counts(xs, ys, ops, bits=8):
    xs, ys, ops = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (xs, ys, ops))
    tr = np.zeros((2, 1 << (2 * bits)), np.uint32)
    l = lib()
    l.ora_bitwise_lookup_counts.restype = C.c_size_t
    l.ora_bitwise_lookup_counts.argtypes = [u32p_t(), u32p_t(), u32p_t(), C.c_size_t, C.c_uint, u32p_t(), C.c_int]
    bad = l.ora_bitwise_lookup_counts(p32(xs), p32(ys), p32(ops), len(xs), bits, p32(tr), 0)
    return tr, int(bad)


def rv32_alu_trace(opc, bs, cs, log_height, xor_counts=None):
    opc, bs, cs = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (opc, bs, cs))
    tr = np.zeros((18, 1 << log_height), np.uint32)
    xc = np.zeros(1 << 16, np.uint32) if xor_counts is None else np.ascontiguousarray(xor_counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_rv32_alu_trace.restype = C.c_size_t
    l.ora_rv32_alu_trace.argtypes = [u32p_t()] * 3 + [C.c_size_t, C.c_uint, u32p_t(), u32p_t()]
    bad = l.ora_rv32_alu_trace(p32(opc), p32(bs), p32(cs), len(opc), log_height, p32(tr), p32(xc))
    return tr, xc, int(bad)


def memory_access_trace(as_, ptr, prev_data, prev_ts, data, ts, is_read, log_height):
    cols = [np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (as_, ptr, prev_data, prev_ts, data, ts, is_read)]
    tr = np.zeros((10, 1 << log_height), np.uint32)
    l = lib()
    l.ora_memory_access_trace.restype = C.c_size_t
    l.ora_memory_access_trace.argtypes = [u32p_t()] * 7 + [C.c_size_t, C.c_uint, u32p_t()]
    bad = l.ora_memory_access_trace(*[p32(c) for c in cols], cols[0].size, log_height, p32(tr))
    return tr, int(bad)


def rv32_shift_trace(opc, bs, cs, log_height, range_counts=None, xor_counts=None):
    opc, bs, cs = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (opc, bs, cs))
    tr = np.zeros((32, 1 << log_height), np.uint32)
    rc = np.zeros(1 << 16, np.uint32) if range_counts is None else np.ascontiguousarray(range_counts, dtype=np.uint32).copy()
    xc = np.zeros(1 << 16, np.uint32) if xor_counts is None else np.ascontiguousarray(xor_counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_rv32_shift_trace.restype = C.c_size_t
    l.ora_rv32_shift_trace.argtypes = [u32p_t()] * 3 + [C.c_size_t, C.c_uint, u32p_t(), u32p_t(), u32p_t()]
    bad = l.ora_rv32_shift_trace(p32(opc), p32(bs), p32(cs), len(opc), log_height, p32(tr), p32(rc), p32(xc))
    return tr, rc, xc, int(bad)


def rv32_branch_eq_trace(opc, a, b, imm, log_height):
    opc, a, b, imm = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (opc, a, b, imm))
    tr = np.zeros((17, 1 << log_height), np.uint32)
    l = lib()
    l.ora_rv32_branch_eq_trace.restype = C.c_size_t
    l.ora_rv32_branch_eq_trace.argtypes = [u32p_t()] * 4 + [C.c_size_t, C.c_uint, u32p_t()]
    bad = l.ora_rv32_branch_eq_trace(p32(opc), p32(a), p32(b), p32(imm), len(opc), log_height, p32(tr))
    return tr, int(bad)


def rv32_branch_lt_trace(opc, a, b, imm, log_height, range_counts=None):
    opc, a, b, imm = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (opc, a, b, imm))
    tr = np.zeros((23, 1 << log_height), np.uint32)
    rc = np.zeros(1 << 16, np.uint32) if range_counts is None else np.ascontiguousarray(range_counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_rv32_branch_lt_trace.restype = C.c_size_t
    l.ora_rv32_branch_lt_trace.argtypes = [u32p_t()] * 4 + [C.c_size_t, C.c_uint, u32p_t(), u32p_t()]
    bad = l.ora_rv32_branch_lt_trace(p32(opc), p32(a), p32(b), p32(imm), len(opc), log_height, p32(tr), p32(rc))
    return tr, rc, int(bad)


def _records_trace(fn, width, cols, log_height, range_counts):
    cols = [np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in cols]
    tr = np.zeros((width, 1 << log_height), np.uint32)
    rc = np.zeros(1 << 16, np.uint32) if range_counts is None else np.ascontiguousarray(range_counts, dtype=np.uint32).copy()
    f = getattr(lib(), fn)
    f.restype = C.c_size_t
    f.argtypes = [u32p_t()] * len(cols) + [C.c_size_t, C.c_uint, u32p_t(), u32p_t()]
    bad = f(*[p32(c) for c in cols], len(cols[0]), log_height, p32(tr), p32(rc))
    return tr, rc, int(bad)


def rv32_jal_lui_trace(opc, pc, imm, log_height, range_counts=None):
    return _records_trace("ora_rv32_jal_lui_trace", 9, (opc, pc, imm), log_height, range_counts)


def rv32_auipc_trace(pc, imm, log_height, range_counts=None):
    return _records_trace("ora_rv32_auipc_trace", 14, (pc, imm), log_height, range_counts)


def rv32_loadstore_trace(case, read, prev, log_height, range_counts=None):
    return _records_trace("ora_rv32_loadstore_trace", 33, (case, read, prev), log_height, range_counts)


def rv32_jalr_trace(pc, rs1, imm, log_height, range_counts=None):
    return _records_trace("ora_rv32_jalr_trace", 20, (pc, rs1, imm), log_height, range_counts)


def rv32_mulh_trace(opc, bs, cs, log_height, size_x=256, size_y=2048, tuple_counts=None, range_counts=None):
    opc, bs, cs = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (opc, bs, cs))
    tr = np.zeros((21, 1 << log_height), np.uint32)
    tc = np.zeros(size_x * size_y, np.uint32) if tuple_counts is None else np.ascontiguousarray(tuple_counts, dtype=np.uint32).copy()
    rc = np.zeros(1 << 16, np.uint32) if range_counts is None else np.ascontiguousarray(range_counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_rv32_mulh_trace.restype = C.c_size_t
    l.ora_rv32_mulh_trace.argtypes = [u32p_t()] * 3 + [C.c_size_t, C.c_uint, u32p_t(), u32p_t(), C.c_uint32, u32p_t()]
    bad = l.ora_rv32_mulh_trace(p32(opc), p32(bs), p32(cs), len(opc), log_height, p32(tr), p32(tc), size_y, p32(rc))
    return tr, tc, rc, int(bad)


def rv32_divrem_trace(opc, bs, cs, log_height, size_x=256, size_y=2048, tuple_counts=None, range_counts=None):
    opc, bs, cs = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (opc, bs, cs))
    tr = np.zeros((41, 1 << log_height), np.uint32)
    tc = np.zeros(size_x * size_y, np.uint32) if tuple_counts is None else np.ascontiguousarray(tuple_counts, dtype=np.uint32).copy()
    rc = np.zeros(1 << 16, np.uint32) if range_counts is None else np.ascontiguousarray(range_counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_rv32_divrem_trace.restype = C.c_size_t
    l.ora_rv32_divrem_trace.argtypes = [u32p_t()] * 3 + [C.c_size_t, C.c_uint, u32p_t(), u32p_t(), C.c_uint32, u32p_t()]
    bad = l.ora_rv32_divrem_trace(p32(opc), p32(bs), p32(cs), len(opc), log_height, p32(tr), p32(tc), size_y, p32(rc))
    return tr, tc, rc, int(bad)


def mmcs_path_trace(leaf, index, path_start, step_kind, step_digest, log_height):
    """-> (trace [39, N], hash_inputs [rows, 16], claims [n_claims, 18], bad)"""
    leaf, index, path_start, step_kind, step_digest = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1)
                                                       for v in (leaf, index, path_start, step_kind, step_digest))
    n_paths, rows = len(index), int(path_start[-1]) if len(path_start) else 0
    tr = np.zeros((39, 1 << log_height), np.uint32)
    hin = np.zeros((max(rows, 1), 16), np.uint32)
    claims = np.zeros((len(step_kind) + n_paths + 1, 18), np.uint32)
    nc = C.c_size_t(0)
    l = lib()
    l.ora_mmcs_path_trace.restype = C.c_size_t
    l.ora_mmcs_path_trace.argtypes = [u32p_t()] * 5 + [C.c_size_t, C.c_uint, u32p_t(), u32p_t(), u32p_t(), C.POINTER(C.c_size_t)]
    bad = l.ora_mmcs_path_trace(p32(leaf), p32(index), p32(path_start), p32(step_kind), p32(step_digest), n_paths, log_height, p32(tr), p32(hin),
                                p32(claims), C.byref(nc))
    return tr, hin[:rows], claims[:nc.value], int(bad)


def _plain_trace(fn, width, cols, n, log_height):
    cols = [np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in cols]
    tr = np.zeros((width, 1 << log_height), np.uint32)
    f = getattr(lib(), fn)
    f.restype = C.c_size_t
    f.argtypes = [u32p_t()] * len(cols) + [C.c_size_t, C.c_uint, u32p_t()]
    bad = f(*[p32(c) for c in cols], n, log_height, p32(tr))
    return tr, int(bad)


def field_arith_trace(opc, b, c, log_height):
    return _plain_trace("ora_field_arith_trace", 8, (opc, b, c), len(np.asarray(opc).reshape(-1)), log_height)


def field_ext_trace(opc, x, y, log_height):
    return _plain_trace("ora_field_ext_trace", 20, (opc, x, y), len(np.asarray(opc).reshape(-1)), log_height)


def var_range_counts(values, bits, max_bits, counts=None):
    """bits: an array (one per request) or an int (the same for every request)"""
    values = np.ascontiguousarray(values, dtype=np.uint32).reshape(-1)
    c = np.zeros(1 << (max_bits + 1), np.uint32) if counts is None else np.ascontiguousarray(counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_var_range_counts.restype = C.c_size_t
    l.ora_var_range_counts.argtypes = [u32p_t(), u32p_t(), C.c_uint32, C.c_size_t, C.c_uint, u32p_t()]
    if isinstance(bits, (int, np.integer)):
        bad = l.ora_var_range_counts(p32(values), None, int(bits), len(values), max_bits, p32(c))
    else:
        bits = np.ascontiguousarray(bits, dtype=np.uint32).reshape(-1)
        bad = l.ora_var_range_counts(p32(values), p32(bits), 0, len(values), max_bits, p32(c))
    return c, int(bad)


def castf_trace(xs, log_height, max_bits, counts=None):
    xs = np.ascontiguousarray(xs, dtype=np.uint32).reshape(-1)
    tr = np.zeros((6, 1 << log_height), np.uint32)
    c = np.zeros(1 << (max_bits + 1), np.uint32) if counts is None else np.ascontiguousarray(counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_castf_trace.restype = C.c_size_t
    l.ora_castf_trace.argtypes = [u32p_t(), C.c_size_t, C.c_uint, u32p_t(), u32p_t()]
    bad = l.ora_castf_trace(p32(xs), len(xs), log_height, p32(tr), p32(c))
    return tr, c, int(bad)


def rv32_lt_trace(opc, bs, cs, log_height, range_counts=None):
    opc, bs, cs = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (opc, bs, cs))
    tr = np.zeros((18, 1 << log_height), np.uint32)
    rc = np.zeros(1 << 16, np.uint32) if range_counts is None else np.ascontiguousarray(range_counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_rv32_lt_trace.restype = C.c_size_t
    l.ora_rv32_lt_trace.argtypes = [u32p_t()] * 3 + [C.c_size_t, C.c_uint, u32p_t(), u32p_t()]
    bad = l.ora_rv32_lt_trace(p32(opc), p32(bs), p32(cs), len(opc), log_height, p32(tr), p32(rc))
    return tr, rc, int(bad)


def rv32_mul_trace(bs, cs, log_height, size_x=256, size_y=8192, tuple_counts=None):
    bs, cs = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (bs, cs))
    tr = np.zeros((13, 1 << log_height), np.uint32)
    tc = np.zeros(size_x * size_y, np.uint32) if tuple_counts is None else np.ascontiguousarray(tuple_counts, dtype=np.uint32).copy()
    l = lib()
    l.ora_rv32_mul_trace.restype = None
    l.ora_rv32_mul_trace.argtypes = [u32p_t(), u32p_t(), C.c_size_t, C.c_uint, u32p_t(), u32p_t(), C.c_uint32]
    l.ora_rv32_mul_trace(p32(bs), p32(cs), len(bs), log_height, p32(tr), p32(tc), size_y)
    return tr, tc


def program_freq_trace(idx, log_height):
    idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1)
    out = np.zeros(1 << log_height, np.uint32)
    l = lib()
    l.ora_program_freq_trace.restype = C.c_size_t
    l.ora_program_freq_trace.argtypes = [u32p_t(), C.c_size_t, C.c_uint, u32p_t()]
    bad = l.ora_program_freq_trace(p32(idx), idx.size, log_height, p32(out))
    return out, bad


def exec_frame_trace(idx, program, log_height):
    """program: [9, n_program] canonical; returns ([10, 2^log_height] canonical, bad)"""
    idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1)
    prog = np.ascontiguousarray(program, dtype=np.uint32)
    out = np.zeros((10, 1 << log_height), np.uint32)
    l = lib()
    l.ora_exec_frame_trace.restype = C.c_size_t
    l.ora_exec_frame_trace.argtypes = [u32p_t(), C.c_size_t, u32p_t(), C.c_size_t, C.c_uint, u32p_t()]
    bad = l.ora_exec_frame_trace(p32(idx), idx.size, p32(prog), prog.shape[1], log_height, p32(out))
    return out, bad


def memory_boundary_trace(as_, ptr, init, fin, ts, as_bits, ptr_bits, log_height):
    as_, ptr, init, fin, ts = (np.ascontiguousarray(v, dtype=np.uint32).reshape(-1) for v in (as_, ptr, init, fin, ts))
    tr = np.zeros((8, 1 << log_height), np.uint32)
    l = lib()
    l.ora_memory_boundary_trace.restype = C.c_size_t
    l.ora_memory_boundary_trace.argtypes = [u32p_t()] * 5 + [C.c_size_t, C.c_uint, C.c_uint, C.c_uint, u32p_t()]
    bad = l.ora_memory_boundary_trace(p32(as_), p32(ptr), p32(init), p32(fin), p32(ts), len(as_), as_bits, ptr_bits, log_height, p32(tr))
    return tr, int(bad)


def poseidon2_air_trace(inputs, log_height):
    """Oracle trace (298 x 2^log_height, canonical, column-major) of the Poseidon2 AIR for inputs [n][16]."""
    inputs = np.ascontiguousarray(inputs, dtype=np.uint32).reshape(-1, 16)
    out = np.zeros((298, 1 << log_height), dtype=np.uint32)
    lib().ora_poseidon2_air_trace(p32(inputs), inputs.shape[0], log_height, p32(out))
    return out


class OraMatrix(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_size_t), ("log_height", C.c_uint), ("width", C.c_size_t)]


class OraChallenger(C.Structure):
    _fields_ = [("state", C.c_uint32 * 16), ("in_buf", C.c_uint32 * 8), ("n_in", C.c_uint),
                ("out_buf", C.c_uint32 * 8), ("n_out", C.c_uint)]


class OraParams(C.Structure):
    _fields_ = [("log_blowup", C.c_uint), ("log_final_poly_len", C.c_uint), ("num_queries", C.c_uint),
                ("commit_pow_bits", C.c_uint), ("query_pow_bits", C.c_uint)]


class OraAir(C.Structure):
    _fields_ = [("program", C.POINTER(C.c_uint32)), ("program_len", C.c_size_t), ("log_height", C.c_uint),
                ("width", C.c_size_t), ("trace", C.POINTER(C.c_uint32)), ("pvs", C.POINTER(C.c_uint32)),
                ("n_pvs", C.c_size_t), ("prep", C.POINTER(C.c_uint32)), ("prep_commit", C.POINTER(C.c_uint32))]


def build():
    so = os.path.join(_ORA_DIR, "liboracle.so")
    srcs = [os.path.join(_ORA_DIR, f) for f in os.listdir(_ORA_DIR) if f.endswith((".c", ".h", "Makefile"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _ORA_DIR], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        l = C.CDLL(build())
        u32p, sz = C.POINTER(C.c_uint32), C.c_size_t
        l.ora_mul.restype = C.c_uint32
        l.ora_mul.argtypes = [C.c_uint32, C.c_uint32]
        l.ora_inv.restype = C.c_uint32
        l.ora_inv.argtypes = [C.c_uint32]
        l.ora_pow.restype = C.c_uint32
        l.ora_pow.argtypes = [C.c_uint32, C.c_uint64]
        l.ora_two_adic_generator.restype = C.c_uint32
        l.ora_two_adic_generator.argtypes = [C.c_uint]
        l.ora_ext_mul.argtypes = [u32p, u32p, u32p]
        l.ora_ext_inv.argtypes = [u32p, u32p]
        l.ora_dft_batch.argtypes = [u32p, C.c_uint, sz, sz, C.c_int]
        l.ora_dft_naive.argtypes = [u32p, u32p, C.c_uint, C.c_int]
        l.ora_coset_lde_batch.argtypes = [u32p, sz, u32p, sz, C.c_uint, C.c_uint, sz, C.c_uint32, C.c_int]
        l.ora_poseidon2_round_constants.restype = u32p
        l.ora_poseidon2_permute.argtypes = [u32p]
        l.ora_hash_slice.argtypes = [u32p, sz, u32p]
        l.ora_compress.argtypes = [u32p, u32p, u32p]
        l.ora_poseidon2_air_trace.argtypes = [u32p, sz, C.c_uint, u32p]
        l.ora_range_counts.argtypes = [u32p, sz, C.c_uint, u32p, C.c_int]
        l.ora_range_counts.restype = sz
        l.ora_mmcs_commit.restype = C.c_void_p
        l.ora_mmcs_commit.argtypes = [C.POINTER(OraMatrix), sz, u32p]
        l.ora_tree_log_height.restype = C.c_uint
        l.ora_tree_log_height.argtypes = [C.c_void_p]
        l.ora_tree_layer.restype = u32p
        l.ora_tree_layer.argtypes = [C.c_void_p, C.c_uint]
        l.ora_mmcs_open.restype = sz
        l.ora_mmcs_open.argtypes = [C.c_void_p, sz, u32p]
        l.ora_mmcs_verify.restype = C.c_int
        l.ora_mmcs_verify.argtypes = [u32p, C.POINTER(C.c_uint), C.POINTER(sz), sz, sz, u32p]
        l.ora_tree_free.argtypes = [C.c_void_p]
        l.ora_ch_init.argtypes = [C.POINTER(OraChallenger)]
        l.ora_ch_observe.argtypes = [C.POINTER(OraChallenger), u32p, sz]
        l.ora_ch_sample.restype = C.c_uint32
        l.ora_ch_sample.argtypes = [C.POINTER(OraChallenger)]
        l.ora_ch_sample_bits.restype = C.c_uint32
        l.ora_ch_sample_bits.argtypes = [C.POINTER(OraChallenger), C.c_uint]
        l.ora_ch_grind.restype = C.c_uint32
        l.ora_ch_grind.argtypes = [C.POINTER(OraChallenger), C.c_uint]
        l.ora_fri_fold.argtypes = [u32p, u32p, C.c_uint, u32p]
        l.ora_ext_batch_inverse.argtypes = [u32p, u32p, sz]
        l.ora_logup_running_sum.argtypes = [u32p, u32p, sz, u32p]
        l.ora_mle_fold.argtypes = [u32p, u32p, sz, u32p]
        l.ora_sumcheck_round.argtypes = [C.POINTER(u32p), sz, sz, u32p]
        if hasattr(l, "ora_stark_prove"):
            l.ora_stark_prove.restype = sz
            l.ora_stark_prove.argtypes = [C.POINTER(OraParams), C.POINTER(OraAir), sz, u32p, sz]
            l.ora_stark_verify.restype = C.c_int
            l.ora_stark_verify.argtypes = [C.POINTER(OraParams), C.POINTER(OraAir), sz, u32p, sz]
        _LIB = l
    return _LIB


def u32p_t():
    return C.POINTER(C.c_uint32)


def p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def rand_field(rng, shape):
    return rng.integers(0, P, size=shape, dtype=np.uint64).astype(np.uint32)


def dft_batch(mat_colmajor, log_n, inverse=False):
    """mat: [width, n] uint32 canonical (row c = column c). Returns transformed copy."""
    a = np.ascontiguousarray(mat_colmajor, dtype=np.uint32).copy()
    lib().ora_dft_batch(p32(a), log_n, a.shape[0], a.shape[1], int(inverse))
    return a


def coset_lde_batch(mat_colmajor, log_n, added_bits, shift, bitrev_out=True):
    a = np.ascontiguousarray(mat_colmajor, dtype=np.uint32)
    w = a.shape[0]
    out = np.zeros((w, 1 << (log_n + added_bits)), dtype=np.uint32)
    lib().ora_coset_lde_batch(p32(a), a.shape[1], p32(out), out.shape[1], log_n, added_bits, w, shift,
                              int(bitrev_out))
    return out


def permute(state):
    s = np.ascontiguousarray(state, dtype=np.uint32).copy()
    lib().ora_poseidon2_permute(p32(s))
    return s


class Tree:
    def __init__(self, mats):
        """mats: list of [width, height] uint32 canonical arrays (column-major storage)."""
        self.mats = [np.ascontiguousarray(m, dtype=np.uint32) for m in mats]
        arr = (OraMatrix * len(mats))()
        for i, m in enumerate(self.mats):
            arr[i] = OraMatrix(m.ctypes.data, m.shape[1], int(np.log2(m.shape[1])), m.shape[0])
        self.root = np.zeros(8, dtype=np.uint32)
        self.h = lib().ora_mmcs_commit(arr, len(mats), p32(self.root))
        self.log_height = lib().ora_tree_log_height(self.h)
        self.total_width = sum(m.shape[0] for m in self.mats)

    def layer(self, l):
        n = 8 << (self.log_height - l)
        ptr = lib().ora_tree_layer(self.h, l)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy().reshape(-1, 8)

    def open(self, index):
        out = np.zeros(self.total_width + 8 * self.log_height, dtype=np.uint32)
        n = lib().ora_mmcs_open(self.h, index, p32(out))
        assert n == out.size
        return out

    def verify(self, index, opening):
        lhs = (C.c_uint * len(self.mats))(*[int(np.log2(m.shape[1])) for m in self.mats])
        ws = (C.c_size_t * len(self.mats))(*[m.shape[0] for m in self.mats])
        op = np.ascontiguousarray(opening, dtype=np.uint32)
        return bool(lib().ora_mmcs_verify(p32(self.root), lhs, ws, len(self.mats), index, p32(op)))

    def __del__(self):
        try:
            lib().ora_tree_free(self.h)
        except Exception:
            pass


class Challenger:
    def __init__(self):
        self.c = OraChallenger()
        lib().ora_ch_init(C.byref(self.c))

    def observe(self, vals):
        a = np.ascontiguousarray(vals, dtype=np.uint32)
        lib().ora_ch_observe(C.byref(self.c), p32(a), a.size)

    def sample(self, n=1):
        return np.array([lib().ora_ch_sample(C.byref(self.c)) for _ in range(n)], dtype=np.uint32)

    def sample_bits(self, bits):
        return lib().ora_ch_sample_bits(C.byref(self.c), bits)

    def grind(self, bits):
        return lib().ora_ch_grind(C.byref(self.c), bits)


def fri_fold(vals, log_n_out, beta):
    a = np.ascontiguousarray(vals, dtype=np.uint32)
    out = np.zeros(4 << log_n_out, dtype=np.uint32)
    b = np.asarray(beta, dtype=np.uint32)
    lib().ora_fri_fold(p32(a), p32(out), log_n_out, p32(b))
    return out


def _air_array(airs):
    """airs: list of dicts {program, log_height, width, trace ([width,n] or None), pvs}."""
    arr = (OraAir * len(airs))()
    keep = []
    for i, a in enumerate(airs):
        prog = np.ascontiguousarray(a["program"], dtype=np.uint32)
        pvs = np.ascontiguousarray(a["pvs"], dtype=np.uint32)
        tr = None if a.get("trace") is None else np.ascontiguousarray(a["trace"], dtype=np.uint32)
        prep = None if a.get("prep") is None else np.ascontiguousarray(a["prep"], dtype=np.uint32)
        pc = None if a.get("prep_commit") is None else np.ascontiguousarray(a["prep_commit"], dtype=np.uint32)
        keep += [prog, pvs, tr, prep, pc]
        arr[i] = OraAir(p32(prog), prog.size, a["log_height"], a["width"],
                        p32(tr) if tr is not None else None, p32(pvs) if pvs.size else None, pvs.size,
                        p32(prep) if prep is not None else None, p32(pc) if pc is not None else None)
    return arr, keep


def stark_prove(params, airs, cap_words=1 << 24):
    prm = OraParams(*params)
    arr, keep = _air_array(airs)
    out = np.zeros(cap_words, dtype=np.uint32)
    n = lib().ora_stark_prove(C.byref(prm), arr, len(airs), p32(out), out.size)
    if n == 0:
        raise RuntimeError("oracle prover failed")
    return out[:n].copy()


def constraint_eval(program, log_height, log_blowup, width, lde, pvs, alpha):
    """Quotient values [4, 2^(lh+b)] (canonical) of one AIR on its committed LDE ([width, 2^(lh+b)] canonical)."""
    prog = np.ascontiguousarray(program, dtype=np.uint32)
    l_ = np.ascontiguousarray(lde, dtype=np.uint32)
    pv = np.ascontiguousarray(pvs, dtype=np.uint32)
    al = np.ascontiguousarray(alpha, dtype=np.uint32)
    q = np.zeros((4, 1 << (log_height + log_blowup)), np.uint32)
    l = lib()
    l.ora_constraint_eval.restype = C.c_int
    l.ora_constraint_eval.argtypes = [u32p_t(), C.c_size_t, C.c_uint, C.c_uint, C.c_size_t, u32p_t(), u32p_t(), C.c_size_t, u32p_t(), u32p_t()]
    if l.ora_constraint_eval(p32(prog), prog.size, log_height, log_blowup, width, p32(l_), p32(pv) if pv.size else None, pv.size, p32(al), p32(q)) != 0:
        raise RuntimeError("ora_constraint_eval refused the program")
    return q


def prep_commit(params, a):
    """8-word commitment of one AIR's preprocessed trace (the verifying-key entry)."""
    prm = OraParams(*params)
    arr, keep = _air_array([a])
    out = np.zeros(8, dtype=np.uint32)
    l = lib()
    l.ora_prep_commit.restype = C.c_int
    l.ora_prep_commit.argtypes = [C.POINTER(OraParams), C.POINTER(OraAir), u32p_t()]
    if l.ora_prep_commit(C.byref(prm), arr, p32(out)) != 0:
        raise RuntimeError("ora_prep_commit failed")
    return out


def stark_verify(params, airs, proof):
    prm = OraParams(*params)
    arr, keep = _air_array(airs)
    pr = np.ascontiguousarray(proof, dtype=np.uint32)
    return lib().ora_stark_verify(C.byref(prm), arr, len(airs), p32(pr), pr.size)



def ext_batch_inverse(vals):
    a = np.ascontiguousarray(vals, dtype=np.uint32).reshape(-1)
    out = np.zeros_like(a)
    lib().ora_ext_batch_inverse(p32(a), p32(out), a.size // 4)
    return out


def logup_running_sum(den, num):
    d = np.ascontiguousarray(den, dtype=np.uint32).reshape(-1)
    m = np.ascontiguousarray(num, dtype=np.uint32)
    out = np.zeros_like(d)
    lib().ora_logup_running_sum(p32(d), p32(m), m.size, p32(out))
    return out


def mle_fold(vals, r):
    a = np.ascontiguousarray(vals, dtype=np.uint32).reshape(-1)
    out = np.zeros(a.size // 2, dtype=np.uint32)
    lib().ora_mle_fold(p32(a), p32(out), a.size // 8, p32(np.asarray(r, dtype=np.uint32)))
    return out


def sumcheck_round(tables):
    tabs = [np.ascontiguousarray(t, dtype=np.uint32).reshape(-1) for t in tables]
    arr = (C.POINTER(C.c_uint32) * len(tabs))(*[p32(t) for t in tabs])
    out = np.zeros(4 * (len(tabs) + 1), dtype=np.uint32)
    lib().ora_sumcheck_round(arr, len(tabs), tabs[0].size // 8, p32(out))
    return out


# ---- oracle/fast: the optimised CPU prover (same proofs; bench baseline + full-size checks) ----
_FAST = None


def fast_lib():
    """Builds (with -march=native, on this machine) and loads oracle/fast/libfastoracle.so."""
    global _FAST
    if _FAST is None:
        lib()  # builds liboracle.so, which the fast library links
        d = os.path.join(_ORA_DIR, "fast")
        so = os.path.join(d, "libfastoracle.so")
        srcs = [os.path.join(d, "fast_stark.c"), os.path.join(d, "Makefile"), os.path.join(_ORA_DIR, "liboracle.so")]
        # -march=native: a library built on another machine (it travels with the repository snapshot) may use
        # instructions this host lacks, so the build is stamped with the host's CPU flags and redone when they differ
        import hashlib

        try:
            with open("/proc/cpuinfo") as f:
                flags = next((ln for ln in f if ln.startswith("flags")), "")
        except OSError:
            flags = ""
        stamp, stamp_path = hashlib.sha256(flags.encode()).hexdigest(), so + ".host"
        have = open(stamp_path).read().strip() if os.path.exists(stamp_path) else ""
        if not os.path.exists(so) or have != stamp or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-C", d, "clean"], stdout=subprocess.DEVNULL)
            subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
            with open(stamp_path, "w") as f:
                f.write(stamp)
        l = C.CDLL(so)
        u32p, sz = C.POINTER(C.c_uint32), C.c_size_t
        l.fast_stark_prove.restype = sz
        l.fast_stark_prove.argtypes = [C.POINTER(OraParams), C.POINTER(OraAir), sz, u32p, sz]
        l.fast_vector_lanes.restype = C.c_int
        l.fast_warmup.argtypes = [C.c_uint]
        l.fast_poseidon2_permute_many.argtypes = [u32p, sz]
        l.fast_coset_lde_batch.argtypes = [u32p, sz, u32p, sz, C.c_uint, C.c_uint, sz, C.c_uint32]
        l.fast_mmcs_root.argtypes = [C.POINTER(OraMatrix), sz, u32p]
        _FAST = l
    return _FAST


def fast_stark_prove(params, airs, cap_words=1 << 24):
    prm = OraParams(*params)
    arr, keep = _air_array(airs)
    out = np.zeros(cap_words, dtype=np.uint32)
    n = fast_lib().fast_stark_prove(C.byref(prm), arr, len(airs), p32(out), out.size)
    if n == 0:
        raise RuntimeError("fast oracle prover failed (unsupported AIR set or unsatisfied constraints)")
    return out[:n].copy()
