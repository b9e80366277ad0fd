"""Randomised parity campaign for the instruction chips' device trace generators: random record sets (operand distributions
that exercise equal / near-equal / extreme operands, small divisors, every opcode and case) -> device trace and lookup-table
counts == oracle/tracegen.c, cell for cell.  Test infrastructure (uses oracle/ as the checker).
Usage: python tests/chip_fuzz.py [n_rounds] [first_seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle_lib as ora
import zkvm_prover_amd as z

P = 2013265921
n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
zk = z.Context(0)
dev = zk.device
as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint32).view(np.int32)).to(dev)  # noqa: E731
SX, SY = 256, 2048


def words(rng, n):
    """32-bit operands: uniform, small, near the sign boundary, copies of each other"""
    w = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    kind = rng.integers(0, 6, n)
    w[kind == 1] = rng.integers(0, 1 << 8, int((kind == 1).sum())).astype(np.uint32)
    w[kind == 2] = (0 - rng.integers(0, 1 << 8, int((kind == 2).sum()))).astype(np.uint32)
    w[kind == 3] = rng.choice(np.array([0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0x80000001, 0xFFFF, 0x10000], np.uint32), int((kind == 3).sum()))
    return w


def offsets(rng, n, bits):
    off = rng.integers(-(1 << bits), 1 << bits, n) * 2
    return np.where(off < 0, P + off, off).astype(np.uint32)


def check(name, got, exp):
    if not (got == exp).all():
        bad = np.argwhere(got != exp)[:3]
        raise SystemExit(f"MISMATCH {name}: {bad.tolist()}")


t0, rows = time.time(), 0
for r in range(n_rounds):
    seed = seed0 + r
    rng = np.random.default_rng(seed)
    lh = int(rng.integers(0, 13))
    n = int(rng.integers(0, (1 << lh) + 1))
    a, b = words(rng, n), words(rng, n)
    same = rng.random(n) < 0.2
    b[same] = a[same]
    bw = lambda: torch.zeros(2 << 16, dtype=torch.int32, device=dev)      # noqa: E731
    tab = lambda: torch.zeros(SX * SY, dtype=torch.int32, device=dev)     # noqa: E731
    dl = lambda t, w: zk.download(t).reshape(w, -1)                       # noqa: E731
    # ALU, less-than, shift
    op = rng.integers(0, 5, n).astype(np.uint32)
    d = bw()
    exp, xc, _ = ora.rv32_alu_trace(op, a, b, lh)
    check("alu", dl(zk.rv32_alu_tracegen(as_dev(op), as_dev(a), as_dev(b), lh, d), 18), exp)
    check("alu xor", dl(d, 2)[1], xc)
    op = rng.integers(0, 2, n).astype(np.uint32)
    d = bw()
    exp, rc, _ = ora.rv32_lt_trace(op, a, b, lh)
    check("lt", dl(zk.rv32_lt_tracegen(as_dev(op), as_dev(a), as_dev(b), lh, d), 18), exp)
    check("lt range", dl(d, 2)[0], rc)
    op = rng.integers(0, 3, n).astype(np.uint32)
    d = bw()
    exp, rc, xc, _ = ora.rv32_shift_trace(op, a, b, lh)
    check("shift", dl(zk.rv32_shift_tracegen(as_dev(op), as_dev(a), as_dev(b), lh, d), 32), exp)
    check("shift tables", dl(d, 2), np.stack([rc, xc]))
    # multiplications and division
    t = tab()
    exp, tc = ora.rv32_mul_trace(a, b, lh, SX, SY)
    check("mul", dl(zk.rv32_mul_tracegen(as_dev(a), as_dev(b), lh, t, SX, SY), 13), exp)
    check("mul tuple", zk.download(t), tc)
    op = rng.integers(0, 3, n).astype(np.uint32)
    t, d = tab(), bw()
    exp, tc, rc, bad = ora.rv32_mulh_trace(op, a, b, lh, SX, SY)
    assert bad == 0
    check("mulh", dl(zk.rv32_mulh_tracegen(as_dev(op), as_dev(a), as_dev(b), lh, t, d, SX, SY), 21), exp)
    check("mulh tuple", zk.download(t), tc)
    check("mulh range", dl(d, 2)[0], rc)
    op = rng.integers(0, 4, n).astype(np.uint32)
    t, d = tab(), bw()
    exp, tc, rc, bad = ora.rv32_divrem_trace(op, a, b, lh, SX, SY)
    assert bad == 0
    check("divrem", dl(zk.rv32_divrem_tracegen(as_dev(op), as_dev(a), as_dev(b), lh, t, d, SX, SY), 41), exp)
    check("divrem tuple", zk.download(t), tc)
    check("divrem range", dl(d, 2)[0], rc)
    # branches
    op = rng.integers(0, 2, n).astype(np.uint32)
    imm = offsets(rng, n, 11)
    exp, _ = ora.rv32_branch_eq_trace(op, a, b, imm, lh)
    check("beq", dl(zk.rv32_branch_eq_tracegen(as_dev(op), as_dev(a), as_dev(b), as_dev(imm), lh), 17), exp)
    op = rng.integers(0, 4, n).astype(np.uint32)
    d = bw()
    exp, rc, _ = ora.rv32_branch_lt_trace(op, a, b, imm, lh)
    check("blt", dl(zk.rv32_branch_lt_tracegen(as_dev(op), as_dev(a), as_dev(b), as_dev(imm), lh, d), 23), exp)
    check("blt range", dl(d, 2)[0], rc)
    # jumps and upper immediates
    pc = (rng.integers(0, (1 << 28) - 1, n) * 4).astype(np.uint32)
    op = rng.integers(0, 2, n).astype(np.uint32)
    imm = np.where(op == 0, offsets(rng, n, 19), rng.integers(0, 1 << 20, n)).astype(np.uint32)
    d = bw()
    exp, rc, bad = ora.rv32_jal_lui_trace(op, pc, imm, lh)
    assert bad == 0
    check("jal_lui", dl(zk.rv32_jal_lui_tracegen(as_dev(op), as_dev(pc), as_dev(imm), lh, d), 9), exp)
    check("jal_lui range", dl(d, 2)[0], rc)
    imm = rng.integers(0, 1 << 20, n).astype(np.uint32)
    d = bw()
    exp, rc, bad = ora.rv32_auipc_trace(pc, imm, lh)
    assert bad == 0
    check("auipc", dl(zk.rv32_auipc_tracegen(as_dev(pc), as_dev(imm), lh, d), 14), exp)
    check("auipc range", dl(d, 2)[0], rc)
    rs1 = (a >> 3).astype(np.uint32) + 4096   # targets stay below 2^30 (the JALR chip bounds them: commit a94e43c; a >> 2 reached 2^30 for a = 0xFFFFFFFF)
    imm = rng.integers(0, 1 << 12, n).astype(np.uint32)
    d = bw()
    exp, rc, bad = ora.rv32_jalr_trace(pc, rs1, imm, lh)
    assert bad == 0
    check("jalr", dl(zk.rv32_jalr_tracegen(as_dev(pc), as_dev(rs1), as_dev(imm), lh, d), 20), exp)
    check("jalr range", dl(d, 2)[0], rc)
    # loads and stores
    cs = rng.integers(0, 20, n).astype(np.uint32)
    d = bw()
    exp, rc, _ = ora.rv32_loadstore_trace(cs, a, b, lh)
    check("loadstore", dl(zk.rv32_loadstore_tracegen(as_dev(cs), as_dev(a), as_dev(b), lh, d), 33), exp)
    check("loadstore range", dl(d, 2)[0], rc)
    # native field chips (operands are field elements; divisors non-zero)
    op = rng.integers(0, 4, n).astype(np.uint32)
    fb, fc = (a % P).astype(np.uint32), (b % (P - 1) + 1).astype(np.uint32)
    exp, bad = ora.field_arith_trace(op, fb, fc, lh)
    assert bad == 0
    check("field_arith", dl(zk.field_arith_tracegen(as_dev(op), as_dev(fb), as_dev(fc), lh), 8), exp)
    if n:
        fx = rng.integers(0, P, (n, 4)).astype(np.uint32)
        fy = rng.integers(0, P, (n, 4)).astype(np.uint32)
        fy[:, 0] |= 1
        exp, bad = ora.field_ext_trace(op, fx, fy, lh)
        assert bad == 0
        check("field_ext", dl(zk.field_ext_tracegen(as_dev(op), as_dev(fx), as_dev(fy), lh), 20), exp)
    rows += 15 * n
print(f"{n_rounds} rounds, {rows} records, 0 mismatches, {time.time() - t0:.1f} s")
