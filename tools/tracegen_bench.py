"""Throughput of the device trace generators at 2^22 records (one MI355X): rows/s and the HBM bytes they move (records read +
trace written; the table generators also do one atomic per request).  Usage: python tools/tracegen_bench.py [log_rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import zkvm_prover_amd as z

lh = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lh
zk = z.Context(0)
dev = zk.device
g = torch.Generator(device=dev)
g.manual_seed(1)
ri = lambda hi, k=n: torch.randint(0, hi, (k,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)  # noqa: E731
b, c = ri(1 << 31), ri(1 << 31)
d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=dev)
d_tab = torch.zeros(256 * 8192, dtype=torch.int32, device=dev)
n_prog = 1 << 16
d_prog = ri(2013265921, 9 * n_prog)
idx = ri(n_prog)
ts = torch.arange(1, n + 1, device=dev, dtype=torch.int32)
pts = torch.clamp(ts - ri(1000) - 1, min=0)
data = ri(65536)
op5, op2, as3, ptr, ones = b % 5, b % 2, b % 3, c >> 2, torch.ones_like(b)


def timed(name, fn, bytes_moved, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-28s %7.3f ms  %6.2f G rows/s  %6.2f TB/s (%d B per row)" % (name, dt * 1e3, n / dt / 1e9, bytes_moved / dt / 1e12, bytes_moved // n))


timed("rv32_alu (18 cols)", lambda: zk.rv32_alu_tracegen(op5, b, c, lh, d_bw), n * (12 + 72))
timed("rv32_lt (18 cols)", lambda: zk.rv32_lt_tracegen(op2, b, c, lh, d_bw), n * (12 + 72))
timed("rv32_mul (13 cols)", lambda: zk.rv32_mul_tracegen(b, c, lh, d_tab, 256, 8192), n * (8 + 52))
timed("exec_frame (10 cols)", lambda: zk.exec_frame_tracegen(idx, d_prog, n_prog, lh), n * (4 + 40))
timed("program_freq", lambda: zk.program_freq_tracegen(idx, 16), n * 4)
timed("memory_access (10 cols)", lambda: zk.memory_access_tracegen(as3, ptr, data, pts, data, ts, ones, lh), n * (28 + 40))
P = 2013265921
d_tab2 = torch.zeros(256 * 2048, dtype=torch.int32, device=dev)
op3, op4, imm, pc = b % 3, b % 4, ri(P), (ri(1 << 27) << 2)
timed("rv32_shift (32 cols)", lambda: zk.rv32_shift_tracegen(op3, b, c, lh, d_bw), n * (12 + 128))
timed("rv32_mulh (21 cols)", lambda: zk.rv32_mulh_tracegen(op3, b, c, lh, d_tab2, d_bw, 256, 2048), n * (12 + 84))
timed("rv32_divrem (41 cols)", lambda: zk.rv32_divrem_tracegen(op4, b, c, lh, d_tab2, d_bw, 256, 2048), n * (12 + 164))
timed("rv32_branch_eq (17 cols)", lambda: zk.rv32_branch_eq_tracegen(op2, b, c, imm, lh), n * (16 + 68))
timed("rv32_branch_lt (23 cols)", lambda: zk.rv32_branch_lt_tracegen(op4, b, c, imm, lh, d_bw), n * (16 + 92))
timed("rv32_jal_lui (9 cols)", lambda: zk.rv32_jal_lui_tracegen(op2, pc, torch.where(op2 == 0, imm, imm & 0xFFFFF), lh, d_bw), n * (12 + 36))
timed("rv32_auipc (14 cols)", lambda: zk.rv32_auipc_tracegen(pc, imm & 0xFFFFF, lh, d_bw), n * (8 + 56))
timed("rv32_jalr (20 cols)", lambda: zk.rv32_jalr_tracegen(pc, (b >> 2) + 4096, imm & 0xFFF, lh, d_bw), n * (12 + 80))
timed("rv32_loadstore (33 cols)", lambda: zk.rv32_loadstore_tracegen(b % 20, b, c, lh, d_bw), n * (12 + 132))
fb, fc = b % P, (c % (P - 1)) + 1
timed("field_arith (8 cols)", lambda: zk.field_arith_tracegen(op4, fb, fc, lh), n * (12 + 32))
if lh <= 22:
    fx, fy = ri(P, 4 * n), ri(P - 1, 4 * n) + 1
    timed("field_ext (20 cols)", lambda: zk.field_ext_tracegen(op4, fx, fy, lh), n * (36 + 80))
perm_in = ri(2013265921, 16 * n)
out = torch.empty(299 * n, dtype=torch.int32, device=dev) if lh <= 22 else None
if out is not None:
    timed("poseidon2_air (298 cols)", lambda: zk.poseidon2_air_tracegen(perm_in, lh, out), n * (64 + 1192), reps=3)
