"""GPU: three more device-side trace generators (csrc/tracegen_tables.hip) -- range-tuple table, bitwise-operation lookup,
volatile memory boundary -- cell for cell against oracle/tracegen.c, and proven end to end from the device-resident traces
(proof bytes == the oracle's for the same AIR set)."""
import numpy as np
import pytest
import torch

import zkvm_prover_amd as z
from zkvm_prover_amd import air

pytestmark = pytest.mark.gpu
P = 2013265921
NOPV = np.zeros(0, np.uint32)
PARAMS = (1, 0, 12, 4, 4)


def test_range_tuple_counts(zk, ora):
    rng = np.random.default_rng(0)
    for size_x, size_y, n in [(256, 8192, 300000), (4, 8, 1000), (16, 16, 0)]:
        xs = rng.integers(0, size_x, n).astype(np.uint32)
        ys = rng.integers(0, size_y, n).astype(np.uint32)
        exp, bad = ora.range_tuple_counts(xs, ys, size_x, size_y)
        assert bad == 0
        dx, dy = zk.upload(xs), zk.upload(ys)
        got = zk.download(zk.range_tuple_counts_tracegen(dx, dy, size_x, size_y))
        assert (got == exp).all() and int(exp.astype(np.int64).sum()) == n
        # accumulate a second batch of requests on top
        got2 = zk.download(zk.range_tuple_counts_tracegen(dy % size_x if False else dx, dy, size_x, size_y,
                                                          t_counts=zk.upload(exp), accumulate=True))
        assert (got2 == ora.range_tuple_counts(xs, ys, size_x, size_y, counts=exp)[0]).all()
    with pytest.raises(z.ZkhipError):
        zk.range_tuple_counts_tracegen(zk.upload(np.array([3, 256], np.uint32)), zk.upload(np.array([1, 1], np.uint32)), 256, 8192)


def test_bitwise_lookup_counts(zk, ora):
    rng = np.random.default_rng(1)
    n = 200000
    xs, ys = rng.integers(0, 256, n).astype(np.uint32), rng.integers(0, 256, n).astype(np.uint32)
    ops = rng.integers(0, 2, n).astype(np.uint32)
    exp, bad = ora.bitwise_lookup_counts(xs, ys, ops, 8)
    assert bad == 0
    got = zk.download(zk.bitwise_lookup_tracegen(zk.upload(xs), zk.upload(ys), zk.upload(ops), 8)).reshape(2, -1)
    assert (got == exp).all()
    assert int(got[1].astype(np.int64).sum()) == int(ops.sum())
    with pytest.raises(z.ZkhipError):
        zk.bitwise_lookup_tracegen(zk.upload(np.array([256], np.uint32)), zk.upload(np.array([0], np.uint32)),
                                   zk.upload(np.array([0], np.uint32)), 8)


def test_count_tables_kept_canonical_across_generators(zk, ora):
    """zkhip_tables_canonical: while on, the generators take the shared count tables as canonical counts and leave them so (a segment's
    ~20 generators otherwise convert each table from Montgomery form and back around their increments); one conversion at the end gives
    the table the default mode builds."""
    import ctypes as C
    import torch

    rng = np.random.default_rng(9)
    n = 50000
    xs, ys = rng.integers(0, 256, n).astype(np.uint32), rng.integers(0, 256, n).astype(np.uint32)
    ops = rng.integers(0, 2, n).astype(np.uint32)
    tx, ty = rng.integers(0, 256, n).astype(np.uint32), rng.integers(0, 2048, n).astype(np.uint32)
    dx, dy, dop, dtx, dty = (zk.upload(v) for v in (xs, ys, ops, tx, ty))
    # default mode: two batches accumulated
    bw = zk.bitwise_lookup_tracegen(dx, dy, dop, 8)
    bw = zk.bitwise_lookup_tracegen(dy, dx, dop, 8, t_trace=bw, accumulate=True)
    tup = zk.range_tuple_counts_tracegen(dtx, dty, 256, 2048)
    tup = zk.range_tuple_counts_tracegen(dtx, dty, 256, 2048, t_counts=tup, accumulate=True)
    # canonical mode: tables zeroed, the same calls, one conversion
    bw2 = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    tup2 = torch.zeros(256 * 2048, dtype=torch.int32, device=zk.device)
    assert zk.lib.zkhip_tables_canonical(zk.h, 1) == 0
    try:
        zk.bitwise_lookup_tracegen(dx, dy, dop, 8, t_trace=bw2, accumulate=True)
        zk.bitwise_lookup_tracegen(dy, dx, dop, 8, t_trace=bw2, accumulate=True)
        zk.range_tuple_counts_tracegen(dtx, dty, 256, 2048, t_counts=tup2, accumulate=True)
        zk.range_tuple_counts_tracegen(dtx, dty, 256, 2048, t_counts=tup2, accumulate=True)
        zk.sync()
        raw = bw2.cpu().numpy().view(np.uint32)                      # canonical counts as they are
        exp = ora.bitwise_lookup_counts(xs, ys, ops, 8)[0] + ora.bitwise_lookup_counts(ys, xs, ops, 8)[0]
        assert (raw.reshape(2, -1) == exp).all()
    finally:
        assert zk.lib.zkhip_tables_canonical(zk.h, 0) == 0
    for t in (bw2, tup2):
        zk._check(zk.lib.zkhip_to_monty(zk.h, C.c_void_p(t.data_ptr()), t.numel()))
    assert torch.equal(bw, bw2) and torch.equal(tup, tup2)


@pytest.mark.parametrize("n,log_height", [(0, 3), (1, 0), (5, 3), (1000, 10), (70000, 17)])
def test_memory_boundary_trace_sorted_on_device(zk, ora, n, log_height):
    rng = np.random.default_rng(n)
    keys = rng.choice(1 << 22, size=n, replace=False) if n else np.zeros(0, np.int64)
    as_ = (keys >> 20).astype(np.uint32) + 1           # address spaces 1..4
    ptr = ((keys & ((1 << 20) - 1)) * 37 % (1 << 29)).astype(np.uint32)
    # distinct (as, ptr) pairs are guaranteed by distinct keys only if the map is injective per address space
    _, first = np.unique((as_.astype(np.int64) << 32) | ptr, return_index=True)
    keep = np.sort(first)
    as_, ptr = as_[keep], ptr[keep]
    n = len(as_)
    init, fin = ora.rand_field(rng, n), ora.rand_field(rng, n)
    ts = rng.integers(0, 1 << 29, n).astype(np.uint32)
    exp, bad = ora.memory_boundary_trace(as_, ptr, init, fin, ts, 3, 29, log_height)
    assert bad == 0
    dev = zk.device
    t_as = torch.from_numpy(as_.view(np.int32)).to(dev)
    t_ptr = torch.from_numpy(ptr.view(np.int32)).to(dev)
    t_ts = torch.from_numpy(ts.view(np.int32)).to(dev)
    got = zk.download(zk.memory_boundary_tracegen(t_as, t_ptr, zk.upload(init), zk.upload(fin), t_ts, 3, 29, log_height))
    got = got.reshape(8, -1)
    assert (got == exp).all()
    # rows are sorted and the AIR's own constraints hold on the generated trace
    k = got[0, :n].astype(np.int64) * (1 << 29) + got[1, :n]
    assert (np.diff(k) > 0).all() and (got[5, :n] == 1).all() and (got[:, n:] == 0).all()
    if log_height <= 10:
        assert air.check_trace(air.memory_boundary_air(29).program(), got, NOPV) == []


def test_memory_boundary_rejects_duplicates_and_overflow(zk):
    dev = zk.device
    as_ = torch.tensor([1, 1], dtype=torch.int32, device=dev)
    ptr = torch.tensor([8, 8], dtype=torch.int32, device=dev)
    v = zk.upload(np.array([1, 2], np.uint32))
    with pytest.raises(z.ZkhipError):
        zk.memory_boundary_tracegen(as_, ptr, v, v, ptr, 3, 29, 2)
    with pytest.raises(z.ZkhipError):   # more records than rows
        zk.memory_boundary_tracegen(as_, torch.tensor([8, 9], dtype=torch.int32, device=dev), v, v, ptr, 3, 29, 0)


def test_tables_proven_from_device_generated_traces(zk, ora):
    """Requester chips + the two lookup tables, the tables' traces generated on the device from the requesters' columns where
    they lie: the proof equals the oracle's proof of the same AIR set with host-generated tables."""
    rng = np.random.default_rng(5)
    sx, sy, bits = 16, 64, 4
    nu = 1 << 9
    u = np.zeros((3, nu), np.uint32)
    u[0], u[1] = rng.integers(0, sx, nu), rng.integers(0, sy, nu)
    u[2] = (u[0].astype(np.int64) * u[1] % P).astype(np.uint32)
    w = np.zeros((4, nu), np.uint32)
    w[0], w[1], w[3] = rng.integers(0, 1 << bits, nu), rng.integers(0, 1 << bits, nu), rng.integers(0, 2, nu)
    w[2] = (w[0] ^ w[1]) * w[3]
    tuple_prep, bw_prep = air.range_tuple_prep(sx, sy), air.bitwise_lookup_prep(bits)
    d_u, d_w = zk.upload(u.reshape(-1)), zk.upload(w.reshape(-1))
    d_tuple = zk.range_tuple_counts_tracegen(d_u[0:nu], d_u[nu:2 * nu], sx, sy)
    d_bw = zk.bitwise_lookup_tracegen(d_w[0:nu], d_w[nu:2 * nu], d_w[3 * nu:4 * nu], bits)
    t_tuple = zk.download(d_tuple).reshape(1, -1)
    t_bw = zk.download(d_bw).reshape(2, -1)
    assert (t_tuple == ora.range_tuple_counts(u[0], u[1], sx, sy)[0]).all()
    assert (t_bw == ora.bitwise_lookup_counts(w[0], w[1], w[3], bits)[0]).all()
    airs = [dict(program=air.range_tuple_user_air().program(), log_height=9, width=3, n_pvs=0, trace=u, pvs=NOPV),
            dict(program=air.range_tuple_table_air(sx, sy).program(), log_height=10, width=1, n_pvs=0, trace=t_tuple, pvs=NOPV, prep=tuple_prep),
            dict(program=air.bitwise_user_air().program(), log_height=9, width=4, n_pvs=0, trace=w, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(bits).program(), log_height=2 * bits, width=2, n_pvs=0, trace=t_bw, pvs=NOPV, prep=bw_prep)]
    for a in airs:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a.get("prep")) == []
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_u, d_tuple, d_w, d_bw], [NOPV] * 4)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 4, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def test_rv32_alu_chip_from_execution_records(zk, ora):
    """An instruction chip: the RV32 base ALU core (ADD / SUB / XOR / OR / AND on 8-bit limbs, OpenVM BaseAluCoreAir) filled on
    the device from execution records, its bitwise-lookup requests counted into the lookup table's trace in the same pass;
    cell for cell == oracle, and ALU chip + table prove end to end (proof bytes == oracle) with both traces device-resident."""
    rng = np.random.default_rng(7)
    n, lh = 3000, 12
    opc = rng.integers(0, 5, n).astype(np.uint32)
    b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    c = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    b[:5] = [0xFFFFFFFF, 0, 0x80000000, 1, 0xFF]
    c[:5] = [1, 1, 0x80000000, 0xFFFFFFFF, 0xFF01]
    opc[:5] = [0, 1, 0, 1, 0]                                    # carries through every limb, borrows, wrap-around
    exp, xc, bad = ora.rv32_alu_trace(opc, b, c, lh)
    assert bad == 0
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(v.view(np.int32)).to(dev)  # noqa: E731
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=dev)
    d_alu = zk.rv32_alu_tracegen(as_dev(opc), as_dev(b), as_dev(c), lh, d_bw)
    got = zk.download(d_alu).reshape(18, -1)
    bw = zk.download(d_bw).reshape(2, -1)
    assert (got == exp).all()
    assert (bw[1] == xc).all() and (bw[0] == 0).all() and int(bw[1].astype(np.int64).sum()) == 4 * n
    prog = air.rv32_alu_core_air().program()
    assert air.check_trace(prog, got, NOPV) == []
    # a second batch of records accumulates into the same table
    d_alu2 = zk.rv32_alu_tracegen(as_dev(opc[:100]), as_dev(c[:100]), as_dev(b[:100]), 7, d_bw)
    exp2, xc2, _ = ora.rv32_alu_trace(opc[:100], c[:100], b[:100], 7, xor_counts=xc)
    assert (zk.download(d_alu2).reshape(18, -1) == exp2).all() and (zk.download(d_bw).reshape(2, -1)[1] == xc2).all()
    with pytest.raises(z.ZkhipError):
        zk.rv32_alu_tracegen(as_dev(np.array([5], np.uint32)), as_dev(b[:1]), as_dev(c[:1]), 0, d_bw)
    # end to end: both ALU batches + the table they counted into
    bw2 = zk.download(d_bw).reshape(2, -1)
    airs = [dict(program=prog, log_height=lh, width=18, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=prog, log_height=7, width=18, n_pvs=0, trace=exp2, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8).program(), log_height=16, width=2, n_pvs=0, trace=bw2, pvs=NOPV,
                 prep=air.bitwise_lookup_prep(8))]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_alu, d_alu2, d_bw], [NOPV] * 3)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 3, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def test_rv32_mul_chip_with_the_range_tuple_table(zk, ora):
    """RV32 multiplication core (OpenVM MultiplicationCoreAir) from records on the device, its (limb, carry) requests counted into
    the range-tuple table in the same pass: cells == oracle with the reference's table sizes [256, 8192]; MUL chip + table proven end
    to end from the device-resident traces (proof bytes == oracle) with a smaller table."""
    rng = np.random.default_rng(11)
    n, lh = 5000, 13
    b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    c = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    b[:3], c[:3] = [0xFFFFFFFF, 0, 0x10001], [0xFFFFFFFF, 5, 0xFFFF]
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(v.view(np.int32)).to(dev)  # noqa: E731
    for sx, sy in ((256, 8192), (256, 2048)):
        exp, tc = ora.rv32_mul_trace(b, c, lh, sx, sy)
        d_tab = torch.zeros(sx * sy, dtype=torch.int32, device=dev)
        d_mul = zk.rv32_mul_tracegen(as_dev(b), as_dev(c), lh, d_tab, sx, sy)
        assert (zk.download(d_mul).reshape(13, -1) == exp).all()
        assert (zk.download(d_tab) == tc).all() and int(tc.astype(np.int64).sum()) == 4 * n
    prog = air.rv32_mul_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    airs = [dict(program=prog, log_height=lh, width=13, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=air.range_tuple_table_air(256, 2048).program(), log_height=19, width=1, n_pvs=0, trace=tc.reshape(1, -1), pvs=NOPV,
                 prep=air.range_tuple_prep(256, 2048))]
    params = (1, 0, 6, 2, 2)
    pk = z.ProvingKey(zk, params, airs)
    proof = pk.prove([d_mul, d_tab], [NOPV] * 2)
    assert z.verify(params, pk.verifying_airs(), [NOPV] * 2, proof) == 0
    assert proof == ora.stark_prove(params, airs).tobytes()
    pk.close()
    with pytest.raises(z.ZkhipError):   # a table that cannot hold the carries
        zk.rv32_mul_tracegen(as_dev(b), as_dev(c), lh, d_tab, 256, 512)


def _program(n_program, seed=0):
    """[9, n_program] canonical: pc = 4 * row, the other fields random."""
    rng = np.random.default_rng(seed)
    prog = rng.integers(0, P, size=(air.PROGRAM_FIELDS, n_program)).astype(np.uint32)
    prog[0] = 4 * np.arange(n_program, dtype=np.uint32)
    return prog


@pytest.mark.parametrize("log_program,log_frames,n", [(3, 4, 11), (10, 12, 4096), (12, 16, 60001), (5, 5, 0)])
def test_program_chip_and_execution_frames(zk, ora, log_program, log_frames, n):
    """Program chip (cached program of width 9 + frequency column) and the execution frames that send each executed instruction:
    both traces generated on the device from the list of executed instruction indices == oracle/tracegen.c; the pair proves from
    the device-resident traces with the program as a cached main partition, bytes == oracle."""
    rng = np.random.default_rng(log_frames)
    n_program = 1 << log_program
    prog = _program(n_program, seed=log_program)
    # a skewed execution profile: loops revisit a few instructions
    idx = (rng.integers(0, n_program, n) * rng.integers(0, 2, n) + rng.integers(0, min(8, n_program), n)) % n_program
    idx = idx.astype(np.uint32)
    exp_freq, bad = ora.program_freq_trace(idx, log_program)
    exp_frames, bad2 = ora.exec_frame_trace(idx, prog, log_frames)
    assert bad == 0 and bad2 == 0 and int(exp_freq.astype(np.int64).sum()) == n
    d_idx = torch.from_numpy(idx.astype(np.int32)).cuda()   # plain integers (zk.upload would convert to Montgomery form)
    d_prog = zk.upload(prog.reshape(-1))            # canonical -> Montgomery on upload: the cached partition as the prover reads it
    d_freq = zk.program_freq_tracegen(d_idx, log_program)
    d_frames = zk.exec_frame_tracegen(d_idx, d_prog, n_program, log_frames)
    assert (zk.download(d_freq) == exp_freq).all()
    assert (zk.download(d_frames).reshape(10, -1) == exp_frames).all()
    airs = [dict(program=air.program_air().program(), log_height=log_program, width=10, n_pvs=0,
                 trace=np.concatenate([prog, exp_freq.reshape(1, -1)]), pvs=NOPV),
            dict(program=air.exec_frame_air().program(), log_height=log_frames, width=10, n_pvs=0, trace=exp_frames, pvs=NOPV)]
    assert air.quotient_chunks(airs[0]["program"]) == 1   # degree 2, like the stored proofs' program chip
    exp = ora.stark_prove(PARAMS, airs).tobytes()
    pk = z.ProvingKey(zk, PARAMS, airs)
    d_program_trace = torch.cat([d_prog, d_freq])   # [cached 9 columns | frequency]
    got = pk.prove([d_program_trace, d_frames], [NOPV, NOPV])
    assert got == exp and z.verify(PARAMS, pk.verifying_airs(), [NOPV, NOPV], got) == 0
    lay = z.proof_layout(PARAMS, pk.verifying_airs())
    assert lay["n_cached"] == 1
    pk.close()
    if n:
        with pytest.raises(z.ZkhipError):
            zk.program_freq_tracegen(torch.tensor([n_program], dtype=torch.int32, device="cuda"), log_program)
        with pytest.raises(z.ZkhipError):
            zk.exec_frame_tracegen(torch.tensor([0, n_program], dtype=torch.int32, device="cuda"), d_prog, n_program, log_frames)


def test_rv32_less_than_chip(zk, ora):
    """RV32 less-than core (SLT / SLTU, OpenVM LessThanCoreAir<4, 8>) filled on the device from execution records, its range
    requests counted into the bitwise lookup table's range column in the same pass: cells == oracle, cmp == the integers'
    comparison, the oracle's trace satisfies the AIR, and chip + table prove from the device-resident traces (bytes == oracle)."""
    rng = np.random.default_rng(13)
    n, lh = 6000, 13
    opc = rng.integers(0, 2, n).astype(np.uint32)
    b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    c = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    b[:8] = [0, 5, 0x80000000, 0x7FFFFFFF, 0xFFFFFFFF, 0x80000000, 7, 0x12345678]
    c[:8] = [0, 5, 0x7FFFFFFF, 0x80000000, 0, 0x80000001, 7, 0x12345679]
    c[100:200] = b[100:200]                 # equal operands: no marker, cmp = 0
    c[200:300] = b[200:300] ^ 0x100         # first difference in limb 1
    c[300:400] = b[300:400] ^ 0x80000000    # sign flips
    exp, rc, bad = ora.rv32_lt_trace(opc, b, c, lh)
    assert bad == 0
    want = np.where(opc == 0, b.view(np.int32) < c.view(np.int32), b < c).astype(np.uint32)
    assert (exp[8][:n] == want).all()
    prog = air.rv32_lt_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(v.view(np.int32)).to(dev)  # noqa: E731
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=dev)
    d_lt = zk.rv32_lt_tracegen(as_dev(opc), as_dev(b), as_dev(c), lh, d_bw)
    assert (zk.download(d_lt).reshape(18, -1) == exp).all()
    bw = zk.download(d_bw).reshape(2, -1)
    assert (bw[0] == rc).all() and (bw[1] == 0).all() and int(rc.astype(np.int64).sum()) == n + int((b != c).sum())
    with pytest.raises(z.ZkhipError):
        zk.rv32_lt_tracegen(as_dev(np.array([2], np.uint32)), as_dev(b[:1]), as_dev(c[:1]), 0, d_bw)
    # a cheating witness (cmp flipped on one row) violates the AIR
    bad_tr = exp.copy()
    bad_tr[8][3] ^= 1
    assert air.check_trace(prog, bad_tr, NOPV) != []
    # the ALU chip counts into the XOR column of the same table: both instruction chips + the table in one proof
    opa = rng.integers(0, 5, 1000).astype(np.uint32)
    d_alu = zk.rv32_alu_tracegen(as_dev(opa), as_dev(b[:1000]), as_dev(c[:1000]), 10, d_bw)
    exp_alu, xc, _ = ora.rv32_alu_trace(opa, b[:1000], c[:1000], 10)
    bw = zk.download(d_bw).reshape(2, -1)
    assert (bw[0] == rc).all() and (bw[1] == xc).all()
    airs = [dict(program=prog, log_height=lh, width=18, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=air.rv32_alu_core_air().program(), log_height=10, width=18, n_pvs=0, trace=exp_alu, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8).program(), log_height=16, width=2, n_pvs=0, trace=bw, pvs=NOPV,
                 prep=air.bitwise_lookup_prep(8))]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_lt, d_alu, d_bw], [NOPV] * 3)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 3, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def test_memory_access_tracegen_refuses_inconsistent_records(zk, ora):
    """The generator's own checks (a 16-bit cell, time moving forward, a read that leaves its cell alone) and parity with the oracle
    on a hand-made log, including an empty one."""
    dev = zk.device
    t = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)  # noqa: E731
    good = dict(as_=[1, 1, 2], ptr=[4, 4, 9], pd=[0, 7, 65535], pts=[0, 1, 0], d=[7, 7, 1], ts=[1, 2, 3], rd=[0, 1, 0])
    tr = zk.memory_access_tracegen(*[t(good[k]) for k in ("as_", "ptr", "pd", "pts", "d", "ts", "rd")], 2)
    exp, bad = ora.memory_access_trace(*[np.array(good[k], np.uint32) for k in ("as_", "ptr", "pd", "pts", "d", "ts", "rd")], 2)
    assert bad == 0 and (zk.download(tr).reshape(10, -1) == exp).all()
    assert air.check_trace(air.memory_access_air().program(), exp, NOPV) == []
    empty = zk.memory_access_tracegen(*[torch.empty(0, dtype=torch.int32, device=dev)] * 7, 1)
    assert (zk.download(empty) == 0).all()
    for key, val in (("d", [7, 7, 65536]), ("ts", [1, 1, 3]), ("d", [7, 8, 1]), ("rd", [0, 2, 0]), ("pts", [0, 5, 0])):
        rec = dict(good)
        rec[key] = val
        with pytest.raises(z.ZkhipError):
            zk.memory_access_tracegen(*[t(rec[k]) for k in ("as_", "ptr", "pd", "pts", "d", "ts", "rd")], 2)
        assert ora.memory_access_trace(*[np.array(rec[k], np.uint32) for k in ("as_", "ptr", "pd", "pts", "d", "ts", "rd")], 2)[1] >= 1
    with pytest.raises(z.ZkhipError):   # more records than rows
        zk.memory_access_tracegen(*[t(good[k]) for k in ("as_", "ptr", "pd", "pts", "d", "ts", "rd")], 1)


def test_rv32_shift_chip(zk, ora):
    """RV32 shift core (SLL / SRL / SRA, OpenVM ShiftCoreAir<4, 8>) filled on the device from execution records, its eight lookup
    requests per row counted into the bitwise table in the same pass: cells == oracle (whose result limbs come from a limb-wise
    restatement, not from C shifts), results == the integers' shifts for every shift amount, the oracle's trace satisfies the AIR
    and a wrong result does not; chip + table prove from the device-resident traces (bytes == oracle)."""
    rng = np.random.default_rng(17)
    n, lh = 5000, 13
    opc = rng.integers(0, 3, n).astype(np.uint32)
    b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    c = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    c[:144] = np.tile(np.arange(48), 3)
    b[:144] = np.repeat(np.array([0x80000001, 0x7FFFFFFF, 0xFFFFFFFF], np.uint32), 48)
    opc[:144] = np.tile(np.repeat([0, 1, 2], 16), 3)
    exp, rc, xc, bad = ora.rv32_shift_trace(opc, b, c, lh)
    assert bad == 0
    res = (exp[0].astype(np.uint64) | (exp[1].astype(np.uint64) << 8) | (exp[2].astype(np.uint64) << 16) | (exp[3].astype(np.uint64) << 24))[:n]
    s = (c & 31).astype(np.uint64)
    want = np.where(opc == 0, (b.astype(np.uint64) << s) & 0xFFFFFFFF,
                    np.where(opc == 1, b.astype(np.uint64) >> s, (b.view(np.int32).astype(np.int64) >> s.astype(np.int64)) & 0xFFFFFFFF))
    assert (res == want.astype(np.uint64)).all()
    prog = air.rv32_shift_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    wrong = exp.copy()
    wrong[0][5] = (int(wrong[0][5]) + 1) % 256
    assert air.check_trace(prog, wrong, NOPV) != []
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(v.view(np.int32)).to(dev)  # noqa: E731
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=dev)
    d_sh = zk.rv32_shift_tracegen(as_dev(opc), as_dev(b), as_dev(c), lh, d_bw)
    assert (zk.download(d_sh).reshape(32, -1) == exp).all()
    bw = zk.download(d_bw).reshape(2, -1)
    assert (bw[0] == rc).all() and (bw[1] == xc).all()
    assert int(rc.astype(np.int64).sum()) == 7 * n and int(xc.astype(np.int64).sum()) == int((opc == 2).sum())
    with pytest.raises(z.ZkhipError):
        zk.rv32_shift_tracegen(as_dev(np.array([3], np.uint32)), as_dev(b[:1]), as_dev(c[:1]), 0, d_bw)
    airs = [dict(program=prog, log_height=lh, width=32, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8).program(), log_height=16, width=2, n_pvs=0, trace=bw, pvs=NOPV,
                 prep=air.bitwise_lookup_prep(8))]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_sh, d_bw], [NOPV] * 2)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def test_rv32_branch_equal_chip(zk, ora):
    """RV32 branch-equal core (BEQ / BNE, OpenVM BranchEqualCoreAir<4>) filled on the device from execution records, the inverse of
    the first limb difference computed on the device: cells == oracle, `taken` == the integers' comparison, the oracle's trace
    satisfies the AIR and tampered decisions do not; the chip proves from the device-resident trace (bytes == oracle)."""
    rng = np.random.default_rng(23)
    n, lh = 4000, 12
    opc = rng.integers(0, 2, n).astype(np.uint32)
    a = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    b = a.copy()
    kind = rng.integers(0, 3, n)
    b[kind == 0] = rng.integers(0, 1 << 32, int((kind == 0).sum()), dtype=np.uint64).astype(np.uint32)
    b[kind == 1] ^= (1 << rng.integers(0, 32, int((kind == 1).sum()))).astype(np.uint32)    # differ in exactly one bit
    off = rng.integers(-2048, 2048, n) * 2
    imm = np.where(off < 0, P + off, off).astype(np.uint32)
    exp, bad = ora.rv32_branch_eq_trace(opc, a, b, imm, lh)
    assert bad == 0
    taken = np.where(opc == 0, a == b, a != b)
    assert (exp[8][:n] == taken).all() and (exp[16][:n] == np.where(taken, imm, 4)).all()
    prog = air.rv32_branch_eq_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    for col, row in ((8, 0), (16, 5), (12, int(np.nonzero(a != b)[0][0]))):
        w = exp.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != []
    as_dev = lambda v: torch.from_numpy(v.view(np.int32)).to(zk.device)  # noqa: E731
    d_tr = zk.rv32_branch_eq_tracegen(as_dev(opc), as_dev(a), as_dev(b), as_dev(imm), lh)
    assert (zk.download(d_tr).reshape(17, -1) == exp).all()
    with pytest.raises(z.ZkhipError):
        zk.rv32_branch_eq_tracegen(as_dev(np.array([2], np.uint32)), as_dev(a[:1]), as_dev(b[:1]), as_dev(imm[:1]), 0)
    with pytest.raises(z.ZkhipError):   # an offset that is not a field element
        zk.rv32_branch_eq_tracegen(as_dev(opc[:1]), as_dev(a[:1]), as_dev(b[:1]), as_dev(np.array([P], np.uint32)), 0)
    airs = [dict(program=prog, log_height=lh, width=17, n_pvs=0, trace=exp, pvs=NOPV)]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_tr], [NOPV])
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV], proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def test_rv32_branch_less_than_chip(zk, ora):
    """RV32 branch-less-than core (BLT / BLTU / BGE / BGEU, OpenVM BranchLessThanCoreAir<4, 8>) filled on the device from execution
    records, its range requests counted into the bitwise table in the same pass: cells == oracle (whose comparison is on the
    integers, its marker scan on the limbs), `taken` == the integers' comparison incl. equal operands and operands that differ
    only in the sign bit, the oracle's trace satisfies the AIR and tampered decisions do not; chip + table prove from the
    device-resident traces (bytes == oracle)."""
    rng = np.random.default_rng(29)
    n, lh = 4000, 12
    opc = rng.integers(0, 4, n).astype(np.uint32)
    a = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    b[:500] = a[:500]
    b[500:900] = a[500:900] ^ 0x80000000
    b[900:1200] = a[900:1200] ^ 0x100
    off = rng.integers(-2048, 2048, n) * 2
    imm = np.where(off < 0, P + off, off).astype(np.uint32)
    exp, rc, bad = ora.rv32_branch_lt_trace(opc, a, b, imm, lh)
    assert bad == 0
    lt = np.where(opc % 2 == 0, a.view(np.int32) < b.view(np.int32), a < b)
    taken = np.where(opc >= 2, ~lt, lt)
    assert (exp[8][:n] == lt).all() and (exp[9][:n] == taken).all() and (exp[22][:n] == np.where(taken, imm, 4)).all()
    assert int(rc.astype(np.int64).sum()) == n + int((a != b).sum())
    prog = air.rv32_branch_lt_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    for col, row in ((9, 0), (8, 600), (22, 5), (21, 1000)):
        w = exp.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != []
    as_dev = lambda v: torch.from_numpy(v.view(np.int32)).to(zk.device)  # noqa: E731
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tr = zk.rv32_branch_lt_tracegen(as_dev(opc), as_dev(a), as_dev(b), as_dev(imm), lh, d_bw)
    assert (zk.download(d_tr).reshape(23, -1) == exp).all()
    bw = zk.download(d_bw).reshape(2, -1)
    assert (bw[0] == rc).all() and (bw[1] == 0).all()
    with pytest.raises(z.ZkhipError):
        zk.rv32_branch_lt_tracegen(as_dev(np.array([4], np.uint32)), as_dev(a[:1]), as_dev(b[:1]), as_dev(imm[:1]), 0, d_bw)
    with pytest.raises(z.ZkhipError):   # an offset that is not a field element
        zk.rv32_branch_lt_tracegen(as_dev(opc[:1]), as_dev(a[:1]), as_dev(b[:1]), as_dev(np.array([P], np.uint32)), 0, d_bw)
    airs = [dict(program=prog, log_height=lh, width=23, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8).program(), log_height=16, width=2, n_pvs=0, trace=bw, pvs=NOPV,
                 prep=air.bitwise_lookup_prep(8))]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_tr, d_bw], [NOPV] * 2)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def _prove_chip_with_bitwise_table(zk, ora, prog, width, lh, exp, d_tr, d_bw, rc):
    bw = zk.download(d_bw).reshape(2, -1)
    assert (bw[0] == rc).all() and (bw[1] == 0).all()
    airs = [dict(program=prog, log_height=lh, width=width, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=air.bitwise_lookup_air(8).program(), log_height=16, width=2, n_pvs=0, trace=bw, pvs=NOPV,
                 prep=air.bitwise_lookup_prep(8))]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_tr, d_bw], [NOPV] * 2)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def test_rv32_jal_lui_chip(zk, ora):
    """RV32 JAL / LUI core (OpenVM Rv32JalLuiCoreAir) from (opcode, pc, immediate) records: cells == oracle, rd == pc + 4 resp.
    imm << 12 as integers, the oracle's trace satisfies the AIR and tampered cells do not, records out of range are refused;
    chip + bitwise table prove from the device-resident traces (bytes == oracle)."""
    rng = np.random.default_rng(31)
    n, lh = 3000, 12
    opc = rng.integers(0, 2, n).astype(np.uint32)
    pc = (rng.integers(0, 1 << 28, n) * 4).astype(np.uint32)
    pc[:2] = [0, (1 << 30) - 8]
    off = rng.integers(-(1 << 19), 1 << 19, n) * 2
    imm = np.where(opc == 0, np.where(off < 0, P + off, off), rng.integers(0, 1 << 20, n)).astype(np.uint32)
    exp, rc, bad = ora.rv32_jal_lui_trace(opc, pc, imm, lh)
    assert bad == 0 and int(rc.astype(np.int64).sum()) == 2 * n + int((opc == 0).sum())
    rd = sum(exp[2 + i][:n].astype(np.uint64) << (8 * i) for i in range(4))
    assert (rd == np.where(opc == 0, pc.astype(np.uint64) + 4, (imm.astype(np.uint64) << 12) & 0xFFFFFFFF)).all()
    prog = air.rv32_jal_lui_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    for col, row in ((2, 0), (8, 1), (5, 2)):
        w = exp.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != []
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(zk.device)  # noqa: E731
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tr = zk.rv32_jal_lui_tracegen(as_dev(opc), as_dev(pc), as_dev(imm), lh, d_bw)
    assert (zk.download(d_tr).reshape(9, -1) == exp).all()
    one = lambda v: as_dev(np.array([v], np.uint32))   # noqa: E731
    for o, p_, i_ in ((2, 0, 0), (1, 0, 1 << 20), (0, (1 << 30) - 4, 8), (0, 0, P)):
        with pytest.raises(z.ZkhipError):
            zk.rv32_jal_lui_tracegen(one(o), one(p_), one(i_), 0, torch.zeros(2 << 16, dtype=torch.int32, device=zk.device))
        assert ora.rv32_jal_lui_trace([o], [p_], [i_], 0)[2] == 1
    _prove_chip_with_bitwise_table(zk, ora, prog, 9, lh, exp, d_tr, d_bw, rc)


def test_rv32_auipc_chip(zk, ora):
    """RV32 AUIPC core (OpenVM Rv32AuipcCoreAir) from (pc, 20-bit immediate) records: cells == oracle, rd == pc + (imm << 12)
    mod 2^32 as integers incl. wrap-around, AIR satisfied / tampering caught; proves with the bitwise table (bytes == oracle)."""
    rng = np.random.default_rng(37)
    n, lh = 3000, 12
    pc = rng.integers(0, 1 << 30, n).astype(np.uint32)
    imm = rng.integers(0, 1 << 20, n).astype(np.uint32)
    pc[:10], imm[:10] = (1 << 30) - 1, (1 << 20) - 1
    exp, rc, bad = ora.rv32_auipc_trace(pc, imm, lh)
    assert bad == 0 and int(rc.astype(np.int64).sum()) == 5 * n
    rd = sum(exp[9 + i][:n].astype(np.uint64) << (8 * i) for i in range(4))
    assert (rd == (pc.astype(np.uint64) + (imm.astype(np.uint64) << 12)) & 0xFFFFFFFF).all()
    prog = air.rv32_auipc_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    for col, row in ((9, 0), (10, 1), (12, 2), (6, 3)):
        w = exp.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != []
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(zk.device)  # noqa: E731
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tr = zk.rv32_auipc_tracegen(as_dev(pc), as_dev(imm), lh, d_bw)
    assert (zk.download(d_tr).reshape(14, -1) == exp).all()
    for p_, i_ in ((0, 1 << 20), (1 << 30, 0)):   # an immediate or a pc too wide
        with pytest.raises(z.ZkhipError):
            zk.rv32_auipc_tracegen(as_dev(np.array([p_], np.uint32)), as_dev(np.array([i_], np.uint32)), 0,
                                   torch.zeros(2 << 16, dtype=torch.int32, device=zk.device))
        assert ora.rv32_auipc_trace([p_], [i_], 0)[2] == 1
    _prove_chip_with_bitwise_table(zk, ora, prog, 14, lh, exp, d_tr, d_bw, rc)


def test_rv32_jalr_chip(zk, ora):
    """RV32 JALR core (OpenVM Rv32JalrCoreAir) from (pc, rs1, raw 12-bit immediate) records: cells == oracle, to_pc ==
    (rs1 + sext(imm)) & ~1 as integers incl. negative immediates and wrap-around, AIR satisfied / tampering caught, a target
    that is not a field element refused; proves with the bitwise table (bytes == oracle)."""
    rng = np.random.default_rng(41)
    n, lh = 3000, 12
    pc = (rng.integers(0, 1 << 28, n) * 4).astype(np.uint32)
    rs1 = rng.integers(0, 1 << 30, n).astype(np.uint32)
    imm = rng.integers(0, 1 << 12, n).astype(np.uint32)
    rs1[:5] = [4096, 1, 0x7FF, 0xFFFFFFFF, 0x800]
    imm[:5] = [0xFFF, 0x001, 0x7FF, 1, 0x800]
    exp, rc, bad = ora.rv32_jalr_trace(pc, rs1, imm, lh)
    assert bad == 0 and int(rc.astype(np.int64).sum()) == 5 * n
    ext = np.where(imm >= 2048, imm.astype(np.int64) - 4096, imm.astype(np.int64))
    assert (exp[18][:n] == ((rs1.astype(np.int64) + ext) & 0xFFFFFFFF) & ~1).all()
    prog = air.rv32_jalr_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    for col, row in ((18, 5), (17, 6), (9, 7), (4, 8), (13, 9)):
        w = exp.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != []
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(zk.device)  # noqa: E731
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tr = zk.rv32_jalr_tracegen(as_dev(pc), as_dev(rs1), as_dev(imm), lh, d_bw)
    assert (zk.download(d_tr).reshape(20, -1) == exp).all()
    one = lambda v: as_dev(np.array([v], np.uint32))   # noqa: E731
    for p_, r_, i_ in ((0, 0, 0xFFF), (0, 0, 1 << 12), ((1 << 30) - 4, 0, 0)):   # target 0xffffffff & ~1 >= p; immediate too wide; pc too large
        with pytest.raises(z.ZkhipError):
            zk.rv32_jalr_tracegen(one(p_), one(r_), one(i_), 0, torch.zeros(2 << 16, dtype=torch.int32, device=zk.device))
        assert ora.rv32_jalr_trace([p_], [r_], [i_], 0)[2] == 1
    _prove_chip_with_bitwise_table(zk, ora, prog, 20, lh, exp, d_tr, d_bw, rc)


def test_rv32_mulh_chip(zk, ora):
    """RV32 high-multiplication core (OpenVM MulHCoreAir<4, 8>: MULH / MULHSU / MULHU) from records on the device, its eight
    (limb, carry) requests counted into the range-tuple table and its sign requests into the bitwise table in the same pass:
    cells == oracle (whose result limbs come from the 64-bit integer product), high word == the integers' product for every
    signedness incl. the extreme operands; chip + both tables prove from the device-resident traces (bytes == oracle); a wrong
    result limb leaves the tuple bus unbalanced and the verifier refuses."""
    rng = np.random.default_rng(43)
    n, lh = 4000, 12
    opc = rng.integers(0, 3, n).astype(np.uint32)
    b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    c = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    ext = np.array([0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0, 1], np.uint32)
    k = 0
    for o in range(3):
        for x in ext:
            for y in ext:
                opc[k], b[k], c[k] = o, x, y
                k += 1
    sx, sy = 256, 2048
    exp, tc, rc, bad = ora.rv32_mulh_trace(opc, b, c, lh, sx, sy)
    assert bad == 0 and int(tc.astype(np.int64).sum()) == 8 * n and int(rc.astype(np.int64).sum()) == int((opc != 2).sum() + (opc == 0).sum())
    sb = np.where(opc != 2, b.view(np.int32).astype(object), b.astype(object))
    sc = np.where(opc == 0, c.view(np.int32).astype(object), c.astype(object))
    want = np.array([((int(x) * int(y)) >> 32) & 0xFFFFFFFF for x, y in zip(sb, sc)], dtype=np.uint64)
    assert (sum(exp[i][:n].astype(np.uint64) << (8 * i) for i in range(4)) == want).all()
    prog = air.rv32_mulh_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(dev)  # noqa: E731
    d_tab = torch.zeros(sx * sy, dtype=torch.int32, device=dev)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=dev)
    d_tr = zk.rv32_mulh_tracegen(as_dev(opc), as_dev(b), as_dev(c), lh, d_tab, d_bw, sx, sy)
    assert (zk.download(d_tr).reshape(21, -1) == exp).all()
    assert (zk.download(d_tab) == tc).all()
    bw = zk.download(d_bw).reshape(2, -1)
    assert (bw[0] == rc).all() and (bw[1] == 0).all()
    with pytest.raises(z.ZkhipError):
        zk.rv32_mulh_tracegen(as_dev(np.array([3], np.uint32)), as_dev(b[:1]), as_dev(c[:1]), 0, d_tab.clone(), d_bw.clone(), sx, sy)
    with pytest.raises(z.ZkhipError):   # a table that cannot hold the carries
        zk.rv32_mulh_tracegen(as_dev(opc), as_dev(b), as_dev(c), lh, d_tab.clone(), d_bw.clone(), 256, 1024)
    airs = [dict(program=prog, log_height=lh, width=21, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=air.range_tuple_table_air(sx, sy).program(), log_height=19, width=1, n_pvs=0, trace=tc.reshape(1, -1), pvs=NOPV,
                 prep=air.range_tuple_prep(sx, sy)),
            dict(program=air.bitwise_lookup_air(8).program(), log_height=16, width=2, n_pvs=0, trace=bw, pvs=NOPV, prep=air.bitwise_lookup_prep(8))]
    params = (1, 0, 6, 2, 2)
    pk = z.ProvingKey(zk, params, airs)
    proof = pk.prove([d_tr, d_tab, d_bw], [NOPV] * 3)
    assert z.verify(params, pk.verifying_airs(), [NOPV] * 3, proof) == 0
    assert proof == ora.stark_prove(params, airs).tobytes()
    wrong = exp.copy()
    wrong[0][7] = (int(wrong[0][7]) + 1) % 256
    assert z.verify(params, pk.verifying_airs(), [NOPV] * 3, pk.prove([zk.upload(wrong.reshape(-1)), d_tab, d_bw], [NOPV] * 3)) != 0
    pk.close()


def test_rv32_loadstore_chip(zk, ora):
    """RV32 load/store cores (OpenVM LoadStoreCoreAir<4> + LoadSignExtendCoreAir<4, 8> in one chip) from (case, read word, prev
    word) records: cells == oracle, the written word == the Python statement of each of the 20 (opcode, byte offset) cases, the
    oracle's trace satisfies the AIR and tampered cells do not; proves with the bitwise table (bytes == oracle)."""
    rng = np.random.default_rng(47)
    n, lh = 4000, 12
    cs = rng.integers(0, 20, n).astype(np.uint32)
    cs[:20] = np.arange(20)
    rd = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    pv = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    rd[20:40] = 0x80FF7F00
    cs[20:40] = np.arange(20)
    exp, rc, bad = ora.rv32_loadstore_trace(cs, rd, pv, lh)
    assert bad == 0 and int(rc.astype(np.int64).sum()) == int((cs >= 14).sum())
    for i in range(n):
        kind, s = air.RV32_LOADSTORE_CASES[cs[i]]
        r, p_ = int(rd[i]), int(pv[i])
        if kind in ("lw", "sw"):
            w = r
        elif kind in ("lhu", "lh"):
            w = (r >> (8 * s)) & 0xFFFF
            w |= 0xFFFF0000 if kind == "lh" and w & 0x8000 else 0
        elif kind in ("lbu", "lb"):
            w = (r >> (8 * s)) & 0xFF
            w |= 0xFFFFFF00 if kind == "lb" and w & 0x80 else 0
        elif kind == "sh":
            w = (p_ & ~(0xFFFF << (8 * s)) & 0xFFFFFFFF) | ((r & 0xFFFF) << (8 * s))
        else:
            w = (p_ & ~(0xFF << (8 * s)) & 0xFFFFFFFF) | ((r & 0xFF) << (8 * s))
        assert sum(int(exp[8 + k][i]) << (8 * k) for k in range(4)) == w, (i, kind, s)
    prog = air.rv32_loadstore_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    for col, row in ((8, 0), (9, 1), (11, 14), (32, 16), (10, 9), (12, 5)):
        w = exp.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != []
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(zk.device)  # noqa: E731
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tr = zk.rv32_loadstore_tracegen(as_dev(cs), as_dev(rd), as_dev(pv), lh, d_bw)
    assert (zk.download(d_tr).reshape(33, -1) == exp).all()
    with pytest.raises(z.ZkhipError):
        zk.rv32_loadstore_tracegen(as_dev(np.array([20], np.uint32)), as_dev(rd[:1]), as_dev(pv[:1]), 0, torch.zeros(2 << 16, dtype=torch.int32, device=zk.device))
    _prove_chip_with_bitwise_table(zk, ora, prog, 33, lh, exp, d_tr, d_bw, rc)


def test_rv32_divrem_chip(zk, ora):
    """RV32 division core (DIV / DIVU / REM / REMU, the job of OpenVM's DivRemCoreAir<4, 8>) from records on the device: cells ==
    oracle (quotient and remainder from C's division with RISC-V's two exceptions), q and r == Python's integers for every
    opcode incl. division by zero, -2^31 / -1 and the extreme operands; the oracle's trace satisfies the AIR and tampered cells do
    not; chip + tuple table + bitwise table prove from the device-resident traces (bytes == oracle); a wrong quotient limb leaves
    the tuple bus unbalanced and the verifier refuses."""
    rng = np.random.default_rng(53)
    n, lh = 4000, 12
    opc = rng.integers(0, 4, n).astype(np.uint32)
    b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    c = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    c[500:1500] = rng.integers(0, 1 << 16, 1000).astype(np.uint32)
    c[1500:2000] = (0 - rng.integers(1, 1 << 12, 500)).astype(np.uint32)
    c[2000:2200] = rng.integers(0, 4, 200).astype(np.uint32)
    ext = np.array([0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0, 1, 2, 0xFFFFFFFE, 0x80000001], np.uint32)
    k = 0
    for o in range(4):
        for x in ext:
            for y in ext:
                opc[k], b[k], c[k] = o, x, y
                k += 1
    sx, sy = 256, 2048
    exp, tc, rc, bad = ora.rv32_divrem_trace(opc, b, c, lh, sx, sy)
    assert bad == 0 and int(tc.astype(np.int64).sum()) == 8 * n
    s32 = lambda v: v - (1 << 32) if v >= 1 << 31 else v   # noqa: E731
    for i in range(n):
        o, bv, cv = int(opc[i]), int(b[i]), int(c[i])
        if cv == 0:
            q, r = 0xFFFFFFFF, bv
        elif o in (0, 2):
            sb, sc = s32(bv), s32(cv)
            if sb == -(1 << 31) and sc == -1:
                q, r = bv, 0
            else:
                q = abs(sb) // abs(sc)
                q = -q if (sb < 0) != (sc < 0) else q
                r = (sb - q * sc) & 0xFFFFFFFF
                q &= 0xFFFFFFFF
        else:
            q, r = bv // cv, bv % cv
        assert sum(int(exp[8 + j][i]) << (8 * j) for j in range(4)) == q and sum(int(exp[12 + j][i]) << (8 * j) for j in range(4)) == r, (o, bv, cv)
    prog = air.rv32_divrem_core_air().program()
    assert air.check_trace(prog, exp, NOPV) == []
    for col, row in ((12, 301), (30, 303), (36, 304), (16, 1600), (27, 305)):   # b_sign, q_sign and the q limbs are held by the lookups alone
        w = exp.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != []
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(dev)  # noqa: E731
    d_tab = torch.zeros(sx * sy, dtype=torch.int32, device=dev)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=dev)
    d_tr = zk.rv32_divrem_tracegen(as_dev(opc), as_dev(b), as_dev(c), lh, d_tab, d_bw, sx, sy)
    got = zk.download(d_tr).reshape(41, -1)
    assert (got == exp).all(), [int(q) for q in range(41) if (got[q] != exp[q]).any()]
    assert (zk.download(d_tab) == tc).all()
    bw = zk.download(d_bw).reshape(2, -1)
    assert (bw[0] == rc).all() and (bw[1] == 0).all()
    with pytest.raises(z.ZkhipError):
        zk.rv32_divrem_tracegen(as_dev(np.array([4], np.uint32)), as_dev(b[:1]), as_dev(c[:1]), 0, d_tab.clone(), d_bw.clone(), sx, sy)
    airs = [dict(program=prog, log_height=lh, width=41, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=air.range_tuple_table_air(sx, sy).program(), log_height=19, width=1, n_pvs=0, trace=tc.reshape(1, -1), pvs=NOPV,
                 prep=air.range_tuple_prep(sx, sy)),
            dict(program=air.bitwise_lookup_air(8).program(), log_height=16, width=2, n_pvs=0, trace=bw, pvs=NOPV, prep=air.bitwise_lookup_prep(8))]
    params = (1, 0, 6, 2, 2)
    pk = z.ProvingKey(zk, params, airs)
    proof = pk.prove([d_tr, d_tab, d_bw], [NOPV] * 3)
    assert z.verify(params, pk.verifying_airs(), [NOPV] * 3, proof) == 0
    assert proof == ora.stark_prove(params, airs).tobytes()
    wrong = exp.copy()
    wrong[8][300] = (int(wrong[8][300]) + 1) % 256
    assert z.verify(params, pk.verifying_airs(), [NOPV] * 3, pk.prove([zk.upload(wrong.reshape(-1)), d_tab, d_bw], [NOPV] * 3)) != 0
    pk.close()


def test_native_field_chips(zk, ora):
    """The native field-arithmetic and field-extension chips (OpenVM native FieldArithmeticCoreAir / FieldExtensionCoreAir: what
    the recursion programs of the aggregation circuits compute with) from records on the device, one field inversion per division:
    cells == oracle, results == Python's modular arithmetic, the AIRs hold and catch a wrong result, division by zero is refused;
    both chips prove from the device-resident traces (bytes == oracle)."""
    rng = np.random.default_rng(59)
    n, lh = 3000, 12
    op = rng.integers(0, 4, n).astype(np.uint32)
    b = rng.integers(0, P, n).astype(np.uint32)
    c = rng.integers(1, P, n).astype(np.uint32)
    b[:4], c[:4] = [0, P - 1, 1, P - 1], [1, P - 1, P - 1, 2]
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint32).view(np.int32)).to(zk.device)  # noqa: E731
    exp, bad = ora.field_arith_trace(op, b, c, lh)
    want = [(int(x) + int(y)) % P if o == 0 else (int(x) - int(y)) % P if o == 1 else int(x) * int(y) % P if o == 2 else int(x) * pow(int(y), P - 2, P) % P
            for o, x, y in zip(op, b, c)]
    assert bad == 0 and (exp[0][:n] == np.array(want, np.uint32)).all()
    prog_a = air.field_arith_air().program()
    assert air.check_trace(prog_a, exp, NOPV) == []
    w = exp.copy()
    w[0][7] = (int(w[0][7]) + 1) % P
    assert air.check_trace(prog_a, w, NOPV) != []
    d_a = zk.field_arith_tracegen(as_dev(op), as_dev(b), as_dev(c), lh)
    assert (zk.download(d_a).reshape(8, -1) == exp).all()
    with pytest.raises(z.ZkhipError):
        zk.field_arith_tracegen(as_dev([3]), as_dev([5]), as_dev([0]), 0)
    with pytest.raises(z.ZkhipError):
        zk.field_arith_tracegen(as_dev([0]), as_dev([P]), as_dev([1]), 0)
    x = rng.integers(0, P, (n, 4)).astype(np.uint32)
    y = rng.integers(0, P, (n, 4)).astype(np.uint32)
    y[:3] = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, P - 1]]
    exp_e, bad = ora.field_ext_trace(op, x, y, lh)
    prog_e = air.field_ext_air().program()
    assert bad == 0 and air.check_trace(prog_e, exp_e, NOPV) == []
    # division really is the inverse of multiplication: (x / y) * y == x through the oracle's own extension product
    k = int(np.nonzero(op == 3)[0][0])
    back = np.zeros(4, np.uint32)
    ora.lib().ora_ext_mul(ora.p32(np.ascontiguousarray(exp_e[8:12, k])), ora.p32(np.ascontiguousarray(y[k])), ora.p32(back))
    assert back.tolist() == x[k].tolist()
    w = exp_e.copy()
    w[9][k] = (int(w[9][k]) + 1) % P
    assert air.check_trace(prog_e, w, NOPV) != []
    d_e = zk.field_ext_tracegen(as_dev(op), as_dev(x), as_dev(y), lh)
    assert (zk.download(d_e).reshape(20, -1) == exp_e).all()
    with pytest.raises(z.ZkhipError):
        zk.field_ext_tracegen(as_dev([3]), as_dev([[1, 2, 3, 4]]), as_dev([[0, 0, 0, 0]]), 0)
    airs = [dict(program=prog_a, log_height=lh, width=8, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=prog_e, log_height=lh, width=20, n_pvs=0, trace=exp_e, pvs=NOPV)]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_a, d_e], [NOPV] * 2)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def test_variable_range_checker(zk, ora):
    """OpenVM's VariableRangeCheckerChip: one table for every check value < 2^bits, bits <= max_bits.  Multiplicities counted on
    the device from the requesting columns (a column of bit counts, or one constant) == oracle; requests outside the table are
    refused; a user chip sending (value, bits) pairs and the table prove from the device-resident traces (bytes == oracle)."""
    rng = np.random.default_rng(61)
    B, lh = 12, 11
    n = 1 << lh
    bits = rng.integers(0, B + 1, n).astype(np.uint32)
    vals = (rng.integers(0, 1 << 30, n) % (1 << bits.astype(np.uint64))).astype(np.uint32)
    vals[:40], bits[:40] = 0, 0                      # a hot entry
    prep = air.var_range_prep(B)
    assert prep.shape == (2, 1 << (B + 1)) and (prep[0] < (1 << prep[1].astype(np.uint64))).all()
    assert prep[:, 0].tolist() == [0, 0] and prep[:, 1].tolist() == [0, 1] and prep[:, 2].tolist() == [1, 1] and prep[:, -1].tolist() == [0, B + 1]
    cnt, bad = ora.var_range_counts(vals, bits, B)
    assert bad == 0 and int(cnt.astype(np.int64).sum()) == n
    for i in (0, 77, 500):
        assert cnt[(1 << int(bits[i])) - 1 + int(vals[i])] >= 1
    user = np.stack([vals, bits, (vals.astype(np.uint64) * bits % P).astype(np.uint32)])
    d_user = zk.upload(user.reshape(-1))
    d_cnt = zk.var_range_counts_tracegen(d_user[:n], d_user[n:2 * n], B)
    assert (zk.download(d_cnt) == cnt).all()
    # a second requesting column with one constant bit count, accumulated
    more = rng.integers(0, 1 << 7, 999).astype(np.uint32)
    cnt2, _ = ora.var_range_counts(more, 7, B, counts=cnt)
    d_cnt2 = zk.var_range_counts_tracegen(zk.upload(more), 7, B, d_cnt.clone(), accumulate=True)
    assert (zk.download(d_cnt2) == cnt2).all()
    for v, b_ in ((4, 2), (0, B + 1)):
        with pytest.raises(z.ZkhipError):
            zk.var_range_counts_tracegen(zk.upload(np.array([v], np.uint32)), b_, B)
        assert ora.var_range_counts([v], b_, B)[1] == 1
    airs = [dict(program=air.var_range_user_air(7).program(), log_height=lh, width=3, n_pvs=0, trace=user, pvs=NOPV),
            dict(program=air.var_range_table_air(7).program(), log_height=B + 1, width=1, n_pvs=0, trace=cnt.reshape(1, -1), pvs=NOPV, prep=prep)]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_user, d_cnt], [NOPV] * 2)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    pk.close()


def test_native_castf_chip_with_the_variable_range_checker(zk, ora):
    """Native CASTF (a field element below 2^30 to limbs of 8, 8, 8, 6 bits) filled on the device, its four limb checks counted
    into the variable range checker's table in the same pass: cells and counts == oracle, the AIR holds, a value >= 2^30 is
    refused; chip + table prove from the device-resident traces (bytes == oracle); a limb moved between positions leaves the bus
    unbalanced."""
    rng = np.random.default_rng(67)
    n, lh, B = 3000, 12, 10
    xs = rng.integers(0, 1 << 30, n).astype(np.uint32)
    xs[:4] = [0, (1 << 30) - 1, 255, 1 << 24]
    exp, cnt, bad = ora.castf_trace(xs, lh, B)
    assert bad == 0 and int(cnt.astype(np.int64).sum()) == 4 * n
    assert (sum(exp[1 + i][:n].astype(np.uint64) << (8 * i) for i in range(4)) == xs).all() and exp[4][:n].max() < 64
    prog = air.castf_air(7).program()
    assert air.check_trace(prog, exp, NOPV) == []
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint32).view(np.int32)).to(zk.device)  # noqa: E731
    d_cnt = torch.zeros(1 << (B + 1), dtype=torch.int32, device=zk.device)
    d_tr = zk.castf_tracegen(as_dev(xs), lh, d_cnt, B)
    assert (zk.download(d_tr).reshape(6, -1) == exp).all() and (zk.download(d_cnt) == cnt).all()
    with pytest.raises(z.ZkhipError):
        zk.castf_tracegen(as_dev([1 << 30]), 0, d_cnt.clone(), B)
    airs = [dict(program=prog, log_height=lh, width=6, n_pvs=0, trace=exp, pvs=NOPV),
            dict(program=air.var_range_table_air(7).program(), log_height=B + 1, width=1, n_pvs=0, trace=cnt.reshape(1, -1), pvs=NOPV,
                 prep=air.var_range_prep(B))]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_tr, d_cnt], [NOPV] * 2)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    wrong = exp.copy()     # 2^24 as limbs (0, 0, 256, 0) instead of (0, 0, 0, 1): the sum still holds, the range check does not
    wrong[3][3], wrong[4][3] = 256, 0
    assert air.check_trace(prog, wrong, NOPV) == []
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, pk.prove([zk.upload(wrong.reshape(-1)), d_cnt], [NOPV] * 2)) != 0
    pk.close()
