"""GPU: the 256-bit ALU chip on the device: zkhip_int256_alu_tracegen == the oracle twin cell for cell (101 columns) and count for count
in the bitwise table's XOR column, a = b op c is Python's, the HIP proof of the chip with its table == the oracle's."""
import json
import os

import numpy as np
import pytest
import torch

import zkvm_prover_amd as z

import int256_util as iu

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_int256_trace_and_proof(zk, ora):
    kat = [(c["op"], int(c["b"], 16), int(c["c"], 16), int(c["a"], 16)) for c in json.load(open(os.path.join(HERE, "golden", "int256_kat.json")))["cases"]]
    kat_mul = [(b, c, a) for op, b, c, a in kat if op == 5]
    kat = [k for k in kat if k[0] < 5]
    cases = [(op, b, c) for op, b, c, _ in kat]
    log_h = 8
    d_recs = torch.from_numpy(iu.records(cases).reshape(-1).view(np.int32)).to(zk.device)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tr = zk.int256_alu_tracegen(d_recs, len(cases), log_h, d_bw)
    got = zk.download(d_tr).reshape(iu.WIDTH, -1)
    tr, xc, bad = iu.ora_trace(ora, cases, log_h)
    assert bad == 0 and (got == tr).all()
    bw = zk.download(d_bw)
    assert (bw[1 << 16:] == xc).all() and not bw[:1 << 16].any()
    for row, (_, _, _, a) in enumerate(kat):
        assert bytes(got[0:32, row].astype(np.uint8)) == a.to_bytes(32, "little")
    inst = iu.instance(got, xc, log_h)
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [iu.NOPV] * 2
    proof = pk.prove([d_tr, d_bw], pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    bad_tr = d_tr.clone()
    bad_tr[3] ^= 1   # another result limb
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, pk.prove([bad_tr, d_bw], pvs)) != 0
    pk.close()
    # the multiplication chip with its two tables
    pairs = [(b, c) for b, c, _ in kat_mul]
    recs = np.array([iu.words(b) + iu.words(c) for b, c in pairs], dtype=np.uint32)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tup = torch.zeros(iu.SX * iu.SY, dtype=torch.int32, device=zk.device)
    d_tr = zk.int256_mul_tracegen(torch.from_numpy(recs.reshape(-1).view(np.int32)).to(zk.device), len(pairs), 5, d_bw, d_tup, iu.SX, iu.SY)
    got = zk.download(d_tr).reshape(iu.MUL_WIDTH, -1)
    tr, bw, tup, bad = iu.ora_mul_trace(ora, pairs, 5)
    assert bad == 0 and (got == tr).all() and (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_tup) == tup).all()
    for row, (_, _, a) in enumerate(kat_mul):
        assert bytes(got[0:32, row].astype(np.uint8)) == a.to_bytes(32, "little")
    inst = iu.mul_instance(got, bw, tup, 5)
    pk = z.ProvingKey(zk, PARAMS, inst)
    proof = pk.prove([d_tr, d_bw, d_tup], [iu.NOPV] * 3)
    assert z.verify(PARAMS, pk.verifying_airs(), [iu.NOPV] * 3, proof) == 0 and proof == ora.stark_prove(PARAMS, inst).tobytes()
    pk.close()


def test_int256_chip_throughput(zk):
    """2^18 operations: trace generation and proof at the reference's parameters (reported, not asserted on time)."""
    import time
    from zkvm_prover_amd import air

    log_h, n = 18, 1 << 18
    rng = np.random.default_rng(1)
    recs = rng.integers(0, 1 << 32, size=(n, 17), dtype=np.uint64).astype(np.uint32)
    recs[:, 0] %= 5
    d_recs = torch.from_numpy(recs.reshape(-1).view(np.int32)).to(zk.device)
    program, width = z.int256_alu_air(iu.BITWISE_BUS)
    inst = [dict(program=program, log_height=log_h, width=width, n_pvs=0),
            dict(program=air.bitwise_lookup_air(8, iu.BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, prep=air.bitwise_lookup_prep(8))]
    pk = z.ProvingKey(zk, z.DEFAULT_PARAMS, inst)
    pvs = [iu.NOPV] * 2

    def gen():
        d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
        return [zk.int256_alu_tracegen(d_recs, n, log_h, d_bw), d_bw]
    T = gen()
    assert z.verify(z.DEFAULT_PARAMS, pk.verifying_airs(), pvs, pk.prove(T, pvs)) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T = gen()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pk.prove(T, pvs)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("int256 chip: %d operations, %.1f M cells: tracegen %.2f ms, proof %.1f ms" % (n, iu.WIDTH * (1 << log_h) / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    pk.close()


def test_device_comparison_trace_and_proof(zk, ora):
    """zkhip_int256_cmp_tracegen == the tests' twin cell for cell and count for count; the answers are Python's; the HIP proof of the chip
    with the bitwise table == the oracle's; a non-comparison opcode is refused."""
    kat = json.load(open(os.path.join(HERE, "golden", "int256_kat.json")))["cmp"]
    cases = [(c["op"], int(c["b"], 16), int(c["c"], 16)) for c in kat]
    log_h = 8
    recs = iu.records(cases).reshape(-1)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tr = zk.int256_cmp_tracegen(d_recs, len(cases), log_h, d_bw)
    got = zk.download(d_tr).reshape(iu.CMP_WIDTH, -1)
    tr, bw = iu.cmp_twin_trace(cases, log_h)
    assert (got == tr).all() and (zk.download(d_bw)[:1 << 16] == bw).all()
    for row, c in enumerate(kat):
        out = int(got[64, row]) if c["op"] != 8 else 1 - int(got[65:97, row].sum())
        assert out == int(c["a"], 16)
    inst = iu.cmp_instance(got, bw, log_h)
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [iu.NOPV] * 2
    proof = pk.prove([d_tr, d_bw], pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    bad_tr = d_tr.clone()
    bad_tr[64 * (1 << log_h) + 1] ^= 1   # (the Montgomery form of) another answer
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, pk.prove([bad_tr, d_bw], pvs)) != 0
    pk.close()
    with pytest.raises(Exception):
        zk.int256_cmp_tracegen(torch.from_numpy(iu.records([(2, 1, 2)]).reshape(-1).view(np.int32)).to(zk.device), 1, 1,
                               torch.zeros(2 << 16, dtype=torch.int32, device=zk.device))


def test_device_shift_trace_and_proof(zk, ora):
    """zkhip_int256_shift_tracegen == the tests' twin cell for cell and count for count (both columns of the bitwise table); the results are
    Python's; the HIP proof of the chip with the bitwise table == the oracle's; a non-shift opcode is refused."""
    kat = json.load(open(os.path.join(HERE, "golden", "int256_kat.json")))["shift"]
    cases = [(c["op"], int(c["b"], 16), int(c["c"], 16)) for c in kat]
    log_h = 7
    recs = iu.records(cases).reshape(-1)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tr = zk.int256_shift_tracegen(d_recs, len(cases), log_h, d_bw)
    got = zk.download(d_tr).reshape(iu.SH_WIDTH, -1)
    tr, bw, xc = iu.shift_twin_trace(cases, log_h)
    assert (got == tr).all() and (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_bw)[1 << 16:] == xc).all()
    for row, c in enumerate(kat):
        assert bytes(got[0:32, row].astype(np.uint8)) == int(c["a"], 16).to_bytes(32, "little")
    inst = iu.shift_instance(got, bw, xc, log_h)
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [iu.NOPV] * 2
    proof = pk.prove([d_tr, d_bw], pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    bad_tr = d_tr.clone()
    bad_tr[3 * (1 << log_h) + 1] ^= 1   # another result limb
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, pk.prove([bad_tr, d_bw], pvs)) != 0
    pk.close()
    with pytest.raises(Exception):
        zk.int256_shift_tracegen(torch.from_numpy(iu.records([(6, 1, 2)]).reshape(-1).view(np.int32)).to(zk.device), 1, 1,
                                 torch.zeros(2 << 16, dtype=torch.int32, device=zk.device))
