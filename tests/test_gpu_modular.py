"""GPU: the modular-multiplication chip on the device: zkhip_modmul_tracegen == the oracle twin cell for cell (325 columns) and count for
count in both lookup tables, r = a b mod P is Python's, the HIP proof of the chip with its tables == the oracle's."""
import json
import os

import numpy as np
import pytest
import torch

import zkvm_prover_amd as z

import modular_util as mu

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_modmul_trace_and_proof(zk, ora):
    kat = json.load(open(os.path.join(HERE, "golden", "modular_kat.json")))
    for name in ("secp256k1_p", "secp256k1_n", "bn254_p", "bn254_r"):
        p = int(kat["moduli"][name], 16)
        pairs = [(int(c["a"], 16), int(c["b"], 16)) for c in kat["cases"] if c["modulus"] == name]
        log_h = 5
        recs = np.ascontiguousarray(mu.records_bytes(pairs)).view("<u4").reshape(-1).astype(np.uint32)
        d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
        d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
        d_tup = torch.zeros(mu.SX * mu.SY, dtype=torch.int32, device=zk.device)
        d_tr = zk.modmul_tracegen(p, d_recs, len(pairs), log_h, d_bw, d_tup, mu.SX, mu.SY)
        got = zk.download(d_tr).reshape(mu.WIDTH, -1)
        tr, bw, tup, bad = mu.ora_trace(ora, pairs, p, log_h)
        assert bad == 0 and (got == tr).all()
        assert (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_tup) == tup).all()
        for row, (a, b) in enumerate(pairs):
            assert bytes(got[96:128, row].astype(np.uint8)) == (a * b % p).to_bytes(32, "little")
        if name != "secp256k1_p":
            continue
        inst = mu.instance(p, got, bw, tup, log_h)
        pk = z.ProvingKey(zk, PARAMS, inst)
        pvs = [mu.NOPV] * 3
        proof = pk.prove([d_tr, d_bw, d_tup], pvs)
        assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
        assert proof == ora.stark_prove(PARAMS, inst).tobytes()
        bad_tr = d_tr.clone()
        bad_tr[96 * (1 << log_h) + 2] ^= 1   # another residue
        assert z.verify(PARAMS, pk.verifying_airs(), pvs, pk.prove([bad_tr, d_bw, d_tup], pvs)) != 0
        pk.close()


def test_device_modular_operations_trace(zk, ora):
    """zkhip_modular_tracegen (17-word records: op | a | b) with the four operations mixed == the oracle twin cell for cell and count for
    count; the results are Python's; a division record whose quotient is not reduced is refused."""
    kat = json.load(open(os.path.join(HERE, "golden", "modular_kat.json")))
    for name in ("secp256k1_n", "bn254_p"):
        p = int(kat["moduli"][name], 16)
        rows = [(3, int(c["r"], 16), int(c["b"], 16)) for c in kat["div"] if c["modulus"] == name]
        rows += [(c["op"], int(c["a"], 16), int(c["b"], 16)) for c in kat["addsub"] if c["modulus"] == name][:10]
        rows += [(0, int(c["a"], 16), int(c["b"], 16)) for c in kat["cases"] if c["modulus"] == name][:6]
        rows += [(4, 5, 5), (4, 5, 6), (4, p - 1, p - 1), (4, 0, p - 1)]      # equality tests
        log_h = 6
        recs = np.array([[op] + mu.to_bytes(a).view("<u4").tolist() + mu.to_bytes(b).view("<u4").tolist() for op, a, b in rows], dtype=np.uint32).reshape(-1)
        d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
        d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
        d_tup = torch.zeros(mu.SX * mu.SY, dtype=torch.int32, device=zk.device)
        d_tr = zk.modular_tracegen(p, d_recs, len(rows), log_h, d_bw, d_tup, mu.SX, mu.SY)
        got = zk.download(d_tr).reshape(mu.WIDTH, -1)
        tr, bw, tup, bad = mu.ora_trace(ora, [(a, b) for _, a, b in rows], p, log_h, ops=[o for o, _, _ in rows])
        assert bad == 0 and (got == tr).all()
        assert (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_tup) == tup).all()
        divs = [c for c in kat["div"] if c["modulus"] == name]
        for row, c in enumerate(divs):   # the r columns hold the dividend, the a columns the quotient
            assert bytes(got[96:128, row].astype(np.uint8)) == int(c["a"], 16).to_bytes(32, "little")
            assert int.from_bytes(bytes(got[0:32, row].astype(np.uint8)), "little") * int(c["b"], 16) % p == int(c["a"], 16)
    p = int(kat["moduli"]["bn254_p"], 16)
    recs = np.array([3] + mu.to_bytes(p + 5).view("<u4").tolist() + mu.to_bytes(3).view("<u4").tolist(), dtype=np.uint32)
    with pytest.raises(Exception):
        zk.modular_tracegen(p, torch.from_numpy(recs.view(np.int32)).to(zk.device), 1, 1, torch.zeros(2 << 16, dtype=torch.int32, device=zk.device),
                            torch.zeros(mu.SX * mu.SY, dtype=torch.int32, device=zk.device), mu.SX, mu.SY)


def test_modmul_chip_throughput(zk):
    """2^16 multiplications mod the secp256k1 field prime: trace generation and proof at the reference's parameters (reported)."""
    import time
    from zkvm_prover_amd import air

    p = 2**256 - 2**32 - 977
    log_h, n = 16, 1 << 16
    rng = np.random.default_rng(1)
    recs = rng.integers(0, 1 << 32, size=(n, 16), dtype=np.uint64).astype(np.uint32)
    recs[:, 7] &= 0x7FFFFFFF   # operands below 2^255 < p
    recs[:, 15] &= 0x7FFFFFFF
    d_recs = torch.from_numpy(recs.reshape(-1).view(np.int32)).to(zk.device)
    program, width = z.modmul_air(p, mu.BITWISE_BUS, mu.TUPLE_BUS)
    inst = [dict(program=program, log_height=log_h, width=width, n_pvs=0),
            dict(program=air.bitwise_lookup_air(8, mu.BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, prep=air.bitwise_lookup_prep(8)),
            dict(program=air.range_tuple_table_air(mu.SX, mu.SY, mu.TUPLE_BUS).program(), log_height=15, width=1, n_pvs=0, prep=air.range_tuple_prep(mu.SX, mu.SY))]
    pk = z.ProvingKey(zk, z.DEFAULT_PARAMS, inst)
    pvs = [mu.NOPV] * 3

    def gen():
        d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
        d_tup = torch.zeros(mu.SX * mu.SY, dtype=torch.int32, device=zk.device)
        return [zk.modmul_tracegen(p, d_recs, n, log_h, d_bw, d_tup, mu.SX, mu.SY), d_bw, d_tup]
    T = gen()
    proof = pk.prove(T, pvs)
    assert z.verify(z.DEFAULT_PARAMS, pk.verifying_airs(), pvs, proof) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T = gen()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pk.prove(T, pvs)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("modmul chip: %d multiplications, %.1f M cells: tracegen %.2f ms, proof %.1f ms" % (n, mu.WIDTH * (1 << log_h) / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    pk.close()
