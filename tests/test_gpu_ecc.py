"""GPU: the elliptic-curve chip on the device: zkhip_ec_tracegen == the tests' twin cell for cell (772 columns) and count for count in both
lookup tables, the results are Python's (and the published multiples of the generators), the HIP proof of the chip with its tables ==
the oracle's; a record whose slope does not solve the chord identity is refused."""
import json
import os

import numpy as np
import pytest
import torch

import zkvm_prover_amd as z

import ecc_util as eu

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    kat = json.load(open(os.path.join(HERE, "golden", "ecc_kat.json")))
    c = kat["curves"][name]
    p, a = int(c["p"], 16), int(c["a"], 16)
    cases = [(k["op"], (int(k["x1"], 16), int(k["y1"], 16)), (int(k["x2"], 16), int(k["y2"], 16)), int(k["slope"], 16), (int(k["x3"], 16), int(k["y3"], 16)))
             for k in kat["cases"] if k["curve"] == name]
    return p, a, cases


def device_trace(zk, p, a, calls, log_h):
    recs = np.array([eu.record(*c) for c in calls], dtype=np.uint32).reshape(-1)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tup = torch.zeros(eu.SX * eu.SY, dtype=torch.int32, device=zk.device)
    return zk.ec_tracegen(p, a, d_recs, len(calls), log_h, d_bw, d_tup, eu.SX, eu.SY), d_bw, d_tup


@pytest.mark.parametrize("name", ["secp256k1", "p256", "bn254"])
def test_device_ec_trace_and_proof(zk, ora, name):
    p, a, cases = load(name)
    calls = [(op, p1, p2, lam) for op, p1, p2, lam, _ in cases]
    log_h = 4
    d_tr, d_bw, d_tup = device_trace(zk, p, a, calls, log_h)
    got = zk.download(d_tr).reshape(eu.WIDTH, -1)
    tr, bw, tup = eu.twin_trace(calls, p, a, log_h)
    assert (got == tr).all()
    assert (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_tup) == tup).all()
    for row, (_, _, _, _, r) in enumerate(cases):
        assert bytes(got[160:192, row].astype(np.uint8)) == r[0].to_bytes(32, "little") and bytes(got[192:224, row].astype(np.uint8)) == r[1].to_bytes(32, "little")
    if name != "secp256k1":
        return
    inst = eu.instance(p, a, got, bw, tup, log_h)
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [eu.NOPV] * 3
    proof = pk.prove([d_tr, d_bw, d_tup], pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    bad_tr = d_tr.clone()
    bad_tr[160 * (1 << log_h) + 2] ^= 1   # another abscissa
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, pk.prove([bad_tr, d_bw, d_tup], pvs)) != 0
    pk.close()


def test_device_refuses_a_wrong_slope(zk):
    p, a, cases = load("secp256k1")
    op, p1, p2, lam, _ = cases[-1]
    with pytest.raises(Exception):
        device_trace(zk, p, a, [(op, p1, p2, (lam + 1) % p)], 2)


def test_ec_chip_throughput(zk):
    """2^12 point additions on secp256k1: trace generation and proof at the reference's parameters (reported)."""
    import time
    from zkvm_prover_amd import air

    p, a, cases = load("secp256k1")
    log_h, n = 12, 1 << 12
    pts = [c[4] for c in cases] + [c[1] for c in cases]
    rng = np.random.default_rng(1)
    calls = []
    while len(calls) < n:
        p1, p2 = pts[int(rng.integers(len(pts)))], pts[int(rng.integers(len(pts)))]
        if p1[0] != p2[0]:
            calls.append((0, p1, p2, eu.slope_of(0, p, a, p1, p2)))
    recs = np.array([eu.record(*c) for c in calls], dtype=np.uint32).reshape(-1)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
    program, width = z.ec_air(p, a, eu.BITWISE_BUS, eu.TUPLE_BUS)
    inst = [dict(program=program, log_height=log_h, width=width, n_pvs=0),
            dict(program=air.bitwise_lookup_air(8, eu.BITWISE_BUS).program(), log_height=16, width=2, n_pvs=0, prep=air.bitwise_lookup_prep(8)),
            dict(program=air.range_tuple_table_air(eu.SX, eu.SY, eu.TUPLE_BUS).program(), log_height=19, width=1, n_pvs=0, prep=air.range_tuple_prep(eu.SX, eu.SY))]
    pk = z.ProvingKey(zk, z.DEFAULT_PARAMS, inst)
    pvs = [eu.NOPV] * 3

    def gen():
        d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
        d_tup = torch.zeros(eu.SX * eu.SY, dtype=torch.int32, device=zk.device)
        return [zk.ec_tracegen(p, a, d_recs, n, log_h, d_bw, d_tup, eu.SX, eu.SY), d_bw, d_tup]
    T = gen()
    proof = pk.prove(T, pvs)
    assert z.verify(z.DEFAULT_PARAMS, pk.verifying_airs(), pvs, proof) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T = gen()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    proof = pk.prove(T, pvs)
    t2 = time.perf_counter()
    print("\nec chip: %d additions: tracegen %.2f ms, proof %.1f ms (%d bytes)" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(proof)))
    pk.close()
