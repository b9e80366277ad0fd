"""GPU: the Fp2 chip on the device: zkhip_fp2_tracegen == the tests' twin cell for cell (648 columns) and count for count in both lookup
tables, the results are Python's (tests/golden/fp2_kat.json: bn254's Fp2, anchored on the published G2 generator), the HIP proof of the
chip with its tables == the oracle's; a division record whose quotient is not reduced is refused."""
import json
import os

import numpy as np
import pytest
import torch

import zkvm_prover_amd as z

import fp2_util as fu

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_fp2_trace_and_proof(zk, ora):
    d = json.load(open(os.path.join(HERE, "golden", "fp2_kat.json")))
    p = int(d["p"], 16)
    cases = [(c["op"], (int(c["a0"], 16), int(c["a1"], 16)), (int(c["b0"], 16), int(c["b1"], 16)), (int(c["r0"], 16), int(c["r1"], 16))) for c in d["cases"]]
    calls = [(op, r if op == 3 else a, b) for op, a, b, r in cases]      # a division's record holds (quotient, divisor)
    log_h = 7
    recs = np.array([fu.record(*c) for c in calls], dtype=np.uint32).reshape(-1)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(zk.device)
    d_bw = torch.zeros(2 << 16, dtype=torch.int32, device=zk.device)
    d_tup = torch.zeros(fu.SX * fu.SY, dtype=torch.int32, device=zk.device)
    d_tr = zk.fp2_tracegen(p, d_recs, len(calls), log_h, d_bw, d_tup, fu.SX, fu.SY)
    got = zk.download(d_tr).reshape(fu.WIDTH, -1)
    tr, bw, tup = fu.twin_trace(calls, p, log_h)
    assert (got == tr).all()
    assert (zk.download(d_bw)[:1 << 16] == bw).all() and (zk.download(d_tup) == tup).all()
    for row, (op, a, b, r) in enumerate(cases):
        cols = (0, 32) if op == 3 else (128, 160)
        assert tuple(int.from_bytes(bytes(got[c:c + 32, row].astype(np.uint8)), "little") for c in cols) == r
    inst = fu.instance(p, got, bw, tup, log_h)
    pk = z.ProvingKey(zk, PARAMS, inst)
    pvs = [fu.NOPV] * 3
    proof = pk.prove([d_tr, d_bw, d_tup], pvs)
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    bad_tr = d_tr.clone()
    bad_tr[128 * (1 << log_h) + 3] ^= 1   # another result limb
    assert z.verify(PARAMS, pk.verifying_airs(), pvs, pk.prove([bad_tr, d_bw, d_tup], pvs)) != 0
    pk.close()
    bad = np.array(fu.record(3, (p + 1, 0), (3, 0)), dtype=np.uint32)       # a "quotient" above the modulus
    with pytest.raises(Exception):
        zk.fp2_tracegen(p, torch.from_numpy(bad.view(np.int32)).to(zk.device), 1, 1, torch.zeros(2 << 16, dtype=torch.int32, device=zk.device),
                        torch.zeros(fu.SX * fu.SY, dtype=torch.int32, device=zk.device), fu.SX, fu.SY)
