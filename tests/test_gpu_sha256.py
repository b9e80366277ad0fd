"""GPU: the SHA-256 compression chip on the device: zkhip_sha256_tracegen == the oracle twin cell for cell (433 columns), the digest rows
carry SHA-256 digests that hashlib computes (FIPS 180-4: parity anchored outside this repository), the HIP proof == the oracle's."""
import hashlib

import numpy as np
import pytest
import torch

import zkvm_prover_amd as z

from sha256_util import ROWS, WIDTH, chained_records, digest_of_row, ora_trace

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
NOPV = np.zeros(0, np.uint32)


def test_device_sha256_trace_and_proof(zk, ora):
    rng = np.random.default_rng(11)
    msgs = [b"", b"abc"] + [bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in rng.integers(0, 200, size=8)]
    recs, last = chained_records(msgs, z.sha256_compress_host)
    log_h = 11
    assert len(recs) <= (1 << log_h) // ROWS
    d_recs = torch.from_numpy(recs.reshape(-1).view(np.int32)).to(zk.device)
    d_tr = zk.sha256_tracegen(d_recs, len(recs), log_h)
    got = zk.download(d_tr).reshape(WIDTH, -1)
    assert (got == ora_trace(ora, recs, log_h)).all()
    for m, b in zip(msgs, last):
        assert digest_of_row(got, ROWS * b + 64).hex() == hashlib.sha256(m).hexdigest()
    program, width, prep = z.sha256_air(log_h)
    inst = [dict(program=program, log_height=log_h, width=width, n_pvs=0, trace=got, pvs=NOPV, prep=prep)]
    pk = z.ProvingKey(zk, PARAMS, inst)
    proof = pk.prove([d_tr], [NOPV])
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV], proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    # a wrong digest bit does not prove
    bad = d_tr.clone()
    bad[3 * (1 << log_h) + 64] ^= 1
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV], pk.prove([bad], [NOPV])) != 0
    pk.close()


def test_sha256_chip_throughput(zk):
    """2^16 rows = 1008 blocks: trace generation and proof at the reference's parameters (reported, not asserted on time)."""
    import time

    log_h, n = 16, (1 << 16) // ROWS
    rng = np.random.default_rng(1)
    recs = rng.integers(0, 1 << 32, size=(n, 24), dtype=np.uint64).astype(np.uint32)
    d_recs = torch.from_numpy(recs.reshape(-1).view(np.int32)).to(zk.device)
    program, width, prep = z.sha256_air(log_h)
    inst = [dict(program=program, log_height=log_h, width=width, n_pvs=0, prep=prep)]
    pk = z.ProvingKey(zk, z.DEFAULT_PARAMS, inst)
    d_tr = zk.sha256_tracegen(d_recs, n, log_h)
    proof = pk.prove([d_tr], [NOPV])
    assert z.verify(z.DEFAULT_PARAMS, pk.verifying_airs(), [NOPV], proof) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d_tr = zk.sha256_tracegen(d_recs, n, log_h)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pk.prove([d_tr], [NOPV])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("sha256 chip: %d blocks, %.1f M cells: tracegen %.2f ms, proof %.1f ms" % (n, WIDTH * (1 << log_h) / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    pk.close()
