"""GPU: the Keccak-f[1600] chip on the device: zkhip_keccak_f_tracegen == the oracle twin cell for cell (2633 columns), the export
rows carry SHA3-256 digests that hashlib computes (FIPS 202: parity anchored outside this repository), the HIP proof == the oracle's."""
import hashlib

import numpy as np
import pytest
import torch

import zkvm_prover_amd as z

from test_keccak_cpu import one_block_state, ora_trace

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
NOPV = np.zeros(0, np.uint32)


def test_device_keccak_trace_and_proof(zk, ora):
    rng = np.random.default_rng(11)
    msgs = [bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in rng.integers(0, 136, size=20)]
    states = np.stack([one_block_state(m) for m in msgs])
    d_states = torch.from_numpy(states.view(np.uint32).reshape(-1).view(np.int32)).to(zk.device)
    log_h = 9
    d_tr = zk.keccak_f_tracegen(d_states, len(msgs), log_h)
    got = zk.download(d_tr).reshape(2633, -1)
    assert (got == ora_trace(ora, states, log_h)).all()
    for p, m in enumerate(msgs):
        row = 24 * p + 23
        lanes = [[int(got[2629 + k, row]) for k in range(4)]] + [[int(got[2465 + 4 * x + k, row]) for k in range(4)] for x in (1, 2, 3)]
        assert b"".join(sum(v << (16 * i) for i, v in enumerate(l)).to_bytes(8, "little") for l in lanes).hex() == hashlib.sha3_256(m).hexdigest()
    program, width = z.keccak_f_air()
    inst = [dict(program=program, log_height=log_h, width=width, n_pvs=0, trace=got, pvs=NOPV)]
    pk = z.ProvingKey(zk, PARAMS, inst)
    proof = pk.prove([d_tr], [NOPV])
    assert z.verify(PARAMS, inst, [NOPV], proof) == 0
    assert proof == ora.stark_prove(PARAMS, inst).tobytes()
    # a wrong round output does not prove
    bad = d_tr.clone()
    bad[2465 * (1 << log_h) + 30] ^= 1
    assert z.verify(PARAMS, inst, [NOPV], pk.prove([bad], [NOPV])) != 0
    pk.close()


def test_keccak_chip_throughput(zk):
    """2^16 rows = 2730 permutations: trace generation and proof at the reference's parameters (reported, not asserted on time)."""
    import time

    log_h, n = 16, (1 << 16) // 24
    rng = np.random.default_rng(1)
    states = rng.integers(0, 1 << 32, size=(n, 50), dtype=np.uint32)
    d_states = torch.from_numpy(states.reshape(-1).view(np.int32)).to(zk.device)
    program, width = z.keccak_f_air()
    inst = [dict(program=program, log_height=log_h, width=width, n_pvs=0)]
    pk = z.ProvingKey(zk, z.DEFAULT_PARAMS, inst)
    d_tr = zk.keccak_f_tracegen(d_states, n, log_h)
    proof = pk.prove([d_tr], [NOPV])
    assert z.verify(z.DEFAULT_PARAMS, inst, [NOPV], proof) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d_tr = zk.keccak_f_tracegen(d_states, n, log_h)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pk.prove([d_tr], [NOPV])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("keccak chip: %d permutations, %.1f M cells: tracegen %.2f ms, proof %.1f ms" % (n, 2633 * (1 << log_h) / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    pk.close()
