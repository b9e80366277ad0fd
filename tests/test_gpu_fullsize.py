"""GPU: BIT-EXACT parity at BASELINE.json's full sizes (cfg #2 NTT, cfg #3 Merkle commit, cfg #4 whole proof).

The checkers are the plain oracle (oracle/ntt.c, oracle/merkle.c, all host cores) for the stage-level configs, and for
the whole 2^22 x 300 proof the optimised CPU prover oracle/fast -- itself bit-exact against oracle/stark.c on every
proof of tests/test_fast_oracle_cpu.py and, below, three ways (HIP == fast == plain oracle) on a 2^17 x 300 instance of
the same workload.  Inputs are generated on the device (seeded) and downloaded for the CPU side."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

pytestmark = pytest.mark.gpu
P = 2013265921


def _device_random(zk, n, seed):
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    t = torch.randint(0, P, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    host = t.cpu().numpy().view(np.uint32).copy()          # canonical values
    zk._check(zk.lib.zkhip_to_monty(zk.h, t.data_ptr(), t.numel()))
    return t, host


@pytest.mark.parametrize("log_n,width", [(22, 16), (24, 4), (20, 8)])
def test_cfg2_ntt_full_size_bit_exact(zk, ora, log_n, width):
    """cfg #2: natural-order forward DFT of every column, vs oracle/ntt.c ora_dft_batch (p3-dft's definition)."""
    d, host = _device_random(zk, width << log_n, 1000 + log_n)
    zk.ntt_batch(d, log_n, width)
    got = zk.download(d).reshape(width, -1)
    exp = ora.dft_batch(host.reshape(width, -1), log_n)
    assert (got == exp).all()
    del d
    # and the inverse brings the input back
    d2 = zk.upload(exp.reshape(-1))
    zk.ntt_batch(d2, log_n, width, inverse=True)
    assert (zk.download(d2) == host).all()


def test_cfg3_merkle_commit_full_size_bit_exact(zk, ora):
    """cfg #3: Poseidon2 Merkle commit of a 2^22 x 300 matrix: root AND every digest layer sampled, vs oracle/merkle.c."""
    log_n, width = 22, 300
    d, host = _device_random(zk, width << log_n, 4242)
    tree = zk.merkle_commit([(d, log_n, width)])
    ref = ora.Tree([host.reshape(width, -1)])
    assert tree.root.tolist() == ref.root.tolist()
    for layer in (0, 1, 7, 15, log_n - 1):
        a, b = tree.layer(layer), ref.layer(layer)
        assert (a == b).all(), layer
    idx = [0, 1, 12345, (1 << log_n) - 1]
    op = tree.open(idx)
    for k, i in enumerate(idx):
        assert ref.verify(i, op[k])
        assert (op[k] == ref.open(i)).all()


def _workload(log_n, width=300):
    sa = air.SyntheticAir(width=width, n_free=max(4, width // 5), n_bool=min(16, max(1, width // 20)),
                          n_boundary=min(8, max(1, width // 40)), seed=0)
    fa = air.fibonacci_air()
    airs = [dict(program=sa.program(), log_height=log_n, width=width, n_pvs=sa.n_pvs),
            dict(program=fa.program(), log_height=log_n, width=2, n_pvs=3)]
    return sa, airs


def _prove_on_device(zk, sa, airs, log_n, params, seed):
    tr, pv = sa.gen_trace(log_n, seed=seed, xp="torch", device="cuda")
    d = tr.reshape(-1).contiguous()
    del tr
    zk._check(zk.lib.zkhip_to_monty(zk.h, d.data_ptr(), d.numel()))
    ftr, fpv = air.fibonacci_trace(log_n, a0=3, b0=5)
    df = zk.upload(ftr.reshape(-1))
    pk = z.ProvingKey(zk, params, airs)
    proof = pk.prove([d, df], [pv, fpv])
    host = [zk.download(d).reshape(airs[0]["width"], -1), ftr]
    pk.close()
    return proof, host, [np.asarray(pv, dtype=np.uint32), np.asarray(fpv, dtype=np.uint32)]


def test_three_way_parity_mid_size(zk, ora):
    """The bench workload at 2^17 rows: HIP prover == oracle/fast == oracle/stark.c, byte for byte."""
    params = z.DEFAULT_PARAMS
    sa, airs = _workload(17)
    proof, host, pvs = _prove_on_device(zk, sa, airs, 17, params, seed=5)
    inst = [dict(a, trace=t, pvs=pv) for a, t, pv in zip(airs, host, pvs)]
    fast = ora.fast_stark_prove(params, inst, cap_words=len(proof) // 4 + 16).tobytes()
    assert fast == proof
    plain = ora.stark_prove(params, inst, cap_words=len(proof) // 4 + 16).tobytes()
    assert plain == proof


def test_cfg4_full_chunk_proof_bit_exact(zk, ora):
    """cfg #4 at full size: 2^22 x 300 (+ 2^22 x 2), reference FRI parameters (100 queries, PoW 16 + 16): the HIP
    prover's 1,128,348 proof bytes equal the CPU prover's, and the host verifier accepts them."""
    params = z.DEFAULT_PARAMS
    sa, airs = _workload(22)
    proof, host, pvs = _prove_on_device(zk, sa, airs, 22, params, seed=1000)
    assert len(proof) == 1128348
    import torch

    torch.cuda.empty_cache()
    inst = [dict(a, trace=t, pvs=pv) for a, t, pv in zip(airs, host, pvs)]
    cpu = ora.fast_stark_prove(params, inst, cap_words=len(proof) // 4 + 16).tobytes()
    assert cpu == proof
    assert z.verify(params, airs, pvs, proof) == 0
