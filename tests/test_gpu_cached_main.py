"""GPU: cached main partitions -- the HIP prover commits the leading `cached_width` columns of an AIR in a tree of their own
(OpenVM-v1 cached main; the reference's stored proofs have one), proof bytes equal the oracle's for small and four-step
heights, with LogUp buses and preprocessed tables in the same proof, interpreter and compiled constraint kernels alike."""
import os

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

pytestmark = pytest.mark.gpu
NOPV = np.zeros(0, np.uint32)


def _case(lh_big, lh_small):
    sa = air.SyntheticAir(width=20, n_free=7, n_bool=3, n_boundary=2, seed=3)
    sa.builder.cached_width = 9          # the reference's cached partition is 9 columns wide (the program ROM)
    tr, pv = sa.gen_trace(lh_big, seed=1)
    sb = air.SyntheticAir(width=12, n_free=5, n_bool=2, n_boundary=2, seed=8)   # same height, no cached part
    trb, pvb = sb.gen_trace(lh_big, seed=2)
    ftr, fpv = air.fibonacci_trace(lh_small)
    fb = air.fibonacci_air()
    fb.cached_width = 1
    s, t = air.lookup_traces(lh_big - 1, 6, seed=1)
    lb = air.lookup_sender_air()
    lb.cached_width = 2
    u, m, prep = air.range_traces(lh_small, 4, seed=2)
    return [dict(program=sa.program(), log_height=lh_big, width=20, n_pvs=len(pv), trace=tr, pvs=pv),
            dict(program=fb.program(), log_height=lh_small, width=2, n_pvs=3, trace=ftr, pvs=fpv),
            dict(program=sb.program(), log_height=lh_big, width=12, n_pvs=len(pvb), trace=trb, pvs=pvb),
            dict(program=lb.program(), log_height=lh_big - 1, width=3, n_pvs=0, trace=s, pvs=NOPV),
            dict(program=air.lookup_table_air().program(), log_height=6, width=3, n_pvs=0, trace=t, pvs=NOPV),
            dict(program=air.range_user_air().program(), log_height=lh_small, width=4, n_pvs=0, trace=u, pvs=NOPV),
            dict(program=air.range_table_air().program(), log_height=4, width=1, n_pvs=0, trace=m, pvs=NOPV, prep=prep)]


@pytest.mark.parametrize("lh_big,lh_small,params", [(7, 5, (1, 0, 10, 3, 4)), (13, 9, (1, 0, 20, 6, 6)), (14, 12, (2, 1, 12, 4, 4))])
@pytest.mark.parametrize("force_jit", [False, True])
def test_cached_partitions_match_the_oracle(zk, ora, lh_big, lh_small, params, force_jit):
    airs = _case(lh_big, lh_small)
    pvs = [a["pvs"] for a in airs]
    if force_jit:
        zk.set_config(jit=2)   # (restored after the test: tests/conftest.py)
    pk = z.ProvingKey(zk, params, airs)
    d = [zk.upload(a["trace"].reshape(-1)) for a in airs]
    proof = pk.prove(d, pvs)
    assert np.frombuffer(proof[:4], np.uint32)[0] == 0x31504B5A + 7
    assert proof == ora.stark_prove(params, airs, cap_words=len(proof) // 4 + 16).tobytes()
    vk = pk.verifying_airs()
    assert z.verify(params, vk, pvs, proof) == 0
    lay = z.proof_layout(params, vk)
    assert lay["n_cached"] == 3 and lay["n_words"] * 4 == len(proof) == pk.proof_size
    # the pipelined commit must stay out of the way when cached partitions exist
    zk.set_commit_pipeline(4)
    try:
        assert pk.prove(d, pvs) == proof
    finally:
        zk.set_commit_pipeline(0)
    pk.close()
