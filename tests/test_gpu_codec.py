"""GPU: a proof made by the HIP prover is re-encoded into the reference's v1 `Proof<SC>` container
(zkhip_proof_to_v1), read by the independent Python reader, decoded back (zkhip_proof_from_v1) to the identical
bytes, and verifies."""
import numpy as np
import pytest

import refproof_v1 as rp
import zkvm_prover_amd as z
from zkvm_prover_amd import air

pytestmark = pytest.mark.gpu
NOPV = np.zeros(0, np.uint32)


@pytest.mark.parametrize("params", [(1, 0, 20, 0, 8), (2, 0, 44, 0, 10)])
def test_hip_proof_into_v1_container_and_back(zk, params):
    sa = air.SyntheticAir(width=24, n_free=8, n_bool=2, n_boundary=2, seed=5)
    tr, pv = sa.gen_trace(12, seed=3)
    s, t = air.lookup_traces(10, 6, seed=2)
    u, m, prep = air.range_traces(9, 5, seed=4)
    airs = [dict(program=sa.program(), log_height=12, width=24, n_pvs=len(pv), trace=tr, pvs=pv),
            dict(program=air.lookup_sender_air().program(), log_height=10, width=3, n_pvs=0, trace=s, pvs=NOPV),
            dict(program=air.lookup_table_air().program(), log_height=6, width=3, n_pvs=0, trace=t, pvs=NOPV),
            dict(program=air.range_user_air().program(), log_height=9, width=4, n_pvs=0, trace=u, pvs=NOPV),
            dict(program=air.range_table_air().program(), log_height=5, width=1, n_pvs=0, trace=m, pvs=NOPV, prep=prep)]
    pk = z.ProvingKey(zk, params, airs)
    d = [zk.upload(a["trace"].reshape(-1)) for a in airs]
    pvs = [a["pvs"] for a in airs]
    proof = pk.prove(d, pvs)
    vk = pk.verifying_airs()
    assert z.verify(params, vk, pvs, proof) == 0
    v1 = z.proof_to_v1(params, vk, pvs, proof)
    p = rp.decode_proofs((1).to_bytes(8, "little") + v1)[0]
    sh = rp.shape_of(p)
    assert sh["n_airs"] == 5 and sh["n_queries"] == params[2] and sh["log_degrees"] == [12, 10, 6, 9, 5]
    # batches in the reference's order: preprocessed tree, main, after-challenge (LogUp permutation), quotient chunks
    n_chunks = [air.quotient_chunks(a["program"]) for a in airs]   # next_pow2(max(degree, 2) - 1) per AIR
    assert [len(q) for q in p["opened"]["quotient"]] == n_chunks and max(n_chunks) <= (1 << params[0])
    assert [len(b["widths"]) for b in sh["batches"]] == [1, 5, 4, sum(n_chunks)]
    assert sh["batches"][1]["log_height"] == 12 + params[0]
    s1 = z.proof_decode_v1(v1, z.V1_SINGLE)
    assert s1["log_blowup"] == params[0] and s1["n_preprocessed"] == 1 and s1["n_after_challenge_commits"] == 1
    back, pvs_back = z.proof_from_v1(params, vk, v1)
    assert back == proof
    assert z.verify(params, vk, pvs_back, back) == 0
    pk.close()
