"""An AIR set with the dimensions of the chunk proof the reference stores (air.ReferenceShapedSet: 17 AIRs, cached main, two
preprocessed traces, the reference's after-challenge widths), heights shrunk for the CPU: the oracle proves it, the host
verifier accepts it, and converted into the reference's v1 container the proof has the reference proof's own STRUCTURE --
every vector length a decoder meets, batch by batch -- up to the heights."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refproof_v1 as rp  # noqa: E402

import zkvm_prover_amd as z  # noqa: E402
from zkvm_prover_amd import air  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_v1_vectors.json")
PARAMS = (2, 0, 44, 0, 16)  # blow-up 4, 44 queries: what the stored proof was made with
SHRINK = 10


@pytest.fixture(scope="module")
def ora():
    import oracle_lib

    return oracle_lib


def ref_shape(name="chunk-proof-feynman.json"):
    with open(GOLDEN) as f:
        return json.load(f)["shapes"][name]


def verifying(ora, airs, params=PARAMS):
    vk = []
    for a in airs:
        v = {k: a[k] for k in ("program", "log_height", "width", "n_pvs")}
        if a.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(params, a)
        vk.append(v)
    return vk


def check_against_reference_shape(shape, shrink, name="chunk-proof-feynman.json"):
    ref = ref_shape(name)
    assert shape["n_airs"] == ref["n_airs"] == 17
    assert shape["log_degrees"] == [1 if i == 1 else max(1, d - shrink) for i, d in enumerate(ref["log_degrees"])]
    assert shape["n_queries"] == ref["n_queries"]
    # the two `feynman` proofs end FRI with one coefficient, like this prover with log_final_poly_len = 0; the six older ones
    # (42 queries) carry four final values after the same number of layers -- an earlier convention of the engine's FRI
    assert shape["n_final_poly"] == 1 and ref["n_final_poly"] == (1 if "feynman" in name else 4)
    assert shape["n_fri_layers"] == ref["n_fri_layers"] - shrink
    got, exp = shape["batches"], ref["batches"]
    assert len(got) == len(exp) == 6
    # preprocessed (2 trees), cached main, common main, after-challenge: the same matrices of the same widths in the same order
    for g, e in zip(got[:5], exp[:5]):
        assert g["widths"] == e["widths"]
    # quotient: 62 width-4 chunk matrices (next_pow2(max_degree - 1) per AIR: 1 + 15 x 4 + 1)
    assert got[5]["widths"] == exp[5]["widths"] == [4] * 62


def test_reference_shaped_set_oracle_to_v1(ora):
    rs = air.ReferenceShapedSet(shrink=SHRINK)
    airs = rs.gen()
    assert [a["width"] for a in airs] == [10] + air.ReferenceShapedSet.MAIN_WIDTHS[1:]
    pvs = [a["pvs"] for a in airs]
    proof = ora.stark_prove(PARAMS, airs).tobytes()
    vk = verifying(ora, airs)
    assert z.verify(PARAMS, vk, pvs, proof) == 0
    v1 = z.proof_to_v1(PARAMS, vk, pvs, proof)
    p = rp.decode_proofs((1).to_bytes(8, "little") + v1)[0]
    assert rp.encode_proofs([p])[8:] == v1
    check_against_reference_shape(rp.shape_of(p), SHRINK)
    assert [len(q) for q in p["opened"]["quotient"]] == [1] + [4] * 15 + [1]   # chunks per AIR, as in the stored proof
    assert len(p["main_trace"]) == 2 and len(p["opened"]["preprocessed"]) == 2   # [cached, common]; two preprocessed AIRs
    back, pvs_back = z.proof_from_v1(PARAMS, vk, v1)
    assert back == proof


@pytest.mark.parametrize("name", ["batch-proof-feynman.json", "prover-chunk-proof.json", "chunk-proof-phase2.json"])
def test_other_stored_proofs_have_the_same_structure(ora, name):
    """The eight stored proofs are proofs of one circuit at different heights: the same AIR set at the heights of three more of
    them gives their structure too (number of queries and FRI layers included; see check_against_reference_shape for the final
    polynomial of the six older ones)."""
    ref = ref_shape(name)
    params = (2, 0, ref["n_queries"], 0, 16)
    airs = air.ReferenceShapedSet(shrink=SHRINK, log_degrees=ref["log_degrees"]).gen()
    pvs = [a["pvs"] for a in airs]
    proof = ora.stark_prove(params, airs).tobytes()
    vk = verifying(ora, airs, params)
    assert z.verify(params, vk, pvs, proof) == 0
    p = rp.decode_proofs((1).to_bytes(8, "little") + z.proof_to_v1(params, vk, pvs, proof))[0]
    check_against_reference_shape(rp.shape_of(p), SHRINK, name)
