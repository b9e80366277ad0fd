"""CPU: the pure-Python verifier (tests/pymodel_verify.py, written from DESIGN.md section 4 on top of the big-int
model) accepts the oracle's proofs for every protocol feature and rejects tampered ones: a third implementation of
verification next to oracle/stark.c and csrc/verifier.hip."""
import importlib.util
import os

import numpy as np
import pytest

import pymodel_verify as pv

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("gen_proof_digests", os.path.join(HERE, "golden", "gen_proof_digests.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


def _vk(ora, params, airs):
    out = []
    for a in airs:
        v = {k: a[k] for k in ("program", "log_height", "width", "n_pvs", "pvs")}
        if a.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(params, a)
        out.append(v)
    return out


@pytest.mark.parametrize("name", ["fib_5", "fib_min", "mixed_heights", "blowup_4", "lookup", "bus_mix", "compound_messages",
                                  "range_table_prep", "chipset_6"])
def test_python_verifier_accepts_oracle_proofs(ora, name):
    params, airs = gen.cases()[name]
    params = (params[0], 0, 3, 2, 2)  # few queries: the Python permutation costs 0.4 ms
    proof = ora.stark_prove(params, airs)
    vk = _vk(ora, params, airs)
    assert pv.verify(params, vk, proof) is True
    # tampering is rejected (header, roots / exposed sums, opened values, FRI, queries)
    rng = np.random.default_rng(1)
    for pos in [1, 5, 14, 30, len(proof) // 2, len(proof) - 1] + rng.integers(0, len(proof), 4).tolist():
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % pv.P
        with pytest.raises(pv.Reject):
            pv.verify(params, vk, bad)


def test_python_verifier_needs_the_right_key(ora):
    params, airs = gen.cases()["range_table_prep"]
    params = (1, 0, 3, 2, 2)
    proof = ora.stark_prove(params, airs)
    vk = _vk(ora, params, airs)
    vk[-1]["prep_commit"] = vk[-1]["prep_commit"].copy()
    vk[-1]["prep_commit"][0] ^= 1
    with pytest.raises(pv.Reject):
        pv.verify(params, vk, proof)
    vk = _vk(ora, params, airs)
    vk[1]["pvs"] = vk[1]["pvs"].copy()
    vk[1]["pvs"][2] = (int(vk[1]["pvs"][2]) + 1) % pv.P
    with pytest.raises(pv.Reject):
        pv.verify(params, vk, proof)
