"""CPU: the product's host verifier (zkhip_verify, Montgomery C++) against proofs produced by the
oracle prover (canonical C) -- two independent implementations of the same protocol -- plus
rejection of tampered proofs.  Needs no GPU."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

PARAMS = (1, 0, 12, 5, 6)


def _case_fib(log_n=5):
    fa = air.fibonacci_air()
    tr, pv = air.fibonacci_trace(log_n)
    return [dict(program=fa.program(), log_height=log_n, width=2, n_pvs=3, trace=tr, pvs=pv)]


def _case_multi():
    sa = air.SyntheticAir(width=40, n_free=10, n_bool=4, n_boundary=3, seed=1)
    tr, pv = sa.gen_trace(7, seed=3)
    sb = air.SyntheticAir(width=12, n_free=5, n_bool=2, n_boundary=2, seed=2)
    tr2, pv2 = sb.gen_trace(4, seed=4)
    return ([dict(program=sa.program(), log_height=7, width=40, n_pvs=len(pv), trace=tr, pvs=pv)] + _case_fib(6) +
            [dict(program=sb.program(), log_height=4, width=12, n_pvs=len(pv2), trace=tr2, pvs=pv2)])


@pytest.mark.parametrize("case", [_case_fib, _case_multi])
def test_product_verifier_accepts_oracle_proofs(ora, case):
    airs = case()
    for a in airs:
        assert air.check_trace(a["program"], a["trace"], a["pvs"]) == []
    proof = ora.stark_prove(PARAMS, airs)
    assert ora.stark_verify(PARAMS, airs, proof) == 0
    pvs = [a["pvs"] for a in airs]
    assert z.verify(PARAMS, airs, pvs, proof.tobytes()) == 0
    # every region of the proof is bound: flip one word at a spread of positions
    rng = np.random.default_rng(0)
    for pos in sorted(set([1, 5, 13, 21, len(proof) - 1] + rng.integers(0, len(proof), 40).tolist())):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % ora.P
        assert z.verify(PARAMS, airs, pvs, bad.tobytes()) != 0, "tampered word %d accepted" % pos
        assert ora.stark_verify(PARAMS, airs, bad) != 0
    # wrong public value, wrong parameters, truncated proof
    pv_bad = [p.copy() for p in pvs]
    pv_bad[0][0] = (int(pv_bad[0][0]) + 1) % ora.P
    assert z.verify(PARAMS, airs, pv_bad, proof.tobytes()) != 0
    assert z.verify((1, 0, 12, 5, 7), airs, pvs, proof.tobytes()) != 0
    assert z.verify(PARAMS, airs, pvs, proof[:-1].tobytes()) != 0


def test_unsatisfied_trace_is_rejected(ora):
    airs = _case_fib(5)
    airs[0]["trace"] = airs[0]["trace"].copy()
    airs[0]["trace"][1, 7] = (int(airs[0]["trace"][1, 7]) + 1) % ora.P
    assert air.check_trace(airs[0]["program"], airs[0]["trace"], airs[0]["pvs"]) != []
    proof = ora.stark_prove(PARAMS, airs)
    assert ora.stark_verify(PARAMS, airs, proof) != 0
    assert z.verify(PARAMS, airs, [a["pvs"] for a in airs], proof.tobytes()) != 0


def test_degree_bound_enforced():
    b = air.AirBuilder(2, 0)
    x = b.var(0)
    b.assert_zero(x * x * x * x - b.var(1))
    airs = [dict(program=b.program(), log_height=3, width=2, n_pvs=0)]
    assert z.verify(PARAMS, airs, [np.zeros(0, np.uint32)], np.zeros(64, np.uint32).tobytes()) == -8
