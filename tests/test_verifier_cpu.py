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


def test_proof_layout_matches_the_proofs(ora):
    """zkhip_proof_layout_of: the static field offsets agree with oracle proofs of AIR sets with every feature
    (mixed heights, preprocessed tree, LogUp phase) -- size, magic-dependent sections, commitments, exposed sums."""
    import zkvm_prover_amd as z
    from zkvm_prover_amd import air
    from test_prep_cpu import range_case, verifying_airs, PARAMS as PP

    nopv = np.zeros(0, np.uint32)
    ftr, fpv = air.fibonacci_trace(6)
    fib = dict(program=air.fibonacci_air().program(), log_height=6, width=2, n_pvs=3, trace=ftr, pvs=fpv)
    tr, pv = air.bus_mix_trace(4, seed=1)
    mix = dict(program=air.bus_mix_air().program(), log_height=4, width=6, n_pvs=1, trace=tr, pvs=pv)
    for params, airs in [((1, 0, 7, 2, 3), [fib]), ((2, 0, 5, 1, 1), [mix, fib]), (PP, range_case())]:
        proof = ora.stark_prove(params, airs)
        lay = z.proof_layout(params, airs)
        assert lay["n_words"] == len(proof)
        assert lay["queries"] + lay["n_queries"] * lay["query_words"] == len(proof) and lay["n_queries"] == params[2]
        assert lay["root_main"] == 4 and lay["n_fri_layers"] == int(proof[3])
        has_lu = any(air.LOGUP_MAGIC in [int(w) for w in a["program"]] for a in airs)
        assert (lay["root_perm"] != 0) == has_lu == bool((int(proof[0]) - 0x31504B5A) & 1)
        if has_lu:
            # the exposed sums of all AIRs with interactions cancel (each is an extension element)
            ex = proof[lay["exposed"]: lay["exposed"] + 4 * lay["n_exposed"]].astype(np.int64).reshape(-1, 4)
            assert (ex.sum(axis=0) % ora.P == 0).all()
        # tampering with the word just before / at the located fields is caught by the verifier
        vk = verifying_airs(ora, airs) if any(a.get("prep") is not None for a in airs) else airs
        pvs = [a["pvs"] for a in airs]
        assert z.verify(params, vk, pvs, proof.tobytes()) == 0
        for field in ("root_main", "root_quot", "opened", "fri_layers", "final_poly", "query_pow", "queries"):
            bad = proof.copy()
            bad[lay[field]] = (int(bad[lay[field]]) + 1) % ora.P
            assert z.verify(params, vk, pvs, bad.tobytes()) != 0, field


@pytest.mark.parametrize("lfp", [1, 2, 3, 4])
def test_final_polynomial_of_positive_length(ora, lfp):
    """log_final_poly_len > 0: the fold loop stops at 2^(b+lfp) values and the proof carries 2^lfp coefficients.  The
    oracle's proofs are accepted by the product's host verifier and by the Python verifier, shrink with lfp, and every
    coefficient is bound (tampering is rejected); an AIR shorter than the final polynomial is refused."""
    import zkvm_prover_amd as z
    from zkvm_prover_amd import air
    import pymodel_verify

    ftr, fpv = air.fibonacci_trace(6)
    fib = dict(program=air.fibonacci_air().program(), log_height=6, width=2, n_pvs=3, trace=ftr, pvs=fpv)
    tr, pv = air.bus_mix_trace(4, seed=1)
    mix = dict(program=air.bus_mix_air().program(), log_height=4, width=6, n_pvs=1, trace=tr, pvs=pv)
    airs, pvs = [mix, fib], [pv, fpv]
    params, params0 = (1, lfp, 6, 2, 3), (1, 0, 6, 2, 3)
    proof = ora.stark_prove(params, airs)
    assert 0 < len(proof) < len(ora.stark_prove(params0, airs))
    assert ora.stark_verify(params, airs, proof) == 0
    assert z.verify(params, airs, pvs, proof.tobytes()) == 0
    assert pymodel_verify.verify(params, airs, proof) is True
    lay = z.proof_layout(params, airs)
    assert lay["n_words"] == len(proof) and lay["n_final_poly"] == 1 << lfp and int(proof[3]) == lay["n_fri_layers"] == 6 + 1 - 1 - lfp
    for j in range(1 << lfp):
        bad = proof.copy()
        w = lay["final_poly"] + 4 * j + (j % 4)
        bad[w] = (int(bad[w]) + 1) % ora.P
        assert ora.stark_verify(params, airs, bad) != 0 and z.verify(params, airs, pvs, bad.tobytes()) != 0
    # a proof for one lfp does not verify under another
    assert z.verify(params0, airs, pvs, proof.tobytes()) != 0
    # a trace shorter than the final polynomial cannot be part of the set
    tiny = dict(program=air.fibonacci_air().program(), log_height=lfp - 1, width=2, n_pvs=3)
    tt, tp = air.fibonacci_trace(lfp - 1) if lfp > 1 else (np.array([[0], [1]], np.uint32), np.array([0, 1, 1], np.uint32))
    tiny.update(trace=tt, pvs=tp)
    with pytest.raises(RuntimeError):
        ora.stark_prove(params, airs + [tiny])
    assert z.verify(params, airs + [tiny], pvs + [tp], proof.tobytes()) != 0


def test_logup_bus_row_bound_and_parameter_validation():
    """multiplicities live in characteristic p, so a bus must carry fewer than p interaction rows in total
    (prover-chosen heights!); and the verifier validates parameters like keygen does."""
    import zkvm_prover_amd as z
    from zkvm_prover_amd import air

    b = air.AirBuilder(2, 0)
    for _ in range(16):
        b.push_interaction(3, [b.var(0)], b.var(1), "send")
        b.push_interaction(3, [b.var(0)], b.var(1), "receive")
    b.finalize_interactions()
    prog = b.program()
    dummy = np.zeros(64, dtype=np.uint32).tobytes()
    params = (1, 0, 4, 0, 0)
    # 32 interactions x 2^26 rows = 2^31 >= p: refused outright, whatever the proof says
    a = [dict(program=prog, log_height=26, width=2, n_pvs=0)]
    assert z.verify(params, a, [np.zeros(0, np.uint32)], dummy) == -3
    # 2^25 rows: below the bound; the (garbage) proof is then rejected as a proof
    a = [dict(program=prog, log_height=25, width=2, n_pvs=0)]
    assert z.verify(params, a, [np.zeros(0, np.uint32)], dummy) == -7
    # the same total spread over two AIRs on the same bus is also refused
    a = [dict(program=prog, log_height=25, width=2, n_pvs=0)] * 2
    assert z.verify(params, a, [np.zeros(0, np.uint32)] * 2, dummy) == -3
    fib = [dict(program=air.fibonacci_air().program(), log_height=4, width=2, n_pvs=3)]
    pv = [np.array([0, 1, 1], dtype=np.uint32)]
    for bad in [(1, 0, 0, 0, 0), (1, 0, 4, 31, 0), (1, 0, 4, 0, 31), (5, 0, 4, 0, 0), (0, 0, 4, 0, 0)]:
        assert z.verify(bad, fib, pv, dummy) == -3, bad
    tall = [dict(program=air.fibonacci_air().program(), log_height=27, width=2, n_pvs=3)]
    assert z.verify((1, 0, 4, 0, 0), tall, pv, dummy) == -3   # log_height + log_blowup > 27
    with pytest.raises(z.ZkhipError):
        z.proof_layout((1, 0, 4, 0, 0), tall)
