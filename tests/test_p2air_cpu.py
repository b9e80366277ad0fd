"""CPU: the Poseidon2 AIR (one permutation per row, p3-poseidon2-air shape with one S-box register).  The oracle's
trace generator satisfies the constraint program row by row, its output columns are the permutation, any flipped
cell breaks a constraint, and oracle proofs of the AIR are accepted by the product's host verifier and by the
independent Python verifier."""
import numpy as np

import zkvm_prover_amd as z
from zkvm_prover_amd import air
import oracle_lib as O
import pymodel_verify

PARAMS = (1, 0, 10, 4, 5)
NOPV = np.zeros(0, np.uint32)


def _case(log_n, n_perms, seed=0):
    rng = np.random.default_rng(seed)
    inputs = O.rand_field(rng, (n_perms, 16))
    return inputs, O.poseidon2_air_trace(inputs, log_n)


def test_round_constants_and_shape():
    rc = air.poseidon2_round_constants()
    assert len(rc) == 141 and rc == np.ctypeslib.as_array(O.lib().ora_poseidon2_round_constants(), shape=(141,)).tolist()
    b = air.poseidon2_air()
    b.program()
    assert b.width == air.POSEIDON2_AIR_WIDTH == 298 and len(b.cons) == 282 and b.max_degree() == 3


def test_oracle_trace_satisfies_the_air_and_outputs_the_permutation():
    prog = air.poseidon2_air().program()
    for log_n, n in [(0, 1), (3, 8), (4, 11), (5, 0)]:
        inputs, tr = _case(log_n, n, seed=log_n)
        assert air.check_trace(prog, tr, NOPV) == []
        for r in range(1 << log_n):
            s = inputs[r].copy() if r < n else np.zeros(16, np.uint32)
            assert (tr[:16, r] == s).all()
            assert (tr[-16:, r] == O.permute(s)).all()


def test_every_column_is_constrained():
    prog = air.poseidon2_air().program()
    _, tr = _case(2, 3, seed=9)
    for col in range(16, 298):  # inputs are free; every other cell is pinned by a constraint
        bad = tr.copy()
        bad[col, 1] = (int(bad[col, 1]) + 1) % air.P
        assert air.check_trace(prog, bad, NOPV) != [], "column %d unconstrained" % col


def test_oracle_proofs_verify_everywhere(ora):
    prog = air.poseidon2_air().program()
    _, tr = _case(4, 13, seed=2)
    ftr, fpv = air.fibonacci_trace(6)
    airs = [dict(program=prog, log_height=4, width=298, n_pvs=0, trace=tr, pvs=NOPV),
            dict(program=air.fibonacci_air().program(), log_height=6, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
    pvs = [a["pvs"] for a in airs]
    proof = ora.stark_prove(PARAMS, airs)
    assert ora.stark_verify(PARAMS, airs, proof) == 0
    assert z.verify(PARAMS, airs, pvs, proof.tobytes()) == 0
    assert pymodel_verify.verify(PARAMS, airs, proof) is True
    # a wrong S-box register anywhere makes the (self-consistent) proof unverifiable
    bad = [dict(a) for a in airs]
    bad[0]["trace"] = tr.copy()
    bad[0]["trace"][200, 7] ^= 1
    p2 = ora.stark_prove(PARAMS, bad)
    assert ora.stark_verify(PARAMS, bad, p2) != 0 and z.verify(PARAMS, airs, pvs, p2.tobytes()) != 0


def hasher_pair(log_user=4, n_req=11, log_chip=4, seed=3, bus=9):
    """A requester chip and the Poseidon2 chip on one bus with 24-field messages; one request is made twice
    (multiplicity 2 on the chip's side)."""
    rng = np.random.default_rng(seed)
    left, right = O.rand_field(rng, (n_req, 8)), O.rand_field(rng, (n_req, 8))
    left[1], right[1] = left[0], right[0]
    states = np.concatenate([left, right], axis=1)
    out = np.stack([O.permute(s.copy())[:8] for s in states])
    user = air.hasher_user_trace(log_user, left, right, out)
    uniq = states[1:]                                   # request 0 == request 1: served by one row
    chip = np.zeros((299, 1 << log_chip), np.uint32)
    chip[:298] = O.poseidon2_air_trace(uniq, log_chip)
    chip[298, :len(uniq)] = 1
    chip[298, 0] = 2
    return [dict(program=air.hasher_user_air(bus).program(), log_height=log_user, width=25, n_pvs=0, trace=user, pvs=NOPV),
            dict(program=air.poseidon2_air(bus).program(), log_height=log_chip, width=299, n_pvs=0, trace=chip, pvs=NOPV)], uniq


def test_poseidon2_chip_serves_a_requester_over_a_24_field_bus(ora):
    airs, _ = hasher_pair()
    pvs = [NOPV, NOPV]
    for a in airs:
        assert air.check_trace(a["program"], a["trace"], a["pvs"]) == []
    proof = ora.stark_prove(PARAMS, airs)
    assert ora.stark_verify(PARAMS, airs, proof) == 0
    assert z.verify(PARAMS, airs, pvs, proof.tobytes()) == 0
    assert pymodel_verify.verify(PARAMS, airs, proof) is True
    # a requester claiming a wrong digest word: every AIR is satisfied row by row, but the bus does not balance
    bad = [dict(a) for a in airs]
    bad[0]["trace"] = airs[0]["trace"].copy()
    bad[0]["trace"][19, 2] ^= 1
    assert air.check_trace(bad[0]["program"], bad[0]["trace"], NOPV) == []
    p2 = ora.stark_prove(PARAMS, bad)
    assert ora.stark_verify(PARAMS, bad, p2) != 0 and z.verify(PARAMS, airs, pvs, p2.tobytes()) != 0
    # so does a multiplicity that is off by one
    bad = [dict(a) for a in airs]
    bad[1]["trace"] = airs[1]["trace"].copy()
    bad[1]["trace"][298, 0] = 1
    p3 = ora.stark_prove(PARAMS, bad)
    assert ora.stark_verify(PARAMS, bad, p3) != 0 and z.verify(PARAMS, airs, pvs, p3.tobytes()) != 0


def test_range_counts_oracle_matches_numpy_and_balances_the_range_bus(ora):
    rng = np.random.default_rng(4)
    vals = rng.integers(0, 1 << 6, 500).astype(np.uint32)
    c, bad = O.range_counts(vals, 6)
    assert bad == 0 and (c == np.bincount(vals, minlength=64)).all()
    c2, bad = O.range_counts(np.array([3, 64, 3, 2**31 - 5], np.uint32), 6, counts=c)
    assert bad == 2 and c2[3] == c[3] + 2 and c2.sum() == c.sum() + 2
    # the table chip's trace of air.range_traces is exactly this histogram of the user's column
    u, m, prep = air.range_traces(6, 4, seed=2)
    c3, bad = O.range_counts(u[0], 4)
    assert bad == 0 and (c3 == m[0]).all()
