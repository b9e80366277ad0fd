"""CPU: bus interactions (LogUp after-challenge phase).  The oracle prover (canonical C) produces proofs
for AIR sets that talk over a bus; the oracle verifier and the product's host verifier (Montgomery C++,
no shared code) must both accept them, reject every tampered word, and reject unbalanced buses."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

PARAMS = (1, 0, 12, 5, 6)
NOPV = np.zeros(0, np.uint32)


def lookup_case(log_s=6, log_t=4, seed=1, with_fib=True):
    s, t = air.lookup_traces(log_s, log_t, seed=seed)
    airs = [dict(program=air.lookup_sender_air().program(), log_height=log_s, width=3, n_pvs=0, trace=s, pvs=NOPV)]
    if with_fib:
        ftr, fpv = air.fibonacci_trace(5)
        airs.append(dict(program=air.fibonacci_air().program(), log_height=5, width=2, n_pvs=3, trace=ftr, pvs=fpv))
    airs.append(dict(program=air.lookup_table_air().program(), log_height=log_t, width=3, n_pvs=0, trace=t, pvs=NOPV))
    return airs


def test_logup_program_shape():
    b = air.lookup_sender_air()
    prog = b.program()
    assert b.max_degree() == 2
    assert air.LOGUP_MAGIC in [int(x) for x in prog]
    # 4 coordinate constraints per interaction + 3 x 4 for the running sum, plus the AIR's own one
    assert int(prog[2]) == 1 + 4 + 12
    assert air.check_trace(prog, air.lookup_traces(4, 3)[0], NOPV) == []


@pytest.mark.parametrize("with_fib", [True, False])
def test_logup_proofs_verify_and_bind(ora, with_fib):
    airs = lookup_case(with_fib=with_fib)
    pvs = [a["pvs"] for a in airs]
    proof = ora.stark_prove(PARAMS, airs)
    assert int(proof[0]) == 0x31504B5B
    assert ora.stark_verify(PARAMS, airs, proof) == 0
    assert z.verify(PARAMS, airs, pvs, proof.tobytes()) == 0
    rng = np.random.default_rng(0)
    for pos in sorted(set([1, 5, 13, 21, 22, 29, len(proof) - 1] + rng.integers(0, len(proof), 60).tolist())):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % ora.P
        assert ora.stark_verify(PARAMS, airs, bad) != 0, "oracle accepted tampered word %d" % pos
        assert z.verify(PARAMS, airs, pvs, bad.tobytes()) != 0, "product accepted tampered word %d" % pos
    assert z.verify(PARAMS, airs, pvs, proof[:-1].tobytes()) != 0


def test_logup_compound_messages(ora):
    """Bus messages that are expressions of the row (lo + 256*hi, is_valid * value, gated multiplicities)."""
    b = air.limb_air()
    tr = air.limb_trace(5, seed=3)
    prog = b.program()
    assert b.max_degree() == 3
    assert air.check_trace(prog, tr, NOPV) == []
    airs = lookup_case() + [dict(program=prog, log_height=5, width=4, n_pvs=0, trace=tr, pvs=NOPV)]
    pvs = [a["pvs"] for a in airs]
    proof = ora.stark_prove(PARAMS, airs)
    assert ora.stark_verify(PARAMS, airs, proof) == 0
    assert z.verify(PARAMS, airs, pvs, proof.tobytes()) == 0
    # a wrong limb on a valid row: the AIR's own constraint and the bus both break
    bad = [dict(a) for a in airs]
    t2 = tr.copy()
    row = int(np.nonzero(t2[3])[0][0])
    t2[0, row] = (int(t2[0, row]) + 1) % 256
    bad[-1]["trace"] = t2
    proof = ora.stark_prove(PARAMS, bad)
    assert ora.stark_verify(PARAMS, bad, proof) != 0
    assert z.verify(PARAMS, bad, pvs, proof.tobytes()) != 0
    # operands that look at the next row or at selectors are refused by both parsers
    bb = air.AirBuilder(2, 0)
    bb.interactions = [(1, 0, bb.const(1), [bb.var(0, 1)])]
    prog_bad = bb.program()
    a_bad = [dict(program=prog_bad, log_height=3, width=2, n_pvs=0, trace=np.zeros((2, 8), np.uint32), pvs=NOPV)]
    with pytest.raises(RuntimeError):
        ora.stark_prove(PARAMS, a_bad)
    assert z.verify(PARAMS, a_bad, [NOPV], proof.tobytes()) != 0


def test_logup_twelve_field_message(ora):
    a = [dict(program=air.program_bus_air().program(), log_height=4, width=13, n_pvs=0, trace=air.program_bus_trace(4, 1), pvs=NOPV)]
    proof = ora.stark_prove(PARAMS, a)
    assert ora.stark_verify(PARAMS, a, proof) == 0
    assert z.verify(PARAMS, a, [NOPV], proof.tobytes()) == 0


def test_logup_unbalanced_bus_is_rejected(ora):
    airs = lookup_case()
    pvs = [a["pvs"] for a in airs]
    # a multiplicity that is off by one: per-AIR constraints hold, the exposed sums no longer cancel
    t2 = airs[2]["trace"].copy()
    t2[2, 0] = (int(t2[2, 0]) + 1) % ora.P
    bad = [dict(a) for a in airs]
    bad[2]["trace"] = t2
    proof = ora.stark_prove(PARAMS, bad)
    assert ora.stark_verify(PARAMS, bad, proof) != 0
    assert z.verify(PARAMS, bad, pvs, proof.tobytes()) != 0
    # a looked-up value that is not in the table
    s2 = airs[0]["trace"].copy()
    s2[1, 3] = (int(s2[1, 3]) + 1) % ora.P
    bad = [dict(a) for a in airs]
    bad[0]["trace"] = s2
    proof = ora.stark_prove(PARAMS, bad)
    assert ora.stark_verify(PARAMS, bad, proof) != 0
    assert z.verify(PARAMS, bad, pvs, proof.tobytes()) != 0


def test_logup_proof_is_not_accepted_for_interaction_free_airs(ora):
    """Dropping the interactions section changes the program digest and the proof layout."""
    airs = lookup_case(with_fib=False)
    proof = ora.stark_prove(PARAMS, airs)
    stripped = []
    for a in airs:
        w = [int(x) for x in a["program"]]
        cut = w.index(air.LOGUP_MAGIC)
        d = dict(a)
        d["program"] = np.array(w[:cut], dtype=np.uint32)
        stripped.append(d)
    assert z.verify(PARAMS, stripped, [a["pvs"] for a in airs], proof.tobytes()) != 0
