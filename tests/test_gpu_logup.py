"""GPU: bus interactions (LogUp phase) -- the HIP prover's proofs equal the CPU oracle's byte for byte,
with the interpreter and with the keygen-compiled constraint kernel, over mixed heights; the product
verifier accepts them; unbalanced buses are rejected; a larger lookup verifies."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

pytestmark = pytest.mark.gpu
NOPV = np.zeros(0, np.uint32)


def _fib(log_n):
    tr, pv = air.fibonacci_trace(log_n)
    return dict(program=air.fibonacci_air().program(), log_height=log_n, width=2, n_pvs=3, trace=tr, pvs=pv)


def _lookup(log_s, log_t, seed=1, sender_width=3):
    s, t = air.lookup_traces(log_s, log_t, seed=seed, sender_width=sender_width)
    return (dict(program=air.lookup_sender_air(sender_width).program(), log_height=log_s, width=sender_width, n_pvs=0,
                 trace=s, pvs=NOPV),
            dict(program=air.lookup_table_air().program(), log_height=log_t, width=3, n_pvs=0, trace=t, pvs=NOPV))


def _mix(log_n, seed=3):
    tr, pv = air.bus_mix_trace(log_n, seed)
    return dict(program=air.bus_mix_air().program(), log_height=log_n, width=6, n_pvs=1, trace=tr, pvs=pv)


def _limb(log_n, seed=3):
    return dict(program=air.limb_air().program(), log_height=log_n, width=4, n_pvs=0, trace=air.limb_trace(log_n, seed), pvs=NOPV)


def _prove_gpu(zk, params, airs):
    pk = z.ProvingKey(zk, params, airs)
    d_traces = [zk.upload(a["trace"].reshape(-1)) for a in airs]
    proof = pk.prove(d_traces, [a["pvs"] for a in airs])
    assert len(proof) == pk.proof_size
    return proof, pk, d_traces


def _cases():
    s, t = _lookup(6, 4)
    s2, t2 = _lookup(9, 5, seed=2, sender_width=5)
    return {
        "lookup_pair": [s, t],
        "lookup_tall_table_last": [s, _fib(8), t],          # permutation tree shorter than the main tree
        "lookup_fib_first": [_fib(5), t, s],
        "mix_only": [_mix(5)],
        "mix_and_lookup": [_mix(7), s2, _fib(4), t2],
        "mix_min_height": [_mix(1)],
        "compound_messages": [_limb(6), s, t],
        "compound_and_mix": [_mix(4), _limb(8, seed=5), _fib(6)],
        "twelve_fields": [dict(program=air.program_bus_air().program(), log_height=5, width=13, n_pvs=0,
                               trace=air.program_bus_trace(5, 2), pvs=NOPV), _fib(3)],
    }


@pytest.mark.parametrize("name", sorted(_cases()))
@pytest.mark.parametrize("jit", ["interpreter", "jit"])
def test_logup_proof_bytes_equal_oracle(zk, ora, name, jit, monkeypatch):
    zk.set_config(jit=0 if jit == "interpreter" else 2)   # (restored after the test: tests/conftest.py)
    params = (1, 0, 8, 3, 4)
    airs = _cases()[name]
    for a in airs:
        assert air.check_trace(a["program"], a["trace"], a["pvs"]) == []
    exp = ora.stark_prove(params, airs)
    assert ora.stark_verify(params, airs, exp) == 0
    got, pk, d_traces = _prove_gpu(zk, params, airs)
    got_words = np.frombuffer(got, dtype=np.uint32)
    assert len(got_words) == len(exp)
    if not (got_words == exp).all():
        pytest.fail("proof differs from oracle at word %d of %d" % (int(np.nonzero(got_words != exp)[0][0]), len(exp)))
    assert z.verify(params, airs, [a["pvs"] for a in airs], got) == 0
    # deterministic, traces untouched
    assert pk.prove(d_traces, [a["pvs"] for a in airs]) == got
    assert (zk.download(d_traces[0]).reshape(airs[0]["width"], -1) == airs[0]["trace"]).all()


def test_logup_blowup_4_and_reference_parameters(zk, ora):
    s, t = _lookup(7, 5, seed=4)
    for params in [(2, 0, 5, 2, 3), z.DEFAULT_PARAMS]:
        airs = [s, t, _mix(6)]
        exp = ora.stark_prove(params, airs)
        got, _, _ = _prove_gpu(zk, params, airs)
        assert got == exp.tobytes()
        assert z.verify(params, airs, [a["pvs"] for a in airs], got) == 0


def test_logup_unbalanced_bus_fails_verification(zk):
    params = (1, 0, 8, 3, 4)
    s, t = _lookup(6, 4)
    t = dict(t)
    t["trace"] = t["trace"].copy()
    t["trace"][2, 1] = (int(t["trace"][2, 1]) + 1) % air.P
    airs = [s, t]
    got, _, _ = _prove_gpu(zk, params, airs)
    assert z.verify(params, airs, [a["pvs"] for a in airs], got) != 0


def test_logup_large_lookup_verifies(zk):
    """2^18 lookups into a 2^12-row table, next to a 2^16-row AIR with six interactions of up to eight
    fields: beyond what the oracle proves quickly; checked with the host verifier and tamper rejection."""
    params = z.DEFAULT_PARAMS
    s, t = _lookup(18, 12, seed=9, sender_width=16)
    airs = [s, _mix(16, seed=5), t]
    got, pk, _ = _prove_gpu(zk, params, airs)
    pvs = [a["pvs"] for a in airs]
    assert z.verify(params, airs, pvs, got) == 0
    words = np.frombuffer(got, dtype=np.uint32).copy()
    rng = np.random.default_rng(1)
    for pos in [4, 13, 21, 29] + rng.integers(0, len(words), 12).tolist():
        bad = words.copy()
        bad[pos] = (int(bad[pos]) + 1) % air.P
        assert z.verify(params, airs, pvs, bad.tobytes()) != 0


def _random_bus_air(seed, log_n):
    """A SyntheticAir with a random preprocessed matrix and random bus messages: every message is sent and received
    by the same chip (balanced by construction), its fields and multiplicity are random expressions of the row --
    sums, products, negations of main / preprocessed cells, public values and constants up to degree 2."""
    rng = np.random.default_rng(seed)
    width = int(rng.integers(8, 20))
    sa = air.SyntheticAir(width=width, n_free=max(4, width // 3), n_bool=2, n_boundary=1, seed=seed)
    b = sa.builder
    prep_w = int(rng.integers(0, 4))
    b.prep_width = prep_w

    def leaf():
        k = int(rng.integers(0, 4 if prep_w else 3))
        if k == 0:
            return b.var(int(rng.integers(0, width)))
        if k == 1:
            return b.pub(int(rng.integers(0, sa.n_pvs)))
        if k == 2:
            return b.const(int(rng.integers(0, air.P)))
        return b.prep(int(rng.integers(0, prep_w)))

    def expr(deg):
        e = leaf()
        for _ in range(int(rng.integers(0, 4))):
            op = int(rng.integers(0, 4))
            o = leaf()
            if op == 0:
                e = e + o
            elif op == 1:
                e = e - o
            elif op == 2 and deg > 1 and e.deg + o.deg <= deg:
                e = e * o
            else:
                e = -e
        return e

    for j in range(int(rng.integers(1, 5))):
        msg = [expr(2) for _ in range(int(rng.integers(1, 7)))]
        cnt = expr(2)
        bus = int(rng.integers(0, 50))
        b.push_interaction(bus, msg, cnt, "send")
        b.push_interaction(bus, msg, cnt, "receive")
    if prep_w:
        b.assert_zero(b.prep(0) * b.var(0) - b.var(0) * b.prep(0))
    tr, pv = sa.gen_trace(log_n, seed=seed + 7)
    d = dict(program=sa.program(), log_height=log_n, width=width, n_pvs=len(pv), trace=tr, pvs=pv)
    if prep_w:
        d["prep"] = rng.integers(0, air.P, size=(prep_w, 1 << log_n)).astype(np.uint32)
    return d


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_bus_expressions_match_oracle(zk, ora, seed):
    params = (1, 0, 6, 3, 3)
    airs = [_random_bus_air(seed, 5 + seed % 3), _random_bus_air(100 + seed, 3 + seed % 4), _fib(4)]
    for a in airs:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a.get("prep")) == []
    exp = ora.stark_prove(params, airs)
    assert ora.stark_verify(params, airs, exp) == 0
    pk = z.ProvingKey(zk, params, airs)
    pvs = [a["pvs"] for a in airs]
    got = pk.prove([zk.upload(a["trace"].reshape(-1)) for a in airs], pvs)
    assert got == exp.tobytes()
    assert z.verify(params, pk.verifying_airs(), pvs, got) == 0


@pytest.mark.parametrize("jit", ["interpreter", "jit"])
def test_four_interactions_per_column_group_at_blowup_4(zk, ora, jit, monkeypatch):
    """With blow-up 4 the degree budget is 5: four degree-1 interactions share one group of permutation columns
    (constraint phi * d1 d2 d3 d4 = sum +-count * prod of the other three, degree 5).  Proof bytes equal the oracle's."""
    zk.set_config(jit=0 if jit == "interpreter" else 2)   # (restored after the test: tests/conftest.py)
    b = air.bus_mix_air()
    b.max_constraint_degree = 5
    prog = b.program()
    assert b.interaction_groups == [0, 0, 0, 0, 1, 1] and b.max_degree() == 5
    tr, pv = air.bus_mix_trace(7, seed=3)
    ftr, fpv = air.fibonacci_trace(5)
    airs = [dict(program=prog, log_height=7, width=6, n_pvs=1, trace=tr, pvs=pv),
            dict(program=air.fibonacci_air().program(), log_height=5, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
    params = (2, 0, 6, 2, 3)
    exp = ora.stark_prove(params, airs)
    assert ora.stark_verify(params, airs, exp) == 0
    pk = z.ProvingKey(zk, params, airs)
    got = pk.prove([zk.upload(a["trace"].reshape(-1)) for a in airs], [a["pvs"] for a in airs])
    assert got == exp.tobytes()
    assert z.verify(params, airs, [a["pvs"] for a in airs], got) == 0
    # the same program is over the budget of blow-up 2 and must be refused there
    with pytest.raises(Exception):
        z.ProvingKey(zk, (1, 0, 6, 2, 3), airs)
