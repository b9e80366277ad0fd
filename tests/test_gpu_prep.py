"""GPU: preprocessed traces -- keygen commits the table (commitment equal to the oracle's), proofs equal the
oracle's byte for byte with both constraint kernels, and verify against the commitment alone."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

pytestmark = pytest.mark.gpu
NOPV = np.zeros(0, np.uint32)


def _fib(log_n):
    tr, pv = air.fibonacci_trace(log_n)
    return dict(program=air.fibonacci_air().program(), log_height=log_n, width=2, n_pvs=3, trace=tr, pvs=pv)


def _range(log_u, log_t, seed=1, user_width=4):
    u, m, prep = air.range_traces(log_u, log_t, seed=seed, user_width=user_width)
    return (dict(program=air.range_user_air(user_width).program(), log_height=log_u, width=user_width, n_pvs=0, trace=u, pvs=NOPV),
            dict(program=air.range_table_air().program(), log_height=log_t, width=1, n_pvs=0, trace=m, pvs=NOPV, prep=prep))


def _cases():
    u, t = _range(6, 4)
    u2, t2 = _range(9, 7, seed=2, user_width=7)
    u3, t3 = _range(5, 5, seed=3)
    return {
        "range_pair": [u, t],
        "table_first": [t, _fib(7), u],            # preprocessed tree shorter than the main tree
        "two_tables": [u2, t, t2, u, _fib(3)],     # two AIRs with their own preprocessed commitments
        "same_height": [u3, t3],
    }


@pytest.mark.parametrize("name", sorted(_cases()))
@pytest.mark.parametrize("jit", ["interpreter", "jit"])
def test_prep_proof_bytes_equal_oracle(zk, ora, name, jit, monkeypatch):
    zk.set_config(jit=0 if jit == "interpreter" else 2)   # (restored after the test: tests/conftest.py)
    params = (1, 0, 8, 3, 4)
    airs = _cases()[name]
    pvs = [a["pvs"] for a in airs]
    exp = ora.stark_prove(params, airs)
    assert ora.stark_verify(params, airs, exp) == 0
    pk = z.ProvingKey(zk, params, airs)
    for i, a in enumerate(airs):
        if a.get("prep") is not None:
            assert (pk.prep_commitment(i) == ora.prep_commit(params, a)).all()
    d_traces = [zk.upload(a["trace"].reshape(-1)) for a in airs]
    got = pk.prove(d_traces, pvs)
    got_words = np.frombuffer(got, dtype=np.uint32)
    assert len(got_words) == len(exp)
    if not (got_words == exp).all():
        pytest.fail("proof differs from oracle at word %d of %d" % (int(np.nonzero(got_words != exp)[0][0]), len(exp)))
    vk = pk.verifying_airs()
    assert all("prep" not in v for v in vk)
    assert z.verify(params, vk, pvs, got) == 0
    assert pk.prove(d_traces, pvs) == got


def test_prep_blowup_4_and_reference_parameters(zk, ora):
    u, t = _range(7, 5, seed=4)
    for params in [(2, 0, 5, 2, 3), z.DEFAULT_PARAMS]:
        airs = [u, t, _fib(6)]
        exp = ora.stark_prove(params, airs)
        pk = z.ProvingKey(zk, params, airs)
        got = pk.prove([zk.upload(a["trace"].reshape(-1)) for a in airs], [a["pvs"] for a in airs])
        assert got == exp.tobytes()
        assert z.verify(params, pk.verifying_airs(), [a["pvs"] for a in airs], got) == 0


def test_prep_keygen_requires_the_table(zk):
    u, t = _range(6, 4)
    t = dict(t)
    del t["prep"]
    with pytest.raises(z.ZkhipError):
        z.ProvingKey(zk, (1, 0, 8, 3, 4), [u, t])


def test_prep_large_range_check_verifies(zk):
    """2^20 lookups into a preprocessed 2^16-entry range table."""
    params = z.DEFAULT_PARAMS
    u, t = _range(20, 16, seed=9, user_width=8)
    airs = [u, t]
    pk = z.ProvingKey(zk, params, airs)
    pvs = [a["pvs"] for a in airs]
    got = pk.prove([zk.upload(a["trace"].reshape(-1)) for a in airs], pvs)
    vk = pk.verifying_airs()
    assert z.verify(params, vk, pvs, got) == 0
    words = np.frombuffer(got, dtype=np.uint32).copy()
    rng = np.random.default_rng(1)
    for pos in [4, 13, 21, 29] + rng.integers(0, len(words), 10).tolist():
        bad = words.copy()
        bad[pos] = (int(bad[pos]) + 1) % air.P
        assert z.verify(params, vk, pvs, bad.tobytes()) != 0
