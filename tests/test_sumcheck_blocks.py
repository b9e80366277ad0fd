"""LogUp / sum-check building blocks (K6, K7): oracle vs the independent model on CPU, HIP vs oracle
and golden vectors on the GPU, plus size-independent properties at large sizes."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
P = 2013265921


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "kat_v1.json")) as f:
        return json.load(f)


def u32(x):
    return np.asarray(x, dtype=np.uint32)


def test_oracle_blocks_match_golden(ora, kat):
    for d, inv in kat["ext_batch_inverse"]:
        assert ora.ext_batch_inverse(u32(d)).tolist() == inv
    c = kat["logup_running_sum"]
    assert ora.logup_running_sum(u32(c["den"]), u32(c["num"])).reshape(-1, 4).tolist() == c["out"]
    c = kat["mle_fold"]
    assert ora.mle_fold(u32(c["in"]), c["r"]).reshape(-1, 4).tolist() == c["out"]
    for c in kat["sumcheck_round"]:
        assert ora.sumcheck_round([u32(t) for t in c["tables"]]).reshape(-1, 4).tolist() == c["out"]


@pytest.mark.gpu
def test_hip_blocks_match_golden(zk, kat):
    den = u32([d for d, _ in kat["ext_batch_inverse"]]).reshape(-1)
    got = zk.download(zk.ext_batch_inverse(zk.upload(den), len(den) // 4)).reshape(-1, 4)
    assert got.tolist() == [inv for _, inv in kat["ext_batch_inverse"]]
    c = kat["logup_running_sum"]
    out, total = zk.logup_running_sum(zk.upload(u32(c["den"]).reshape(-1)), zk.upload(u32(c["num"])), len(c["num"]))
    assert zk.download(out).reshape(-1, 4).tolist() == c["out"] and total.tolist() == c["out"][-1]
    c = kat["mle_fold"]
    assert zk.download(zk.mle_fold(zk.upload(u32(c["in"]).reshape(-1)), len(c["in"]) // 2, c["r"])).reshape(-1, 4).tolist() == c["out"]
    for c in kat["sumcheck_round"]:
        tabs = [zk.upload(u32(t).reshape(-1)) for t in c["tables"]]
        assert zk.sumcheck_round(tabs, len(c["tables"][0]) // 2).reshape(-1, 4).tolist() == c["out"]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 8, 9, 255, 2048, 2049, 100003])
def test_hip_running_sum_and_inverse_vs_oracle(zk, ora, n):
    rng = np.random.default_rng(n)
    den = ora.rand_field(rng, 4 * n)
    den[::4] |= 1  # never the zero element
    num = ora.rand_field(rng, n)
    num[:: max(1, n // 5)] = 0  # zero multiplicities are common in LogUp
    assert (zk.download(zk.ext_batch_inverse(zk.upload(den), n)) == ora.ext_batch_inverse(den)).all()
    out, total = zk.logup_running_sum(zk.upload(den), zk.upload(num), n)
    exp = ora.logup_running_sum(den, num)
    assert (zk.download(out) == exp).all() and (total == exp[-4:]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("k,log_n", [(1, 0), (2, 3), (3, 10), (4, 13), (2, 17)])
def test_hip_sumcheck_round_and_fold_vs_oracle(zk, ora, k, log_n):
    rng = np.random.default_rng(100 * k + log_n)
    n = 1 << log_n
    tabs = [ora.rand_field(rng, 8 * n) for _ in range(k)]
    d_tabs = [zk.upload(t) for t in tabs]
    got = zk.sumcheck_round(d_tabs, n)
    assert (got == ora.sumcheck_round(tabs)).all()
    r = ora.rand_field(rng, 4)
    folded = [zk.mle_fold(t, n, r) for t in d_tabs]
    for f, t in zip(folded, tabs):
        assert (zk.download(f) == ora.mle_fold(t, r)).all()


@pytest.mark.gpu
def test_sumcheck_protocol_invariant_large(zk, ora):
    """2^22-entry tables (too slow for the oracle): the sum-check identity s_i(0) + s_i(1) = claim_{i}
    and claim_{i+1} = s_i(r_i) must hold round after round down to a single entry, where the claim
    equals the product of the fully folded tables."""
    import torch

    from pymodel import ext_mul

    log_n, k = 22, 3
    n = 1 << log_n
    tabs = [torch.randint(0, P, (4 * n,), dtype=torch.int32, device="cuda") for _ in range(k)]  # Montgomery residues
    rng = np.random.default_rng(1)

    def interp(evals, r):  # Lagrange interpolation of degree-k s(X) at nodes 0..k, evaluated at ext r (python ints)
        acc = [0, 0, 0, 0]
        for t, e in enumerate(evals):
            num, den = [1, 0, 0, 0], 1
            for u in range(len(evals)):
                if u != t:
                    num = ext_mul(num, [(r[0] - u) % P, r[1], r[2], r[3]])
                    den = den * (t - u) % P
            c = pow(den, P - 2, P)
            term = ext_mul(num, [int(x) for x in e])
            acc = [(a + x * c) % P for a, x in zip(acc, term)]
        return acc

    claim = None
    size = n
    while size > 1:
        half = size // 2
        s = zk.sumcheck_round(tabs, half).reshape(-1, 4)
        s01 = [(int(a) + int(b)) % P for a, b in zip(s[0], s[1])]
        if claim is not None:
            assert s01 == claim
        r = ora.rand_field(rng, 4)
        claim = interp(s, [int(x) for x in r])
        tabs = [zk.mle_fold(t, half, r) for t in tabs]
        size = half
    finals = [zk.download(t)[:4].tolist() for t in tabs]
    prod = [1, 0, 0, 0]
    for f in finals:
        prod = ext_mul(prod, [int(x) for x in f])
    assert prod == claim


@pytest.mark.gpu
def test_hip_batch_inverse_with_zero_elements(zk, ora):
    """A vanishing LogUp denominator must not poison its batch-inversion group: zero -> zero (what the oracle's
    element-wise Fermat inversion gives), every other element still inverted exactly."""
    rng = np.random.default_rng(9)
    n = 5000
    den = ora.rand_field(rng, (n, 4))
    for i in (0, 1, 7, 8, 2047, 2048, 3333, n - 1):
        den[i] = 0
    den[100:120] = 0  # a whole run of zeros, several per lane group
    got = zk.download(zk.ext_batch_inverse(zk.upload(den.reshape(-1)), n)).reshape(-1, 4)
    exp = ora.ext_batch_inverse(den.reshape(-1)).reshape(-1, 4)
    assert (got == exp).all()
    assert (got[100:120] == 0).all() and (got[50] != 0).any()
    num = ora.rand_field(rng, n)
    out, total = zk.logup_running_sum(zk.upload(den.reshape(-1)), zk.upload(num), n)
    assert (zk.download(out) == ora.logup_running_sum(den.reshape(-1), num)).all()
