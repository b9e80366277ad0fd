"""CPU: the FRI fold chip (air.fri_fold_air -- one arity-2 folding step per row, in-circuit) on the sibling pairs stored in the
REFERENCE'S OWN proofs: every (e0, e1, beta, folded) triple of the fixture, at the point its index stands for, satisfies the AIR;
a wrong folded value, a wrong point or swapped siblings do not."""
import json
import os

import numpy as np
import pytest

from zkvm_prover_amd import air

HERE = os.path.dirname(os.path.abspath(__file__))
NOPV = np.zeros(0, np.uint32)
P = 2013265921
G27 = 0x1A427A41   # generator of the 2^27 two-adic subgroup


def bitrev(k, bits):
    return int(format(k, "0%db" % bits)[::-1], 2) if bits else 0


def fold_rows(vec):
    """[19, n] canonical rows from the fixture's FRI layers: x = g^bitrev(k) in the subgroup of order 2 * 2^log_n_out"""
    rows = []
    for lay in vec["fri_layers"]:
        lo = lay["log_n_out"]
        g = pow(G27, 1 << (27 - (lo + 1)), P)
        for t in lay["triples"]:
            xinv = pow(pow(g, bitrev(t["k"], lo), P), P - 2, P)
            rows.append(t["e0"] + t["e1"] + lay["beta"] + [xinv] + t["folded"] + [1, t["k"]])
    return np.array(rows, np.uint32).T


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(HERE, "golden", "ref_v1_vectors.json")) as f:
        return json.load(f)


def fold_trace(vec):
    rows = fold_rows(vec)
    lh = int(np.ceil(np.log2(rows.shape[1])))
    tr = np.zeros((19, 1 << lh), np.uint32)
    tr[:, :rows.shape[1]] = rows
    return tr, rows.shape[1], lh


def test_reference_fold_steps_satisfy_the_air(vec):
    tr, n, lh = fold_trace(vec)
    assert n >= 50
    prog = air.fri_fold_air().program()
    assert air.check_trace(prog, tr, NOPV) == [] and air.quotient_chunks(prog) == 2
    for col, row in ((13, 0), (16, 5), (12, 7), (8, 11)):
        w = tr.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != []
    w = tr.copy()
    w[0:4, 3], w[4:8, 3] = tr[4:8, 3], tr[0:4, 3]    # siblings swapped: the fold at x is not the fold at -x
    assert air.check_trace(prog, w, NOPV) != []


def point_table(tr, n):
    """domain_point_air trace for the distinct pair indices of the fold rows (with how often each is used)"""
    from collections import Counter

    cnt = Counter(int(k) for k in tr[18][:n])
    lh = max(1, int(np.ceil(np.log2(len(cnt)))))
    w = air.domain_point_inverse_roots()
    t = np.zeros((air.DOMAIN_POINT_WIDTH, 1 << lh), np.uint32)
    t[1 + air.DOMAIN_POINT_BITS:1 + 2 * air.DOMAIN_POINT_BITS, :] = 1      # padding rows: k = 0, x = 1, multiplicity 0
    for r, (k, m) in enumerate(sorted(cnt.items())):
        acc = 1
        t[0, r] = k
        for j in range(air.DOMAIN_POINT_BITS):
            bit = (k >> j) & 1
            acc = acc * w[j] % P if bit else acc
            t[1 + j, r], t[1 + air.DOMAIN_POINT_BITS + j, r] = bit, acc
        t[1 + 2 * air.DOMAIN_POINT_BITS, r] = m
    return t


def test_fold_points_follow_from_the_pair_indices(vec):
    """x^-1 of every fold row of the fixture equals the product the domain-point chip builds from the bits of the pair index --
    for every layer size with the same per-bit constants -- and its trace satisfies the AIR."""
    tr, n, lh = fold_trace(vec)
    t = point_table(tr, n)
    prog = air.domain_point_air(11).program()
    assert air.check_trace(prog, t, NOPV) == []
    last = t[2 * air.DOMAIN_POINT_BITS]
    by_k = {int(k): int(x) for k, x in zip(t[0], last)}
    assert all(by_k[int(tr[18][i])] == int(tr[12][i]) for i in range(n))
    assert len({lay["log_n_out"] for lay in vec["fri_layers"]}) >= 3         # several layer sizes, one table
    w = t.copy()
    w[1 + 3][0] ^= 1                                                         # a flipped bit: k no longer matches
    assert air.check_trace(prog, w, NOPV) != []
