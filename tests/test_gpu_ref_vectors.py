"""GPU: the HIP kernels (through the C ABI) against vectors taken from the REFERENCE'S OWN stored proofs
(tests/golden/ref_v1_vectors.json; see tests/golden/gen_ref_vectors.py).  No expected value here comes from code of
this repository, and the oracle is not in the loop: the device recomputes the reference's commitments.

 * zkhip_poseidon2_permute_batch on L || R of every commitment of all eight proofs -> the commitment;
 * zkhip_merkle_commit (row sponge kernel, ragged multi-matrix rows) + zkhip_poseidon2_permute_batch (compress) replay
   p3's MerkleTreeMmcs::verify_batch on the device for every stored opening -- leaf digests from the opened rows, one
   batched compression per tree level, injected digests of the shorter matrices -- and must arrive at the root in the proof;
 * zkhip_fri_fold on the sibling pairs of the reference's FRI layers with the beta they determine -> the folded values.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
P = 2013265921


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(HERE, "golden", "ref_v1_vectors.json")) as f:
        return json.load(f)


def test_permutation_reproduces_reference_commitments(zk, vec):
    states = np.array([t["L"] + t["R"] for t in vec["compress"]], dtype=np.uint32)
    d = zk.upload(states.reshape(-1))
    zk.poseidon2_permute_batch(d, len(states))
    got = zk.download(d).reshape(-1, 16)[:, :8]
    assert got.tolist() == [t["root"] for t in vec["compress"]]


def _device_row_digests(zk, rows_per_matrix):
    """rows_per_matrix: list (matrices of ONE height group) of [n_queries][width] canonical rows.
    Commits them as n-row matrices on the device and returns the leaf digests [n_queries][8]: the sponge over the
    concatenated rows, exactly what the tree kernel hashes."""
    n = len(rows_per_matrix[0])
    lh = max(1, int(np.ceil(np.log2(max(n, 2)))))
    mats, keep = [], []
    for rows in rows_per_matrix:
        w = len(rows[0])
        m = np.zeros((w, 1 << lh), dtype=np.uint32)  # column-major: column c = m[c]
        m[:, :n] = np.array(rows, dtype=np.uint32).T
        t = zk.upload(m.reshape(-1))
        keep.append(t)
        mats.append((t, lh, w))
    tree = zk.merkle_commit(mats)
    dig = tree.layer(0)[:n].copy()
    tree.close()
    return dig


def _device_compress(zk, left, right):
    st = np.concatenate([np.asarray(left, dtype=np.uint32), np.asarray(right, dtype=np.uint32)], axis=1)
    d = zk.upload(st.reshape(-1))
    zk.poseidon2_permute_batch(d, st.shape[0])
    return zk.download(d).reshape(-1, 16)[:, :8].copy()


def _device_mmcs_roots(zk, log_heights, widths, indices, openings):
    """verify_batch for several openings of the same commitment shape at once, every hash on the device."""
    H = max(log_heights)
    total = sum(widths)
    offs = np.cumsum([0] + list(widths))
    ops = np.array(openings, dtype=np.uint32)
    n = len(indices)

    def group_digests(level):
        ms = [m for m in range(len(widths)) if log_heights[m] == level]
        if not ms:
            return None
        return _device_row_digests(zk, [[ops[q, offs[m]:offs[m + 1]].tolist() for q in range(n)] for m in ms])

    cur = group_digests(H)
    for l in range(H):
        sib = ops[:, total + 8 * l: total + 8 * l + 8]
        bit = np.array([(indices[q] >> l) & 1 for q in range(n)], dtype=bool)
        left = np.where(bit[:, None], sib, cur)
        right = np.where(bit[:, None], cur, sib)
        cur = _device_compress(zk, left, right)
        inj = group_digests(H - l - 1)
        if inj is not None:
            cur = _device_compress(zk, cur, inj)
    return cur


def test_device_replays_reference_openings(zk, vec):
    files = sorted({q["file"] for q in vec["openings"]})
    checked = 0
    for f in files:
        qs = [q for q in vec["openings"] if q["file"] == f]
        for b in range(len(qs[0]["batches"])):
            bt = qs[0]["batches"][b]
            lhs, ws = bt["log_heights"], bt["widths"]
            idx = [q["index"] >> (q["log_max_height"] - max(lhs)) for q in qs]
            roots = _device_mmcs_roots(zk, lhs, ws, idx, [q["batches"][b]["opening"] for q in qs])
            for r in roots:
                assert r.tolist() == bt["root"], (f, bt["commit"])
            checked += len(qs)
    assert checked == 72


def test_device_fri_layer_leaves(zk, vec):
    for lf in vec["fri_leaves"]:
        r = _device_mmcs_roots(zk, [lf["log_height"]], [8], [lf["index"]], [lf["opening"]])
        assert r[0].tolist() == lf["root"], (lf["file"], lf["layer"])


def test_device_fri_fold_reproduces_reference_layers(zk, vec):
    n = 0
    for lay in vec["fri_layers"]:
        lo = lay["log_n_out"]
        vals = np.zeros((2 << lo, 4), dtype=np.uint32)
        for t in lay["triples"]:
            vals[2 * t["k"]] = t["e0"]
            vals[2 * t["k"] + 1] = t["e1"]
        d = zk.upload(vals.reshape(-1))
        out = zk.download(zk.fri_fold(d, lo, lay["beta"])).reshape(-1, 4)
        for t in lay["triples"]:
            assert out[t["k"]].tolist() == t["folded"], (lay["file"], lay["layer"], t["k"])
            n += 1
    assert n >= 50


def test_device_running_sum_over_the_reference_exposed_sums_is_zero(zk):
    """The LogUp kernels (zkhip_ext_batch_inverse, zkhip_logup_running_sum: prefix sums of multiplicity / denominator in the quartic
    extension) over the exposed cumulative sums the reference's stored proofs carry: with denominators 1 / S_k and unit
    multiplicities the running total is S_1 + ... + S_n -- the verifier's bus check -- and must end at zero."""
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_v1_vectors.json")) as f:
        vec = json.load(f)
    for e in vec["logup_exposed"]:
        ex = np.array([v for v in e["exposed"] if any(v)], dtype=np.uint32)
        n = len(ex)
        den = zk.ext_batch_inverse(zk.upload(ex.reshape(-1)), n)
        out, total = zk.logup_running_sum(den, zk.upload(np.ones(n, np.uint32)), n)
        assert np.asarray(total).tolist() == [0, 0, 0, 0], e["file"]
        run = zk.download(out).reshape(-1, 4)
        assert (run[-1] == 0).all() and (run[0] == ex[0]).all()
        assert (run[1] == (ex[0].astype(np.int64) + ex[1]) % P).all()
