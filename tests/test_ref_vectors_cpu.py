"""CPU: the oracle AND the product's host verifier primitives against vectors taken from the REFERENCE'S OWN stored
proofs (tests/golden/ref_v1_vectors.json, made by tests/golden/gen_ref_vectors.py from
/root/reference/crates/{verifier/testdata/proofs,prover/testdata}/*.json).  This is what pins the Poseidon2 permutation,
TruncatedPermutation, PaddingFreeSponge, MerkleTreeMmcs (mixed heights) and the FRI fold to the reference's prover:
none of the expected values below was produced by code of this repository."""
import base64
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import refproof_v1 as rp

HERE = os.path.dirname(os.path.abspath(__file__))
P = 2013265921


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(HERE, "golden", "ref_v1_vectors.json")) as f:
        return json.load(f)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def test_fixture_covers_every_stored_proof(vec):
    names = {s["name"] for s in vec["sources"]}
    assert len(names) == 8
    per_file = {n: sum(1 for t in vec["compress"] if t["file"] == n) for n in names}
    # main x2, after-challenge, quotient + every FRI layer commitment of every file
    for n, cnt in per_file.items():
        assert cnt == 4 + vec["shapes"][n]["n_fri_layers"], (n, cnt)
    assert len(vec["compress"]) == 193
    assert len(vec["openings"]) == 12 and all(len(q["batches"]) == 6 for q in vec["openings"])
    assert sum(len(l["triples"]) - 1 for l in vec["fri_layers"]) >= 40  # triples beyond the one that fixes beta


def test_oracle_compress_reproduces_reference_commitments(ora, vec):
    for t in vec["compress"]:
        out = np.zeros(8, np.uint32)
        ora.lib().ora_compress(ora.p32(_u32(t["L"])), ora.p32(_u32(t["R"])), ora.p32(out))
        assert out.tolist() == t["root"], (t["file"], t["commit"])
        # and as a bare permutation of L || R (TruncatedPermutation = first 8 lanes)
        st = ora.permute(_u32(t["L"] + t["R"]))
        assert st[:8].tolist() == t["root"]


def test_product_host_permutation_reproduces_reference_commitments(vec):
    import zkvm_prover_amd as z

    for t in vec["compress"]:
        st = z.poseidon2_permute_host(t["L"] + t["R"])
        assert st[:8].tolist() == t["root"], (t["file"], t["commit"])


def _ora_verify(ora, root, lhs, ws, index, opening):
    a = (C.c_uint * len(lhs))(*lhs)
    w = (C.c_size_t * len(ws))(*ws)
    return bool(ora.lib().ora_mmcs_verify(ora.p32(_u32(root)), a, w, len(lhs), index, ora.p32(_u32(opening))))


def test_mmcs_openings_of_reference_queries(ora, vec):
    """Every input-batch opening of the stored queries (single matrices, the 17-matrix mixed-height main and
    after-challenge batches, 62 quotient chunks) leads to the commitment in the proof -- oracle and product."""
    import zkvm_prover_amd as z

    n = 0
    for q in vec["openings"]:
        H = q["log_max_height"]
        for b in q["batches"]:
            idx = q["index"] >> (H - max(b["log_heights"]))
            assert _ora_verify(ora, b["root"], b["log_heights"], b["widths"], idx, b["opening"]), (q["file"], b["commit"])
            assert z.mmcs_verify(b["root"], b["log_heights"], b["widths"], idx, b["opening"]) == 0
            n += 1
            # a wrong index, a flipped row word and a flipped path word are all rejected
            bad = list(b["opening"])
            bad[0] = (bad[0] + 1) % P
            assert not _ora_verify(ora, b["root"], b["log_heights"], b["widths"], idx, bad)
            assert z.mmcs_verify(b["root"], b["log_heights"], b["widths"], idx, bad) == -7
            bad = list(b["opening"])
            bad[-1] = (bad[-1] + 1) % P
            assert z.mmcs_verify(b["root"], b["log_heights"], b["widths"], idx, bad) == -7
            if max(b["log_heights"]) > 0:
                assert z.mmcs_verify(b["root"], b["log_heights"], b["widths"], idx ^ 1, b["opening"]) == -7
                assert not _ora_verify(ora, b["root"], b["log_heights"], b["widths"], idx ^ 1, b["opening"])
    assert n == 72
    # the mixed-height batches really are mixed
    assert any(len(set(b["log_heights"])) > 5 for q in vec["openings"] for b in q["batches"])


def test_fri_fold_triples(ora, vec):
    """(e0, e1, beta) -> folded value of the reference's FRI layers: oracle's layer fold and the product's fold_row."""
    import zkvm_prover_amd as z

    for lay in vec["fri_layers"]:
        lo = lay["log_n_out"]
        vals = np.zeros((2 << lo, 4), dtype=np.uint32)
        for t in lay["triples"]:
            vals[2 * t["k"]] = t["e0"]
            vals[2 * t["k"] + 1] = t["e1"]
        out = ora.fri_fold(vals.reshape(-1), lo, lay["beta"]).reshape(-1, 4)
        for t in lay["triples"]:
            assert out[t["k"]].tolist() == t["folded"], (lay["file"], lay["layer"], t["k"])
            assert z.fri_fold_row(t["k"], lo, lay["beta"], t["e0"], t["e1"]).tolist() == t["folded"]


def test_fri_layer_leaves(ora, vec):
    """Pairs of extension values are the leaves of a FRI layer's tree: flattened to 8 base words, one sponge call."""
    import zkvm_prover_amd as z

    assert len(vec["fri_leaves"]) >= 10
    for lf in vec["fri_leaves"]:
        assert _ora_verify(ora, lf["root"], [lf["log_height"]], [8], lf["index"], lf["opening"])
        assert z.mmcs_verify(lf["root"], [lf["log_height"]], [8], lf["index"], lf["opening"]) == 0


REF = "/root/reference/crates"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_fixture_matches_reference_files(vec):
    """In the build container: the fixture's sources are the reference's files, byte for byte, and the Python twin of
    the container codec round-trips them."""
    for s in vec["sources"]:
        d = json.load(open(os.path.join(REF, s["file"])))
        blob = base64.b64decode(d["proof"]["proofs"])
        assert hashlib.sha256(blob).hexdigest() == s["sha256"]
        proofs = rp.decode_proofs(blob)
        assert len(proofs) == 1 and rp.encode_proofs(proofs) == blob
        assert rp.shape_of(proofs[0])["n_queries"] == vec["shapes"][s["name"]]["n_queries"]
        pv = rp.decode_public_values(base64.b64decode(d["proof"]["public_values"]))
        assert len(pv) == 32 and all(rp.from_monty(v) < 256 for v in pv)  # 32 user public values, one byte each


def test_logup_exposed_sums_of_the_reference_proofs_cancel(ora, vec):
    """The verifier's bus check on the reference's own data: in every stored proof the `exposed_values_after_challenge` of the AIRs
    with interactions add up to zero in F_p[X]/(X^4 - 11) -- through the oracle's check and through the product's
    (zkhip_logup_exposed_check, the function zkhip_verify calls); one changed coordinate and both refuse."""
    import zkvm_prover_amd as z

    lib = z.load_library()
    assert len(vec["logup_exposed"]) == 8
    o = ora.lib()
    o.ora_logup_exposed_check.restype = C.c_int
    o.ora_logup_exposed_check.argtypes = [C.POINTER(C.c_uint32), C.c_size_t]
    for e in vec["logup_exposed"]:
        ex = _u32(e["exposed"])
        assert ex.shape == (len(e["air_ids"]), 4) and len(e["air_ids"]) >= 15
        assert (ex.astype(np.int64).sum(axis=0) % P).tolist() == [0, 0, 0, 0]      # coordinate-wise: ext addition is
        assert o.ora_logup_exposed_check(ora.p32(ex), len(ex)) == 0, e["file"]
        assert lib.zkhip_logup_exposed_check(ex.ctypes.data_as(C.POINTER(C.c_uint32)), len(ex)) == 0, e["file"]
        bad = ex.copy()
        bad[3, 2] = (int(bad[3, 2]) + 1) % P
        assert o.ora_logup_exposed_check(ora.p32(bad), len(bad)) != 0
        assert lib.zkhip_logup_exposed_check(bad.ctypes.data_as(C.POINTER(C.c_uint32)), len(bad)) == -7
        assert lib.zkhip_logup_exposed_check(ex[:-1].copy().ctypes.data_as(C.POINTER(C.c_uint32)), len(ex) - 1) == -7
