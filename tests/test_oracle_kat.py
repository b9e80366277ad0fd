"""CPU: pins oracle/ (C restatement) against tests/golden/kat_v1.json (independent big-int model)
and the SURVEY.md A.3 anchors.  No GPU needed."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "kat_v1.json")) as f:
        return json.load(f)


def u32(x):
    return np.asarray(x, dtype=np.uint32)


def test_field_constants(ora, kat):
    # p pinned in the reference tree: scripts/compress_bn254.py:10
    assert kat["p"] == ora.P == 2013265921 == 2 ** 31 - 2 ** 27 + 1
    l = ora.lib()
    for k, g in enumerate(kat["two_adic_generators"]):
        assert l.ora_two_adic_generator(k) == g
        assert l.ora_pow(g, 1 << k) == 1 and (k == 0 or l.ora_pow(g, 1 << (k - 1)) != 1)
    assert l.ora_two_adic_generator(27) == 0x1A427A41 == l.ora_pow(31, 15)


def test_field_mul_inv(ora, kat):
    l = ora.lib()
    for a, b, c in kat["field_mul"]:
        assert l.ora_mul(a, b) == c
    for a, ai in kat["field_inv"]:
        assert l.ora_inv(a) == ai and l.ora_mul(a, ai) == 1


def test_ext(ora, kat):
    l = ora.lib()
    for a, b, c in kat["ext_mul"]:
        out = np.zeros(4, np.uint32)
        l.ora_ext_mul(ora.p32(u32(a)), ora.p32(u32(b)), ora.p32(out))
        assert out.tolist() == c
    for a, ai in kat["ext_inv"]:
        out = np.zeros(4, np.uint32)
        l.ora_ext_inv(ora.p32(u32(a)), ora.p32(out))
        assert out.tolist() == ai


def test_poseidon2_round_constants(ora, kat):
    rc = np.ctypeslib.as_array(ora.lib().ora_poseidon2_round_constants(), shape=(141,)).tolist()
    assert rc == kat["poseidon2_rc"]
    # anchors of SURVEY.md A.3 (zkhash RC16 / p3 BABYBEAR_RC16_* tables)
    assert rc[:8] == [0x69CBB6AF, 0x46AD93F9, 0x60A00F4E, 0x6B1297CD, 0x23189AFE, 0x732E7BEF, 0x72C246DE, 0x2C941900]
    assert rc[64:77] == [0x5A8053C0, 0x693BE639, 0x3858867D, 0x19334F6B, 0x128F0FD8, 0x4E2B1CCB, 0x61210CE0,
                         0x3C318939, 0x0B5B2F22, 0x2EDB11D5, 0x213EFFDF, 0x0CAC4606, 0x241AF16D]
    assert rc[77:81] == [0x7290A80D, 0x6F7E5329, 0x598EC8A8, 0x76A859A0] and rc[140] == 0x608758B8


def test_poseidon2_permutation(ora, kat):
    for s, exp in kat["poseidon2_perm"]:
        assert ora.permute(s).tolist() == exp
    # SURVEY.md A.3 self-consistency anchors
    assert ora.permute(list(range(16))).tolist()[:4] == [1906786279, 1737026427, 1959749225, 700325316]
    assert ora.permute([0] * 16).tolist()[:8] == [1168947398, 128782440, 747404447, 883925857, 360581875,
                                                   1704698758, 1878363991, 1054281681]


def test_sponge_and_compress(ora, kat):
    l = ora.lib()
    for xs, exp in kat["hash_slice"]:
        out = np.zeros(8, np.uint32)
        a = u32(xs) if xs else np.zeros(1, np.uint32)
        l.ora_hash_slice(ora.p32(a), len(xs), ora.p32(out))
        assert out.tolist() == exp
    for lft, r, exp in kat["compress"]:
        out = np.zeros(8, np.uint32)
        l.ora_compress(ora.p32(u32(lft)), ora.p32(u32(r)), ora.p32(out))
        assert out.tolist() == exp


def test_dft_against_definition(ora, kat):
    for case in kat["dft"]:
        x = u32(case["in"])[None, :]
        assert ora.dft_batch(x, case["log_n"]).tolist()[0] == case["fwd"]
        assert ora.dft_batch(x, case["log_n"], inverse=True).tolist()[0] == case["inv"]
    # fast vs naive O(n^2) inside the oracle, and round trip, on a multi-column matrix
    rng = np.random.default_rng(1)
    m = ora.rand_field(rng, (3, 1 << 9))
    f = ora.dft_batch(m, 9)
    for c in range(3):
        out = np.zeros(1 << 9, np.uint32)
        ora.lib().ora_dft_naive(ora.p32(np.ascontiguousarray(m[c])), ora.p32(out), 9, 0)
        assert (out == f[c]).all()
    assert (ora.dft_batch(f, 9, inverse=True) == m).all()


def test_coset_lde(ora, kat):
    for case in kat["coset_lde"]:
        x = u32(case["in"])[None, :]
        nat = ora.coset_lde_batch(x, case["log_n"], case["added_bits"], case["shift"], bitrev_out=False)
        br = ora.coset_lde_batch(x, case["log_n"], case["added_bits"], case["shift"], bitrev_out=True)
        assert nat.tolist()[0] == case["natural"] and br.tolist()[0] == case["bitrev"]


def test_merkle_mixed_heights(ora, kat):
    for case in kat["merkle"]:
        mats = [np.array(mm["rows"], dtype=np.uint32).T.copy() for mm in case["mats"]]  # -> [width, height]
        t = ora.Tree(mats)
        assert t.root.tolist() == case["root"]
        for idx in {0, (1 << t.log_height) - 1, (1 << t.log_height) // 3}:
            op = t.open(idx)
            assert t.verify(idx, op)
            bad = op.copy()
            bad[0] = (int(bad[0]) + 1) % ora.P
            assert not t.verify(idx, bad)


def test_challenger_script(ora, kat):
    ch = ora.Challenger()
    for step in kat["challenger"]:
        if step["op"] == "observe":
            ch.observe(step["vals"])
        elif step["op"] == "sample":
            assert ch.sample(step["n"]).tolist() == step["out"]
        elif step["op"] == "sample_bits":
            assert ch.sample_bits(step["bits"]) == step["out"]
        else:
            assert ch.grind(step["bits"]) == step["witness"]


def test_fri_fold(ora, kat):
    for case in kat["fri_fold"]:
        flat = u32(case["in"]).reshape(-1)
        out = ora.fri_fold(flat, case["log_n_out"], case["beta"])
        assert out.reshape(-1, 4).tolist() == case["out"]


def test_proof_digests_are_stable(ora):
    """The oracle's proof bytes for a fixed set of small AIR sets (tests/golden/proof_digests_v3.json)."""
    import importlib.util
    import json
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("gen_proof_digests", os.path.join(here, "gen_proof_digests.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = json.load(open(os.path.join(here, "proof_digests_v3.json")))
    got = gen.digests(ora)
    assert got == want
