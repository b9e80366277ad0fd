"""GPU: bit-exact parity of each HIP stage (through the C ABI) against the CPU oracle and the
committed golden vectors."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
P = 2013265921


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "kat_v1.json")) as f:
        return json.load(f)


def test_monty_roundtrip(zk):
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.integers(0, P, 10000, dtype=np.uint64).astype(np.uint32),
                        np.array([0, 1, P - 1], dtype=np.uint32)])
    t = zk.upload(a)
    assert (zk.download(t) == a).all()
    # device holds Montgomery form: x * 2^32 mod p
    raw = t.cpu().numpy().view(np.uint32).astype(np.uint64)
    assert (raw == (a.astype(np.uint64) << np.uint64(32)) % np.uint64(P)).all()


def test_poseidon2_permutation_kat(zk, ora, kat):
    states = np.array([s for s, _ in kat["poseidon2_perm"]], dtype=np.uint32)
    t = zk.upload(states.reshape(-1))
    zk.poseidon2_permute_batch(t, len(states))
    got = zk.download(t).reshape(-1, 16)
    assert got.tolist() == [e for _, e in kat["poseidon2_perm"]]
    rng = np.random.default_rng(5)
    rs = ora.rand_field(rng, (1000, 16))
    t = zk.upload(rs.reshape(-1))
    zk.poseidon2_permute_batch(t, 1000)
    got = zk.download(t).reshape(-1, 16)
    for i in range(0, 1000, 37):
        assert (got[i] == ora.permute(rs[i])).all()


@pytest.mark.parametrize("log_n,width", [(0, 3), (1, 2), (3, 5), (8, 3), (10, 4), (11, 2), (12, 16), (13, 3),
                                         (16, 2), (20, 1), (21, 2), (23, 1), (24, 1)])
def test_ntt_forward_inverse(zk, ora, log_n, width):
    rng = np.random.default_rng(log_n * 100 + width)
    m = ora.rand_field(rng, (width, 1 << log_n))
    exp = ora.dft_batch(m, log_n)
    t = zk.upload(m.reshape(-1))
    zk.ntt_batch(t, log_n, width)
    got = zk.download(t).reshape(width, -1)
    assert (got == exp).all()
    # bit-reversed output mode
    t2 = zk.upload(m.reshape(-1))
    zk.ntt_batch(t2, log_n, width, bitrev_out=True)
    br = zk.download(t2).reshape(width, -1)
    idx = np.array([int(format(i, "0%db" % log_n)[::-1], 2) if log_n else 0 for i in range(1 << log_n)])
    assert (br == exp[:, idx]).all()
    # inverse round trip
    zk.ntt_batch(t, log_n, width, inverse=True)
    assert (zk.download(t).reshape(width, -1) == m).all()


def test_ntt_golden(zk, kat):
    for case in kat["dft"]:
        t = zk.upload(np.array(case["in"], dtype=np.uint32))
        zk.ntt_batch(t, case["log_n"], 1)
        assert zk.download(t).tolist() == case["fwd"]
        t = zk.upload(np.array(case["in"], dtype=np.uint32))
        zk.ntt_batch(t, case["log_n"], 1, inverse=True)
        assert zk.download(t).tolist() == case["inv"]


def test_ntt_strided_columns(zk, ora):
    log_n, width, stride = 9, 3, (1 << 9) + 40
    rng = np.random.default_rng(3)
    buf = ora.rand_field(rng, (width, stride))
    exp = ora.dft_batch(np.ascontiguousarray(buf[:, :1 << log_n]), log_n)
    t = zk.upload(buf.reshape(-1))
    zk.ntt_batch(t, log_n, width, stride=stride)
    got = zk.download(t).reshape(width, stride)
    assert (got[:, :1 << log_n] == exp).all() and (got[:, 1 << log_n:] == buf[:, 1 << log_n:]).all()


@pytest.mark.parametrize("log_n,added,width,shift", [(0, 1, 2, 31), (2, 1, 3, 31), (5, 2, 4, 31), (9, 1, 3, 31),
                                                      (10, 1, 5, 31), (12, 1, 7, 31), (13, 2, 3, 31), (14, 1, 3, 7), (17, 1, 2, 31),
                                                      (22, 1, 1, 31), (23, 1, 1, 31)])
def test_coset_lde(zk, ora, log_n, added, width, shift):
    rng = np.random.default_rng(log_n + 7 * width)
    m = ora.rand_field(rng, (width, 1 << log_n))
    exp = ora.coset_lde_batch(m, log_n, added, shift, bitrev_out=True)
    t = zk.upload(m.reshape(-1))
    out = zk.lde_batch(t, log_n, added, width, shift)
    assert (zk.download(out).reshape(width, -1) == exp).all()
    assert (zk.download(t).reshape(width, -1) == m).all()  # input preserved


def test_coset_lde_golden(zk, kat):
    for case in kat["coset_lde"]:
        t = zk.upload(np.array(case["in"], dtype=np.uint32))
        out = zk.lde_batch(t, case["log_n"], case["added_bits"], 1, case["shift"])
        assert zk.download(out).tolist() == case["bitrev"]


def test_lde_properties_large(zk, ora):
    """2^20 x 4: too big for the oracle's comfort; check linearity and that the low-degree part
    (first N rows of the bit-reversed coset LDE) interpolates back to the input."""
    log_n, width = 20, 4
    rng = np.random.default_rng(11)
    a, b = ora.rand_field(rng, (width, 1 << log_n)), ora.rand_field(rng, (width, 1 << log_n))
    s = ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    la = zk.download(zk.lde_batch(zk.upload(a.reshape(-1)), log_n, 1, width, 31)).astype(np.uint64)
    lb = zk.download(zk.lde_batch(zk.upload(b.reshape(-1)), log_n, 1, width, 31)).astype(np.uint64)
    ls = zk.download(zk.lde_batch(zk.upload(s.reshape(-1)), log_n, 1, width, 31))
    assert (((la + lb) % P) == ls).all()
    # one column against the oracle
    exp = ora.coset_lde_batch(a[:1], log_n, 1, 31, bitrev_out=True)
    assert (la.reshape(width, -1)[0] == exp[0]).all()


def _mats(ora, rng, shapes):
    return [ora.rand_field(rng, (w, 1 << lh)) for lh, w in shapes]


@pytest.mark.parametrize("shapes", [[(3, 5)], [(4, 8), (4, 3)], [(4, 9), (2, 3), (0, 2)],
                                    [(5, 20), (5, 1), (3, 17), (1, 8)], [(10, 300)], [(12, 37), (11, 8), (12, 4), (6, 70)],
                                    [(0, 5)], [(1, 16)]])
def test_merkle_commit_and_open(zk, ora, shapes):
    rng = np.random.default_rng(len(shapes) * 31 + shapes[0][1])
    mats = _mats(ora, rng, shapes)
    ot = ora.Tree(mats)
    dev = [(zk.upload(m.reshape(-1)), lh, w) for m, (lh, w) in zip(mats, shapes)]
    t = zk.merkle_commit(dev)
    assert t.root.tolist() == ot.root.tolist()
    for l in range(t.log_height + 1):
        assert (t.layer(l) == ot.layer(l)).all()
    n = 1 << t.log_height
    idx = sorted({0, n - 1, n // 2, n // 3, (5 * n) // 7})
    ops = t.open(idx)
    for i, q in enumerate(idx):
        assert (ops[i] == ot.open(q)).all()
        assert ot.verify(q, ops[i])


def test_merkle_many_heights_wide_short_levels(zk, ora):
    """A tree of fourteen heights -- more than the eight levels the row-sponge launch took before zkhip_config.rows_in_bulk -- with WIDE matrices
    of few rows (the chunk-circuit configuration's Keccak / limb chips: hundreds of permutations per row; they take the 16-lanes-per-row form
    of k_hash_rows_multi) beside narrow tall ones: every layer and a few openings against the oracle, under both settings of the switch."""
    shapes = [(17, 3), (16, 2), (15, 9), (14, 17), (13, 300), (12, 1), (11, 1500), (10, 8), (9, 5), (8, 3), (7, 700), (5, 2), (2, 9), (0, 3)]
    rng = np.random.default_rng(5)
    mats = _mats(ora, rng, shapes)
    ot = ora.Tree(mats)
    dev = [(zk.upload(m.reshape(-1)), lh, w) for m, (lh, w) in zip(mats, shapes)]
    cfg0 = zk.config()
    try:
        for bulk, coop_log in ((1, 15), (1, 0), (0, 15)):
            zk.set_config(rows_in_bulk=bulk, rows_coop_max_log=coop_log)
            t = zk.merkle_commit(dev)
            assert t.root.tolist() == ot.root.tolist(), (bulk, coop_log)
            for l in range(t.log_height + 1):
                assert (t.layer(l) == ot.layer(l)).all(), (bulk, coop_log, l)
            n = 1 << t.log_height
            for i, q in enumerate(sorted({0, n - 1, n // 3})):
                assert (t.open([q])[0] == ot.open(q)).all()
    finally:
        zk.set_config(cfg0)


def test_merkle_golden(zk, kat):
    for case in kat["merkle"]:
        dev = []
        for mm in case["mats"]:
            rows = np.array(mm["rows"], dtype=np.uint32)  # [height, width]
            dev.append((zk.upload(rows.T.copy().reshape(-1)), mm["log_height"], rows.shape[1]))
        assert zk.merkle_commit(dev).root.tolist() == case["root"]


def test_merkle_ragged_width_and_stride(zk, ora):
    # widths around the sponge rate, and a padded stride
    rng = np.random.default_rng(17)
    for w in (1, 7, 8, 9, 15, 16, 17):
        m = ora.rand_field(rng, (w, 64 + 8))
        ot = ora.Tree([np.ascontiguousarray(m[:, :64])])
        t = zk.merkle_commit([(zk.upload(m.reshape(-1)), 6, w, 72)])
        assert t.root.tolist() == ot.root.tolist()


def test_transcript_script(zk, kat):
    tr = zk.transcript()
    for step in kat["challenger"]:
        if step["op"] == "observe":
            tr.observe(step["vals"])
        elif step["op"] == "sample":
            assert tr.sample(step["n"]).tolist() == step["out"]
        elif step["op"] == "sample_bits":
            assert int(tr.sample(1)[0]) & ((1 << step["bits"]) - 1) == step["out"]
        else:
            assert tr.grind(step["bits"]) == step["witness"]


def test_grind_16_bits_matches_oracle(zk, ora):
    rng = np.random.default_rng(23)
    for trial in range(3):
        pre = ora.rand_field(rng, 5 + 3 * trial)
        tr, oc = zk.transcript(), ora.Challenger()
        tr.observe(pre), oc.observe(pre)
        w = tr.grind(16)
        assert w == oc.grind(16)
        assert tr.sample(4).tolist() == oc.sample(4).tolist()


@pytest.mark.parametrize("log_n_out", [0, 1, 3, 5, 10, 16])
def test_fri_fold(zk, ora, log_n_out):
    rng = np.random.default_rng(log_n_out)
    vals = ora.rand_field(rng, 8 << log_n_out)
    beta = ora.rand_field(rng, 4)
    exp = ora.fri_fold(vals, log_n_out, beta)
    got = zk.download(zk.fri_fold(zk.upload(vals), log_n_out, beta))
    assert (got == exp).all()


def test_fri_fold_golden(zk, kat):
    for case in kat["fri_fold"]:
        flat = np.array(case["in"], dtype=np.uint32).reshape(-1)
        got = zk.download(zk.fri_fold(zk.upload(flat), case["log_n_out"], case["beta"]))
        assert got.reshape(-1, 4).tolist() == case["out"]


@pytest.mark.parametrize("log_n,log_blowup,width", [(1, 1, 6), (6, 1, 12), (10, 2, 30), (13, 1, 40), (12, 3, 9)])
def test_constraint_eval_stage(zk, ora, log_n, log_blowup, width):
    """K5 as a stage through the C ABI (zkhip_constraint_eval): quotient values of one AIR over its committed LDE == oracle; on
    a satisfying trace the quotient is a polynomial of degree < 2^b * N - N (its chunks' LDEs are what zkhip_prove commits)."""
    from zkvm_prover_amd import air

    sa = air.SyntheticAir(width=width, n_free=max(4, width // 3), n_bool=2, n_boundary=1, seed=log_n)
    tr, pv = sa.gen_trace(log_n, seed=3)
    lde = ora.coset_lde_batch(tr, log_n, log_blowup, 31)
    alpha = ora.rand_field(np.random.default_rng(5), 4)
    exp = ora.constraint_eval(sa.program(), log_n, log_blowup, width, lde, pv, alpha)
    d_lde = zk.lde_batch(zk.upload(tr.reshape(-1)), log_n, log_blowup, width, 31)
    got = zk.download(zk.constraint_eval(sa.program(), log_n, log_blowup, width, d_lde, pv, alpha)).reshape(4, -1)
    assert (got == exp).all()
    # an unsatisfying trace still evaluates (to something else); a bus AIR is refused
    tr2 = tr.copy()
    tr2[0, 0] = (int(tr2[0, 0]) + 1) % P
    got2 = zk.download(zk.constraint_eval(sa.program(), log_n, log_blowup, width, zk.lde_batch(zk.upload(tr2.reshape(-1)), log_n, log_blowup, width, 31), pv, alpha))
    assert (got2.reshape(4, -1) == ora.constraint_eval(sa.program(), log_n, log_blowup, width, ora.coset_lde_batch(tr2, log_n, log_blowup, 31), pv, alpha)).all()
    import zkvm_prover_amd as z

    with pytest.raises(z.ZkhipError):
        zk.constraint_eval(air.lookup_sender_air().program(), log_n, log_blowup, 3, d_lde, np.zeros(0, np.uint32), alpha)
