"""GPU: the MMCS path chip -- in-circuit verification of the Merkle openings stored in the REFERENCE'S OWN proofs (a piece of
the recursion circuit).  Path rows and the Poseidon2 chip's 298 permutation columns are generated on the
device; path chip + Poseidon2 chip + claims table prove together: every opening of the fixture leads to the commitment its proof
carries, and the chip's claims are exactly the digests of the opened rows.  Proof bytes == oracle."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mmcs_path_util as mu  # noqa: E402

import zkvm_prover_amd as z  # noqa: E402
from zkvm_prover_amd import air  # noqa: E402

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NOPV = np.zeros(0, np.uint32)
PARAMS = (1, 0, 12, 4, 4)
P = 2013265921


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(HERE, "golden", "ref_v1_vectors.json")) as f:
        return json.load(f)


def _airs(ora, leaf, idx, starts, kinds, digs, claims_rows=None):
    rows = int(starts[-1])
    lh = max(1, int(np.ceil(np.log2(max(rows, 2)))))
    tr, hin, claims, bad = ora.mmcs_path_trace(leaf, idx, starts, kinds, digs, lh)
    assert bad == 0
    chip = np.zeros((299, 1 << lh), np.uint32)
    full = np.zeros((1 << lh, 16), np.uint32)
    full[:rows] = hin
    chip[:298] = ora.poseidon2_air_trace(full, lh)
    chip[298, :rows] = 1
    table = mu.claims_table(claims if claims_rows is None else claims_rows, int(np.ceil(np.log2(max(len(claims), 2)))))
    A = lambda prog, w, t: dict(program=prog, log_height=int(np.log2(t.shape[1])), width=w, n_pvs=0, trace=t, pvs=NOPV)  # noqa: E731
    return [A(air.mmcs_path_air(9, 10).program(), 39, tr), A(air.poseidon2_air(9).program(), 299, chip),
            A(air.mmcs_claims_air(10).program(), 19, table)], lh, rows, claims


def test_reference_openings_verified_in_circuit(zk, ora, vec):
    leaf, idx, starts, kinds, digs, want = mu.records_of_fixture(ora, vec)
    airs, lh, rows, claims = _airs(ora, leaf, idx, starts, kinds, digs)
    assert sorted((tuple(c[:8].tolist()), int(c[8]), int(c[9]), tuple(c[10:].tolist())) for c in claims) == sorted(want)
    dev = zk.device
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint32).view(np.int32)).to(dev)  # noqa: E731
    d_tr, d_hin = zk.mmcs_path_tracegen(as_dev(leaf), as_dev(idx), as_dev(starts), as_dev(kinds), as_dev(digs), lh)
    assert (zk.download(d_tr).reshape(39, -1) == airs[0]["trace"]).all()
    N = 1 << lh
    d_chip = torch.empty(299 * N, dtype=torch.int32, device=dev)
    zk.poseidon2_air_tracegen(d_hin, lh, d_chip)
    d_chip[298 * N:] = zk.upload(airs[1]["trace"][298])
    assert (zk.download(d_chip).reshape(299, N) == airs[1]["trace"]).all()
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_tr, d_chip, zk.upload(airs[2]["trace"].reshape(-1))], [NOPV] * 3)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 3, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    # one flipped word in one sibling digest: the walk ends in another root, the claims no longer match the table
    bad = digs.copy()
    bad[5][0] = (int(bad[5][0]) + 1) % P
    d_tr2, d_hin2 = zk.mmcs_path_tracegen(as_dev(leaf), as_dev(idx), as_dev(starts), as_dev(kinds), as_dev(bad), lh)
    zk.poseidon2_air_tracegen(d_hin2, lh, d_chip)
    d_chip[298 * N:] = zk.upload(airs[1]["trace"][298])
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 3, pk.prove([d_tr2, d_chip, zk.upload(airs[2]["trace"].reshape(-1))], [NOPV] * 3)) != 0
    pk.close()


def test_generator_edge_cases(zk, ora):
    dev = zk.device
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint32).view(np.int32)).to(dev)  # noqa: E731
    leaf = np.arange(8, dtype=np.uint32).reshape(1, 8)
    digs = np.arange(16, dtype=np.uint32).reshape(2, 8) + 100
    # one path of two sibling steps at index 2 into a 4-row trace: rows 0..1 used
    tr, hin, claims, bad = ora.mmcs_path_trace(leaf, [2], [0, 2], [0, 0], digs, 2)
    d_tr, d_hin = zk.mmcs_path_tracegen(t(leaf), t([2]), t([0, 2]), t([0, 0]), t(digs), 2)
    assert bad == 0 and (zk.download(d_tr).reshape(39, -1) == tr).all()
    assert (zk.download(d_hin).reshape(-1, 16)[:2] == hin).all() and (zk.download(d_hin).reshape(-1, 16)[2:] == 0).all()
    assert claims[0][8] == 2 and claims[0][9] == 2
    for starts, kinds in (([0, 2], [1, 0]), ([0, 5], [0] * 5)):   # bottom step an injection; more steps than rows
        with pytest.raises(z.ZkhipError):
            zk.mmcs_path_tracegen(t(leaf), t([0]), t(starts), t(kinds), t(np.zeros((len(kinds), 8), np.uint32)), 2)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    d_tr, _ = zk.mmcs_path_tracegen(e, e, t([0]), e, e, 1)
    assert (zk.download(d_tr) == 0).all()


def test_reference_fri_fold_steps_proven(zk, ora, vec):
    """The FRI fold chip (air.fri_fold_air) on the sibling pairs stored in the reference's proofs: the rows prove (JIT and
    interpreter paths alike: bytes == oracle) and a wrong folded value is refused at keygen-free proving time by the verifier."""
    from test_fri_fold_chip_cpu import fold_trace

    tr, n, lh = fold_trace(vec)
    # the rows from the device generator: records = the sibling pairs, the layer's beta, the pair's index and layer size
    recs = [(t["e0"], t["e1"], lay["beta"], t["k"], lay["log_n_out"]) for lay in vec["fri_layers"] for t in lay["triples"]]
    t32 = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint32).view(np.int32)).to(zk.device)  # noqa: E731
    d_tr = zk.fri_fold_chip_tracegen(t32([r[0] for r in recs]), t32([r[1] for r in recs]), t32([r[2] for r in recs]), t32([r[3] for r in recs]),
                                     t32([r[4] for r in recs]), lh)
    assert (zk.download(d_tr).reshape(19, -1) == tr).all()       # incl. folded == the value the reference's proof holds
    with pytest.raises(z.ZkhipError):                            # an index outside its layer
        zk.fri_fold_chip_tracegen(t32([[1, 2, 3, 4]]), t32([[1, 2, 3, 4]]), t32([[1, 0, 0, 0]]), t32([8]), t32([3]), 0)
    # the evaluation points come from the pair indices: the domain-point chip (device-generated) holds every (k, x^-1) the fold rows send
    from test_fri_fold_chip_cpu import point_table
    pt = point_table(tr, n)
    cnt = sorted({int(k): 0 for k in tr[18][:n]})
    mult = [int((tr[18][:n] == k).sum()) for k in cnt]
    lpt = int(np.log2(pt.shape[1]))
    d_pt = zk.domain_point_tracegen(t32(cnt), t32(mult), lpt)
    assert (zk.download(d_pt).reshape(air.DOMAIN_POINT_WIDTH, -1) == pt).all()
    with pytest.raises(z.ZkhipError):
        zk.domain_point_tracegen(t32([1 << 26]), t32([1]), 0)
    airs = [dict(program=air.fri_fold_air(11).program(), log_height=lh, width=19, n_pvs=0, trace=tr, pvs=NOPV),
            dict(program=air.domain_point_air(11).program(), log_height=lpt, width=air.DOMAIN_POINT_WIDTH, n_pvs=0, trace=pt, pvs=NOPV)]
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_tr, d_pt], [NOPV] * 2)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    w = tr.copy()
    w[13][0] = (int(w[13][0]) + 1) % P                       # a wrong folded value: the fold constraint
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, pk.prove([zk.upload(w.reshape(-1)), d_pt], [NOPV] * 2)) != 0
    w = tr.copy()
    w[18][0] = int(w[18][0]) ^ 1                             # a fold row that claims another pair's point: the point bus
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 2, pk.prove([zk.upload(w.reshape(-1)), d_pt], [NOPV] * 2)) != 0
    pk.close()


def test_transcript_chip_over_the_poseidon2_chip(zk, ora):
    """The transcript chip (air.duplex_air: the DuplexChallenger in-circuit) with the Poseidon2 chip serving its 32-field hash bus
    and a table stating what was observed and sampled: the Poseidon2 chip's 298 columns are generated on the device from the
    duplexing inputs, the three AIRs prove together (bytes == oracle), and a table that claims a different sample is refused."""
    from test_duplex_chip_cpu import random_script
    import duplex_util as du

    rng = np.random.default_rng(9)
    script = random_script(rng, 80)
    tr, io, samples = du.run_script(ora, script)
    rows = len(tr)
    lh = int(np.ceil(np.log2(rows + 1)))
    N = 1 << lh
    t = du.padded(tr, lh)
    inputs = np.zeros((N, 16), np.uint32)
    inputs[:rows] = tr[:, :16]
    chip = np.zeros((299, N), np.uint32)
    chip[:298] = ora.poseidon2_air_trace(inputs, lh)
    chip[298, :rows] = 1
    lio = int(np.ceil(np.log2(len(set(io)) + 1)))
    table = du.io_table(io, lio)
    A = lambda prog, w, x: dict(program=prog, log_height=int(np.log2(x.shape[1])), width=w, n_pvs=0, trace=x, pvs=NOPV)  # noqa: E731
    airs = [A(air.duplex_air(9, 10).program(), 50, t), A(air.poseidon2_air(9, out_lanes=16).program(), 299, chip),
            A(air.duplex_io_air(10).program(), 5, table)]
    # the chip's rows from the device: one record per duplexing (observed values, how many lanes were sampled afterwards)
    t32 = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint32).view(np.int32)).to(zk.device)  # noqa: E731
    n_obs = tr[:, 32:40].sum(axis=1).astype(np.uint32)
    obs = np.where(tr[:, 32:40] == 1, tr[:, 0:8], 0).astype(np.uint32)
    d_t, d_hin = zk.duplex_tracegen(t32(n_obs), t32(obs), t32(tr[:, 40:48].sum(axis=1)), lh)
    assert (zk.download(d_t).reshape(50, N) == t).all() and (zk.download(d_hin).reshape(N, 16) == inputs).all()
    with pytest.raises(z.ZkhipError):
        zk.duplex_tracegen(t32([9]), t32(np.zeros((1, 8))), t32([0]), 0)
    d_chip = torch.empty(299 * N, dtype=torch.int32, device=zk.device)
    zk.poseidon2_air_tracegen(d_hin, lh, d_chip)
    d_chip[298 * N:] = zk.upload(chip[298])
    assert (zk.download(d_chip).reshape(299, N) == chip).all()
    pk = z.ProvingKey(zk, PARAMS, airs)
    proof = pk.prove([d_t, d_chip, zk.upload(table.reshape(-1))], [NOPV] * 3)
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 3, proof) == 0
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    wrong = table.copy()
    r = int(np.nonzero(table[3] == 1)[0][0])     # a sampled value
    wrong[2][r] = (int(wrong[2][r]) + 1) % P
    assert z.verify(PARAMS, pk.verifying_airs(), [NOPV] * 3, pk.prove([zk.upload(t.reshape(-1)), d_chip, zk.upload(wrong.reshape(-1))], [NOPV] * 3)) != 0
    pk.close()
