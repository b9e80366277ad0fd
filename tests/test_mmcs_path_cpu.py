"""CPU: the MMCS path chip (air.mmcs_path_air -- in-circuit verification of mixed-height Merkle openings, a piece of the
recursion circuit) on the openings stored in the REFERENCE'S OWN proofs: the oracle's generator walks every opening of the
fixture to the commitment the proof carries, its trace satisfies the AIR, the claims it emits are exactly the row digests of
the opened matrices at their heights and indices, and tampering is caught."""
import json
import os

import numpy as np
import pytest

from zkvm_prover_amd import air
import mmcs_path_util as mu

HERE = os.path.dirname(os.path.abspath(__file__))
NOPV = np.zeros(0, np.uint32)
P = 2013265921


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(HERE, "golden", "ref_v1_vectors.json")) as f:
        return json.load(f)


def test_reference_openings_walk_to_their_commitments(ora, vec):
    leaf, idx, starts, kinds, digs, want = mu.records_of_fixture(ora, vec)
    rows = int(starts[-1])
    lh = int(np.ceil(np.log2(rows)))
    tr, hin, claims, bad = ora.mmcs_path_trace(leaf, idx, starts, kinds, digs, lh)
    assert bad == 0 and len(idx) >= 60 and int(kinds.sum()) > 100          # mixed-height batches: many injections
    # every path ends in the commitment of the stored proof, every claim is an opened matrix group's digest at its place
    got = sorted((tuple(c[:8].tolist()), int(c[8]), int(c[9]), tuple(c[10:].tolist())) for c in claims)
    assert got == sorted(want)
    prog = air.mmcs_path_air(9, 10).program()
    assert air.check_trace(prog, tr, NOPV) == []
    # the compressions the chip asks the Poseidon2 chip for are true
    for r in (0, 1, rows // 2, rows - 1):
        assert ora.permute(hin[r])[:8].tolist() == tr[8:16, r].tolist()
    # tampering: a wrong parent, a wrong bit, a path cut short, a root swapped mid-path, wrong position counters
    for col, row in ((8, 3), (32, 5), (35, 2), (0, 7), (37, 4), (38, 6), (36, rows - 1)):   # (the digests a, b themselves are held by the hash bus)
        w = tr.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != [], (col, row)


def test_generator_refuses_bad_paths(ora):
    leaf = np.zeros((1, 8), np.uint32)
    # a path whose bottom step is an injection, one that does not fit the trace
    assert ora.mmcs_path_trace(leaf, [0], [0, 2], [1, 0], np.zeros((2, 8), np.uint32), 2)[3] == 1
    assert ora.mmcs_path_trace(leaf, [0], [0, 5], [0] * 5, np.zeros((5, 8), np.uint32), 2)[3] == 1
    tr, hin, claims, bad = ora.mmcs_path_trace(leaf, [], [0], [], np.zeros((0, 8), np.uint32), 1)
    assert bad == 0 and not tr.any() and len(claims) == 0
