"""The sizes of the leaf verifier circuits the aggregation tree's common heights rest on (docs/round5_b.md 3, 4): `prove_cli leaf-stats` builds the
circuits on the host.  A leaf node that outgrows 2^20 gate rows / 2^17 permutations doubles EVERY node of every tree -- this test is the guard."""
import json
import os
import subprocess

import prover_mirror_util as pm
from test_vm_cpu import batch_circuit_toml, chunk_circuit_toml

REF = (1, 0, 100, 16, 16)


def stats(cfg, frame, children):
    r = subprocess.run([pm.CLI, "leaf-stats", cfg, str(frame), str(children)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_two_base_segment_proofs_fit_the_common_node():
    """base chips (with and without the batch circuit's native extension), frames of 2^20: a leaf over TWO segment proofs fits 2^20 / 2^17"""
    for s in stats("-", 20, 2):
        assert s["gate_rows"] <= 0.97 * (1 << 20) and s["permutations"] <= (1 << 17), s


def test_the_reference_configurations(tmp_path):
    chunk, batch = tmp_path / "chunk.toml", tmp_path / "batch.toml"
    chunk.write_text(chunk_circuit_toml(REF))
    batch.write_text(batch_circuit_toml(REF))
    c = stats(str(chunk), 20, 1)
    assert [s["chips"] for s in c] == [22, 22, 26, 51]
    # the 26-chip shape enters the tree without a wrapper; the 51-chip leaf circuit is a quarter of round 4's 2^23 rows
    assert c[2]["gate_rows"] <= 0.97 * (1 << 20) and c[2]["permutations"] <= 0.98 * (1 << 17), c[2]
    assert (1 << 20) < c[3]["gate_rows"] <= (1 << 21) and (1 << 18) < c[3]["permutations"] <= (1 << 19), c[3]
    b2 = stats(str(batch), 20, 2)
    assert b2[0]["chips"] == 25 and b2[0]["gate_rows"] <= 0.97 * (1 << 20) and b2[0]["permutations"] <= (1 << 17), b2[0]
