"""GPU: the aggregation layer on the device.
  * zkhip_recursion_tracegen == the numpy / oracle restatement of the node's traces, cell for cell; the HIP proof of a node
    equals the oracle's proof of the same instance byte for byte;
  * `prove_cli prove-agg`: 69 segment proofs -> 18 leaf nodes (arity 4) -> 6 -> 2 -> 1 root, every node a REAL verifier
    circuit; the root verifies under root.vk alone; its statement is the chained state of all 69 segments; node proofs equal
    the oracle's; one flipped byte in one segment proof and the node's witness is unsatisfiable."""
import json
import os
import subprocess

import numpy as np
import pytest

import zkvm_prover_amd as z

import prover_mirror_util as pm
import recursion_util as ru

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
NOPV = ru.NOPV


def test_device_traces_and_node_proof_match_the_oracle(zk, ora):
    kids = [ru.counter_segment(s, seed=i) for i, s in enumerate([3, 10, 17, 24])]
    vk = ru.verifying(PARAMS, kids[0])
    pk = z.ProvingKey(zk, PARAMS, kids[0])
    proofs = []
    for k in kids:
        tr = [zk.upload(a["trace"].reshape(-1)) for a in k]
        proofs.append(pk.prove(tr, [a["pvs"] for a in k]))
        assert proofs[-1] == ora.stark_prove(PARAMS, k).tobytes()
    pvs = [[a["pvs"] for a in k] for k in kids]
    rc = z.RecursionCircuit(PARAMS, vk, 4, stmt=ru.COUNTER_STMT)
    st, npv = rc.witness(proofs, pvs)
    assert st == 0, rc.last_error()
    assert int(npv[8]) == 3 and int(npv[9]) == 31
    node = ru.node_instance(rc, npv)
    d_traces = rc.tracegen(zk)
    for a, d in zip(node, d_traces):
        assert (zk.download(d).reshape(a["width"], -1) == a["trace"]).all()
    npk = z.ProvingKey(zk, PARAMS, node)
    proof = npk.prove(d_traces, [NOPV, NOPV, npv])
    assert proof == ora.stark_prove(PARAMS, node).tobytes()
    assert z.verify(PARAMS, npk.verifying_airs(), [NOPV, NOPV, npv], proof) == 0
    # a tampered child: the traces are still generated, and their proof is refused
    bad = bytearray(proofs[2])
    bad[4 * 200] ^= 1
    st, npv2 = rc.witness([proofs[0], proofs[1], bytes(bad)], pvs[:3])
    assert st == -7
    proof2 = npk.prove(rc.tracegen(zk), [NOPV, NOPV, npv2])
    assert z.verify(PARAMS, npk.verifying_airs(), [NOPV, NOPV, npv2], proof2) != 0


def test_leaf_and_internal_parameter_sets(zk, ora):
    """The reference proves its aggregation levels under parameter sets of their own (crates/prover/src/prover/mod.rs:47-52: leaf /
    internal).  A circuit is built for its CHILD's parameters and proven under its own: segments under PARAMS, the leaf node under
    blow-up 2 with 6 queries, the internal node (built for THAT child) under blow-up 4 with 3 queries; every proof == the oracle's."""
    LEAF, INTERNAL = (1, 0, 6, 2, 2), (2, 0, 3, 1, 2)
    kids = [ru.counter_segment(s, seed=i) for i, s in enumerate([3, 10, 17])]
    vk = ru.verifying(PARAMS, kids[0])
    pk = z.ProvingKey(zk, PARAMS, kids[0])
    proofs = [pk.prove([zk.upload(a["trace"].reshape(-1)) for a in k], [a["pvs"] for a in k]) for k in kids]
    pvs = [[a["pvs"] for a in k] for k in kids]
    leaf_rc = z.RecursionCircuit(PARAMS, vk, 4, stmt=ru.COUNTER_STMT)          # verifies proofs made under PARAMS
    leaf_proofs, leaf_pvs, leaf_vk = [], [], None
    for group in ([0, 1], [2]):
        st, npv = leaf_rc.witness([proofs[i] for i in group], [pvs[i] for i in group])
        assert st == 0, leaf_rc.last_error()
        node = ru.node_instance(leaf_rc, npv)
        npk = z.ProvingKey(zk, LEAF, node)                                      # ... and is proven under the leaf set
        proof = npk.prove(leaf_rc.tracegen(zk), [NOPV, NOPV, npv])
        assert proof == ora.stark_prove(LEAF, node).tobytes()
        assert z.verify(LEAF, npk.verifying_airs(), [NOPV, NOPV, npv], proof) == 0
        assert z.verify(PARAMS, npk.verifying_airs(), [NOPV, NOPV, npv], proof) != 0   # (not a proof under the app's parameters)
        leaf_proofs.append(proof), leaf_pvs.append([NOPV, NOPV, npv])
        leaf_vk = npk.verifying_airs()
    top_rc = z.RecursionCircuit(LEAF, leaf_vk, 3, stmt="node")                   # built for children proven under the leaf set
    st, rpv = top_rc.witness(leaf_proofs, leaf_pvs)
    assert st == 0, top_rc.last_error()
    assert int(rpv[8]) == 3 and int(rpv[9]) == 24                               # the chained state: from segment 0's start to segment 2's end
    root = ru.node_instance(top_rc, rpv)
    rpk = z.ProvingKey(zk, INTERNAL, root)
    proof = rpk.prove(top_rc.tracegen(zk), [NOPV, NOPV, rpv])
    assert proof == ora.stark_prove(INTERNAL, root).tobytes()
    assert z.verify(INTERNAL, rpk.verifying_airs(), [NOPV, NOPV, rpv], proof) == 0
    # a circuit built for other child parameters than the proofs were made under has no witness
    wrong = z.RecursionCircuit(PARAMS, leaf_vk, 3, stmt="node")
    assert wrong.witness(leaf_proofs, leaf_pvs)[0] != 0


def test_prove_agg_cli_69_segments_to_one_root(ora, tmp_path):
    n_seg = 69
    start, tasks, segs = 1000, [], []
    for i in range(n_seg):
        k = ru.counter_segment(start, seed=i)
        segs.append(k)
        d = tmp_path / ("s%d" % i)
        d.mkdir()
        tasks.append(pm.write_task(str(d), k, identifier="seg-%d" % i))
        start += 7
    exe, cfg = pm.write_app(str(tmp_path), segs[0], PARAMS)
    out = tmp_path / "out"
    out.mkdir()
    r = pm.run_cli("prove-agg", exe, cfg, str(out), "2", "0:0/0:1", *tasks)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert info["segments"] == 69 and info["nodes"] == 18 + 6 + 2 + 1 and info["levels"] == 4
    # the root verifies with the root vk alone
    assert pm.run_cli("verify", str(out / "root.vk"), cfg, str(out / "root.json")).returncode == 0
    root = json.loads((out / "root.json").read_text())
    rpv = np.frombuffer(pm.un_b64_bincode(root["user_pvs_proof"]), dtype=np.uint32)
    assert rpv.tolist() == info["root_public_values"]
    assert int(rpv[8]) == 1000 and int(rpv[9]) == 1000 + 7 * n_seg        # start of segment 0, end of segment 68
    # the accumulator, recomputed from the segments' public values through the tree's shape
    accs = [ru.leaf_accumulator([[a["pvs"] for a in s] for s in segs[i:i + 4]]) for i in range(0, n_seg, 4)]
    while len(accs) > 1:
        accs = [ru.internal_accumulator(accs[i:i + 3]) for i in range(0, len(accs), 3)]
    assert rpv[10:18].tolist() == accs[0].tolist()
    # a leaf node's proof against the oracle: same circuit, same witness -> same bytes; so is an internal node's (the one circuit
    # that verifies leaf proofs and proofs of itself)
    vk = ru.verifying(PARAMS, segs[0])
    rc, irc = ru.one_key_circuits(PARAMS, vk, ru.COUNTER_STMT)
    assert rpv[:8].tolist() == rc.child_vk_digest().tolist()
    seg_proofs = [ora.stark_prove(PARAMS, s).tobytes() for s in segs[4:8]]
    st, npv = rc.witness(seg_proofs, [[a["pvs"] for a in s] for s in segs[4:8]])
    assert st == 0
    leaf1 = json.loads((out / "agg-0-1.json").read_text())
    assert pm.un_b64_bincode(leaf1["proof"]) == ora.stark_prove(PARAMS, ru.node_instance(rc, npv)).tobytes()
    leaf_pcs, LC = ru.node_key_commits(PARAMS, rc.airs())
    int_pcs, IC = ru.node_key_commits(PARAMS, irc.airs())
    assert rpv[18:26].tolist() == LC.tolist() and rpv[26:34].tolist() == IC.tolist()
    kids = [json.loads((out / ("agg-0-%d.json" % i)).read_text()) for i in range(3)]
    kid_pvs = [np.frombuffer(pm.un_b64_bincode(k["user_pvs_proof"]), dtype=np.uint32) for k in kids]
    st, ipv = irc.witness([pm.un_b64_bincode(k["proof"]) for k in kids], [[NOPV, NOPV, p] for p in kid_pvs], prep_commits=[leaf_pcs] * 3,
                          is_leaf=[1, 1, 1], leaf_commit=LC, internal_commit=IC)
    assert st == 0, irc.last_error()
    int0 = json.loads((out / "agg-1-0.json").read_text())
    assert pm.un_b64_bincode(int0["proof"]) == ora.stark_prove(PARAMS, ru.node_instance(irc, ipv)).tobytes()
    # ONE key: a node of any level above the leaves is a proof of the internal circuit (its statement is about ITS segments); a leaf
    # proof is not; a flipped public value is refused
    for name in ("agg-1-0.json", "agg-2-1.json"):
        assert pm.run_cli("verify", str(out / "root.vk"), cfg, str(out / name)).returncode == 0
    assert pm.run_cli("verify", str(out / "root.vk"), cfg, str(out / "agg-0-0.json")).returncode != 0
    lie = dict(root)
    w = bytearray(pm.un_b64_bincode(root["user_pvs_proof"]))
    w[4 * 9] ^= 1
    lie["user_pvs_proof"] = pm.b64_bincode(bytes(w))
    (out / "lie.json").write_text(json.dumps(lie))
    assert pm.run_cli("verify", str(out / "root.vk"), cfg, str(out / "lie.json")).returncode != 0


def test_root_key_does_not_depend_on_the_depth(tmp_path):
    """3 segments (one leaf node under one internal node) and 40 segments (10 + 4 + 2 + 1 nodes): root.vk is the same file, each root
    verifies under the other tree's key.  With round 3's per-depth keys (ZKHIP_AGG_PER_DEPTH_KEYS=1) the two keys differ."""
    import os
    import subprocess

    keys = {}
    for mode, env in (("one", {}), ("per_depth", {"ZKHIP_AGG_PER_DEPTH_KEYS": "1"})):
        for n_seg in (3, 40):
            start, tasks, segs = 50, [], []
            d0 = tmp_path / ("%s-%d" % (mode, n_seg))
            d0.mkdir()
            for i in range(n_seg):
                k = ru.counter_segment(start, seed=i)
                segs.append(k)
                d = d0 / ("s%d" % i)
                d.mkdir()
                tasks.append(pm.write_task(str(d), k, identifier="seg-%d" % i))
                start += 7
            exe, cfg = pm.write_app(str(d0), segs[0], PARAMS)
            r = subprocess.run([pm.CLI, "prove-agg", exe, cfg, str(d0), "2", "0:0/0:1", *tasks], capture_output=True, text=True, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-3000:]
            info = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            keys[(mode, n_seg)] = dict(vk=(d0 / "root.vk").read_bytes(), dir=d0, cfg=cfg, info=info)
    assert keys[("one", 3)]["info"]["levels"] == 2 and keys[("one", 40)]["info"]["levels"] == 4
    assert keys[("per_depth", 3)]["info"]["levels"] == 1 and keys[("per_depth", 40)]["info"]["levels"] == 4
    assert keys[("one", 3)]["vk"] == keys[("one", 40)]["vk"]
    assert keys[("per_depth", 3)]["vk"] != keys[("per_depth", 40)]["vk"]
    a, b = keys[("one", 3)], keys[("one", 40)]
    assert pm.run_cli("verify", str(a["dir"] / "root.vk"), a["cfg"], str(b["dir"] / "root.json")).returncode == 0
    assert pm.run_cli("verify", str(b["dir"] / "root.vk"), b["cfg"], str(a["dir"] / "root.json")).returncode == 0
    a, b = keys[("per_depth", 3)], keys[("per_depth", 40)]
    assert pm.run_cli("verify", str(a["dir"] / "root.vk"), a["cfg"], str(b["dir"] / "root.json")).returncode != 0


def test_prove_agg_refuses_a_broken_hand_over(tmp_path):
    tasks, segs = [], []
    for i, s in enumerate([10, 17, 25, 32, 39]):   # segment 2 does not start where segment 1 ends
        k = ru.counter_segment(s, seed=i)
        segs.append(k)
        d = tmp_path / ("s%d" % i)
        d.mkdir()
        tasks.append(pm.write_task(str(d), k, identifier="seg-%d" % i))
    exe, cfg = pm.write_app(str(tmp_path), segs[0], PARAMS)
    r = pm.run_cli("prove-agg", exe, cfg, str(tmp_path), "1", "0:0/0:1", *tasks)
    assert r.returncode == 1 and "not satisfied" in r.stderr, r.stderr[-2000:]


def test_the_greedy_fold_reports_errors_and_folds_to_the_same_statement(tmp_path):
    """ZKHIP_AGG_GREEDY=1: `prove-agg` with TreeStream's fold without a fixed shape (what the guest flow runs).  Nine segments fold to a root
    that states the same app, states and commitments as the fixed grouping's, under the same key; a hand-over that breaks INSIDE a leaf
    node and one that breaks BETWEEN two leaf nodes (noticed by the fold that joins them) both end the run with the error -- no hang."""
    import os

    def run(name, starts, greedy=True, arity="0:0/0:1"):
        d = tmp_path / name
        d.mkdir()
        tasks, segs = [], []
        for i, s in enumerate(starts):
            k = ru.counter_segment(s, seed=i)
            segs.append(k)
            sd = d / ("s%d" % i)
            sd.mkdir()
            tasks.append(pm.write_task(str(sd), k, identifier="seg-%d" % i))
        exe, cfg = pm.write_app(str(d), segs[0], PARAMS)
        env = dict(os.environ, ZKHIP_AGG_GREEDY="1") if greedy else dict(os.environ)
        r = subprocess.run([pm.CLI, "prove-agg", exe, cfg, str(d), "1", arity] + tasks, capture_output=True, text=True, env=env, timeout=600)
        return d, cfg, r

    good = [10 + 7 * i for i in range(9)]
    d, cfg, r = run("greedy", good)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["segments"] == 9 and info["levels"] == 2
    d2, cfg2, r2 = run("fixed", good, greedy=False)
    assert r2.returncode == 0, r2.stderr[-3000:]
    a, b = info["root_public_values"], json.loads(r2.stdout.strip().splitlines()[-1])["root_public_values"]
    n = len(a)
    assert a[:n - 24] == b[:n - 24] and a[n - 16:] == b[n - 16:]          # all but the accumulator over the children
    assert (d / "root.vk").read_bytes() == (d2 / "root.vk").read_bytes()
    assert pm.run_cli("verify", str(d2 / "root.vk"), cfg2, str(d / "root.json")).returncode == 0
    # segment 2 does not start where segment 1 ends: inside the first leaf node
    bad = list(good)
    bad[2] += 1
    _, _, r = run("bad_leaf", bad)
    assert r.returncode == 1 and "not satisfied" in r.stderr, r.stderr[-2000:]
    # segment 4 does not start where segment 3 ends: every leaf node (four segments each) is fine, the fold that joins leaf nodes 0 and 1 is not
    bad = list(good)
    for i in range(4, 9):
        bad[i] += 1
    _, _, r = run("bad_fold", bad)
    assert r.returncode == 1 and "not satisfied" in r.stderr, r.stderr[-2000:]
