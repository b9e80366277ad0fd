"""GPU: Prover::setup -> gen_proof_universal through the C++ mirror produces the reference-shaped
StarkProof JSON whose proof bytes equal the oracle's, self-verifies, and rejects a bad witness."""
import json

import numpy as np
import pytest

import prover_mirror_util as pm
import zkvm_prover_amd as z
from zkvm_prover_amd import air

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 8, 4, 4)


def _airs():
    sa = air.SyntheticAir(width=20, n_free=7, n_bool=3, n_boundary=2, seed=9)
    tr, pv = sa.gen_trace(7, seed=1)
    fa = air.fibonacci_air()
    ftr, fpv = air.fibonacci_trace(5)
    return [dict(program=sa.program(), log_height=7, width=20, n_pvs=len(pv), trace=tr, pvs=pv),
            dict(program=fa.program(), log_height=5, width=2, n_pvs=3, trace=ftr, pvs=fpv)]


def test_gen_proof_universal_cli(ora, tmp_path):
    airs = _airs()
    exe, cfg = pm.write_app(str(tmp_path), airs, PARAMS)
    task = pm.write_task(str(tmp_path), airs, identifier="chunk-42")
    out = tmp_path / "proof.json"
    r = pm.run_cli("prove", exe, cfg, task, str(out))
    assert r.returncode == 0 and "proved chunk-42" in r.stdout, r.stderr
    js = json.loads(out.read_text())
    assert set(js) == {"proof", "user_pvs_proof", "baseline", "deferral_merkle_proofs", "stat"}
    assert set(js["stat"]) == {"total_cycles", "execution_time_mills", "proving_time_mills"}
    proof = pm.un_b64_bincode(js["proof"])
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    assert pm.un_b64_bincode(js["baseline"]) == bytes([7, 5])
    assert pm.run_cli("verify", exe, cfg, str(out)).returncode == 0


def test_unsatisfied_witness_fails_self_verification(tmp_path):
    airs = _airs()
    airs[0]["trace"] = airs[0]["trace"].copy()
    airs[0]["trace"][12, 3] = (int(airs[0]["trace"][12, 3]) + 1) % air.P
    exe, cfg = pm.write_app(str(tmp_path), airs, PARAMS)
    task = pm.write_task(str(tmp_path), airs)
    r = pm.run_cli("prove", exe, cfg, task, str(tmp_path / "o.json"))
    assert r.returncode == 1 and "kind 4" in r.stderr  # Error::VerifyProof from the mandatory self-check


def test_prove_many_reuses_keys_and_survives_reset(ora, tmp_path):
    """Three tasks through ONE Prover (the reference's multi-chunk test shape): same app, different
    witnesses and even different trace heights (keys are re-planned), reset() in between."""
    tasks, all_airs = [], []
    for i, (lh_a, lh_b) in enumerate([(7, 5), (7, 5), (6, 4)]):
        sa = air.SyntheticAir(width=20, n_free=7, n_bool=3, n_boundary=2, seed=9)
        tr, pv = sa.gen_trace(lh_a, seed=10 + i)
        fa = air.fibonacci_air()
        ftr, fpv = air.fibonacci_trace(lh_b, a0=i)
        airs = [dict(program=sa.program(), log_height=lh_a, width=20, n_pvs=len(pv), trace=tr, pvs=pv),
                dict(program=fa.program(), log_height=lh_b, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
        all_airs.append(airs)
        d = tmp_path / ("t%d" % i)
        d.mkdir()
        tasks.append(pm.write_task(str(d), airs, identifier="chunk-%d" % i))
    exe, cfg = pm.write_app(str(tmp_path), all_airs[0], PARAMS)
    r = pm.run_cli("prove-many", exe, cfg, str(tmp_path), *tasks)
    assert r.returncode == 0, r.stderr
    for i, airs in enumerate(all_airs):
        js = json.loads((tmp_path / ("chunk-%d.json" % i)).read_text())
        assert pm.un_b64_bincode(js["proof"]) == ora.stark_prove(PARAMS, airs).tobytes()


def _range_airs():
    u, m, prep = air.range_traces(7, 5, seed=3)
    nopv = np.zeros(0, np.uint32)
    return [dict(program=air.range_user_air().program(), log_height=7, width=4, n_pvs=0, trace=u, pvs=nopv),
            dict(program=air.range_table_air().program(), log_height=5, width=1, n_pvs=0, trace=m, pvs=nopv, prep=prep)]


def test_app_with_preprocessed_table_and_buses(ora, tmp_path):
    """The app file carries the range table; keygen commits it; the proof equals the oracle's; a verifier's app file
    with only the commitment accepts it; an app whose commitment does not match its table fails at keygen."""
    airs = _range_airs()
    exe, cfg = pm.write_app(str(tmp_path), airs, PARAMS)
    task = pm.write_task(str(tmp_path), airs, identifier="chunk-7")
    out = tmp_path / "proof.json"
    r = pm.run_cli("prove", exe, cfg, task, str(out))
    assert r.returncode == 0 and "proved chunk-7" in r.stdout, r.stderr
    proof = pm.un_b64_bincode(json.loads(out.read_text())["proof"])
    assert proof == ora.stark_prove(PARAMS, airs).tobytes()
    vk = [dict(a) for a in airs]
    vk[1]["prep_commit"] = ora.prep_commit(PARAMS, airs[1])
    (tmp_path / "vk").mkdir()
    exe_vk, _ = pm.write_app(str(tmp_path / "vk"), vk, PARAMS, with_tables=False)
    assert pm.run_cli("verify", exe_vk, cfg, str(out)).returncode == 0
    bad = [dict(a) for a in vk]
    bad[1]["prep_commit"] = bad[1]["prep_commit"].copy()
    bad[1]["prep_commit"][5] ^= 4
    (tmp_path / "bad").mkdir()
    exe_bad, _ = pm.write_app(str(tmp_path / "bad"), bad, PARAMS)
    r = pm.run_cli("prove", exe_bad, cfg, task, str(tmp_path / "o2.json"))
    assert r.returncode == 1 and "does not match the app's commitment" in r.stderr


def test_batch_prover_queue_and_api_throughput(ora, tmp_path):
    """BatchProver: five tasks queued over 2 Provers in flight on one GPU (the replacement of the reference's sequential
    chunk loop, crates/integration/src/testers/batch.rs:97-107) -- every proof byte-equal to the oracle's, results in task
    order; and the throughput form (witness resident, n proofs, self-verified) reports through the same API."""
    tasks, all_airs = [], []
    for i in range(5):
        sa = air.SyntheticAir(width=20, n_free=7, n_bool=3, n_boundary=2, seed=9)
        tr, pv = sa.gen_trace(8, seed=30 + i)
        ftr, fpv = air.fibonacci_trace(6, a0=i)
        airs = [dict(program=sa.program(), log_height=8, width=20, n_pvs=len(pv), trace=tr, pvs=pv),
                dict(program=air.fibonacci_air().program(), log_height=6, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
        all_airs.append(airs)
        d = tmp_path / ("t%d" % i)
        d.mkdir()
        tasks.append(pm.write_task(str(d), airs, identifier="seg-%d" % i))
    exe, cfg = pm.write_app(str(tmp_path), all_airs[0], PARAMS)
    r = pm.run_cli("prove-batch", exe, cfg, str(tmp_path), "2", *tasks)
    assert r.returncode == 0 and "batch: 5 proofs on 2 lanes" in r.stdout, r.stderr
    for i, airs in enumerate(all_airs):
        js = json.loads((tmp_path / ("seg-%d.json" % i)).read_text())
        assert pm.un_b64_bincode(js["proof"]) == ora.stark_prove(PARAMS, airs).tobytes()
    r = pm.run_cli("bench-many", exe, cfg, tasks[0], "12", "3")
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["proofs"] == 12 and out["inflight_per_gpu"] == 3 and out["proofs_per_s"] > 0 and out["self_verified"] is True
    # a bad witness in the queue surfaces as the reference's error kind (VerifyProof from the self-check)
    bad = [dict(a) for a in all_airs[1]]
    bad[0]["trace"] = bad[0]["trace"].copy()
    bad[0]["trace"][12, 3] = (int(bad[0]["trace"][12, 3]) + 1) % air.P
    (tmp_path / "bad").mkdir()
    tbad = pm.write_task(str(tmp_path / "bad"), bad, identifier="seg-bad")
    r = pm.run_cli("prove-batch", exe, cfg, str(tmp_path), "2", tasks[0], tbad, tasks[2])
    assert r.returncode == 1 and "kind 4" in r.stderr


def test_aggregation_tree_over_a_batch(ora, tmp_path):
    """Six independent segment proofs (no chained state) through a BatchProver, then the reference's aggregation tree shape (leaf
    arity 4, internal 3: 2 leaf nodes + 1 root) on REAL node circuits (tests/test_gpu_recursion.py covers the circuit itself): the
    root verifies under root.vk alone and its accumulator is the Poseidon2 chain over the segments' public values."""
    import recursion_util as ru

    tasks, all_airs = [], []
    for i in range(6):
        sa = air.SyntheticAir(width=20, n_free=7, n_bool=3, n_boundary=2, seed=9)
        tr, pv = sa.gen_trace(7, seed=50 + i)
        ftr, fpv = air.fibonacci_trace(5, a0=i)
        airs = [dict(program=sa.program(), log_height=7, width=20, n_pvs=len(pv), trace=tr, pvs=pv),
                dict(program=air.fibonacci_air().program(), log_height=5, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
        all_airs.append(airs)
        d = tmp_path / ("t%d" % i)
        d.mkdir()
        tasks.append(pm.write_task(str(d), airs, identifier="seg-%d" % i))
    seg_exe, cfg = pm.write_app(str(tmp_path), all_airs[0], PARAMS)
    out = tmp_path / "out"
    out.mkdir()
    r = pm.run_cli("prove-agg", seg_exe, cfg, str(out), "2", "-", *tasks)
    assert r.returncode == 0, r.stderr[-2000:]
    info = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert info["segments"] == 6 and info["nodes"] == 3 and info["levels"] == 2
    assert pm.run_cli("verify", str(out / "root.vk"), cfg, str(out / "root.json")).returncode == 0
    accs = [ru.leaf_accumulator([[a["pvs"] for a in s] for s in all_airs[i:i + 4]]) for i in (0, 4)]
    assert info["root_public_values"][8:16] == ru.internal_accumulator(accs).tolist()
    # the first leaf node's proof, re-derived: same circuit (the leaf circuit of the one aggregation key), same witness, the oracle's prover
    rc, _ = ru.one_key_circuits(PARAMS, [{k: a[k] for k in ("program", "log_height", "width", "n_pvs")} for a in all_airs[0]], None)
    st, npv = rc.witness([ora.stark_prove(PARAMS, a).tobytes() for a in all_airs[:4]], [[a["pvs"] for a in s] for s in all_airs[:4]])
    assert st == 0
    leaf0 = json.loads((out / "agg-0-0.json").read_text())
    assert pm.un_b64_bincode(leaf0["proof"]) == ora.stark_prove(PARAMS, ru.node_instance(rc, npv)).tobytes()
    # bad command lines are errors, not crashes
    assert pm.run_cli("prove-agg", seg_exe, cfg, str(out), "two", "-", *tasks).returncode in (1, 2)


def test_batch_prover_multi_device_queue_on_one_gpu(ora, tmp_path):
    """BatchProver over a device LIST (segments spread over the GPUs of a node): ZKHIP_BATCH_DEVICES=0,0,0 builds
    three lane groups as if there were three GPUs -- all mapped onto this box's one device -- so the cross-device queue (contexts
    bound per call, one host thread per lane, results in task order) runs in the GPU suite.  Seven tasks, six lanes: every proof
    equals the oracle's."""
    import os
    import subprocess

    tasks, all_airs = [], []
    for i in range(7):
        sa = air.SyntheticAir(width=20, n_free=7, n_bool=3, n_boundary=2, seed=9)
        tr, pv = sa.gen_trace(7, seed=60 + i)
        ftr, fpv = air.fibonacci_trace(5, a0=i)
        airs = [dict(program=sa.program(), log_height=7, width=20, n_pvs=len(pv), trace=tr, pvs=pv),
                dict(program=air.fibonacci_air().program(), log_height=5, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
        all_airs.append(airs)
        d = tmp_path / ("t%d" % i)
        d.mkdir()
        tasks.append(pm.write_task(str(d), airs, identifier="seg-%d" % i))
    exe, cfg = pm.write_app(str(tmp_path), all_airs[0], PARAMS)
    env = dict(os.environ, ZKHIP_BATCH_DEVICES="0,0,0")
    r = subprocess.run([pm.CLI, "prove-batch", exe, cfg, str(tmp_path), "2", *tasks], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "7 proofs on 6 lanes" in r.stdout, r.stderr
    for i, airs in enumerate(all_airs):
        js = json.loads((tmp_path / ("seg-%d.json" % i)).read_text())
        assert pm.un_b64_bincode(js["proof"]) == ora.stark_prove(PARAMS, airs).tobytes()
    # a device that does not exist is refused by that lane, and the whole batch reports it
    env = dict(os.environ, ZKHIP_BATCH_DEVICES="0,63")
    r = subprocess.run([pm.CLI, "prove-batch", exe, cfg, str(tmp_path), "1", *tasks[:2]], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "gfx950" in r.stderr
