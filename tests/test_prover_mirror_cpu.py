"""CPU: the C++ mirror of the reference's Prover API (include/zkhip_prover.hpp) -- file loading
(setup), the StarkProof JSON container (base64(bincode(Vec<u8>)) fields, crates/types/src/proof.rs:52-67)
and verify_stark_proof -- exercised through prove_cli on proofs made by the oracle."""
import json
import os

import numpy as np
import pytest

import prover_mirror_util as pm
from zkvm_prover_amd import air

PARAMS = (1, 0, 8, 4, 4)


def _airs():
    sa = air.SyntheticAir(width=20, n_free=7, n_bool=3, n_boundary=2, seed=9)
    tr, pv = sa.gen_trace(6, seed=1)
    fa = air.fibonacci_air()
    ftr, fpv = air.fibonacci_trace(5)
    return [dict(program=sa.program(), log_height=6, width=20, n_pvs=len(pv), trace=tr, pvs=pv),
            dict(program=fa.program(), log_height=5, width=2, n_pvs=3, trace=ftr, pvs=fpv)]


def test_cli_verifies_oracle_proof_in_reference_container(ora, tmp_path):
    assert os.path.exists(pm.CLI), "prove_cli not built (python -c 'import __graft_entry__ as g; g.build()')"
    airs = _airs()
    proof = ora.stark_prove(PARAMS, airs).tobytes()
    exe, cfg = pm.write_app(str(tmp_path), airs, PARAMS)
    pj = tmp_path / "proof.json"
    pj.write_text(pm.stark_proof_json(proof, airs, proving_ms=12))
    r = pm.run_cli("verify", exe, cfg, str(pj))
    assert r.returncode == 0 and "verified" in r.stdout and "proving_time_mills=12" in r.stdout, r.stderr
    # a flipped proof byte is Error::VerifyProof (kind 4); mismatching public values too
    bad = bytearray(proof)
    bad[200] ^= 1
    pj.write_text(pm.stark_proof_json(bytes(bad), airs))
    r = pm.run_cli("verify", exe, cfg, str(pj))
    assert r.returncode == 1 and "kind 4" in r.stderr
    airs2 = _airs()
    airs2[1]["pvs"] = airs2[1]["pvs"].copy()
    airs2[1]["pvs"][2] += 1
    pj.write_text(pm.stark_proof_json(proof, airs2))
    assert pm.run_cli("verify", exe, cfg, str(pj)).returncode == 1


def _range_airs():
    u, m, prep = air.range_traces(6, 4, seed=3)
    nopv = np.zeros(0, np.uint32)
    return [dict(program=air.range_user_air().program(), log_height=6, width=4, n_pvs=0, trace=u, pvs=nopv),
            dict(program=air.range_table_air().program(), log_height=4, width=1, n_pvs=0, trace=m, pvs=nopv, prep=prep)]


def test_cli_verifies_with_the_apps_preprocessed_commitment(ora, tmp_path):
    """A verifier's app file holds the commitment of the range table, not the table (the exe-commit analogue,
    crates/verifier/src/verifier.rs:77-80)."""
    airs = _range_airs()
    proof = ora.stark_prove(PARAMS, airs).tobytes()
    vk_airs = [dict(a) for a in airs]
    vk_airs[1]["prep_commit"] = ora.prep_commit(PARAMS, airs[1])
    exe, cfg = pm.write_app(str(tmp_path), vk_airs, PARAMS, with_tables=False)
    pj = tmp_path / "proof.json"
    pj.write_text(pm.stark_proof_json(proof, airs))
    r = pm.run_cli("verify", exe, cfg, str(pj))
    assert r.returncode == 0 and "verified" in r.stdout, r.stderr
    # another table's commitment, or none at all: Error::VerifyProof
    vk_airs[1]["prep_commit"] = vk_airs[1]["prep_commit"].copy()
    vk_airs[1]["prep_commit"][0] ^= 1
    exe2, _ = pm.write_app(str(tmp_path / "b"), vk_airs, PARAMS, with_tables=False) if os.makedirs(tmp_path / "b") is None else None
    r = pm.run_cli("verify", exe2, cfg, str(pj))
    assert r.returncode == 1 and "kind 4" in r.stderr
    del vk_airs[1]["prep_commit"]
    os.makedirs(tmp_path / "c")
    exe3, _ = pm.write_app(str(tmp_path / "c"), vk_airs, PARAMS, with_tables=True)
    r = pm.run_cli("verify", exe3, cfg, str(pj))
    assert r.returncode == 1 and "kind 4" in r.stderr and "commitment" in r.stderr


def test_setup_errors_match_reference_kinds(tmp_path):
    airs = _airs()
    exe, cfg = pm.write_app(str(tmp_path), airs, PARAMS)
    # missing exe / config -> Error::Setup (kind 1), like read_app_exe / read_app_config failures
    r = pm.run_cli("verify", str(tmp_path / "nope.zkair"), cfg, str(tmp_path / "p.json"))
    assert r.returncode == 1 and "kind 1" in r.stderr and "failed to read or deserialize" in r.stderr
    (tmp_path / "bad.toml").write_text("[app_fri_params.fri_params]\nlog_blowup = 1\n")
    r = pm.run_cli("verify", exe, str(tmp_path / "bad.toml"), str(tmp_path / "p.json"))
    assert r.returncode == 1 and "kind 1" in r.stderr and "num_queries" in r.stderr or "log_final_poly_len" in r.stderr


def test_prove_without_gpu_fails_loudly(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    airs = _airs()
    exe, cfg = pm.write_app(str(tmp_path), airs, PARAMS)
    task = pm.write_task(str(tmp_path), airs)
    r = pm.run_cli("prove", exe, cfg, task, str(tmp_path / "out.json"))
    assert r.returncode == 1 and "kind 2" in r.stderr and "gfx950" in r.stderr  # Error::Keygen, no CPU fallback


def test_aggregation_plan_shape():
    """AggregationPlan::build with the reference's tree (crates/prover/src/prover/mod.rs:57-60: leaf arity 4, internal 3):
    every segment is consumed exactly once, every node of a level exactly once by the level above, one root."""
    import json

    import prover_mirror_util as pm

    for n in (1, 2, 4, 5, 12, 13, 37, 100):
        r = pm.run_cli("agg-plan", str(n))
        assert r.returncode == 0, r.stderr
        plan = json.loads(r.stdout)
        lv = plan["levels"]
        assert sorted(c for node in lv[0] for c in node) == list(range(n))
        assert all(1 <= len(node) <= 4 for node in lv[0]) and all(len(node) == 4 for node in lv[0][:-1])
        for below, above in zip(lv, lv[1:]):
            assert sorted(c for node in above for c in node) == list(range(len(below)))
            assert all(1 <= len(node) <= 3 for node in above)
        assert len(lv[-1]) == 1 and plan["n_nodes"] == sum(len(l) for l in lv)
    r = pm.run_cli("agg-plan", "9", "2", "2")
    assert [len(l) for l in json.loads(r.stdout)["levels"]] == [5, 3, 2, 1]
