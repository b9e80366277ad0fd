"""CPU: the product's codec of the reference's stored-proof container (include/zkhip_codec.hpp through the C ABI).

 * every stored reference proof decodes to the last byte and re-encodes to the identical bytes (all eight in the build
   container from /root/reference; one committed copy everywhere), and the summary agrees with the independent
   Python reader (tests/refproof_v1.py);
 * a proof of this backend (made by the oracle here -- byte-identical to the HIP prover's, tests/test_gpu_stark.py)
   converts to the v1 container and back without loss, for every proof flavour (plain, mixed heights, preprocessed,
   LogUp), and what comes back verifies;
 * malformed containers are rejected, never crash.
"""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import refproof_v1 as rp
import zkvm_prover_amd as z
from zkvm_prover_amd import air

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/crates"
NOPV = np.zeros(0, np.uint32)
PARAMS = (1, 0, 10, 0, 5)   # commit_pow_bits = 0: the v1 container carries no commit-phase witnesses


def _committed_blob():
    with open(os.path.join(HERE, "golden", "ref_proofs", "chunk-proof-feynman.proofs.bin"), "rb") as f:
        return f.read()


def _check_blob(blob):
    s = z.proof_decode_v1(blob, z.V1_VEC)
    assert z.proof_reencode_v1(blob, z.V1_VEC) == blob
    p = rp.decode_proofs(blob)[0]
    sh = rp.shape_of(p)
    assert s["n_proofs"] == 1 and s["n_airs"] == sh["n_airs"] and s["log_degree"] == sh["log_degrees"]
    assert s["n_queries"] == sh["n_queries"] and s["n_fri_layers"] == sh["n_fri_layers"]
    assert s["n_final_poly"] == sh["n_final_poly"] and s["n_input_batches"] == len(sh["batches"])
    assert s["n_main_commits"] == len(p["main_trace"]) == 2 and s["n_after_challenge_commits"] == 1
    assert s["log_blowup"] == 2 and s["has_logup_pow"] == 1
    assert s["log_max_height"] == max(b["log_height"] for b in sh["batches"])
    return s


def test_committed_reference_proof_roundtrips_byte_exactly():
    blob = _committed_blob()
    with open(os.path.join(HERE, "golden", "ref_v1_vectors.json")) as f:
        src = {s["name"]: s for s in json.load(f)["sources"]}
    assert hashlib.sha256(blob).hexdigest() == src["chunk-proof-feynman.json"]["sha256"]
    s = _check_blob(blob)
    assert s["n_airs"] == 17 and s["n_queries"] == 44 and s["n_fri_layers"] == 21 and s["log_max_height"] == 23
    # the single proof inside also round-trips as bincode(Proof<SC>)
    assert z.proof_reencode_v1(blob[8:], z.V1_SINGLE) == blob[8:]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_all_eight_reference_proofs_roundtrip_byte_exactly():
    with open(os.path.join(HERE, "golden", "ref_v1_vectors.json")) as f:
        sources = json.load(f)["sources"]
    assert len(sources) == 8
    for s in sources:
        d = json.load(open(os.path.join(REF, s["file"])))
        blob = base64.b64decode(d["proof"]["proofs"])
        assert hashlib.sha256(blob).hexdigest() == s["sha256"]
        _check_blob(blob)


def test_malformed_containers_are_rejected():
    blob = _committed_blob()
    for bad in (blob[:-1], blob + b"\0", blob[:1000], b"", b"\xff" * 64):
        with pytest.raises(z.ZkhipError):
            z.proof_decode_v1(bad, z.V1_VEC)
    # a length prefix that promises more than the input holds
    bad = bytearray(blob)
    bad[8:16] = (1 << 40).to_bytes(8, "little")
    with pytest.raises(z.ZkhipError):
        z.proof_decode_v1(bytes(bad), z.V1_VEC)
    # a word >= p
    bad = bytearray(blob)
    bad[24:28] = (0xFFFFFFFF).to_bytes(4, "little")
    with pytest.raises(z.ZkhipError):
        z.proof_decode_v1(bytes(bad), z.V1_VEC)
    rng = np.random.default_rng(0)
    for _ in range(200):  # random corruption of structure bytes: error or success, never a crash
        bad = bytearray(blob[:20000])
        pos = int(rng.integers(0, 600))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            z.proof_decode_v1(bytes(bad), z.V1_VEC)
        except z.ZkhipError:
            pass


def _cases():
    sa = air.SyntheticAir(width=12, n_free=5, n_bool=2, n_boundary=2, seed=3)
    tr, pv = sa.gen_trace(6, seed=11)
    ftr, fpv = air.fibonacci_trace(4)
    plain = [dict(program=sa.program(), log_height=6, width=12, n_pvs=len(pv), trace=tr, pvs=pv),
             dict(program=air.fibonacci_air().program(), log_height=4, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
    s, t = air.lookup_traces(6, 4, seed=1)
    logup = [dict(program=air.lookup_sender_air().program(), log_height=6, width=3, n_pvs=0, trace=s, pvs=NOPV),
             dict(program=air.fibonacci_air().program(), log_height=4, width=2, n_pvs=3, trace=ftr, pvs=fpv),
             dict(program=air.lookup_table_air().program(), log_height=4, width=3, n_pvs=0, trace=t, pvs=NOPV)]
    u, m, prep = air.range_traces(6, 4, seed=1)
    prepc = [dict(program=air.range_user_air().program(), log_height=6, width=4, n_pvs=0, trace=u, pvs=NOPV),
             dict(program=air.range_table_air().program(), log_height=4, width=1, n_pvs=0, trace=m, pvs=NOPV, prep=prep)]
    return {"plain": plain, "logup": logup, "prep": prepc}


@pytest.mark.parametrize("name", ["plain", "logup", "prep"])
@pytest.mark.parametrize("params", [PARAMS, (2, 1, 7, 0, 3)])
def test_backend_proof_to_v1_and_back(ora, name, params):
    airs = _cases()[name]
    pvs = [a["pvs"] for a in airs]
    proof = ora.stark_prove(params, airs).tobytes()
    vk = []
    for a in airs:
        v = {k: a[k] for k in ("program", "log_height", "width", "n_pvs")}
        if a.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(params, a)
        vk.append(v)
    assert z.verify(params, vk, pvs, proof) == 0
    v1 = z.proof_to_v1(params, vk, pvs, proof)
    # the product's writer against the independent Python reader
    p = rp.decode_proofs((1).to_bytes(8, "little") + v1)[0]
    lay = z.proof_layout(params, vk)
    words = np.frombuffer(proof, dtype=np.uint32)
    assert [rp.from_monty(x) for x in p["main_trace"][0]] == words[lay["root_main"]:lay["root_main"] + 8].tolist()
    assert [rp.from_monty(x) for x in p["quotient"]] == words[lay["root_quot"]:lay["root_quot"] + 8].tolist()
    assert len(p["fri"]["query_proofs"]) == params[2] and len(p["fri"]["final_poly"]) == 1 << params[1]
    assert [a["degree"] for a in p["per_air"]] == [1 << a["log_height"] for a in airs]
    assert [[rp.from_monty(x) for x in a["pvs"]] for a in p["per_air"]] == [list(map(int, pv)) for pv in pvs]
    n_prep = sum(1 for a in airs if a.get("prep") is not None)
    has_lu = name == "logup" or name == "prep"
    assert len(p["opened"]["preprocessed"]) == n_prep and len(p["after_challenge"]) == (1 if has_lu else 0)
    assert len(p["fri"]["query_proofs"][0]["input_proof"]) == n_prep + 2 + (1 if has_lu else 0)
    s = z.proof_decode_v1(v1, z.V1_SINGLE)
    assert s["log_blowup"] == params[0] and s["n_airs"] == len(airs)
    assert z.proof_reencode_v1(v1, z.V1_SINGLE) == v1
    # and back: identical bytes, identical public values, still verifies
    back, pvs_back = z.proof_from_v1(params, vk, v1)
    assert back == proof
    assert [list(map(int, x)) for x in pvs_back] == [list(map(int, x)) for x in pvs]
    assert z.verify(params, vk, pvs_back, back) == 0
    # a container of the wrong shape for this key is refused
    other = _cases()["plain" if name != "plain" else "logup"]
    ovk = [{k: a[k] for k in ("program", "log_height", "width", "n_pvs")} for a in other]
    if name != "prep" and not any(a.get("prep") is not None for a in other):
        with pytest.raises(z.ZkhipError):
            z.proof_from_v1(params, ovk, v1)


def test_to_v1_refuses_commit_phase_pow(ora):
    airs = _cases()["plain"]
    params = (1, 0, 4, 3, 3)
    proof = ora.stark_prove(params, airs).tobytes()
    vk = [{k: a[k] for k in ("program", "log_height", "width", "n_pvs")} for a in airs]
    with pytest.raises(z.ZkhipError):
        z.proof_to_v1(params, vk, [a["pvs"] for a in airs], proof)


def test_cli_decodes_a_reference_proof_file(tmp_path):
    """`prove_cli decode-v1` on a proof file shaped like the reference's (JSON with base64 `proofs` / `public_values`)."""
    import subprocess

    blob = _committed_blob()
    with open(os.path.join(HERE, "golden", "ref_proofs", "chunk-proof-feynman.public_values.bin"), "rb") as f:
        pv = f.read()
    path = tmp_path / "chunk-proof.json"
    path.write_text(json.dumps({"metadata": {}, "proof": {"proofs": base64.b64encode(blob).decode(),
                                                          "public_values": base64.b64encode(pv).decode()}, "vk": "", "git_version": "x"}))
    cli = os.path.join(os.path.dirname(HERE), "zkvm-prover_amd", "prove_cli")
    r = subprocess.run([cli, "decode-v1", str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["roundtrip_exact"] is True and out["n_airs"] == 17 and out["n_queries"] == 44 and out["n_main_commits"] == 2
    assert out["log_blowup"] == 2 and out["user_public_values"] == 32 and out["log_max_height"] == 23
