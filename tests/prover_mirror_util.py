"""Helpers to drive the C++ Prover mirror (include/zkhip_prover.hpp via zkvm-prover_amd/prove_cli)."""
import base64
import json
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "zkvm-prover_amd", "prove_cli")
AIRSET_MAGIC = 0x58414B5A
AIRSET_MAGIC_V2 = 0x58414B5B   # + per chip {has_prep, [log_height, prep_len, prep..., has_commit, commit(8)]}

# same keys as the reference's circuit config (crates/circuits/chunk-circuit/openvm.toml:1-6)
TOML = """[app_fri_params.fri_params]
log_blowup = {0}
log_final_poly_len = {1}
num_queries = {2}
commit_proof_of_work_bits = {3}
query_proof_of_work_bits = {4}

[app_vm_config.rv32i]
"""


def write_app(tmp, airs, params, with_tables=True):
    """airs: dicts with program/width/n_pvs[/prep/prep_commit/log_height].  with_tables=False writes a verifier's
    app: commitments of the preprocessed traces but not the tables."""
    v2 = any(a.get("prep") is not None or a.get("prep_commit") is not None for a in airs)
    words = [AIRSET_MAGIC_V2 if v2 else AIRSET_MAGIC, len(airs)]
    for a in airs:
        prog = np.asarray(a["program"], dtype=np.uint32)
        words += [a["width"], a["n_pvs"], prog.size] + prog.tolist()
        if not v2:
            continue
        has = a.get("prep") is not None or a.get("prep_commit") is not None
        words.append(1 if has else 0)
        if has:
            prep = np.asarray(a["prep"], dtype=np.uint32).reshape(-1) if (with_tables and a.get("prep") is not None) else np.zeros(0, np.uint32)
            words += [a["log_height"], prep.size] + prep.tolist()
            pc = a.get("prep_commit")
            words += ([1] + np.asarray(pc, dtype=np.uint32).tolist()) if pc is not None else [0]
    exe = os.path.join(tmp, "app.zkair")
    np.array(words, dtype=np.uint32).tofile(exe)
    cfg = os.path.join(tmp, "openvm.toml")
    with open(cfg, "w") as f:
        f.write(TOML.format(*params))
    return exe, cfg


def write_task(tmp, airs, identifier="chunk-0"):
    out = os.path.join(tmp, "task.bin")
    with open(out, "wb") as f:
        idb = identifier.encode()
        f.write(struct.pack("<I", len(idb)) + idb + struct.pack("<I", len(airs)))
        for a in airs:
            w = np.concatenate([np.array([a["log_height"], len(a["pvs"])], dtype=np.uint32),
                                np.asarray(a["pvs"], dtype=np.uint32), np.asarray(a["trace"], dtype=np.uint32).reshape(-1)])
            b = w.tobytes()
            f.write(struct.pack("<Q", len(b)) + b)
    return out


def b64_bincode(b):
    return base64.b64encode(struct.pack("<Q", len(b)) + b).decode()


def un_b64_bincode(s):
    raw = base64.b64decode(s)
    (n,) = struct.unpack("<Q", raw[:8])
    assert n == len(raw) - 8
    return raw[8:]


def stark_proof_json(proof_bytes, airs, proving_ms=0):
    pvs = b"".join(np.asarray(a["pvs"], dtype=np.uint32).tobytes() for a in airs)
    baseline = bytes(a["log_height"] for a in airs)
    return json.dumps({"proof": b64_bincode(proof_bytes), "user_pvs_proof": b64_bincode(pvs),
                       "baseline": b64_bincode(baseline), "deferral_merkle_proofs": b64_bincode(b""),
                       "stat": {"total_cycles": 0, "execution_time_mills": 0, "proving_time_mills": proving_ms}})


def run_cli(*args):
    return subprocess.run([CLI] + list(args), capture_output=True, text=True)


def read_vk(path):
    """A verifying key in app-file form (include/zkhip_prover.hpp encode_app_exe v2, tables omitted) as verifying AIR dicts; an
    aggregation / join key's trailer [magic | leaf commitment (8) | app digest (8)] comes back as the second value (or None)."""
    import numpy as np

    w = np.frombuffer(open(path, "rb").read(), dtype=np.uint32)
    assert w[0] == 0x58414B5B
    n, p, airs = int(w[1]), 2, []
    for _ in range(n):
        width, n_pvs, plen = int(w[p]), int(w[p + 1]), int(w[p + 2])
        p += 3
        a = dict(width=width, n_pvs=n_pvs, program=w[p:p + plen].copy())
        p += plen
        has_prep = int(w[p])
        p += 1
        if has_prep:
            a["log_height"], prep_len = int(w[p]), int(w[p + 1])
            p += 2 + prep_len
            has_commit = int(w[p])
            p += 1
            if has_commit:
                a["prep_commit"] = w[p:p + 8].copy()
                p += 8
        airs.append(a)
    trailer = None
    if len(w) - p == 17:
        trailer = dict(magic=int(w[p]), leaf_commit=w[p + 1:p + 9].copy(), app_digest=w[p + 9:p + 17].copy())
    return airs, trailer
