"""GPU: deferral end to end (crates/prover/src/prover/mod.rs:200-282 `enable_deferral`, crates/integration/src/lib.rs:461-514
`compute_deferral_data`, :556-571 `prove_task_with_deferral`; guest side crates/types/circuit/src/lib.rs:137-154 `verify_stark`).
Three runs of a CHILD guest (Fibonacci; the reference: chunk proofs) are folded to three root proofs under the child's ONE aggregation
key; a PARENT guest (the reference: the batch circuit) states one claim per child -- input commitment, the child app's program
commitments (constants of the parent guest), the child's public values -- in its deferral region; `prove-deferral` proves the parent,
the deferral node over the three child roots, and their join: ONE StarkProof.
  * the parent's root verifies with verify-guest; its statement's deferral accumulator == the independent restatement of the chain
    over the three children's statements (tests/recursion_util.py), i.e. it binds the three child commitments;
  * the deferral node's proof == the oracle's proof of the same circuit and witness, byte for byte;
  * a flipped byte in a child proof, a child of another executable image (another exe commitment), reordered children, a tampered
    opening of the claims and a tampered accumulator all fail."""
import json
import subprocess

import numpy as np
import pytest

import zkvm_prover_amd as z

import prover_mirror_util as pm
import recursion_util as ru
import rv32_model as rv
from test_vm_cpu import deferral_guest_program, deferral_guest_stdin, fib_program

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 4, 3, 3)
NOPV = ru.NOPV
N_STMT = 50


def prove_child(d, elf, cfg, n):
    d.mkdir()
    (d / "stdin.bin").write_bytes(int(n).to_bytes(4, "little"))
    r = subprocess.run([pm.CLI, "prove-elf", str(elf), str(d / "stdin.bin"), str(d), str(cfg), "9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    root = json.loads((d / "root.json").read_text())
    upv = pm.un_b64_bincode(root["user_pvs_proof"])
    return dict(dir=d, json=str(d / "root.json"), root=root, stmt=np.frombuffer(upv[:4 * N_STMT], dtype=np.uint32), pvs=upv[4 * N_STMT:4 * N_STMT + 32],
                openings=np.frombuffer(upv[4 * N_STMT + 32:], dtype=np.uint32))


def test_three_children_one_parent_root(ora, tmp_path):
    cfg = tmp_path / "openvm.toml"
    cfg.write_text(pm.TOML.format(*PARAMS))
    child_elf = tmp_path / "child.elf"
    child_elf.write_bytes(rv.elf_bytes(fib_program()))
    kids = [prove_child(tmp_path / ("c%d" % i), child_elf, cfg, n) for i, n in enumerate((100, 150, 210))]
    child_vk = str(kids[0]["dir"] / "root.vk")
    assert all((k["dir"] / "root.vk").read_bytes() == (kids[0]["dir"] / "root.vk").read_bytes() for k in kids)   # one key
    # the child app's program commitments: constants of the parent guest
    r = pm.run_cli("program-commit", str(child_elf), child_vk, str(cfg))
    assert r.returncode == 0, r.stderr
    pc = json.loads(r.stdout)
    data = b"".join(int(x).to_bytes(4, "little") for x in pc["exe"] + pc["vm"])
    parent = deferral_guest_program()
    parent_elf = tmp_path / "parent.elf"
    parent_elf.write_bytes(rv.elf_bytes(parent, data=data))
    r = pm.run_cli("program-commit", str(parent_elf), child_vk, str(cfg))     # (only for the region's address: it follows the data image)
    assert json.loads(r.stdout)["deferral_base"] == 0x00402000
    witness = tmp_path / "witness.bin"
    witness.write_bytes(b"".join(k["pvs"] for k in kids))
    out = tmp_path / "parent"
    out.mkdir()

    def prove(outdir, jsons, wit=witness, elf=parent_elf):
        return subprocess.run([pm.CLI, "prove-deferral", str(elf), str(cfg), str(outdir), "9", child_vk, str(cfg), str(wit)] + jsons, capture_output=True, text=True)

    r = prove(out, [k["json"] for k in kids])
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["children"] == 3 and info["verified"]
    # the whole statement about the parent guest, under the JOIN key
    v = pm.run_cli("verify-guest", str(parent_elf), str(out / "root.vk"), str(cfg), str(out / "root.json"))
    assert v.returncode == 0, v.stderr
    root = json.loads((out / "root.json").read_text())
    upv = pm.un_b64_bincode(root["user_pvs_proof"])
    stmt = np.frombuffer(upv[:4 * 58], dtype=np.uint32)
    pv = upv[4 * 58:4 * 58 + 32]
    fib = lambda n: (lambda f: [f := (f[1], (f[0] + f[1]) & 0xFFFFFFFF) for _ in range(n)][-1][0])((0, 1))   # noqa: E731
    assert int.from_bytes(pv[:4], "little") == 3 and int.from_bytes(pv[4:8], "little") == (fib(100) + fib(150) + fib(210)) & 0xFFFFFFFF
    # the deferral accumulator binds the three children: the chain over their statements, restated independently
    cells = [np.array([k["pvs"][2 * j] | (k["pvs"][2 * j + 1] << 8) for j in range(16)], np.uint32) for k in kids]
    want = ru.deferral_chain(np.zeros(8, np.uint32), [ru.deferral_claim(k["stmt"], c) for k, c in zip(kids, cells)])
    assert stmt[50:58].tolist() == want.tolist() == info["deferral_state"]
    # ... and the claims use the program commitments of the child app
    claim0 = ru.deferral_claim(kids[0]["stmt"], cells[0])
    assert claim0[1].tolist() == pc["exe"] and claim0[2].tolist() == pc["vm"]
    # the join key's trailer pins the parent app; the generic verifier checks it too
    assert pm.run_cli("verify", str(out / "root.vk"), str(cfg), str(out / "root.json")).returncode == 0
    # the deferral node's proof == the oracle's proof of the same circuit and witness
    key_airs, trailer = pm.read_vk(child_vk)
    D = z.RecursionCircuit(PARAMS, key_airs, 4, stmt="deferral")
    aux = [np.concatenate([c, k["openings"][8:8 * 28]]) for k, c in zip(kids, cells)]
    st, dpv = D.witness([pm.un_b64_bincode(k["root"]["proof"]) for k in kids], [[NOPV, NOPV, k["stmt"]] for k in kids], aux=aux)
    assert st == 0, D.last_error()
    assert dpv[8:].tolist() == want.tolist()
    dj = json.loads((out / "deferral.json").read_text())
    assert pm.un_b64_bincode(dj["proof"]) == ora.stark_prove(PARAMS, ru.node_instance(D, dpv)).tobytes()

    # ---- refusals ----
    # a flipped byte in a child proof
    bad = dict(kids[1]["root"])
    pb = bytearray(pm.un_b64_bincode(bad["proof"]))
    pb[4 * 500] ^= 1
    bad["proof"] = pm.b64_bincode(bytes(pb))
    (tmp_path / "bad_child.json").write_text(json.dumps(bad))
    x = tmp_path / "x1"
    x.mkdir()
    r = prove(x, [kids[0]["json"], str(tmp_path / "bad_child.json"), kids[2]["json"]])
    assert r.returncode != 0 and "does not verify under the child aggregation key" in r.stderr
    # a child of another executable image: same program (same aggregation key), other initial memory -> another exe commitment than
    # the one the parent guest holds
    other_elf = tmp_path / "other.elf"
    other_elf.write_bytes(rv.elf_bytes(fib_program(), data=b"\\x01\\x02\\x03\\x04"))
    other = prove_child(tmp_path / "o", other_elf, cfg, 150)
    assert (other["dir"] / "root.vk").read_bytes() == (kids[0]["dir"] / "root.vk").read_bytes()
    assert other["pvs"] == kids[1]["pvs"]
    x = tmp_path / "x2"
    x.mkdir()
    r = prove(x, [kids[0]["json"], other["json"], kids[2]["json"]])
    assert r.returncode != 0 and "claims are not the ones the deferral node verified" in r.stderr, r.stderr[-2000:]
    # reordered children (the witness keeps the order the guest was told)
    x = tmp_path / "x3"
    x.mkdir()
    r = prove(x, [kids[1]["json"], kids[0]["json"], kids[2]["json"]])
    assert r.returncode != 0 and "claims are not the ones the deferral node verified" in r.stderr
    # a child of ANOTHER app (the parent guest itself as a child): not under the child aggregation key
    # (its proof is a join: another shape altogether)
    x = tmp_path / "x4"
    x.mkdir()
    r = prove(x, [kids[0]["json"], str(out / "root.json")])
    assert r.returncode != 0
    # the final proof: a tampered opening of the claims, a tampered accumulator, a tampered claim count
    def variant(name, field, pos):
        sp = dict(root)
        b = bytearray(pm.un_b64_bincode(sp[field]))
        b[pos] ^= 1
        sp[field] = pm.b64_bincode(bytes(b))
        (tmp_path / name).write_text(json.dumps(sp))
        return str(tmp_path / name)

    for name, field, pos in (("t1.json", "deferral_merkle_proofs", 4 * 64 + 1), ("t2.json", "deferral_merkle_proofs", 4 * 4096 + 5),
                             ("t3.json", "deferral_merkle_proofs", 0), ("t4.json", "user_pvs_proof", 4 * 52)):
        assert pm.run_cli("verify-guest", str(parent_elf), str(out / "root.vk"), str(cfg), variant(name, field, pos)).returncode != 0, name
    # the parent's proof does not verify under the CHILD's key, nor a child's under the parent's
    assert pm.run_cli("verify-guest", str(parent_elf), child_vk, str(cfg), str(out / "root.json")).returncode != 0
    assert pm.run_cli("verify-guest", str(child_elf), str(out / "root.vk"), str(cfg), kids[0]["json"]).returncode != 0


def test_bundle_over_batches_three_layers(ora, tmp_path):
    """chunk -> batch -> bundle (crates/integration/src/testers: every layer defers the verification of the one below).  Two chunk
    proofs (Fibonacci) -> two BATCH proofs (a deferring guest over chunk roots: proofs under the batch's JOIN key) -> one BUNDLE proof (the
    same guest program with the batch app's commitments in its data segment, over the two batch proofs): the bundle's deferral node takes
    JOIN proofs as children -- it opens each batch's deferral region in that batch's final memory root and chains its claims in the
    circuit, what verify-guest does on the host for a single join.
      * the bundle verifies with verify-guest under ITS join key; its chain == the independent restatement over the batch statements;
      * the bundle's deferral node proof == the oracle's proof of the same circuit and witness, byte for byte;
      * a batch proof with a tampered region opening, a witness with a wrong claim count and a region cell that does not open are refused."""
    cfg = tmp_path / "openvm.toml"
    cfg.write_text(pm.TOML.format(*PARAMS))
    chunk_elf = tmp_path / "chunk.elf"
    chunk_elf.write_bytes(rv.elf_bytes(fib_program()))
    chunks = [prove_child(tmp_path / ("c%d" % i), chunk_elf, cfg, n) for i, n in enumerate((100, 150))]
    chunk_vk = str(chunks[0]["dir"] / "root.vk")
    pc = json.loads(pm.run_cli("program-commit", str(chunk_elf), chunk_vk, str(cfg)).stdout)
    batch_elf = tmp_path / "batch.elf"
    batch_elf.write_bytes(rv.elf_bytes(deferral_guest_program(), data=b"".join(int(x).to_bytes(4, "little") for x in pc["exe"] + pc["vm"])))

    def prove_deferral(outdir, elf, child_vk, jsons, pvs):
        outdir.mkdir()
        (outdir / "witness.bin").write_bytes(b"".join(pvs))
        r = subprocess.run([pm.CLI, "prove-deferral", str(elf), str(cfg), str(outdir), "9", child_vk, str(cfg), str(outdir / "witness.bin")] + jsons,
                           capture_output=True, text=True)
        return r

    batches = []
    for i, ks in enumerate(([chunks[0], chunks[1]], [chunks[1]])):
        d = tmp_path / ("b%d" % i)
        r = prove_deferral(d, batch_elf, chunk_vk, [k["json"] for k in ks], [k["pvs"] for k in ks])
        assert r.returncode == 0, r.stderr[-3000:]
        root = json.loads((d / "root.json").read_text())
        upv = pm.un_b64_bincode(root["user_pvs_proof"])
        batches.append(dict(dir=d, json=str(d / "root.json"), root=root, stmt=np.frombuffer(upv[:4 * 58], dtype=np.uint32), pvs=upv[4 * 58:4 * 58 + 32],
                            openings=np.frombuffer(upv[4 * 58 + 32:], dtype=np.uint32),
                            region=np.frombuffer(pm.un_b64_bincode(root["deferral_merkle_proofs"]), dtype=np.uint32)))
    batch_vk = str(batches[0]["dir"] / "root.vk")
    assert (batches[1]["dir"] / "root.vk").read_bytes() == (batches[0]["dir"] / "root.vk").read_bytes()   # one JOIN key for the batch app
    r = pm.run_cli("program-commit", str(batch_elf), batch_vk, str(cfg))
    assert r.returncode == 0, r.stderr
    bpc = json.loads(r.stdout)
    assert bpc["deferral_base"] == 0x00402000 and (bpc["exe"], bpc["vm"]) != (pc["exe"], pc["vm"])
    bundle_elf = tmp_path / "bundle.elf"
    bundle_elf.write_bytes(rv.elf_bytes(deferral_guest_program(), data=b"".join(int(x).to_bytes(4, "little") for x in bpc["exe"] + bpc["vm"])))
    # a join key as the child key needs the child guest (where its deferral region sits)
    r = prove_deferral(tmp_path / "no_elf", bundle_elf, batch_vk, [b["json"] for b in batches], [b["pvs"] for b in batches])
    assert r.returncode != 0 and "join key" in r.stderr
    out = tmp_path / "bundle"
    r = prove_deferral(out, bundle_elf, batch_vk + "@" + str(batch_elf), [b["json"] for b in batches], [b["pvs"] for b in batches])
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["children"] == 2 and info["verified"]
    v = pm.run_cli("verify-guest", str(bundle_elf), str(out / "root.vk"), str(cfg), str(out / "root.json"))
    assert v.returncode == 0, v.stderr
    assert (out / "root.vk").read_bytes() != (batches[0]["dir"] / "root.vk").read_bytes()
    root = json.loads((out / "root.json").read_text())
    upv = pm.un_b64_bincode(root["user_pvs_proof"])
    stmt, pv = np.frombuffer(upv[:4 * 58], dtype=np.uint32), upv[4 * 58:4 * 58 + 32]
    assert int.from_bytes(pv[:4], "little") == 2 and int.from_bytes(pv[4:8], "little") == 2 + 1   # two batches; of two and of one chunk
    cells = [np.array([b["pvs"][2 * j] | (b["pvs"][2 * j + 1] << 8) for j in range(16)], np.uint32) for b in batches]
    want = ru.deferral_chain(np.zeros(8, np.uint32), [ru.deferral_claim(b["stmt"], c) for b, c in zip(batches, cells)])
    assert stmt[50:58].tolist() == want.tolist() == info["deferral_state"]
    claim0 = ru.deferral_claim(batches[0]["stmt"], cells[0])
    assert claim0[1].tolist() == bpc["exe"] and claim0[2].tolist() == bpc["vm"]
    # each batch's own chain is over ITS chunks (the bundle's node re-derives it from the batch's memory)
    ccells = [np.array([k["pvs"][2 * j] | (k["pvs"][2 * j + 1] << 8) for j in range(16)], np.uint32) for k in chunks]
    assert batches[0]["stmt"][50:58].tolist() == ru.deferral_chain(np.zeros(8, np.uint32), [ru.deferral_claim(k["stmt"], c) for k, c in zip(chunks, ccells)]).tolist()
    # the bundle's deferral node: the oracle proves the same circuit and witness to the same bytes
    key_airs, trailer = pm.read_vk(batch_vk)
    region_index = ((2 << 26) | (bpc["deferral_base"] // 16)) >> 9
    D = z.RecursionCircuit(PARAMS, key_airs, 4, stmt="deferral", region_index=region_index)

    def aux_of(b, c, n_flags=None, region=None):
        reg = b["region"] if region is None else region
        n = int(reg[0]) | (int(reg[1]) << 16)
        flags = np.array([1 if k < (n if n_flags is None else n_flags) else 0 for k in range(63)], np.uint32)
        return np.concatenate([c, b["openings"][8:8 * 28], reg, flags])

    proofs = [pm.un_b64_bincode(b["root"]["proof"]) for b in batches]
    pvs = [[NOPV, NOPV, b["stmt"]] for b in batches]
    st, dpv = D.witness(proofs, pvs, aux=[aux_of(b, c) for b, c in zip(batches, cells)])
    assert st == 0, D.last_error()
    assert dpv[8:].tolist() == want.tolist()
    dj = json.loads((out / "deferral.json").read_text())
    assert pm.un_b64_bincode(dj["proof"]) == ora.stark_prove(PARAMS, ru.node_instance(D, dpv)).tobytes()
    # the circuit refuses: one claim too many / too few counted, a region cell that is not the one in the batch's memory
    for n_flags in (1, 3):
        st, _ = D.witness(proofs, pvs, aux=[aux_of(batches[0], cells[0], n_flags=n_flags), aux_of(batches[1], cells[1])])
        assert st != 0
    reg = batches[0]["region"].copy()
    reg[2 * 40] ^= 1
    st, _ = D.witness(proofs, pvs, aux=[aux_of(batches[0], cells[0], region=reg), aux_of(batches[1], cells[1])])
    assert st != 0
    # ... and so does the host before it: a batch proof whose region opening was tampered with
    bad = dict(batches[0]["root"])
    rb = bytearray(pm.un_b64_bincode(bad["deferral_merkle_proofs"]))
    rb[4 * 2 * 40] ^= 1
    bad["deferral_merkle_proofs"] = pm.b64_bincode(bytes(rb))
    (tmp_path / "bad_batch.json").write_text(json.dumps(bad))
    r = prove_deferral(tmp_path / "x1", bundle_elf, batch_vk + "@" + str(batch_elf), [str(tmp_path / "bad_batch.json"), batches[1]["json"]], [b["pvs"] for b in batches])
    assert r.returncode != 0 and "does not open" in r.stderr
    # a CHUNK proof is no child of the bundle (not under the batch's join key), and a batch proof none of a batch
    r = prove_deferral(tmp_path / "x2", bundle_elf, batch_vk + "@" + str(batch_elf), [chunks[0]["json"]], [chunks[0]["pvs"]])
    assert r.returncode != 0
    r = prove_deferral(tmp_path / "x3", batch_elf, chunk_vk, [batches[0]["json"]], [batches[0]["pvs"]])
    assert r.returncode != 0


def test_a_batch_of_more_children_than_one_deferral_node_takes(tmp_path):
    """A batch holds up to 45 chunks (crates/types/batch/src/payload/v6.rs:10).  With FlowOptions::deferral_nodes > 1 a prover's tasks
    run up to that many deferral nodes, each continuing the chain of the one before, and FOLD them; the join verifies the fold.  Five
    chunk proofs under deferral nodes of two children: three nodes, one fold, one join -- the statement's chain is the chain over all five,
    in order; the same prover's key serves a task with a single child; six children are refused."""
    import os

    cfg = tmp_path / "openvm.toml"
    cfg.write_text(pm.TOML.format(*PARAMS))
    chunk_elf = tmp_path / "chunk.elf"
    chunk_elf.write_bytes(rv.elf_bytes(fib_program()))
    ns = (60, 70, 80, 90, 100)
    chunks = [prove_child(tmp_path / ("c%d" % i), chunk_elf, cfg, n) for i, n in enumerate(ns)]
    chunk_vk = str(chunks[0]["dir"] / "root.vk")
    pc = json.loads(pm.run_cli("program-commit", str(chunk_elf), chunk_vk, str(cfg)).stdout)
    batch_elf = tmp_path / "batch.elf"
    batch_elf.write_bytes(rv.elf_bytes(deferral_guest_program(), data=b"".join(int(x).to_bytes(4, "little") for x in pc["exe"] + pc["vm"])))
    env = dict(os.environ, ZKHIP_DEFERRAL_CHILDREN="2", ZKHIP_DEFERRAL_NODES="3")

    def prove(outdir, ks):
        outdir.mkdir()
        (outdir / "witness.bin").write_bytes(b"".join(k["pvs"] for k in ks))
        return subprocess.run([pm.CLI, "prove-deferral", str(batch_elf), str(cfg), str(outdir), "9", chunk_vk, str(cfg), str(outdir / "witness.bin")] + [k["json"] for k in ks],
                              capture_output=True, text=True, env=env)

    out = tmp_path / "five"
    r = prove(out, chunks)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["children"] == 5 and info["verified"]
    assert pm.run_cli("verify-guest", str(batch_elf), str(out / "root.vk"), str(cfg), str(out / "root.json")).returncode == 0
    upv = pm.un_b64_bincode(json.loads((out / "root.json").read_text())["user_pvs_proof"])
    stmt, pv = np.frombuffer(upv[:4 * 58], dtype=np.uint32), upv[4 * 58:4 * 58 + 32]
    fib = lambda n: (lambda f: [f := (f[1], (f[0] + f[1]) & 0xFFFFFFFF) for _ in range(n)][-1][0])((0, 1))   # noqa: E731
    assert int.from_bytes(pv[:4], "little") == 5 and int.from_bytes(pv[4:8], "little") == sum(fib(n) for n in ns) & 0xFFFFFFFF
    cells = [np.array([k["pvs"][2 * j] | (k["pvs"][2 * j + 1] << 8) for j in range(16)], np.uint32) for k in chunks]
    want = ru.deferral_chain(np.zeros(8, np.uint32), [ru.deferral_claim(k["stmt"], c) for k, c in zip(chunks, cells)])
    assert stmt[50:58].tolist() == want.tolist() == info["deferral_state"]
    # what the join verified beside the root is a FOLD: [key digest | chain before = 0 | chain after | accumulator]
    fold = np.frombuffer(pm.un_b64_bincode(json.loads((out / "deferral.json").read_text())["user_pvs_proof"]), dtype=np.uint32)
    assert fold.size == 32 and fold[8:16].tolist() == [0] * 8 and fold[16:24].tolist() == want.tolist()
    # one child under the same prover configuration: the same key
    one = tmp_path / "one"
    r = prove(one, chunks[:1])
    assert r.returncode == 0, r.stderr[-3000:]
    assert (one / "root.vk").read_bytes() == (out / "root.vk").read_bytes()
    assert pm.run_cli("verify-guest", str(batch_elf), str(out / "root.vk"), str(cfg), str(one / "root.json")).returncode == 0
    # ... which is not the key of a prover without the fold
    plain = tmp_path / "plain"
    plain.mkdir()
    (plain / "witness.bin").write_bytes(chunks[0]["pvs"])
    r = subprocess.run([pm.CLI, "prove-deferral", str(batch_elf), str(cfg), str(plain), "9", chunk_vk, str(cfg), str(plain / "witness.bin"), chunks[0]["json"]],
                       capture_output=True, text=True)
    assert r.returncode == 0 and (plain / "root.vk").read_bytes() != (out / "root.vk").read_bytes()
    # seven children: more than 3 x 2
    r = prove(tmp_path / "seven", chunks + chunks[:2])
    assert r.returncode != 0 and "take 6" in r.stderr


def test_three_layers_at_the_reference_parameters(tmp_path):
    """chunk -> batch -> bundle once more at the reference's FRI parameters (crates/circuits/*/openvm.toml: blow-up 2, 100 queries, 16 + 16
    proof-of-work bits) instead of the toy ones the byte-parity tests above use: every proof of the three layers is made and verified under
    the parameters a deployment runs, the chains are the independent restatement's."""
    ref = (1, 0, 100, 16, 16)
    cfg = tmp_path / "openvm.toml"
    cfg.write_text(pm.TOML.format(*ref))
    chunk_elf = tmp_path / "chunk.elf"
    chunk_elf.write_bytes(rv.elf_bytes(fib_program()))
    chunks = [prove_child(tmp_path / ("c%d" % i), chunk_elf, cfg, n) for i, n in enumerate((120, 130))]
    chunk_vk = str(chunks[0]["dir"] / "root.vk")
    pc = json.loads(pm.run_cli("program-commit", str(chunk_elf), chunk_vk, str(cfg)).stdout)
    batch_elf = tmp_path / "batch.elf"
    batch_elf.write_bytes(rv.elf_bytes(deferral_guest_program(), data=b"".join(int(x).to_bytes(4, "little") for x in pc["exe"] + pc["vm"])))

    def prove_deferral(outdir, elf, child_vk, ks):
        outdir.mkdir()
        (outdir / "witness.bin").write_bytes(b"".join(k["pvs"] for k in ks))
        r = subprocess.run([pm.CLI, "prove-deferral", str(elf), str(cfg), str(outdir), "9", child_vk, str(cfg), str(outdir / "witness.bin")] + [k["json"] for k in ks],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        root = json.loads((outdir / "root.json").read_text())
        upv = pm.un_b64_bincode(root["user_pvs_proof"])
        return dict(dir=outdir, json=str(outdir / "root.json"), stmt=np.frombuffer(upv[:4 * 58], dtype=np.uint32), pvs=upv[4 * 58:4 * 58 + 32])

    cells = lambda k: np.array([k["pvs"][2 * j] | (k["pvs"][2 * j + 1] << 8) for j in range(16)], np.uint32)  # noqa: E731
    batch = prove_deferral(tmp_path / "batch", batch_elf, chunk_vk, chunks)
    assert pm.run_cli("verify-guest", str(batch_elf), str(batch["dir"] / "root.vk"), str(cfg), batch["json"]).returncode == 0
    assert batch["stmt"][50:58].tolist() == ru.deferral_chain(np.zeros(8, np.uint32), [ru.deferral_claim(k["stmt"], cells(k)) for k in chunks]).tolist()
    batch_vk = str(batch["dir"] / "root.vk")
    bpc = json.loads(pm.run_cli("program-commit", str(batch_elf), batch_vk, str(cfg)).stdout)
    bundle_elf = tmp_path / "bundle.elf"
    bundle_elf.write_bytes(rv.elf_bytes(deferral_guest_program(), data=b"".join(int(x).to_bytes(4, "little") for x in bpc["exe"] + bpc["vm"])))
    bundle = prove_deferral(tmp_path / "bundle", bundle_elf, batch_vk + "@" + str(batch_elf), [batch])
    assert pm.run_cli("verify-guest", str(bundle_elf), str(bundle["dir"] / "root.vk"), str(cfg), bundle["json"]).returncode == 0
    assert bundle["stmt"][50:58].tolist() == ru.deferral_chain(np.zeros(8, np.uint32), [ru.deferral_claim(batch["stmt"], cells(batch))]).tolist()
    assert int.from_bytes(bundle["pvs"][:4], "little") == 1 and int.from_bytes(bundle["pvs"][4:8], "little") == 2     # one batch, of two chunks
    # a flipped byte in the batch proof is refused before anything is proven
    bad = json.loads((batch["dir"] / "root.json").read_text())
    pb = bytearray(pm.un_b64_bincode(bad["proof"]))
    pb[4 * 777] ^= 1
    bad["proof"] = pm.b64_bincode(bytes(pb))
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    x = tmp_path / "x"
    x.mkdir()
    (x / "witness.bin").write_bytes(batch["pvs"])
    r = subprocess.run([pm.CLI, "prove-deferral", str(bundle_elf), str(cfg), str(x), "9", batch_vk + "@" + str(batch_elf), str(cfg), str(x / "witness.bin"), str(tmp_path / "bad.json")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "does not verify under the child aggregation key" in r.stderr
