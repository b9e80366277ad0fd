#!/usr/bin/env python3
"""Golden vectors from the REFERENCE'S OWN stored proofs -> tests/golden/ref_v1_vectors.json.

Runs in the build container only (reads /root/reference, which does not exist on the GPU box); the output is a
data fixture -- field elements copied out of, or algebraically solved from, the proofs under
  /root/reference/crates/verifier/testdata/proofs/{chunk,batch}-proof-{feynman,phase1,phase2}.json
  /root/reference/crates/prover/testdata/{chunk,batch}-proof.json
which are complete OpenVM-v1 (quotient + FRI) `Proof<SC>` blobs produced by the reference's prover
(container layout: tests/refproof_v1.py).  Everything written is CANONICAL (the files hold Montgomery words).

What the proofs pin, and how it is extracted without the verifying key (the transcript cannot be replayed
without it, so query indices and challenges are not available directly):

 1. compress triples.  Every Merkle path of every query ends in a child of the root; over ~40 queries both
    children appear, so for each of the commitments (main x2, after-challenge, quotient, every FRI layer) the pair
    (L, R) with compress(L, R) == commitment is read off the proof.  Pins the Poseidon2 permutation, its round
    constants and TruncatedPermutation on ~25 independent states per file.
 2. batch openings.  For each query the leaf index is recovered by exhaustive search on the tallest single-matrix
    opening (oracle helper ora_mmcs_find_index: exactly one index verifies), extended by the remaining low bits
    against the 17-matrix mixed-height common-main opening (exactly one extension verifies); all six openings of a
    query (2 preprocessed, cached main, common main, after-challenge, 62 quotient chunks) must then verify at
    index >> (23 - their height).  Pins PaddingFreeSponge, the concatenation order of rows, the injection of shorter
    matrices and the index shifting of MerkleTreeMmcs.
 3. FRI fold triples.  Two queries whose positions in a layer are siblings reveal each other's value, so the pair
    (e0, e1) of that layer is known; the folded value is known when it is itself another query's sibling, the final
    polynomial, or -- once the layer's beta has been solved from ONE triple -- back-propagated along every query.
    Each layer where no shorter matrix joins ("pure") and that has >= 2 triples pins the fold formula, the
    g^bitrev(i) twiddle convention and the x^4 = 11 extension arithmetic: the first triple solves beta, the others
    must agree.  Layers where a shorter matrix joins add beta^2 * (its reduced opening) (p3-fri), which is not
    known without the transcript: they are skipped and break the back-propagation chain.
    Known (e0, e1) pairs are also FRI-layer Merkle leaves: their opening must verify against the layer commitment.

Not extractable offline (needs the verifying key's pre-hash, which is not in the tree): transcript order,
proof-of-work witnesses, the quotient identity, the reduced openings.
"""
import base64
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ora  # noqa: E402  (index search + self-check only)
import refproof_v1 as rp  # noqa: E402

P = rp.P
REF = "/root/reference/crates"
FILES = [
    "verifier/testdata/proofs/chunk-proof-feynman.json",
    "verifier/testdata/proofs/batch-proof-feynman.json",
    "verifier/testdata/proofs/chunk-proof-phase1.json",
    "verifier/testdata/proofs/chunk-proof-phase2.json",
    "verifier/testdata/proofs/batch-proof-phase1.json",
    "verifier/testdata/proofs/batch-proof-phase2.json",
    "prover/testdata/chunk-proof.json",
    "prover/testdata/batch-proof.json",
]
DEEP = {"chunk-proof-feynman.json": 8, "batch-proof-feynman.json": 4}  # files analysed per query: #queries whose openings are stored


def canon(ws):
    return [rp.from_monty(w) for w in ws]


# ---- extension field F_p[x]/(x^4 - 11), written here independently of oracle/ and the product ----
W = 11


def emul(a, b):
    r = [0] * 7
    for i in range(4):
        for j in range(4):
            r[i + j] += a[i] * b[j]
    return [(r[k] + W * (r[k + 4] if k < 3 else 0)) % P for k in range(4)]


def eadd(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def esub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def escale(a, s):
    return [x * s % P for x in a]


def einv(a):
    # a^(p^4 - 2)
    e = P ** 4 - 2
    r, b = [1, 0, 0, 0], a
    while e:
        if e & 1:
            r = emul(r, b)
        b = emul(b, b)
        e >>= 1
    return r


TWO_ADIC_GEN_27 = 0x1A427A41


def two_adic_generator(bits):
    return pow(TWO_ADIC_GEN_27, 1 << (27 - bits), P)


def bitrev(i, n):
    return int(format(i, "0%db" % n)[::-1], 2) if n else 0


def fold_x(k, log_n_out):
    """x of pair k of a layer that folds 2^(log_n_out+1) -> 2^log_n_out values (p3-fri fold_row)."""
    return pow(two_adic_generator(log_n_out + 1), bitrev(k, log_n_out), P)


def fold(e0, e1, x, beta):
    # e0 + (beta - x)(e1 - e0)/(-2x)
    c = escale(esub(beta, [x, 0, 0, 0]), pow((-2 * x) % P, -1, P))
    return eadd(e0, emul(c, esub(e1, e0)))


def solve_beta(e0, e1, x, folded):
    d = esub(e1, e0)
    c = emul(esub(folded, e0), einv(d))
    return eadd(escale(c, (-2 * x) % P), [x, 0, 0, 0])


def unfold(folded, sib, sib_is_e1, x, beta):
    """the other member of the pair, given the folded value, one member and beta"""
    c = escale(esub(beta, [x, 0, 0, 0]), pow((-2 * x) % P, -1, P))
    one_c = esub([1, 0, 0, 0], c)
    if sib_is_e1:  # folded = (1-c) e0 + c e1  ->  e0
        return emul(esub(folded, emul(c, sib)), einv(one_c))
    return emul(esub(folded, emul(one_c, sib)), einv(c))


def load(rel):
    d = json.load(open(os.path.join(REF, rel)))
    blob = base64.b64decode(d["proof"]["proofs"])
    proofs = rp.decode_proofs(blob)
    assert rp.encode_proofs(proofs) == blob
    return d, blob, proofs[0]


def top_pairs(p):
    """{commitment name: (root, set of top siblings)}"""
    out = {}
    nb = len(p["fri"]["query_proofs"][0]["input_proof"])
    ov = p["opened"]
    names = ["preprocessed[%d]" % i for i in range(len(ov["preprocessed"]))]
    names += ["main_trace[%d]" % i for i in range(len(p["main_trace"]))]
    names += ["after_challenge[%d]" % i for i in range(len(p["after_challenge"]))] + ["quotient"]
    assert len(names) == nb
    roots = [None] * len(ov["preprocessed"]) + p["main_trace"] + p["after_challenge"] + [p["quotient"]]
    for b in range(nb):
        if roots[b] is None:
            continue  # preprocessed commitments live in the verifying key, not in the proof
        out[names[b]] = (roots[b], {tuple(q["input_proof"][b]["path"][-1]) for q in p["fri"]["query_proofs"]})
    for l, root in enumerate(p["fri"]["commit_phase_commits"]):
        out["fri_layer[%d]" % l] = (root, {tuple(q["commit_phase_openings"][l]["path"][-1])
                                           for q in p["fri"]["query_proofs"]})
    return out, names


def compress_triples(fname, p):
    res = []
    pairs, _ = top_pairs(p)
    for name, (root, sibs) in pairs.items():
        sibs = [canon(s) for s in sibs]
        rootc = np.array(canon(root), dtype=np.uint32)
        hit = None
        for a in sibs:
            for b in sibs:
                o = np.zeros(8, np.uint32)
                ora.lib().ora_compress(ora.p32(np.array(a, np.uint32)), ora.p32(np.array(b, np.uint32)), ora.p32(o))
                if (o == rootc).all():
                    hit = (a, b)
        if hit is None:
            print("  !! %s: no (L,R) among %d top siblings" % (name, len(sibs)))
            continue
        res.append({"file": fname, "commit": name, "L": hit[0], "R": hit[1], "root": canon(root)})
    return res


def batch_layout(p):
    """per input batch: (log_heights, widths) of its matrices in opening order"""
    q0 = p["fri"]["query_proofs"][0]
    ov = p["opened"]
    logdeg = [a["degree"].bit_length() - 1 for a in p["per_air"]]
    n_prep, n_main = len(ov["preprocessed"]), len(p["main_trace"])
    common = n_prep + n_main - 1
    H = len(q0["input_proof"][common]["path"])
    log_blowup = H - max(logdeg)
    lay = []
    for b, bt in enumerate(q0["input_proof"]):
        widths = [len(r) for r in bt["opened_values"]]
        if len(widths) == 1:
            lhs = [len(bt["path"])]
        elif b == len(q0["input_proof"]) - 1:  # quotient: chunks of AIR i share its LDE height
            lhs = []
            for i, chunks in enumerate(ov["quotient"]):
                lhs += [logdeg[i] + log_blowup] * len(chunks)
        else:
            lhs = [d + log_blowup for d in logdeg]
        assert len(lhs) == len(widths)
        lay.append((lhs, widths))
    return lay, H, log_blowup


def flat_opening(bt):
    rows = [w for r in bt["opened_values"] for w in r]
    path = [w for d in bt["path"] for w in d]
    return np.array(canon(rows) + canon(path), dtype=np.uint32)


def mmcs_verify(root, lhs, widths, index, opening):
    l = ora.lib()
    a = (C.c_uint * len(lhs))(*lhs)
    w = (C.c_size_t * len(widths))(*widths)
    r = np.array(root, dtype=np.uint32)
    return bool(l.ora_mmcs_verify(ora.p32(r), a, w, len(lhs), index, ora.p32(opening)))


def recover_indices(fname, p, lay, H, roots_by_batch):
    l = ora.lib()
    l.ora_mmcs_find_index.restype = C.c_size_t
    l.ora_mmcs_find_index.argtypes = [ora.u32p_t(), C.c_uint, C.c_size_t, ora.u32p_t(), C.POINTER(C.c_size_t)]
    nb = len(lay)
    singles = [b for b in range(nb) if len(lay[b][0]) == 1 and roots_by_batch[b] is not None]
    bs = max(singles, key=lambda b: lay[b][0][0])  # tallest single-matrix opening whose commitment is in the proof
    common = max(b for b in range(nb - 1) if len(lay[b][0]) > 1 and roots_by_batch[b] is not None and b < nb - 2)
    hs = lay[bs][0][0]
    idxs = []
    for qi, q in enumerate(p["fri"]["query_proofs"]):
        op = flat_opening(q["input_proof"][bs])
        out = C.c_size_t(0)
        root = np.array(roots_by_batch[bs], dtype=np.uint32)
        n = l.ora_mmcs_find_index(ora.p32(root), hs, lay[bs][1][0], ora.p32(op), C.byref(out))
        assert n == 1, (fname, qi, n)
        hi = out.value
        opc = flat_opening(q["input_proof"][common])
        ext = [e for e in range(1 << (H - hs))
               if mmcs_verify(roots_by_batch[common], lay[common][0], lay[common][1], (hi << (H - hs)) | e, opc)]
        assert len(ext) == 1, (fname, qi, ext)
        idxs.append((hi << (H - hs)) | ext[0])
    return idxs


def fri_analysis(fname, p, idxs, H, lay):
    """returns (layers: list of {layer, log_n_out, beta, triples}, leaf_checks)"""
    fri = p["fri"]
    nl = len(fri["commit_phase_commits"])
    heights = {h for lhs, _ in lay for h in lhs}
    final = canon(fri["final_poly"][0])
    assert all(c == [0, 0, 0, 0] for c in fri["final_poly"][1:]), "non-constant final polynomial"
    known = [dict() for _ in range(nl + 1)]  # known[l][pos] = value of layer l (domain 2^(H-l)) at position pos
    for qi, q in enumerate(fri["query_proofs"]):
        for l, st in enumerate(q["commit_phase_openings"]):
            pos = (idxs[qi] >> l) ^ 1
            v = canon(st["sibling"])
            assert known[l].get(pos, v) == v
            known[l][pos] = v
        known[nl][idxs[qi] >> nl] = final
    layers, leaf_checks = [], []
    for l in range(nl - 1, -1, -1):
        log_n_out = H - l - 1
        pure = log_n_out not in heights
        trip = []
        for k in sorted({pos >> 1 for pos in known[l]}):
            if 2 * k in known[l] and 2 * k + 1 in known[l] and k in known[l + 1]:
                trip.append((k, known[l][2 * k], known[l][2 * k + 1], known[l + 1][k]))
        if not pure or not trip:
            print("  layer %2d -> 2^%-2d : %s, %d candidate triples, skipped" % (l, log_n_out, "pure" if pure else "a matrix joins", len(trip)))
            continue
        beta = None
        for k, e0, e1, f in trip:
            if e0 != e1:
                beta = solve_beta(e0, e1, fold_x(k, log_n_out), f)
                break
        if beta is None:
            continue
        good = [t for t in trip if fold(t[1], t[2], fold_x(t[0], log_n_out), beta) == t[3]]
        print("  layer %2d -> 2^%-2d : pure, %d triples, %d agree with the beta solved from the first" % (l, log_n_out, len(trip), len(good)))
        assert len(good) == len(trip), "inconsistent triples"
        if len(trip) >= 2:
            layers.append({"file": fname, "layer": l, "log_n_out": log_n_out, "beta": beta,
                           "triples": [{"k": k, "e0": e0, "e1": e1, "folded": f} for k, e0, e1, f in trip]})
        # back-propagate along every query whose folded value is known
        for qi in range(len(idxs)):
            pos, k = idxs[qi] >> l, idxs[qi] >> (l + 1)
            if k in known[l + 1] and (pos ^ 1) in known[l]:
                v = unfold(known[l + 1][k], known[l][pos ^ 1], (pos & 1) == 0, fold_x(k, log_n_out), beta)
                assert known[l].get(pos, v) == v, "back-propagated value contradicts a sibling"
                known[l][pos] = v
        # FRI-layer Merkle leaves: pairs fully known for a query -> its commit-phase opening verifies
        for qi, q in enumerate(fri["query_proofs"]):
            k = idxs[qi] >> (l + 1)
            if 2 * k in known[l] and 2 * k + 1 in known[l]:
                row = known[l][2 * k] + known[l][2 * k + 1]
                path = [w for d in q["commit_phase_openings"][l]["path"] for w in canon(d)]
                op = np.array(row + path, dtype=np.uint32)
                root = canon(fri["commit_phase_commits"][l])
                ok = mmcs_verify(root, [log_n_out], [8], k, op)
                assert ok, "FRI layer leaf does not verify"
                if len([c for c in leaf_checks if c["layer"] == l]) < 2:
                    leaf_checks.append({"file": fname, "layer": l, "log_height": log_n_out, "index": k,
                                        "root": root, "opening": [int(x) for x in op]})
    return layers, leaf_checks


def main():
    out = {"about": "canonical field elements taken / solved from the reference's stored OpenVM-v1 proofs; see gen_ref_vectors.py",
           "sources": [], "compress": [], "openings": [], "fri_layers": [], "fri_leaves": [], "shapes": {}, "logup_exposed": []}
    for rel in FILES:
        fname = os.path.basename(rel) if "prover/" not in rel else "prover-" + os.path.basename(rel)
        d, blob, p = load(rel)
        print(fname, len(blob), "bytes")
        out["sources"].append({"file": rel, "name": fname, "proof_bytes": len(blob), "sha256": hashlib.sha256(blob).hexdigest(),
                               "git_version": d.get("git_version")})
        out["shapes"][fname] = rp.shape_of(p)
        # 4. LogUp exposed sums: per AIR with interactions the cumulative sum of its after-challenge column (one phase); the
        #    verifier's bus check is that they cancel over all AIRs of the proof.
        out["logup_exposed"].append({"file": fname, "air_ids": [a["air_id"] for a in p["per_air"] if a["exposed"] and a["exposed"][0]],
                                     "exposed": [canon(a["exposed"][0][0]) for a in p["per_air"] if a["exposed"] and a["exposed"][0]]})
        trips = compress_triples(fname, p)
        print("  compress triples: %d" % len(trips))
        out["compress"] += trips
        base = os.path.basename(rel)
        if "prover/" in rel or base not in DEEP:
            continue
        lay, H, log_blowup = batch_layout(p)
        _, names = top_pairs(p)
        ov = p["opened"]
        roots = [None] * len(ov["preprocessed"]) + [canon(r) for r in p["main_trace"]] + \
            [canon(r) for r in p["after_challenge"]] + [canon(p["quotient"])]
        idxs = recover_indices(fname, p, lay, H, roots)
        print("  log_blowup %d, LDE height 2^%d, %d query indices recovered (all distinct: %s)" %
              (log_blowup, H, len(idxs), len(set(idxs)) == len(idxs)))
        # every opening of every query verifies at the shifted index (preprocessed roots: recomputed, then cross-checked
        # between queries since the verifying key is not available)
        prep_roots = {}
        for qi, q in enumerate(p["fri"]["query_proofs"]):
            for b, bt in enumerate(q["input_proof"]):
                lhs, widths = lay[b]
                idx_b = idxs[qi] >> (H - max(lhs))
                op = flat_opening(bt)
                if roots[b] is None:
                    # implied root of a preprocessed opening; must be the same for every query
                    cur = np.zeros(8, np.uint32)
                    ora.lib().ora_hash_slice(ora.p32(op), widths[0], ora.p32(cur))
                    for lvl in range(lhs[0]):
                        sib = op[widths[0] + 8 * lvl: widths[0] + 8 * lvl + 8].copy()
                        o = np.zeros(8, np.uint32)
                        if (idx_b >> lvl) & 1:
                            ora.lib().ora_compress(ora.p32(sib), ora.p32(cur), ora.p32(o))
                        else:
                            ora.lib().ora_compress(ora.p32(cur), ora.p32(sib), ora.p32(o))
                        cur = o
                    prep_roots.setdefault(b, set()).add(tuple(int(x) for x in cur))
                else:
                    assert mmcs_verify(roots[b], lhs, widths, idx_b, op), (fname, qi, b)
        for b, s in prep_roots.items():
            assert len(s) == 1, "preprocessed openings imply different roots"
            roots[b] = list(next(iter(s)))
        print("  all %d x %d batch openings verify; implied preprocessed commitments consistent over the queries" %
              (len(idxs), len(lay)))
        for qi in range(DEEP[base]):
            q = p["fri"]["query_proofs"][qi]
            out["openings"].append({
                "file": fname, "query": qi, "index": idxs[qi], "log_max_height": H,
                "batches": [{"commit": names[b], "root": roots[b], "root_source": "proof" if b >= len(ov["preprocessed"]) else "implied by all queries",
                             "log_heights": lay[b][0], "widths": lay[b][1],
                             "opening": [int(x) for x in flat_opening(q["input_proof"][b])]}
                            for b in range(len(lay))]})
        layers, leaves = fri_analysis(fname, p, idxs, H, lay)
        out["fri_layers"] += layers
        out["fri_leaves"] += leaves
        out["shapes"][fname].update({"log_blowup": log_blowup, "log_max_height": H, "query_indices": idxs})
    dst = os.path.join(HERE, "ref_v1_vectors.json")
    with open(dst, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out["compress"]), "compress triples,",
          len(out["openings"]), "queries with openings,", sum(len(l["triples"]) for l in out["fri_layers"]),
          "fold triples in", len(out["fri_layers"]), "layers,", len(out["fri_leaves"]), "FRI leaves")


if __name__ == "__main__":
    main()
