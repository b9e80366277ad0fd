"""Helpers for the MMCS path chip tests: turn the stored openings of the reference's proofs (tests/golden/ref_v1_vectors.json)
into the chip's records -- leaf digest, index, bottom-up steps (sibling digests and injected row digests) -- with the oracle's
sponge; and the claims the chip must emit for them.  Test infrastructure."""
import numpy as np


def hash_slice(ora, words):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.zeros(8, np.uint32)
    ora.lib().ora_hash_slice(ora.p32(w), len(w), ora.p32(out))
    return out


def path_records(ora, batch, index):
    """-> (leaf[8], index, kinds[], digests[][8], expected claims [(lvl, idx, digest)]) for one batch opening at `index`"""
    lhs, ws, opening = batch["log_heights"], batch["widths"], batch["opening"]
    H = max(lhs)
    offs = np.concatenate([[0], np.cumsum(ws)])
    rows_at = lambda level: np.concatenate([opening[offs[m]:offs[m + 1]] for m in range(len(ws)) if lhs[m] == level] or [np.zeros(0)]).astype(np.uint32)  # noqa: E731
    has = lambda level: any(h == level for h in lhs)  # noqa: E731
    path = np.array(opening[offs[-1]:], np.uint32).reshape(-1, 8)
    assert len(path) == H
    leaf = hash_slice(ora, rows_at(H))
    kinds, digests, claims = [], [], [(H, index, leaf)]
    for l in range(H):
        kinds.append(0), digests.append(path[l])
        level = H - l - 1
        if has(level):
            d = hash_slice(ora, rows_at(level))
            kinds.append(1), digests.append(d)
            claims.append((level, index >> (l + 1), d))
    return leaf, index, kinds, digests, claims


def records_of_fixture(ora, vec, limit=None):
    """All batch openings of the fixture's queries as chip records (flattened arrays) + expected claims incl. roots"""
    leaves, idxs, starts, kinds, digs, claims = [], [], [0], [], [], []
    for q in vec["openings"][:limit]:
        H = q["log_max_height"]
        for b in q["batches"]:
            if max(b["log_heights"]) == 0:
                continue   # a single-row batch has no path
            idx = q["index"] >> (H - max(b["log_heights"]))
            leaf, idx, k, d, c = path_records(ora, b, idx)
            leaves.append(leaf), idxs.append(idx), kinds.extend(k), digs.extend(d)
            starts.append(len(kinds))
            claims.extend((tuple(b["root"]), lvl, i, tuple(int(x) for x in dg)) for lvl, i, dg in c)
    return (np.array(leaves, np.uint32), np.array(idxs, np.uint32), np.array(starts, np.uint32), np.array(kinds, np.uint32),
            np.array(digs, np.uint32).reshape(-1, 8), claims)


def claims_table(claims, log_height):
    """mmcs_claims_air trace from a list of claim rows [n][18] (duplicates merged into multiplicities)"""
    from collections import Counter

    cnt = Counter(tuple(int(x) for x in c) for c in claims)
    t = np.zeros((19, 1 << log_height), np.uint32)
    assert len(cnt) <= 1 << log_height
    for r, (c, m) in enumerate(sorted(cnt.items())):
        t[:18, r] = c
        t[18, r] = m
    return t
