#!/usr/bin/env python3
"""Generates tests/golden/kat_v1.json from the independent big-int model tests/pymodel.py.

No upstream (p3 / OpenVM) vectors are reachable offline -- the reference tree holds none at kernel
granularity -- so these fixtures pin oracle/ and the HIP path to the published
definitions as restated by a second, independently written implementation, plus the Poseidon2
anchors listed in SURVEY.md A.3 (round constants, perm([0..15]), perm(0^16)).
Run from the repo root:  python3 tests/golden/gen_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pymodel as m  # noqa: E402

rnd = random.Random(0x5A4B5F4E5454)
P = m.P


def rf(n):
    return [rnd.randrange(P) for _ in range(n)]


out = {"p": P, "two_adic_generator_27": m.two_adic_generator(27)}
out["two_adic_generators"] = [m.two_adic_generator(k) for k in range(0, 28)]
out["field_mul"] = [[a, b, a * b % P] for a, b in zip(rf(32) + [0, 1, P - 1], rf(32) + [P - 1, P - 1, P - 1])]
out["field_inv"] = [[a, m.inv(a)] for a in rf(16) + [1, 2, P - 1]]
ea, eb = [rf(4) for _ in range(8)], [rf(4) for _ in range(8)]
out["ext_mul"] = [[a, b, m.ext_mul(a, b)] for a, b in zip(ea, eb)]
out["ext_inv"] = [[a, m.ext_inv(a)] for a in ea]
out["poseidon2_rc"] = m.RC
out["poseidon2_perm"] = [[s, m.permute(s)] for s in ([list(range(16)), [0] * 16, [P - 1] * 16] + [rf(16) for _ in range(5)])]
out["hash_slice"] = [[xs, m.hash_slice(xs)] for xs in [rf(n) for n in (0, 1, 4, 7, 8, 9, 16, 17, 300)]]
out["compress"] = [[l, r, m.compress(l, r)] for l, r in [(rf(8), rf(8)) for _ in range(4)]]
out["dft"] = []
for log_n in (0, 1, 2, 3, 5, 8):
    xs = rf(1 << log_n)
    out["dft"].append({"log_n": log_n, "in": xs, "fwd": m.dft_naive(xs), "inv": m.dft_naive(xs, inverse=True)})
out["coset_lde"] = []
for log_n, added, shift in ((2, 1, 31), (4, 1, 31), (5, 2, 31), (6, 1, 7)):
    xs = rf(1 << log_n)
    nat = m.coset_lde_naive(xs, added, shift)
    bits = log_n + added
    out["coset_lde"].append({"log_n": log_n, "added_bits": added, "shift": shift, "in": xs, "natural": nat,
                             "bitrev": [nat[m.bitrev(r, bits)] for r in range(1 << bits)]})
out["merkle"] = []
for shape in ([(3, 5)], [(4, 8), (4, 3)], [(4, 9), (2, 3), (0, 2)], [(5, 20), (5, 1), (3, 17), (1, 8)]):
    mats = [(lh, [rf(w) for _ in range(1 << lh)]) for lh, w in shape]
    out["merkle"].append({"mats": [{"log_height": lh, "rows": rows} for lh, rows in mats], "root": m.merkle_root(mats)})
ch = m.Challenger()
script = []
for step in [("observe", rf(3)), ("sample", 2), ("observe", rf(8)), ("observe", rf(13)), ("sample", 9),
             ("sample_bits", 10), ("grind", 6), ("sample", 4), ("observe", rf(1)), ("grind", 9), ("sample", 1)]:
    kind, arg = step
    if kind == "observe":
        ch.observe(arg)
        script.append({"op": kind, "vals": arg})
    elif kind == "sample":
        script.append({"op": kind, "n": arg, "out": [ch.sample() for _ in range(arg)]})
    elif kind == "sample_bits":
        script.append({"op": kind, "bits": arg, "out": ch.sample_bits(arg)})
    else:
        script.append({"op": kind, "bits": arg, "witness": ch.grind(arg)})
out["challenger"] = script
out["fri_fold"] = []
for log_n_out in (0, 1, 3, 5):
    vals = [rf(4) for _ in range(2 << log_n_out)]
    beta = rf(4)
    out["fri_fold"].append({"log_n_out": log_n_out, "in": vals, "beta": beta, "out": m.fri_fold(vals, beta)})

# LogUp / sum-check building blocks (appended last so the earlier vectors keep their random stream)
den = [rf(4) for _ in range(37)]
num = rf(37)
out["ext_batch_inverse"] = [[d, m.ext_inv(d)] for d in den[:9]]
out["logup_running_sum"] = {"den": den, "num": num, "out": m.logup_running_sum(den, num)}
fv = [rf(4) for _ in range(32)]
fr = rf(4)
out["mle_fold"] = {"in": fv, "r": fr, "out": m.mle_fold(fv, fr)}
out["sumcheck_round"] = []
for k in (1, 2, 3, 4):
    tabs = [[rf(4) for _ in range(16)] for _ in range(k)]
    out["sumcheck_round"].append({"tables": tabs, "out": m.sumcheck_round(tabs)})

with open(os.path.join(HERE, "kat_v1.json"), "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("wrote kat_v1.json", os.path.getsize(os.path.join(HERE, "kat_v1.json")), "bytes")
