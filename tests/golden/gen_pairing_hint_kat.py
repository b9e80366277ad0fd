"""Writes tests/golden/pairing_hint_kat.json: final-exponentiation witnesses of BN254 computed with Python integers (tests/pairing_util.py:
every constant derived from the curve parameter x): for seeded f = g^r in the subgroup a Miller loop's output lies in, the pair (c, u) with
c^lambda = f u under the product's uniqueness rule -- and the check of that equation.  The vectors pin include/zkhip_pairing.hpp
(tests/test_phantom_cpu.py).  Usage: python3 tests/golden/gen_pairing_hint_kat.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pairing_util as pu  # noqa: E402

cases = []
for seed in range(1, 7):
    f = pu.sample_f(seed)
    c, u = pu.final_exp_hint(f)
    assert pu.power(c, pu.LAMBDA) == pu.mul(f, u)
    cases.append({"seed": seed, "f": [[hex(a), hex(b)] for a, b in pu.to_sextic(f)], "c": [[hex(a), hex(b)] for a, b in pu.to_sextic(c)],
                  "u": [[hex(a), hex(b)] for a, b in pu.to_sextic(u)], "u_is_one": u == pu.ONE})
bls = []
B = pu.Bls12_381
for seed in range(1, 4):
    f = B.sample_f(seed)
    c, sc = B.final_exp_hint(f)
    assert B.power(c, B.LAMBDA) == B.mul(f, sc)
    bls.append({"seed": seed, "f": [[hex(a), hex(b)] for a, b in B.to_sextic(f)], "c": [[hex(a), hex(b)] for a, b in B.to_sextic(c)],
                "s": [[hex(a), hex(b)] for a, b in B.to_sextic(sc)]})
json.dump({"bls12_381": {"x": B.X, "lambda": hex(B.LAMBDA), "layout": "six Fp2 coefficients [a_i, b_i] of w^i (w^6 = 1 + u, u^2 = -1)", "equation": "c^lambda = f s", "cases": bls},
           "curve": "Bn254", "x": pu.X, "lambda": hex(pu.LAMBDA), "layout": "six Fp2 coefficients [a_i, b_i] of w^i (w^6 = 9 + u, u^2 = -1)",
           "equation": "c^lambda = f u", "cases": cases}, open(os.path.join(HERE, "pairing_hint_kat.json"), "w"), indent=1)
print(len(cases), "cases,", sum(1 for c in cases if not c["u_is_one"]), "with u != 1")
