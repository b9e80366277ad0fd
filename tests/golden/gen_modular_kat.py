#!/usr/bin/env python3
"""Known-answer vectors for 256-bit modular multiplication -> tests/golden/modular_kat.json.  Source of truth OUTSIDE this repository:
Python's arbitrary-precision integers (q, r = divmod(a * b, p)) for the field and scalar moduli of secp256k1 and bn254 (the curves of
the EVM's ecrecover and pairing precompiles; the reference's chunk circuit configures them, crates/circuits/chunk-circuit/openvm.toml),
plus the published secp256k1 generator, whose coordinates satisfy y^2 = x^3 + 7 (SEC 2, 2.4.1)."""
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
MODULI = {
    "secp256k1_p": 2**256 - 2**32 - 977,
    "secp256k1_n": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    "bn254_p": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "bn254_r": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
}
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8


def main():
    rnd = random.Random(20260303)
    out = {"about": "q, r = divmod(a * b, p) from Python integers (generator: tests/golden/gen_modular_kat.py)", "moduli": {}, "cases": [], "addsub": [], "div": [],
           "secp256k1_generator": {"x": hex(GX), "y": hex(GY)}}
    assert (GY * GY - GX * GX * GX - 7) % MODULI["secp256k1_p"] == 0
    for name, p in MODULI.items():
        out["moduli"][name] = hex(p)
        ops = [(0, 0), (1, 1), (p - 1, p - 1), (p - 1, 1), (2, (p + 1) // 2), (1 << 255, 3), (p - 1, 2)]
        ops += [(rnd.randrange(p), rnd.randrange(p)) for _ in range(17)]
        for a, b in ops:
            q, r = divmod(a * b, p)
            assert q < 1 << 256
            out["cases"].append({"modulus": name, "a": hex(a), "b": hex(b), "q": hex(q), "r": hex(r)})
        # addition (a + b = q p + r) and subtraction (a - b + q p = r with q in {0, 1}: reduced operands)
        red = [(0, 0), (p - 1, p - 1), (p - 1, 1), (1, p - 1), (0, 1), (5, 5)] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(10)]
        for a, b in red:
            q, r = divmod(a + b, p)
            out["addsub"].append({"modulus": name, "op": 1, "a": hex(a), "b": hex(b), "q": hex(q), "r": hex(r)})
            r = (a - b) % p
            out["addsub"].append({"modulus": name, "op": 2, "a": hex(a), "b": hex(b), "q": hex(0 if a >= b else 1), "r": hex(r)})
    # division x / y = x y^-1 (a generator of its own: the cases above keep their values); y invertible, x reduced
    rnd_div = random.Random(20261003)
    for name, p in MODULI.items():
        for x, y in [(0, 1), (1, 1), (p - 1, p - 1), (1, 2), (p - 1, 2), (7, p - 1)] + [(rnd_div.randrange(p), rnd_div.randrange(1, p)) for _ in range(8)]:
            r = x * pow(y, -1, p) % p
            assert r * y % p == x
            out["div"].append({"modulus": name, "a": hex(x), "b": hex(y), "r": hex(r)})
    with open(os.path.join(HERE, "modular_kat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
