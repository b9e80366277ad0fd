"""Writes fp2_kat.json: multiplications, divisions, additions and subtractions in Fp[u] / (u^2 + 1) for the field the reference's chunk
circuit configures (crates/circuits/chunk-circuit/openvm.toml:30-33: Bn254Fp2), computed with Python's integers; anchored on bn254's
twist: the published G2 generator satisfies y^2 = x^3 + 3 / (9 + u) when squared, cubed and divided through these formulas
(EIP-197's generator).  Run: python tests/golden/gen_fp2_kat.py"""
import json
import os
import random

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
# EIP-197 G2 generator: x = X0 + X1 u, y = Y0 + Y1 u
G2X = (10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634)
G2Y = (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531)


def mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * n % P, -a[1] * n % P)


def op(o, a, b):
    if o == 0:
        return mul(a, b)
    if o == 1:
        return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
    if o == 2:
        return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
    return mul(a, inv(b))


def main():
    # the twist equation at the generator: y^2 = x^3 + 3 / (9 + u)
    lhs = mul(G2Y, G2Y)
    rhs = op(1, mul(mul(G2X, G2X), G2X), op(3, (3, 0), (9, 1)))
    assert lhs == rhs
    rnd = random.Random(20261005)
    pairs = [((0, 0), (1, 0)), ((1, 0), (1, 0)), ((0, 1), (0, 1)), ((P - 1, P - 1), (P - 1, P - 1)), ((P - 1, 0), (0, P - 1)), ((1, 1), (1, P - 1)), (G2X, G2Y), (G2Y, G2X),
             ((3, 0), (9, 1))]
    pairs += [((rnd.randrange(P), rnd.randrange(P)), (rnd.randrange(P), rnd.randrange(P))) for _ in range(9)]
    cases = []
    for o in range(4):
        for a, b in pairs:
            r = op(o, a, b)
            cases.append({"op": o, "a0": hex(a[0]), "a1": hex(a[1]), "b0": hex(b[0]), "b1": hex(b[1]), "r0": hex(r[0]), "r1": hex(r[1])})
    out = {"about": "Fp2 = Fp[u] / (u^2 + 1) over bn254's base field, from Python integers (generator: tests/golden/gen_fp2_kat.py); op: 0 mul 1 add 2 sub 3 div",
           "p": hex(P), "g2": {"x0": hex(G2X[0]), "x1": hex(G2X[1]), "y0": hex(G2Y[0]), "y1": hex(G2Y[1])}, "cases": cases}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp2_kat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
