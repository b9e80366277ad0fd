"""Writes ecc_kat.json: point additions and doublings on the three curves the reference's chunk circuit configures
(crates/circuits/chunk-circuit/openvm.toml:38-59: secp256k1, P-256, bn254 G1), computed with Python's integers and the textbook chord /
tangent formulas, anchored on published multiples of the standard generators (SEC 2 / NIST / EIP-196 test vectors): 2G and 3G of
secp256k1, 2G of P-256, 2G of bn254 G1.  Run: python tests/golden/gen_ecc_kat.py"""
import json
import os

CURVES = {
    "secp256k1": dict(p=2**256 - 2**32 - 977, a=0, b=7,
                      g=(0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)),
    "p256": dict(p=2**256 - 2**224 + 2**192 + 2**96 - 1, a=2**256 - 2**224 + 2**192 + 2**96 - 4,
                 b=0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
                 g=(0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296, 0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)),
    "bn254": dict(p=21888242871839275222246405745257275088696311157297823662689037894645226208583, a=0, b=3, g=(1, 2)),
}
PUBLISHED = {
    ("secp256k1", 2): (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5, 0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A),
    ("secp256k1", 3): (0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9, 0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672),
    ("p256", 2): (0x7CF27B188D034F7E8A52380304B51AC3C08969E277F21B35A60B48FC47669978, 0x07775510DB8ED040293D9AC69F7430DBBA7DADE63CE982299E04B79D227873D1),
    ("bn254", 2): (1368015179489954701390400359078579693043519447331113978918064868415326638035, 9918110051302171585080402603319702774565515993150576347155970296011118125764),
}


def add(c, p1, p2):
    p = c["p"]
    if p1 == p2:
        lam = (3 * p1[0] * p1[0] + c["a"]) * pow(2 * p1[1], -1, p) % p
    else:
        lam = (p2[1] - p1[1]) * pow(p2[0] - p1[0], -1, p) % p
    x3 = (lam * lam - p1[0] - p2[0]) % p
    return lam, (x3, (lam * (p1[0] - x3) - p1[1]) % p)


def main():
    out = {"curves": {}, "cases": []}
    for name, c in CURVES.items():
        p, g = c["p"], c["g"]
        assert (g[1] ** 2 - g[0] ** 3 - c["a"] * g[0] - c["b"]) % p == 0
        out["curves"][name] = {k: hex(c[k]) for k in ("p", "a", "b")} | {"gx": hex(g[0]), "gy": hex(g[1])}
        mult = {1: g}
        for k in range(2, 12):
            mult[k] = add(c, mult[k - 1], g)[1]
            assert (mult[k][1] ** 2 - mult[k][0] ** 3 - c["a"] * mult[k][0] - c["b"]) % p == 0
        for (cn, k), pt in PUBLISHED.items():
            if cn == name:
                assert mult[k] == pt, (cn, k)
        cases = [(1, k, k) for k in range(1, 6)] + [(0, i, j) for i, j in ((1, 2), (2, 1), (3, 5), (7, 2), (4, 6), (10, 1), (5, 6))]
        for op, i, j in cases:
            lam, r = add(c, mult[i], mult[j])
            assert r == mult[i + j]
            out["cases"].append(dict(curve=name, op=op, i=i, j=j, x1=hex(mult[i][0]), y1=hex(mult[i][1]), x2=hex(mult[j][0]), y2=hex(mult[j][1]), slope=hex(lam),
                                     x3=hex(r[0]), y3=hex(r[1])))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ecc_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
