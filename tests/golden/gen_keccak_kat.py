#!/usr/bin/env python3
"""Known-answer vectors for Keccak -> tests/golden/keccak_kat.json.  Source of truth OUTSIDE this repository: Python's hashlib
(OpenSSL / the XKCP reference code) for SHA3-256 and, through it, the permutation itself: a message shorter than the rate is padded
into ONE block, so SHA3-256(m) is the first 32 bytes of Keccak-f[1600](pad(m)) -- every vector below is therefore also a
(state in, first four lanes out) vector of the permutation.  Plus the published digests every implementation is checked against
(FIPS 202 example values for SHA3-256, the Ethereum Keccak-256 of the empty string, and the Keccak team's zero-state lane)."""
import hashlib
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rnd = random.Random(20260101)
    msgs = [b"", b"abc", b"a" * 135, b"a" * 136, b"a" * 137, bytes(range(200)), b"The quick brown fox jumps over the lazy dog"]
    msgs += [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([1, 31, 64, 100, 135, 136, 271, 272, 273, 1000]))) for _ in range(24)]
    out = {"about": "SHA3-256 digests from hashlib (generator: tests/golden/gen_keccak_kat.py) + published constants",
           "sha3_256": [{"msg": m.hex(), "digest": hashlib.sha3_256(m).hexdigest()} for m in msgs],
           "published": {
               "sha3_256_empty": "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a",
               "sha3_256_abc": "3a985da74fe225b2045c172d6bd390bd855f086e3e9d525b46bfe24511431532",
               "keccak_256_empty": "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470",
               "keccak_f1600_of_zero_state_lane_0_0": "f1258f7940e1dde7"}}
    assert out["sha3_256"][0]["digest"] == out["published"]["sha3_256_empty"] and out["sha3_256"][1]["digest"] == out["published"]["sha3_256_abc"]
    with open(os.path.join(HERE, "keccak_kat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out["sha3_256"]), "vectors")


if __name__ == "__main__":
    main()
