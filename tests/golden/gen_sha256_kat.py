#!/usr/bin/env python3
"""Known-answer vectors for SHA-256 -> tests/golden/sha256_kat.json.  Source of truth OUTSIDE this repository: Python's hashlib (OpenSSL)
and the example digests of FIPS 180-4 / the NIST example-values document (empty string, "abc", the 448-bit two-block message), which the
generator checks hashlib against before writing.  A message is hashed by chaining the compression function over its padded blocks, so
every vector is also a known answer for the compression chip (H_in = IV or the previous block's H_out)."""
import hashlib
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rnd = random.Random(20260202)
    two_block = b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq"
    msgs = [b"", b"abc", two_block, b"a" * 55, b"a" * 56, b"a" * 63, b"a" * 64, b"a" * 65, bytes(range(256)), b"The quick brown fox jumps over the lazy dog"]
    msgs += [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([1, 31, 55, 56, 64, 100, 119, 120, 128, 300, 1000]))) for _ in range(24)]
    out = {"about": "SHA-256 digests from hashlib (generator: tests/golden/gen_sha256_kat.py) + the standard's example digests",
           "sha256": [{"msg": m.hex(), "digest": hashlib.sha256(m).hexdigest()} for m in msgs],
           "published": {
               "sha256_empty": "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855",
               "sha256_abc": "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad",
               "sha256_448_bits": "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"}}
    pub = out["published"]
    assert [v["digest"] for v in out["sha256"][:3]] == [pub["sha256_empty"], pub["sha256_abc"], pub["sha256_448_bits"]]
    with open(os.path.join(HERE, "sha256_kat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out["sha256"]), "vectors")


if __name__ == "__main__":
    main()
