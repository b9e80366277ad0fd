"""CPU: the SHA-256 compression chip (include/zkhip_sha256.hpp) -- parity anchored OUTSIDE this repository.
  * the oracle's compression function / padded hash (oracle/sha256.c) and the product's host compression (zkhip_sha256_compress_host)
    against FIPS 180-4 through hashlib (tests/golden/sha256_kat.json) and the standard's example digests;
  * the chip's trace (oracle twin of zkhip_sha256_tracegen) satisfies the AIR with its preprocessed gates; the digest rows of every
    message's last block hold hashlib's digest; a flipped cell anywhere breaks a constraint; the oracle proves it and both verifiers accept."""
import hashlib
import json
import os

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

from sha256_util import IV, ROWS, WIDTH, chained_records, digest_of_row, ora_compress, ora_sha256, ora_trace, padded_blocks

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = (1, 0, 4, 3, 3)
NOPV = np.zeros(0, np.uint32)


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "sha256_kat.json")) as f:
        return json.load(f)


def test_compression_and_hash_against_fips180(ora, kat):
    pub = kat["published"]
    assert ora_sha256(ora, b"") == pub["sha256_empty"] and ora_sha256(ora, b"abc") == pub["sha256_abc"]
    assert ora_sha256(ora, b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq") == pub["sha256_448_bits"]
    for v in kat["sha256"]:
        msg = bytes.fromhex(v["msg"])
        assert ora_sha256(ora, msg) == v["digest"] == hashlib.sha256(msg).hexdigest()
        h = np.array(IV, np.uint32)
        ho = h.copy()
        for blk in padded_blocks(msg):               # the product's host compression, chained over the padded blocks
            h = z.sha256_compress_host(h, blk)
            ho = ora_compress(ora, ho, blk)
            assert (h == ho).all()
        assert b"".join(int(w).to_bytes(4, "big") for w in h).hex() == v["digest"]


def test_trace_satisfies_the_air_and_exports_the_digests(ora, kat):
    program, width, prep = z.sha256_air(9)
    assert width == WIDTH and prep.shape == (6, 512) and air.quotient_chunks(program) <= 2
    msgs = [bytes.fromhex(v["msg"]) for v in kat["sha256"] if len(v["msg"]) <= 2 * 100][:5]
    recs, last = chained_records(msgs, lambda h, m: ora_compress(ora, h, m))
    assert len(recs) <= 512 // ROWS
    tr = ora_trace(ora, recs, 9)                         # 7 whole blocks fit 512 rows; the rest are zero-record blocks / zero rows
    assert air.check_trace(program, tr, NOPV, prep=prep) == []
    for m, b in zip(msgs, last):
        assert digest_of_row(tr, ROWS * b + 64).hex() == hashlib.sha256(m).hexdigest()
        assert digest_of_row(tr, ROWS * b) == b"".join(int(w).to_bytes(4, "big") for w in recs[b][:8])
    assert int(tr[432].sum()) == ROWS * len(recs)
    rng = np.random.default_rng(3)
    for _ in range(60):
        bad = tr.copy()
        c, r = int(rng.integers(0, WIDTH)), int(rng.integers(0, ROWS * len(recs)))
        bad[c, r] ^= 1
        assert air.check_trace(program, bad, NOPV, prep=prep) != [], (c, r)
    # a wrong digest with consistent bits elsewhere: flip one bit of H_out in the digest row only
    bad = tr.copy()
    bad[5, 64] ^= 1
    assert air.check_trace(program, bad, NOPV, prep=prep) != []


def test_oracle_proves_the_chip(ora):
    program, width, prep = z.sha256_air(8)
    rng = np.random.default_rng(5)
    recs = rng.integers(0, 1 << 32, size=(3, 24), dtype=np.uint64).astype(np.uint32)
    tr = ora_trace(ora, recs, 8)
    inst = [dict(program=program, log_height=8, width=width, n_pvs=0, trace=tr, pvs=NOPV, prep=prep)]
    proof = ora.stark_prove(PARAMS, inst)
    assert ora.stark_verify(PARAMS, inst, proof) == 0
    vk = [dict(program=program, log_height=8, width=width, n_pvs=0, prep_commit=ora.prep_commit(PARAMS, inst[0]))]
    assert z.verify(PARAMS, vk, [NOPV], proof.tobytes()) == 0     # product verifier: the commitment of the gates / constants only
    for p in range(3):
        out = z.sha256_compress_host(recs[p][:8], recs[p][8:])
        assert digest_of_row(tr, ROWS * p + 64) == b"".join(int(w).to_bytes(4, "big") for w in out)
