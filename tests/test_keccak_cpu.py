"""CPU: the Keccak-f[1600] chip (include/zkhip_keccak.hpp) -- the first f3 chip with parity anchored OUTSIDE this repository.
  * the oracle's permutation / sponge (oracle/keccak.c) and the product's host permutation (zkhip_keccak_f1600_host) against FIPS 202
    through hashlib's SHA3-256 (tests/golden/keccak_kat.json) and the published digests;
  * the chip's trace (oracle twin of zkhip_keccak_f_tracegen) satisfies the AIR; its export rows hold exactly those permutations'
    outputs; a flipped bit anywhere breaks a constraint; the oracle proves it and both verifiers accept."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = (1, 0, 4, 3, 3)
NOPV = np.zeros(0, np.uint32)


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "keccak_kat.json")) as f:
        return json.load(f)


def ora_sha3(ora, msg, keccak_padding=False):
    l = ora.lib()
    l.ora_sha3_256.restype = None
    l.ora_sha3_256.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint8), C.c_int]
    out = (C.c_uint8 * 32)()
    l.ora_sha3_256(msg, len(msg), out, 1 if keccak_padding else 0)
    return bytes(out).hex()


def ora_trace(ora, states, log_height):
    l = ora.lib()
    l.ora_keccak_f_trace.restype = None
    l.ora_keccak_f_trace.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32)]
    st = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 25)
    tr = np.zeros((2633, 1 << log_height), np.uint32)
    l.ora_keccak_f_trace(st.ctypes.data_as(C.POINTER(C.c_uint64)), len(st), log_height, ora.p32(tr))
    return tr


def one_block_state(msg):
    """pad10*1 with the SHA-3 domain bits into ONE rate block (messages shorter than 136 bytes) as 25 lanes"""
    assert len(msg) < 136
    blk = bytearray(200)
    blk[:len(msg)] = msg
    blk[len(msg)] ^= 0x06
    blk[135] ^= 0x80
    return np.frombuffer(bytes(blk), dtype="<u8").copy()


def test_permutation_and_sponge_against_fips202(ora, kat):
    pub = kat["published"]
    assert ora_sha3(ora, b"") == pub["sha3_256_empty"] and ora_sha3(ora, b"abc") == pub["sha3_256_abc"]
    assert ora_sha3(ora, b"", keccak_padding=True) == pub["keccak_256_empty"]
    for v in kat["sha3_256"]:
        msg = bytes.fromhex(v["msg"])
        assert ora_sha3(ora, msg) == v["digest"] == hashlib.sha3_256(msg).hexdigest()
        if len(msg) < 136:   # one block: the digest IS the permutation's first four lanes
            out = z.keccak_f1600_host(one_block_state(msg))
            assert out[:4].tobytes().hex() == v["digest"]
    zero = z.keccak_f1600_host(np.zeros(25, np.uint64))
    assert "%016x" % int(zero[0]) == pub["keccak_f1600_of_zero_state_lane_0_0"]
    l = ora.lib()
    l.ora_keccak_f1600.restype = None
    l.ora_keccak_f1600.argtypes = [C.POINTER(C.c_uint64)]
    st = np.zeros(25, np.uint64)
    l.ora_keccak_f1600(st.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert (st == zero).all()


def test_trace_satisfies_the_air_and_exports_the_permutations(ora, kat):
    program, width = z.keccak_f_air()
    assert width == 2633 and air.quotient_chunks(program) <= 2
    msgs = [bytes.fromhex(v["msg"]) for v in kat["sha3_256"] if len(v["msg"]) < 2 * 136][:5]
    states = np.stack([one_block_state(m) for m in msgs])
    tr = ora_trace(ora, states, 7)                       # 5 permutations = 120 rows, 8 rows of zero-state padding
    assert air.check_trace(program, tr, NOPV) == []
    limbs = lambda cols, row: [int(tr[c, row]) for c in cols]   # noqa: E731
    for p, m in enumerate(msgs):
        row = 24 * p + 23
        assert tr[24, row] == 1
        # output lanes 0..3 (y = 0, x = 0..3): a''' for (0, 0), a'' for the others
        lanes = [limbs(range(2629, 2633), row)] + [limbs(range(2465 + 4 * x, 2469 + 4 * x), row) for x in (1, 2, 3)]
        digest = b"".join(sum(v << (16 * i) for i, v in enumerate(l)).to_bytes(8, "little") for l in lanes)
        assert digest.hex() == hashlib.sha3_256(m).hexdigest()
        pre = [sum(int(tr[25 + 4 * i + k, row]) << (16 * k) for k in range(4)) for i in range(25)]
        assert pre == states[p].tolist()
    assert tr[24].sum() == 5
    rng = np.random.default_rng(3)
    for _ in range(40):
        bad = tr.copy()
        c, r = int(rng.integers(0, 2633)), int(rng.integers(0, 120))
        bad[c, r] ^= 1
        assert air.check_trace(program, bad, NOPV) != [], (c, r)


def test_oracle_proves_the_chip(ora):
    program, width = z.keccak_f_air()
    rng = np.random.default_rng(5)
    states = rng.integers(0, 1 << 63, size=(2, 25), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(2, 25), dtype=np.uint64)
    tr = ora_trace(ora, states, 6)
    inst = [dict(program=program, log_height=6, width=width, n_pvs=0, trace=tr, pvs=NOPV)]
    proof = ora.stark_prove(PARAMS, inst)
    assert ora.stark_verify(PARAMS, inst, proof) == 0
    assert z.verify(PARAMS, inst, [NOPV], proof.tobytes()) == 0
    for p in range(2):
        out = z.keccak_f1600_host(states[p])
        got = [sum(int(tr[c + k, 24 * p + 23]) << (16 * k) for k in range(4)) for c in [2629] + [2465 + 4 * i for i in range(1, 25)]]
        assert got == out.tolist()
