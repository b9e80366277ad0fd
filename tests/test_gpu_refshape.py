"""The reference-shaped AIR set (tests/test_refshape_cpu.py) on the GPU: proof bytes == oracle at shrunken heights; at the
reference's real heights (2^1 .. 2^21 rows, 172 M main cells, blow-up 4) the proof verifies and converts to the v1 container
with the stored proof's structure."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refproof_v1 as rp  # noqa: E402
from test_refshape_cpu import PARAMS, check_against_reference_shape  # noqa: E402

import zkvm_prover_amd as z  # noqa: E402
from zkvm_prover_amd import air  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    return z.Context(0)


@pytest.fixture(scope="module")
def ora():
    import oracle_lib

    return oracle_lib


@pytest.mark.parametrize("shrink", [12, 8])
def test_reference_shaped_proof_matches_oracle(zk, ora, shrink):
    airs = air.ReferenceShapedSet(shrink=shrink).gen()
    exp = ora.stark_prove(PARAMS, airs).tobytes()
    pk = z.ProvingKey(zk, PARAMS, airs)
    pvs = [a["pvs"] for a in airs]
    got = pk.prove([zk.upload(a["trace"].reshape(-1)) for a in airs], pvs)
    assert got == exp
    vk = pk.verifying_airs()
    assert z.verify(PARAMS, vk, pvs, got) == 0
    p = rp.decode_proofs((1).to_bytes(8, "little") + z.proof_to_v1(PARAMS, vk, pvs, got))[0]
    check_against_reference_shape(rp.shape_of(p), shrink)
    pk.close()


def test_reference_shaped_proof_full_heights(zk):
    airs = air.ReferenceShapedSet(shrink=0).gen()
    pk = z.ProvingKey(zk, PARAMS, airs)
    pvs = [a["pvs"] for a in airs]
    d_traces = [zk.upload(a["trace"].reshape(-1)) for a in airs]
    got = pk.prove(d_traces, pvs)
    assert pk.prove(d_traces, pvs) == got   # deterministic
    vk = pk.verifying_airs()
    assert z.verify(PARAMS, vk, pvs, got) == 0
    bad = bytearray(got)
    bad[len(bad) // 2] ^= 1
    assert z.verify(PARAMS, vk, pvs, bytes(bad)) != 0
    v1 = z.proof_to_v1(PARAMS, vk, pvs, got)
    p = rp.decode_proofs((1).to_bytes(8, "little") + v1)[0]
    check_against_reference_shape(rp.shape_of(p), 0)
    back, _ = z.proof_from_v1(PARAMS, vk, v1)
    assert back == got
    pk.close()
