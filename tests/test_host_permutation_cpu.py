"""CPU: the host's AVX-512 form of the Poseidon2 permutation (csrc/poseidon2_avx512.cpp: the transcript's long absorptions, the verifier and
the aggregation witness generator hash with it) == the scalar host permutation == the oracle's, on random and extreme states; the sponge
built from it == the oracle's hash of the same words (through zkhip_verify's own path: a proof verifies either way -- covered by the
verifier tests; here the permutation itself)."""
import numpy as np
import pytest

import zkvm_prover_amd as z

P = 2013265921


def test_avx512_permutation_equals_scalar_and_oracle(ora):
    if z.poseidon2_permute_host_avx512(np.zeros(16, np.uint32)) is None:
        pytest.skip("this CPU has no AVX-512: the scalar host permutation and the device sponge are used")
    rng = np.random.default_rng(7)
    states = [np.zeros(16, np.uint32), np.full(16, P - 1, np.uint32), np.arange(16, dtype=np.uint32)]
    e = np.zeros(16, np.uint32)
    e[0] = P - 1
    states.append(e)
    states += [rng.integers(0, P, 16, dtype=np.uint64).astype(np.uint32) for _ in range(500)]
    for s in states:
        fast, slow = z.poseidon2_permute_host_avx512(s), z.poseidon2_permute_host(s)
        assert (fast == slow).all()
    for s in states[:40]:
        assert (z.poseidon2_permute_host_avx512(s) == ora.permute(s.reshape(1, 16))[0]).all()
    # a chain (the sponge's shape): errors would compound
    s = np.arange(16, dtype=np.uint32)
    t = s.copy()
    for _ in range(200):
        s, t = z.poseidon2_permute_host_avx512(s), z.poseidon2_permute_host(t)
    assert (s == t).all()


def test_sixteen_permutations_side_by_side():
    """zkhip_poseidon2_permute16_host (one state word of sixteen independent states per 512-bit register: what the aggregation witness
    generator advances its queries with) == sixteen calls of the scalar permutation, bit for bit."""
    import ctypes as C

    import numpy as np

    import zkvm_prover_amd as z
    from zkvm_prover_amd import _binding as zb

    lib = z.load_library()
    lib.zkhip_poseidon2_permute16_host.argtypes = [C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(16)
    for trial in range(20):
        st = rng.integers(0, z.P, 256, dtype=np.uint64).astype(np.uint32)
        if trial == 0:
            st[:] = 0
        if trial == 1:
            st[:] = z.P - 1
        got = st.copy()
        assert lib.zkhip_poseidon2_permute16_host(zb._u32p(got)) == 0
        for k in range(16):
            s = st[16 * k:16 * k + 16].copy()
            assert lib.zkhip_poseidon2_permute_host(zb._u32p(s)) == 0
            assert (s == got[16 * k:16 * k + 16]).all(), (trial, k)
    bad = np.zeros(256, np.uint32)
    bad[77] = z.P
    assert lib.zkhip_poseidon2_permute16_host(zb._u32p(bad)) != 0
