"""zkvm-prover_amd -- host-side Python binding of libzkhip.so (MI355X STARK backend).

The product is the C-ABI library (include/zkhip.h, sources in csrc/).  This module is the thin
ctypes binding used by tests, bench.py and __graft_entry__; PyTorch only supplies device
memory, streams and torch.distributed.  There is no CPU fallback: if the HIP library is missing
or no gfx950 device is present every call raises.
"""
from ._binding import (  # noqa: F401
    P,
    Context,
    MerkleTree,
    Transcript,
    ZkhipError,
    library_path,
    load_library,
    declared_symbols,
    ProvingKey,
    RecursionCircuit,
    key_commit,
    vk_digest,
    keccak_f_air,
    keccak_f1600_host,
    int256_alu_air,
    int256_alu_host,
    int256_cmp_air,
    int256_shift_air,
    int256_mul_air,
    ec_air,
    fp2_air,
    fp2_host,
    vm_fp2_air,
    ec_host,
    vm_ec_air,
    modmul_air,
    modmul_host,
    modular_host,
    vm_modmul_air,
    sha256_air,
    sha256_compress_host,
    verify,
    proof_layout,
    DEFAULT_PARAMS,
    poseidon2_permute_host,
    poseidon2_permute_host_avx512,
    mmcs_verify,
    fri_fold_row,
    proof_decode_v1,
    proof_reencode_v1,
    proof_to_v1,
    proof_from_v1,
    V1_SINGLE,
    V1_VEC,
)
from . import air  # noqa: F401
