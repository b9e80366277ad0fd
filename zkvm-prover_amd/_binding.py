"""ctypes binding of include/zkhip.h.  Device buffers are torch int32 tensors holding the u32 bit
patterns (Montgomery form on device, see zkhip.h)."""
import ctypes as C
import os
import re

import numpy as np

P = 2013265921
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = None


class ZkhipError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "libzkhip.so")


def declared_symbols():
    """Every function name declared in include/zkhip.h."""
    with open(os.path.join(_ROOT, "include", "zkhip.h")) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", txt)))


class _Matrix(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_size_t), ("log_height", C.c_uint), ("width", C.c_size_t)]


class _Params(C.Structure):
    _fields_ = [("log_blowup", C.c_uint32), ("log_final_poly_len", C.c_uint32), ("num_queries", C.c_uint32),
                ("commit_pow_bits", C.c_uint32), ("query_pow_bits", C.c_uint32)]


class _Air(C.Structure):
    _fields_ = [("program", C.POINTER(C.c_uint32)), ("program_len", C.c_size_t), ("log_height", C.c_uint),
                ("width", C.c_size_t), ("n_pvs", C.c_size_t), ("prep_trace", C.POINTER(C.c_uint32)),
                ("prep_commit", C.POINTER(C.c_uint32))]


class _ProofLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("n_words", "root_main", "roots_cached", "n_cached", "root_perm", "exposed", "n_exposed", "root_quot", "opened",
                                          "n_opened", "fri_layers", "n_fri_layers", "final_poly", "n_final_poly", "query_pow", "queries",
                                          "query_words", "n_queries")]


class _V1Summary(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("n_proofs", "n_airs", "n_queries", "n_fri_layers", "n_final_poly", "n_main_commits",
                                          "n_after_challenge_commits", "n_preprocessed", "n_input_batches")] + \
               [("log_max_height", C.c_uint), ("log_blowup", C.c_uint), ("has_logup_pow", C.c_int), ("log_degree", C.c_uint * 64)]


class _RecursionStmt(C.Structure):
    _fields_ = [("n_state", C.c_size_t), ("start_air", C.POINTER(C.c_uint32)), ("start_idx", C.POINTER(C.c_uint32)),
                ("end_air", C.POINTER(C.c_uint32)), ("end_idx", C.POINTER(C.c_uint32)), ("child_is_node", C.c_int),
                ("uniform", C.c_int), ("min_log_height", C.c_uint * 2), ("n_leaf_shapes", C.c_size_t), ("app_id", C.POINTER(C.c_uint32)),
                ("region_index", C.c_uint32)]


class Config(C.Structure):
    """include/zkhip.h zkhip_config: every switch of the library as a field (the ZKHIP_* variables are overrides read in one place)."""
    _fields_ = [("host_sponge", C.c_int), ("host_sponge_min_words", C.c_uint32), ("jit", C.c_int), ("jit_min_log_work", C.c_uint32),
                ("jit_cache_dir", C.c_char * 256), ("quot_slices", C.c_int), ("grind_sweep_shift", C.c_uint32), ("coop_max_log", C.c_uint32),
                ("coop_inj_max_log", C.c_uint32), ("top_max_log", C.c_uint32), ("commit_parts", C.c_uint32), ("side_cus", C.c_uint32), ("witness_threads", C.c_uint32),
                ("pin_witness", C.c_int), ("parallel_queries", C.c_int), ("self_check", C.c_int), ("tree_store_early", C.c_int), ("hash_block", C.c_uint32), ("coop_fused", C.c_int),
                ("rows_in_bulk", C.c_int), ("rows_coop_max_log", C.c_uint32), ("ntt_log_lanes", C.c_uint32), ("quot_streams", C.c_uint32)]

    @classmethod
    def default(cls):
        c = cls()
        load_library().zkhip_config_default(C.byref(c))
        return c


class _KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


def load_library():
    """Loads libzkhip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ZkhipError("libzkhip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(or make -C zkvm-prover_amd/csrc)")
    # torch first: its wheel bundles a HIP runtime, and the process must end up with ONE runtime.  If libzkhip.so
    # (linked against /opt/rocm) were loaded before torch, two copies of libamdhip64 would be live and the second
    # one to initialise would not see the GPU.
    import torch  # noqa: F401

    lib = C.CDLL(path)
    vp, sz, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)
    sig = {
        "zkhip_version": (C.c_uint32, []),
        "zkhip_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "zkhip_ctx_destroy": (None, [vp]),
        "zkhip_last_error": (C.c_char_p, [vp]),
        "zkhip_set_stream": (C.c_int, [vp, vp]),
        "zkhip_sync": (C.c_int, [vp]),
        "zkhip_set_commit_pipeline": (C.c_int, [vp, C.c_uint]),
        "zkhip_set_cu_partition": (C.c_int, [vp, C.c_uint]),
        "zkhip_malloc": (C.c_int, [vp, sz, C.POINTER(vp)]),
        "zkhip_free": (C.c_int, [vp, vp]),
        "zkhip_h2d": (C.c_int, [vp, vp, vp, sz]),
        "zkhip_d2h": (C.c_int, [vp, vp, vp, sz]),
        "zkhip_to_monty": (C.c_int, [vp, vp, sz]),
        "zkhip_from_monty": (C.c_int, [vp, vp, sz]),
        "zkhip_ntt_batch": (C.c_int, [vp, vp, C.c_uint, sz, sz, C.c_int, C.c_int]),
        "zkhip_lde_batch": (C.c_int, [vp, vp, sz, vp, sz, C.c_uint, C.c_uint, sz, C.c_uint32]),
        "zkhip_poseidon2_permute_batch": (C.c_int, [vp, vp, sz]),
        "zkhip_poseidon2_air_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp]),
        "zkhip_range_counts_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp, C.c_int]),
        "zkhip_range_tuple_counts_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint32, C.c_uint32, vp, C.c_int]),
        "zkhip_bitwise_lookup_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, C.c_int]),
        "zkhip_rv32_alu_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_rv32_lt_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_rv32_shift_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_rv32_branch_eq_tracegen": (C.c_int, [vp, vp, vp, vp, vp, sz, C.c_uint, vp]),
        "zkhip_rv32_branch_lt_tracegen": (C.c_int, [vp, vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_var_range_counts_tracegen": (C.c_int, [vp, vp, vp, C.c_uint32, sz, C.c_uint, vp, C.c_int]),
        "zkhip_domain_point_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp]),
        "zkhip_duplex_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_fri_fold_chip_tracegen": (C.c_int, [vp, vp, vp, vp, vp, vp, sz, C.c_uint, vp]),
        "zkhip_castf_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp, vp, C.c_uint]),
        "zkhip_field_arith_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp]),
        "zkhip_field_ext_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp]),
        "zkhip_mmcs_path_tracegen": (C.c_int, [vp, vp, vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_rv32_divrem_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp, C.c_uint32, C.c_uint32, vp]),
        "zkhip_rv32_mulh_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp, C.c_uint32, C.c_uint32, vp]),
        "zkhip_rv32_loadstore_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_rv32_jal_lui_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_rv32_auipc_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_rv32_jalr_tracegen": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_rv32_mul_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_memory_boundary_tracegen": (C.c_int, [vp, vp, vp, vp, vp, vp, sz, C.c_uint, C.c_uint, C.c_uint, vp]),
        "zkhip_memory_access_tracegen": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, sz, C.c_uint, vp]),
        "zkhip_program_freq_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp]),
        "zkhip_exec_frame_tracegen": (C.c_int, [vp, vp, sz, vp, sz, C.c_uint, vp]),
        "zkhip_merkle_commit": (C.c_int, [vp, C.POINTER(_Matrix), sz, C.POINTER(vp), u32p]),
        "zkhip_tree_root_device": (vp, [vp]),
        "zkhip_tree_log_height": (C.c_uint, [vp]),
        "zkhip_tree_layer": (C.c_int, [vp, vp, C.c_uint, u32p]),
        "zkhip_merkle_opening_words": (sz, [vp]),
        "zkhip_merkle_open": (C.c_int, [vp, vp, C.POINTER(C.c_uint64), sz, u32p, sz]),
        "zkhip_tree_destroy": (None, [vp, vp]),
        "zkhip_constraint_eval": (C.c_int, [vp, u32p, sz, C.c_uint, C.c_uint, sz, vp, u32p, sz, u32p, vp]),
        "zkhip_fri_fold": (C.c_int, [vp, vp, vp, C.c_uint, u32p]),
        "zkhip_ext_batch_inverse": (C.c_int, [vp, vp, vp, sz]),
        "zkhip_logup_running_sum": (C.c_int, [vp, vp, vp, sz, vp, u32p]),
        "zkhip_mle_fold": (C.c_int, [vp, vp, vp, sz, u32p]),
        "zkhip_sumcheck_round": (C.c_int, [vp, C.POINTER(vp), sz, sz, u32p]),
        "zkhip_transcript_create": (C.c_int, [vp, C.POINTER(vp)]),
        "zkhip_transcript_destroy": (None, [vp, vp]),
        "zkhip_transcript_observe": (C.c_int, [vp, vp, u32p, sz]),
        "zkhip_transcript_sample": (C.c_int, [vp, vp, u32p, sz]),
        "zkhip_transcript_grind": (C.c_int, [vp, vp, C.c_uint, u32p]),
        "zkhip_keygen": (C.c_int, [vp, C.POINTER(_Params), C.POINTER(_Air), sz, C.POINTER(vp)]),
        "zkhip_pk_destroy": (None, [vp, vp]),
        "zkhip_pk_prep_commitment": (C.c_int, [vp, vp, sz, u32p]),
        "zkhip_proof_size": (sz, [vp]),
        "zkhip_pk_workspace_bytes": (sz, [vp]),
        "zkhip_prove": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(u32p), C.POINTER(C.c_uint8), sz, C.POINTER(sz)]),
        "zkhip_prove_async": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(u32p)]),
        "zkhip_proof_fetch": (C.c_int, [vp, vp, C.POINTER(C.c_uint8), sz, C.POINTER(sz)]),
        "zkhip_verify": (C.c_int, [C.POINTER(_Params), C.POINTER(_Air), sz, C.POINTER(u32p), C.POINTER(C.c_uint8), sz]),
        "zkhip_verify_where": (C.c_int, [C.POINTER(_Params), C.POINTER(_Air), sz, C.POINTER(u32p), C.POINTER(C.c_uint8), sz, C.POINTER(C.c_int)]),
        "zkhip_poseidon2_permute_host": (C.c_int, [u32p]),
        "zkhip_poseidon2_permute_host_avx512": (C.c_int, [u32p]),
        "zkhip_mmcs_verify": (C.c_int, [u32p, C.POINTER(C.c_uint), C.POINTER(sz), sz, C.c_uint64, u32p]),
        "zkhip_fri_fold_row": (C.c_int, [C.c_uint64, C.c_uint, u32p, u32p, u32p, u32p]),
        "zkhip_logup_exposed_check": (C.c_int, [u32p, sz]),
        "zkhip_proof_decode_v1": (C.c_int, [C.POINTER(C.c_uint8), sz, C.c_int, C.POINTER(_V1Summary)]),
        "zkhip_proof_reencode_v1": (C.c_int, [C.POINTER(C.c_uint8), sz, C.c_int, C.POINTER(C.c_uint8), sz, C.POINTER(sz)]),
        "zkhip_proof_to_v1": (C.c_int, [C.POINTER(_Params), C.POINTER(_Air), sz, C.POINTER(u32p), C.POINTER(C.c_uint8), sz,
                                        C.POINTER(C.c_uint8), sz, C.POINTER(sz)]),
        "zkhip_proof_from_v1": (C.c_int, [C.POINTER(_Params), C.POINTER(_Air), sz, C.POINTER(C.c_uint8), sz,
                                          C.POINTER(C.c_uint8), sz, C.POINTER(sz), C.POINTER(u32p)]),
        "zkhip_proof_layout_of": (C.c_int, [C.POINTER(_Params), C.POINTER(_Air), sz, C.POINTER(_ProofLayout)]),
        "zkhip_tracegen_defer_checks": (C.c_int, [vp, C.c_int]),
        "zkhip_tracegen_check": (C.c_int, [vp]),
        "zkhip_keccak_f_air": (C.c_int, [C.POINTER(_Air)]),
        "zkhip_keccak_f1600_host": (C.c_int, [C.POINTER(C.c_uint64)]),
        "zkhip_keccak_f_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp]),
        "zkhip_int256_alu_air": (C.c_int, [C.c_uint32, C.POINTER(_Air)]),
        "zkhip_int256_alu_host": (C.c_int, [C.c_uint32, u32p, u32p, u32p]),
        "zkhip_int256_alu_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_vm_int256_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_int256_mul_air": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(_Air)]),
        "zkhip_int256_mul_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_vm_mul256_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_int256_cmp_air": (C.c_int, [C.c_uint32, C.POINTER(_Air)]),
        "zkhip_int256_shift_air": (C.c_int, [C.c_uint32, C.POINTER(_Air)]),
        "zkhip_int256_shift_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_vm_shift256_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_vm_native_arith_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp]),
        "zkhip_vm_native_ext_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp]),
        "zkhip_vm_castf_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp]),
        "zkhip_merkle_rebuild": (C.c_int, [vp, vp]),
        "zkhip_tree_check": (C.c_int, [vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "zkhip_int256_cmp_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_vm_cmp256_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp, vp]),
        "zkhip_modmul_air": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.POINTER(_Air)]),
        "zkhip_modmul_air_x": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_Air)]),
        "zkhip_modular_host_x": (C.c_int, [C.c_uint32, C.c_uint32, u32p, u32p, u32p, u32p, u32p]),
        "zkhip_modular_tracegen_x": (C.c_int, [vp, C.c_uint32, u32p, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_vm_modmul_air_x": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint, C.c_int, C.POINTER(_Air)]),
        "zkhip_vm_modmul_tracegen_x": (C.c_int, [vp, C.c_uint32, u32p, vp, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_fp2_air_x": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_Air)]),
        "zkhip_fp2_host_x": (C.c_int, [C.c_uint32, C.c_uint32, u32p, u32p, u32p, u32p]),
        "zkhip_fp2_tracegen_x": (C.c_int, [vp, C.c_uint32, u32p, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_vm_fp2_air_x": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint, C.c_int, C.POINTER(_Air)]),
        "zkhip_vm_fp2_tracegen_x": (C.c_int, [vp, C.c_uint32, u32p, vp, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_ec_air_x": (C.c_int, [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_Air)]),
        "zkhip_ec_host_x": (C.c_int, [C.c_uint32, C.c_uint32] + [u32p] * 9),
        "zkhip_ec_tracegen_x": (C.c_int, [vp, C.c_uint32, u32p, u32p, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_vm_ec_air_x": (C.c_int, [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_uint32, C.c_uint, C.c_int, C.POINTER(_Air)]),
        "zkhip_vm_ec_tracegen_x": (C.c_int, [vp, C.c_uint32, u32p, u32p, vp, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_modmul_host": (C.c_int, [u32p, u32p, u32p, u32p, u32p]),
        "zkhip_modmul_tracegen": (C.c_int, [vp, u32p, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_modular_tracegen": (C.c_int, [vp, u32p, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_modular_host": (C.c_int, [C.c_uint32, u32p, u32p, u32p, u32p, u32p]),
        "zkhip_sha256_air": (C.c_int, [C.c_uint, C.POINTER(_Air)]),
        "zkhip_sha256_compress_host": (C.c_int, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "zkhip_sha256_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp]),
        "zkhip_vm_n_airs": (sz, []),
        "zkhip_vm_air": (C.c_int, [C.c_uint, C.POINTER(_Air), C.POINTER(sz)]),
        "zkhip_vm_decode": (C.c_int, [C.c_uint32, C.c_uint32, u32p, C.POINTER(C.c_int)]),
        "zkhip_vm_program_table": (C.c_int, [u32p, sz, C.c_uint32, C.c_uint, u32p]),
        "zkhip_vm_frame_tracegen": (C.c_int, [vp] * 10 + [sz, vp, sz, C.c_uint, vp]),
        "zkhip_vm_loadstore_tracegen": (C.c_int, [vp] * 8 + [sz, C.c_uint, vp, vp]),
        "zkhip_vm_poseidon2_tracegen": (C.c_int, [vp, vp, sz, C.c_uint, vp]),
        "zkhip_vm_keccak_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp]),
        "zkhip_vm_sha256_tracegen": (C.c_int, [vp, vp, vp, sz, C.c_uint, vp]),
        "zkhip_vm_modmul_air": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint, C.c_int, C.POINTER(_Air)]),
        "zkhip_vm_modmul_tracegen": (C.c_int, [vp, u32p, vp, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_vm_sha256_prep": (C.c_int, [C.c_uint, u32p]),
        "zkhip_fp2_air": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.POINTER(_Air)]),
        "zkhip_fp2_host": (C.c_int, [C.c_uint32, u32p, u32p, u32p, u32p]),
        "zkhip_fp2_tracegen": (C.c_int, [vp, u32p, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_vm_fp2_air": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint, C.c_int, C.POINTER(_Air)]),
        "zkhip_vm_fp2_tracegen": (C.c_int, [vp, u32p, vp, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_ec_air": (C.c_int, [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.POINTER(_Air)]),
        "zkhip_ec_host": (C.c_int, [C.c_uint32] + [u32p] * 9),
        "zkhip_ec_tracegen": (C.c_int, [vp, u32p, u32p, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_vm_ec_air": (C.c_int, [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_uint, C.c_int, C.POINTER(_Air)]),
        "zkhip_vm_ec_tracegen": (C.c_int, [vp, u32p, u32p, vp, vp, sz, C.c_uint, vp, vp, vp, C.c_uint32, C.c_uint32]),
        "zkhip_rows_tracegen": (C.c_int, [vp, vp, sz, sz, C.c_uint, vp, u32p]),
        "zkhip_range_counts_scaled_tracegen": (C.c_int, [vp, vp, sz, C.c_uint32, C.c_uint, vp, C.c_int]),
        "zkhip_recursion_build": (C.c_int, [C.POINTER(_Params), C.POINTER(_Air), sz, sz, C.POINTER(_RecursionStmt), C.POINTER(vp)]),
        "zkhip_recursion_destroy": (None, [vp]),
        "zkhip_recursion_last_error": (C.c_char_p, [vp]),
        "zkhip_recursion_n_airs": (sz, [vp]),
        "zkhip_recursion_n_pvs": (sz, [vp]),
        "zkhip_recursion_n_state": (sz, [vp]),
        "zkhip_recursion_max_children": (sz, [vp]),
        "zkhip_recursion_child_proof_bytes": (sz, [vp]),
        "zkhip_recursion_stats": (C.c_int, [vp, C.POINTER(sz)]),
        "zkhip_recursion_child_vk_digest": (C.c_int, [vp, u32p]),
        "zkhip_recursion_air": (C.c_int, [vp, sz, C.POINTER(_Air)]),
        "zkhip_recursion_witness": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz), C.POINTER(C.POINTER(u32p)), sz, u32p]),
        "zkhip_recursion_wires": (C.c_int, [vp, u32p, sz, C.POINTER(sz)]),
        "zkhip_jit_prewarm": (C.c_int, [C.POINTER(_Air), sz, C.c_uint, C.c_char_p, C.POINTER(sz)]),
        "zkhip_tables_canonical": (C.c_int, [vp, C.c_int]),
        "zkhip_host_cpus": (C.c_uint, []),
        "zkhip_config_default": (None, [C.POINTER(Config)]),
        "zkhip_ctx_get_config": (C.c_int, [vp, C.POINTER(Config)]),
        "zkhip_has_test_kernels": (C.c_int, []),
        "zkhip_ctx_set_config": (C.c_int, [vp, C.POINTER(Config)]),
        "zkhip_set_process_config": (C.c_int, [C.POINTER(Config)]),
        "zkhip_recursion_build_join": (C.c_int, [C.POINTER(_Params), C.POINTER(_Air), sz, C.POINTER(_Params), C.POINTER(_Air), sz, C.POINTER(vp)]),
        "zkhip_recursion_witness_deferral": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz), C.POINTER(C.POINTER(u32p)), u32p, u32p, sz, u32p]),
        "zkhip_recursion_n_aux": (sz, [vp]),
        "zkhip_recursion_fork": (C.c_int, [vp, C.POINTER(vp)]),
        "zkhip_recursion_pad": (C.c_int, [vp, C.POINTER(C.c_uint)]),
        "zkhip_recursion_key_commit": (C.c_int, [u32p, sz, u32p]),
        "zkhip_recursion_vk_digest": (C.c_int, [C.POINTER(_Params), C.POINTER(_Air), sz, u32p]),
        "zkhip_recursion_witness_uniform": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz), C.POINTER(C.POINTER(u32p)), u32p,
                                                       C.POINTER(C.c_int), u32p, u32p, sz, u32p]),
        "zkhip_recursion_tracegen": (C.c_int, [vp, vp, vp, vp, vp]),
        "zkhip_profile_enable": (C.c_int, [vp, C.c_int]),
        "zkhip_profile_read": (C.c_int, [vp, C.POINTER(_KernelStat), sz]),
        "zkhip_profile_reset": (C.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _u32p(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_uint32))


class Context:
    """One per GPU (zkhip_ctx).  Work is issued on the torch stream that is CURRENT WHEN THE CONTEXT IS CREATED (so that it
    is ordered with the torch copies that fill its buffers).  Several contexts meant to run side by side must therefore be
    created under different streams (`with torch.cuda.stream(s): Context(0)`, as bench.py and tools/refshape_bench.py do):
    created under torch's default stream they all share it and their proofs run one after the other.  (The C++ side has no
    such coupling: zkhip_ctx_create gives every context a non-blocking stream of its own.)"""

    def __init__(self, device=0):
        import torch

        self.lib = load_library()
        if not torch.cuda.is_available():
            raise ZkhipError("no GPU visible: libzkhip has no CPU fallback")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        rc = self.lib.zkhip_ctx_create(device, C.byref(h))
        if rc != 0:
            raise ZkhipError("zkhip_ctx_create failed with %d (needs a gfx950 device)" % rc)
        self.h = h
        self.use_torch_stream()

    def use_torch_stream(self):
        import torch

        s = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.zkhip_set_stream(self.h, C.c_void_p(s)))

    def config(self):
        """This context's zkhip_config (a Config structure)."""
        c = Config()
        self._check(self.lib.zkhip_ctx_get_config(self.h, C.byref(c)))
        return c

    def set_config(self, cfg=None, **fields):
        """zkhip_ctx_set_config: a whole Config, or single fields of the current one (set_config(jit=2, quot_slices=0))."""
        c = cfg if cfg is not None else self.config()
        for k, v in fields.items():
            setattr(c, k, v.encode() if isinstance(v, str) else v)
        self._check(self.lib.zkhip_ctx_set_config(self.h, C.byref(c)))

    def close(self):
        if self.h:
            self.lib.zkhip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ZkhipError("zkhip error %d: %s" % (rc, self.lib.zkhip_last_error(self.h).decode()))

    def sync(self):
        self._check(self.lib.zkhip_sync(self.h))

    def set_commit_pipeline(self, parts):
        """Trace commit as a pipeline of `parts` column blocks (LDE of block k+1 beside the sponge of block k); 0 = off."""
        self._check(self.lib.zkhip_set_commit_pipeline(self.h, parts))

    def set_cu_partition(self, side_cus):
        """CU partition of the pipelined commit: `side_cus` CUs for the LDE stream, the rest for the sponge; 0 = off."""
        self._check(self.lib.zkhip_set_cu_partition(self.h, side_cus))

    # ---- data movement -----------------------------------------------------------------------
    def upload(self, arr_canonical):
        """numpy uint32 (canonical) -> device int32 tensor in Montgomery form."""
        import torch

        a = np.ascontiguousarray(arr_canonical, dtype=np.uint32)
        t = torch.from_numpy(a.view(np.int32)).to(self.device)
        self._check(self.lib.zkhip_to_monty(self.h, C.c_void_p(t.data_ptr()), t.numel()))
        return t

    def download(self, t):
        """device Montgomery tensor -> numpy uint32 canonical (tensor left untouched)."""
        c = t.clone()
        self._check(self.lib.zkhip_from_monty(self.h, C.c_void_p(c.data_ptr()), c.numel()))
        self.sync()
        return c.cpu().numpy().view(np.uint32)

    # ---- stages --------------------------------------------------------------------------------
    def ntt_batch(self, t, log_n, width, stride=None, inverse=False, bitrev_out=False):
        stride = stride or (1 << log_n)
        self._check(self.lib.zkhip_ntt_batch(self.h, C.c_void_p(t.data_ptr()), log_n, width, stride,
                                             int(inverse), int(bitrev_out)))

    def lde_batch(self, t_in, log_n, added_bits, width, shift, t_out=None, in_stride=None, out_stride=None):
        import torch

        in_stride = in_stride or (1 << log_n)
        out_stride = out_stride or (1 << (log_n + added_bits))
        if t_out is None:
            t_out = torch.empty(width * out_stride, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_lde_batch(self.h, C.c_void_p(t_in.data_ptr()), in_stride,
                                             C.c_void_p(t_out.data_ptr()), out_stride, log_n, added_bits, width,
                                             shift))
        return t_out

    def poseidon2_permute_batch(self, t, n):
        self._check(self.lib.zkhip_poseidon2_permute_batch(self.h, C.c_void_p(t.data_ptr()), n))

    def poseidon2_air_tracegen(self, t_inputs, log_height, t_out=None):
        """Device trace (298 columns x 2^log_height, column-major, Montgomery) of the Poseidon2 AIR for the
        Montgomery input states t_inputs ([n][16], device)."""
        import torch

        n = t_inputs.numel() // 16
        if t_out is None:
            t_out = torch.empty(298 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_poseidon2_air_tracegen(self.h, C.c_void_p(t_inputs.data_ptr()), n, log_height,
                                                          C.c_void_p(t_out.data_ptr())))
        return t_out

    def range_counts_tracegen(self, t_values, log_table, t_counts=None, accumulate=False):
        """Multiplicity column (2^log_table Montgomery words, device) of the Montgomery values in t_values."""
        import torch

        if t_counts is None:
            t_counts = torch.empty(1 << log_table, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_range_counts_tracegen(self.h, C.c_void_p(t_values.data_ptr()), t_values.numel(), log_table,
                                                         C.c_void_p(t_counts.data_ptr()), 1 if accumulate else 0))
        return t_counts

    def range_tuple_counts_tracegen(self, t_x, t_y, size_x, size_y, t_counts=None, accumulate=False):
        """Multiplicity column (size_x * size_y Montgomery words) of a range-tuple table for the requests (t_x[i], t_y[i])."""
        import torch

        if t_counts is None:
            t_counts = torch.empty(size_x * size_y, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_range_tuple_counts_tracegen(self.h, C.c_void_p(t_x.data_ptr()), C.c_void_p(t_y.data_ptr()), t_x.numel(),
                                                               size_x, size_y, C.c_void_p(t_counts.data_ptr()), 1 if accumulate else 0))
        return t_counts

    def bitwise_lookup_tracegen(self, t_x, t_y, t_op, num_bits=8, t_trace=None, accumulate=False):
        """The two multiplicity columns (range, xor) of a bitwise-operation lookup table: 2 x 2^(2 num_bits) words."""
        import torch

        if t_trace is None:
            t_trace = torch.empty(2 << (2 * num_bits), dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_bitwise_lookup_tracegen(self.h, C.c_void_p(t_x.data_ptr()), C.c_void_p(t_y.data_ptr()),
                                                           C.c_void_p(t_op.data_ptr()), t_x.numel(), num_bits,
                                                           C.c_void_p(t_trace.data_ptr()), 1 if accumulate else 0))
        return t_trace

    def rv32_alu_tracegen(self, t_opcode, t_b, t_c, log_height, t_bitwise_trace):
        """18-column trace of the RV32 base ALU core chip from records (int32 tensors: opcode 0..4, operands b, c); the rows'
        bitwise-lookup requests are added to the XOR column of t_bitwise_trace (2 x 2^16 Montgomery words)."""
        import torch

        out = torch.empty(18 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_rv32_alu_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_b.data_ptr()),
                                                     C.c_void_p(t_c.data_ptr()), t_opcode.numel(), log_height, C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(t_bitwise_trace.data_ptr())))
        return out

    def rv32_lt_tracegen(self, t_opcode, t_b, t_c, log_height, t_bitwise_trace):
        """18-column trace of the RV32 less-than core chip from records (int32 tensors: opcode 0 = SLT, 1 = SLTU; operands b, c); the
        rows' range requests are added to the range column (column 0) of t_bitwise_trace (2 x 2^16 Montgomery words)."""
        import torch

        out = torch.empty(18 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_rv32_lt_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_b.data_ptr()),
                                                    C.c_void_p(t_c.data_ptr()), t_opcode.numel(), log_height, C.c_void_p(out.data_ptr()),
                                                    C.c_void_p(t_bitwise_trace.data_ptr())))
        return out

    def rv32_branch_lt_tracegen(self, t_opcode, t_a, t_b, t_imm, log_height, t_bitwise_trace):
        """23-column trace of the RV32 branch-less-than core chip from records (int32 tensors: opcode 0 = BLT, 1 = BLTU, 2 = BGE, 3 = BGEU;
        operands; the offset as a canonical field element); range requests go to column 0 of t_bitwise_trace."""
        import torch

        out = torch.empty(23 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_rv32_branch_lt_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_a.data_ptr()),
                                                           C.c_void_p(t_b.data_ptr()), C.c_void_p(t_imm.data_ptr()), t_opcode.numel(), log_height,
                                                           C.c_void_p(out.data_ptr()), C.c_void_p(t_bitwise_trace.data_ptr())))
        return out

    def _records_tracegen(self, fn, width, records, log_height, t_bitwise_trace):
        import torch

        out = torch.empty(width << log_height, dtype=torch.int32, device=self.device)
        self._check(fn(self.h, *[C.c_void_p(t.data_ptr()) for t in records], records[0].numel(), log_height, C.c_void_p(out.data_ptr()),
                       C.c_void_p(t_bitwise_trace.data_ptr())))
        return out

    def rv32_jal_lui_tracegen(self, t_opcode, t_pc, t_imm, log_height, t_bitwise_trace):
        """9-column trace of the RV32 JAL / LUI core chip from records (int32 tensors: opcode 0 = JAL, 1 = LUI; pc; JAL's offset as a
        field element or LUI's 20-bit immediate); range requests go to column 0 of t_bitwise_trace."""
        return self._records_tracegen(self.lib.zkhip_rv32_jal_lui_tracegen, 9, (t_opcode, t_pc, t_imm), log_height, t_bitwise_trace)

    def rv32_loadstore_tracegen(self, t_case, t_read, t_prev, log_height, t_bitwise_trace):
        """33-column trace of the RV32 load/store core chip from records (int32 tensors: case 0..19 as air.RV32_LOADSTORE_CASES,
        read word, prev word); the sign requests of LH / LB go to column 0 of t_bitwise_trace."""
        return self._records_tracegen(self.lib.zkhip_rv32_loadstore_tracegen, 33, (t_case, t_read, t_prev), log_height, t_bitwise_trace)

    def rv32_auipc_tracegen(self, t_pc, t_imm, log_height, t_bitwise_trace):
        """14-column trace of the RV32 AUIPC core chip from records (pc, 20-bit immediate)."""
        return self._records_tracegen(self.lib.zkhip_rv32_auipc_tracegen, 14, (t_pc, t_imm), log_height, t_bitwise_trace)

    def rv32_jalr_tracegen(self, t_pc, t_rs1, t_imm, log_height, t_bitwise_trace):
        """20-column trace of the RV32 JALR core chip from records (pc, rs1, raw 12-bit immediate)."""
        return self._records_tracegen(self.lib.zkhip_rv32_jalr_tracegen, 20, (t_pc, t_rs1, t_imm), log_height, t_bitwise_trace)

    def rv32_branch_eq_tracegen(self, t_opcode, t_a, t_b, t_imm, log_height):
        """17-column trace of the RV32 branch-equal core chip from records (int32 tensors: opcode 0 = BEQ, 1 = BNE; operands; the
        branch offset as a canonical field element)."""
        import torch

        out = torch.empty(17 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_rv32_branch_eq_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_a.data_ptr()),
                                                           C.c_void_p(t_b.data_ptr()), C.c_void_p(t_imm.data_ptr()), t_opcode.numel(), log_height,
                                                           C.c_void_p(out.data_ptr())))
        return out

    def rv32_shift_tracegen(self, t_opcode, t_b, t_c, log_height, t_bitwise_trace):
        """32-column trace of the RV32 shift core chip from records (int32 tensors: opcode 0 = SLL, 1 = SRL, 2 = SRA; value b; shift
        operand c); the rows' lookup requests are added to both columns of t_bitwise_trace (2 x 2^16 Montgomery words)."""
        import torch

        out = torch.empty(32 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_rv32_shift_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_b.data_ptr()),
                                                       C.c_void_p(t_c.data_ptr()), t_opcode.numel(), log_height, C.c_void_p(out.data_ptr()),
                                                       C.c_void_p(t_bitwise_trace.data_ptr())))
        return out

    def rv32_mul_tracegen(self, t_b, t_c, log_height, t_tuple_counts, size_x=256, size_y=8192):
        """13-column trace of the RV32 multiplication core chip from records (int32 tensors: operands); the (limb, carry) requests of
        every row are added to t_tuple_counts (size_x * size_y Montgomery words: the range-tuple table's trace)."""
        import torch

        out = torch.empty(13 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_rv32_mul_tracegen(self.h, C.c_void_p(t_b.data_ptr()), C.c_void_p(t_c.data_ptr()), t_b.numel(), log_height,
                                                     C.c_void_p(out.data_ptr()), C.c_void_p(t_tuple_counts.data_ptr()), size_x, size_y))
        return out

    def var_range_counts_tracegen(self, t_values, bits, max_bits, t_counts=None, accumulate=False):
        """Multiplicity column (2^(max_bits + 1) Montgomery words) of the variable range checker for the Montgomery column t_values;
        bits: a Montgomery column of bit counts (tensor) or one int for every request."""
        import torch

        if t_counts is None:
            t_counts = torch.zeros(1 << (max_bits + 1), dtype=torch.int32, device=self.device)
        col = isinstance(bits, torch.Tensor)
        self._check(self.lib.zkhip_var_range_counts_tracegen(self.h, C.c_void_p(t_values.data_ptr()), C.c_void_p(bits.data_ptr()) if col else None,
                                                             0 if col else int(bits), t_values.numel(), max_bits, C.c_void_p(t_counts.data_ptr()),
                                                             1 if accumulate else 0))
        return t_counts

    def domain_point_tracegen(self, t_k, t_mult, log_height):
        """54-column trace of the domain-point chip from (pair index, multiplicity) records (int32 tensors)."""
        import torch

        out = torch.empty(54 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_domain_point_tracegen(self.h, C.c_void_p(t_k.data_ptr()), C.c_void_p(t_mult.data_ptr()), t_k.numel(), log_height,
                                                         C.c_void_p(out.data_ptr())))
        return out

    def duplex_tracegen(self, t_n_observed, t_observed, t_n_sampled, log_height):
        """(50-column trace of the transcript chip, [2^log_height][16] permutation inputs for the Poseidon2 chip) from duplexing
        records (int32 tensors: observed count per row, observed values [n][8] canonical, sampled count per row)."""
        import torch

        tr = torch.empty(50 << log_height, dtype=torch.int32, device=self.device)
        hin = torch.empty(16 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_duplex_tracegen(self.h, C.c_void_p(t_n_observed.data_ptr()), C.c_void_p(t_observed.data_ptr()),
                                                   C.c_void_p(t_n_sampled.data_ptr()), t_n_observed.numel(), log_height, C.c_void_p(tr.data_ptr()),
                                                   C.c_void_p(hin.data_ptr())))
        return tr, hin

    def fri_fold_chip_tracegen(self, t_e0, t_e1, t_beta, t_k, t_log_n_out, log_height):
        """19-column trace of the FRI fold chip from records (int32 tensors: e0, e1, beta [n][4] canonical; pair indices; log2 of the
        folded layer's size)."""
        import torch

        out = torch.empty(19 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_fri_fold_chip_tracegen(self.h, *[C.c_void_p(t.data_ptr()) for t in (t_e0, t_e1, t_beta, t_k, t_log_n_out)], t_k.numel(),
                                                          log_height, C.c_void_p(out.data_ptr())))
        return out

    def castf_tracegen(self, t_x, log_height, t_var_range_counts, max_bits):
        """6-column trace of the native CASTF chip from records (int32 tensor of values < 2^30); the limb checks are added to
        t_var_range_counts (the variable range checker's trace)."""
        import torch

        out = torch.empty(6 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_castf_tracegen(self.h, C.c_void_p(t_x.data_ptr()), t_x.numel(), log_height, C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(t_var_range_counts.data_ptr()), max_bits))
        return out

    def field_arith_tracegen(self, t_opcode, t_b, t_c, log_height):
        """8-column trace of the native field-arithmetic chip from records (int32 tensors: opcode 0 = ADD .. 3 = DIV, canonical operands)."""
        import torch

        out = torch.empty(8 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_field_arith_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_b.data_ptr()), C.c_void_p(t_c.data_ptr()),
                                                        t_opcode.numel(), log_height, C.c_void_p(out.data_ptr())))
        return out

    def field_ext_tracegen(self, t_opcode, t_x, t_y, log_height):
        """20-column trace of the native field-extension chip from records (opcode, x [n][4], y [n][4] canonical)."""
        import torch

        out = torch.empty(20 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_field_ext_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_x.data_ptr()), C.c_void_p(t_y.data_ptr()),
                                                      t_opcode.numel(), log_height, C.c_void_p(out.data_ptr())))
        return out

    def mmcs_path_tracegen(self, t_leaf, t_index, t_path_start, t_step_kind, t_step_digest, log_height):
        """(39-column trace of the MMCS path chip, [2^log_height][16] hash inputs for the Poseidon2 chip) from path records (int32
        tensors of canonical words: leaf digests [n][8], indices [n], step offsets [n + 1], step kinds, step digests [steps][8])."""
        import torch

        tr = torch.empty(39 << log_height, dtype=torch.int32, device=self.device)
        hin = torch.empty(16 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_mmcs_path_tracegen(self.h, *[C.c_void_p(t.data_ptr()) for t in (t_leaf, t_index, t_path_start, t_step_kind, t_step_digest)],
                                                      t_index.numel(), log_height, C.c_void_p(tr.data_ptr()), C.c_void_p(hin.data_ptr())))
        return tr, hin

    def rv32_divrem_tracegen(self, t_opcode, t_b, t_c, log_height, t_tuple_counts, t_bitwise_trace, size_x=256, size_y=2048):
        """41-column trace of the RV32 division core chip from records (int32 tensors: opcode 0 = DIV, 1 = DIVU, 2 = REM, 3 = REMU;
        dividend; divisor); the (limb, carry) requests go to t_tuple_counts, the range requests to column 0 of t_bitwise_trace."""
        import torch

        out = torch.empty(41 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_rv32_divrem_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_b.data_ptr()), C.c_void_p(t_c.data_ptr()),
                                                        t_opcode.numel(), log_height, C.c_void_p(out.data_ptr()), C.c_void_p(t_tuple_counts.data_ptr()),
                                                        size_x, size_y, C.c_void_p(t_bitwise_trace.data_ptr())))
        return out

    def rv32_mulh_tracegen(self, t_opcode, t_b, t_c, log_height, t_tuple_counts, t_bitwise_trace, size_x=256, size_y=2048):
        """21-column trace of the RV32 high-multiplication core chip from records (int32 tensors: opcode 0 = MULH, 1 = MULHSU,
        2 = MULHU; operands); the (limb, carry) requests go to t_tuple_counts, the sign requests to column 0 of t_bitwise_trace."""
        import torch

        out = torch.empty(21 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_rv32_mulh_tracegen(self.h, C.c_void_p(t_opcode.data_ptr()), C.c_void_p(t_b.data_ptr()), C.c_void_p(t_c.data_ptr()),
                                                      t_opcode.numel(), log_height, C.c_void_p(out.data_ptr()), C.c_void_p(t_tuple_counts.data_ptr()),
                                                      size_x, size_y, C.c_void_p(t_bitwise_trace.data_ptr())))
        return out

    def memory_access_tracegen(self, t_as, t_ptr, t_prev_data, t_prev_ts, t_data, t_ts, t_is_read, log_height):
        """10-column trace of the memory access chip from the memory log (int32 tensors of plain integers, one entry per cell access)."""
        import torch

        out = torch.empty(10 << log_height, dtype=torch.int32, device=self.device)
        args = [C.c_void_p(t.data_ptr()) for t in (t_as, t_ptr, t_prev_data, t_prev_ts, t_data, t_ts, t_is_read)]
        self._check(self.lib.zkhip_memory_access_tracegen(self.h, *args, t_as.numel(), log_height, C.c_void_p(out.data_ptr())))
        return out

    def program_freq_tracegen(self, t_pc_index, log_height):
        """Frequency column of the program chip (2^log_height Montgomery words) from the executed instruction indices (int32 tensor)."""
        import torch

        out = torch.empty(1 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_program_freq_tracegen(self.h, C.c_void_p(t_pc_index.data_ptr()), t_pc_index.numel(), log_height,
                                                         C.c_void_p(out.data_ptr())))
        return out

    def exec_frame_tracegen(self, t_pc_index, t_program, n_program, log_height):
        """10-column trace of the execution-frame chip: row i = the 9 fields of instruction t_pc_index[i] (gathered from t_program:
        9 columns of n_program Montgomery words) and is_valid."""
        import torch

        out = torch.empty(10 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_exec_frame_tracegen(self.h, C.c_void_p(t_pc_index.data_ptr()), t_pc_index.numel(),
                                                       C.c_void_p(t_program.data_ptr()), n_program, log_height, C.c_void_p(out.data_ptr())))
        return out

    def memory_boundary_tracegen(self, t_as, t_ptr, t_init, t_final, t_ts, as_bits, pointer_bits, log_height):
        """8-column trace of the volatile memory boundary chip, rows sorted by address on the device.
        t_as / t_ptr / t_ts: plain integers (int32 tensors); t_init / t_final: Montgomery words."""
        import torch

        out = torch.empty(8 << log_height, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_memory_boundary_tracegen(self.h, C.c_void_p(t_as.data_ptr()), C.c_void_p(t_ptr.data_ptr()),
                                                            C.c_void_p(t_init.data_ptr()), C.c_void_p(t_final.data_ptr()),
                                                            C.c_void_p(t_ts.data_ptr()), t_as.numel(), as_bits, pointer_bits, log_height,
                                                            C.c_void_p(out.data_ptr())))
        return out

    # ---- the one-statement VM circuit (include/zkhip_vm_circuit.hpp): generators of the adapter-side chips ----
    def _new(self, words):
        import torch

        return torch.empty(words, dtype=torch.int32, device=self.device)

    def vm_frame_tracegen(self, t_idx, t_x, t_y, t_z, t_rdprev, t_pcinc, t_pts1, t_pts2, t_pts3, t_program, n_program, log_height):
        out = self._new(43 << log_height)
        self._check(self.lib.zkhip_vm_frame_tracegen(self.h, t_idx.data_ptr(), t_x.data_ptr(), t_y.data_ptr(), t_z.data_ptr(), t_rdprev.data_ptr(),
                                                     t_pcinc.data_ptr(), t_pts1.data_ptr(), t_pts2.data_ptr(), t_pts3.data_ptr(), t_idx.numel(),
                                                     t_program.data_ptr(), n_program, log_height, out.data_ptr()))
        return out

    def vm_loadstore_tracegen(self, t_case, t_read, t_prev, t_ts, t_base, t_imm, t_pts, log_height, t_bitwise_trace):
        out = self._new(48 << log_height)
        self._check(self.lib.zkhip_vm_loadstore_tracegen(self.h, t_case.data_ptr(), t_read.data_ptr(), t_prev.data_ptr(), t_ts.data_ptr(), t_base.data_ptr(),
                                                         t_imm.data_ptr(), t_pts.data_ptr(), t_case.numel(), log_height, out.data_ptr(),
                                                         t_bitwise_trace.data_ptr()))
        return out

    def vm_keccak_tracegen(self, t_states, t_ts, n_perms, log_height):
        out = self._new(2634 << log_height)
        self._check(self.lib.zkhip_vm_keccak_tracegen(self.h, t_states.data_ptr() if n_perms else None, t_ts.data_ptr() if n_perms else None, n_perms,
                                                      log_height, out.data_ptr()))
        return out

    def vm_modmul_tracegen(self, modulus, t_records, t_ts, n, log_height, t_bitwise_trace, t_tuple_counts, size_x, size_y):
        nw = limb_words(modulus)
        out = self._new((40 * nw + 6) << log_height)   # 326 columns (485 + 1 for a 48-limb modulus)
        m = np.ascontiguousarray(_int_words(modulus, nw), dtype=np.uint32)
        self._check(self.lib.zkhip_vm_modmul_tracegen_x(self.h, nw, _u32p(m), t_records.data_ptr() if n else None, t_ts.data_ptr() if n else None, n, log_height,
                                                        out.data_ptr(), t_bitwise_trace.data_ptr(), t_tuple_counts.data_ptr(), size_x, size_y))
        return out

    def vm_sha256_tracegen(self, t_blocks, t_ts, n_blocks, log_height):
        out = self._new(434 << log_height)
        self._check(self.lib.zkhip_vm_sha256_tracegen(self.h, t_blocks.data_ptr() if n_blocks else None, t_ts.data_ptr() if n_blocks else None, n_blocks,
                                                      log_height, out.data_ptr()))
        return out

    def vm_poseidon2_tracegen(self, t_inputs_monty, n, log_height):
        out = self._new(299 << log_height)
        self._check(self.lib.zkhip_vm_poseidon2_tracegen(self.h, t_inputs_monty.data_ptr() if n else None, n, log_height, out.data_ptr()))
        return out

    def rows_tracegen(self, t_rows, n, width, log_height, pad_row=None):
        out = self._new(width << log_height)
        pad = None if pad_row is None else np.ascontiguousarray(pad_row, dtype=np.uint32)
        self._check(self.lib.zkhip_rows_tracegen(self.h, t_rows.data_ptr() if n else None, n, width, log_height, out.data_ptr(),
                                                 _u32p(pad) if pad is not None else None))
        return out

    def range_counts_scaled_tracegen(self, t_values, scale, log_table, t_counts, accumulate=True):
        self._check(self.lib.zkhip_range_counts_scaled_tracegen(self.h, t_values.data_ptr(), t_values.numel(), scale, log_table, t_counts.data_ptr(),
                                                                1 if accumulate else 0))
        return t_counts

    def keccak_f_tracegen(self, t_states, n_perms, log_height):
        out = self._new(2633 << log_height)
        self._check(self.lib.zkhip_keccak_f_tracegen(self.h, t_states.data_ptr() if n_perms else None, n_perms, log_height, out.data_ptr()))
        return out

    def modular_tracegen(self, modulus, t_records, n, log_height, t_bitwise_trace, t_tuple_counts, size_x, size_y):
        """records: n x (2 nw + 1) words (op | a | b; op 0 mul, 1 add, 2 sub, 3 div, 4 is_eq), nw = limb_words(modulus)"""
        nw = limb_words(modulus)
        out = self._new((40 * nw + 5) << log_height)   # 325 columns (485 for a 48-limb modulus)
        m = np.ascontiguousarray(_int_words(modulus, nw), dtype=np.uint32)
        self._check(self.lib.zkhip_modular_tracegen_x(self.h, nw, _u32p(m), t_records.data_ptr() if n else None, n, log_height, out.data_ptr(), t_bitwise_trace.data_ptr(),
                                                      t_tuple_counts.data_ptr(), size_x, size_y))
        return out

    def int256_alu_tracegen(self, t_records, n, log_height, t_bitwise_trace):
        out = self._new(101 << log_height)
        self._check(self.lib.zkhip_int256_alu_tracegen(self.h, t_records.data_ptr() if n else None, n, log_height, out.data_ptr(), t_bitwise_trace.data_ptr()))
        return out

    def int256_mul_tracegen(self, t_records, n, log_height, t_bitwise_trace, t_tuple_counts, size_x, size_y):
        out = self._new(161 << log_height)
        self._check(self.lib.zkhip_int256_mul_tracegen(self.h, t_records.data_ptr() if n else None, n, log_height, out.data_ptr(), t_bitwise_trace.data_ptr(),
                                                       t_tuple_counts.data_ptr(), size_x, size_y))
        return out

    def int256_cmp_tracegen(self, t_records, n, log_height, t_bitwise_trace, t_ts=None):
        """records: n x 17 words (op | b | c; op 6 sltu, 7 slt, 8 eq); t_ts: the VM chip (108 columns: + the calls' timestamps and the branch columns
        is_br | neg | taken | opcode -- there the records may also carry the branch opcodes 12 .. 17)"""
        out = self._new((103 if t_ts is None else 108) << log_height)
        if t_ts is None:
            self._check(self.lib.zkhip_int256_cmp_tracegen(self.h, t_records.data_ptr() if n else None, n, log_height, out.data_ptr(), t_bitwise_trace.data_ptr()))
        else:
            self._check(self.lib.zkhip_vm_cmp256_tracegen(self.h, t_records.data_ptr() if n else None, t_ts.data_ptr() if n else None, n, log_height, out.data_ptr(),
                                                          t_bitwise_trace.data_ptr()))
        return out

    def int256_shift_tracegen(self, t_records, n, log_height, t_bitwise_trace, t_ts=None):
        """records: n x 17 words (op | b | c; op 9 sll, 10 srl, 11 sra); t_ts: the VM chip (190 columns: + the calls' timestamps)"""
        out = self._new((189 if t_ts is None else 190) << log_height)
        if t_ts is None:
            self._check(self.lib.zkhip_int256_shift_tracegen(self.h, t_records.data_ptr() if n else None, n, log_height, out.data_ptr(), t_bitwise_trace.data_ptr()))
        else:
            self._check(self.lib.zkhip_vm_shift256_tracegen(self.h, t_records.data_ptr() if n else None, t_ts.data_ptr() if n else None, n, log_height, out.data_ptr(),
                                                            t_bitwise_trace.data_ptr()))
        return out

    def vm_native_tracegen(self, kind, t_records, n, log_height):
        """The rows of a native chip of the one-statement circuit from the executor's call records (include/zkhip.h): kind "arith" (27
        columns, 9 words per call), "ext" (90 columns, 27 words), "castf" (16 columns, 6 words)"""
        width, fn = {"arith": (27, self.lib.zkhip_vm_native_arith_tracegen), "ext": (90, self.lib.zkhip_vm_native_ext_tracegen),
                     "castf": (16, self.lib.zkhip_vm_castf_tracegen)}[kind]
        out = self._new(width << log_height)
        self._check(fn(self.h, t_records.data_ptr() if n else None, n, log_height, out.data_ptr()))
        return out

    def vm_mul256_tracegen(self, t_records, t_ts, n, log_height, t_bitwise_trace, t_tuple_counts, size_x, size_y):
        out = self._new(162 << log_height)
        self._check(self.lib.zkhip_vm_mul256_tracegen(self.h, t_records.data_ptr() if n else None, t_ts.data_ptr() if n else None, n, log_height, out.data_ptr(),
                                                      t_bitwise_trace.data_ptr(), t_tuple_counts.data_ptr(), size_x, size_y))
        return out

    def vm_int256_tracegen(self, t_records, t_ts, n, log_height, t_bitwise_trace):
        out = self._new(102 << log_height)
        self._check(self.lib.zkhip_vm_int256_tracegen(self.h, t_records.data_ptr() if n else None, t_ts.data_ptr() if n else None, n, log_height, out.data_ptr(),
                                                      t_bitwise_trace.data_ptr()))
        return out

    def fp2_tracegen(self, modulus, t_records, n, log_height, t_bitwise_trace, t_tuple_counts, size_x, size_y, t_ts=None):
        """records: n x (4 nw + 1) words (op | a0 a1 | b0 b1); t_ts: the VM chip (+ the calls' timestamps)"""
        nw = limb_words(modulus)
        out = self._new((80 * nw + 8 + (0 if t_ts is None else 1)) << log_height)   # 648 columns (968 for a 48-limb modulus)
        m = np.ascontiguousarray(_int_words(modulus, nw), dtype=np.uint32)
        if t_ts is None:
            self._check(self.lib.zkhip_fp2_tracegen_x(self.h, nw, _u32p(m), t_records.data_ptr() if n else None, n, log_height, out.data_ptr(), t_bitwise_trace.data_ptr(),
                                                      t_tuple_counts.data_ptr(), size_x, size_y))
        else:
            self._check(self.lib.zkhip_vm_fp2_tracegen_x(self.h, nw, _u32p(m), t_records.data_ptr() if n else None, t_ts.data_ptr() if n else None, n, log_height,
                                                         out.data_ptr(), t_bitwise_trace.data_ptr(), t_tuple_counts.data_ptr(), size_x, size_y))
        return out

    def ec_tracegen(self, modulus, coeff_a, t_records, n, log_height, t_bitwise_trace, t_tuple_counts, size_x, size_y, t_ts=None):
        """records: n x (5 nw + 1) words (op | x1 y1 x2 y2 | slope); t_ts: the VM chip (+ the calls' timestamps)"""
        nw = limb_words(modulus)
        out = self._new((96 * nw + 4 + (0 if t_ts is None else 1)) << log_height)   # 772 columns (1156 for a 48-limb modulus)
        m, ca = (np.ascontiguousarray(_int_words(v, nw), dtype=np.uint32) for v in (modulus, coeff_a))
        if t_ts is None:
            self._check(self.lib.zkhip_ec_tracegen_x(self.h, nw, _u32p(m), _u32p(ca), t_records.data_ptr() if n else None, n, log_height, out.data_ptr(),
                                                     t_bitwise_trace.data_ptr(), t_tuple_counts.data_ptr(), size_x, size_y))
        else:
            self._check(self.lib.zkhip_vm_ec_tracegen_x(self.h, nw, _u32p(m), _u32p(ca), t_records.data_ptr() if n else None, t_ts.data_ptr() if n else None, n, log_height,
                                                        out.data_ptr(), t_bitwise_trace.data_ptr(), t_tuple_counts.data_ptr(), size_x, size_y))
        return out

    def modmul_tracegen(self, modulus, t_records, n, log_height, t_bitwise_trace, t_tuple_counts, size_x, size_y):
        out = self._new(325 << log_height)
        m = np.ascontiguousarray(_int_words(modulus), dtype=np.uint32)
        self._check(self.lib.zkhip_modmul_tracegen(self.h, _u32p(m), t_records.data_ptr() if n else None, n, log_height, out.data_ptr(), t_bitwise_trace.data_ptr(),
                                                   t_tuple_counts.data_ptr(), size_x, size_y))
        return out

    def sha256_tracegen(self, t_blocks, n_blocks, log_height):
        out = self._new(433 << log_height)
        self._check(self.lib.zkhip_sha256_tracegen(self.h, t_blocks.data_ptr() if n_blocks else None, n_blocks, log_height, out.data_ptr()))
        return out

    def merkle_commit(self, mats, want_root=True):
        """mats: list of (tensor, log_height, width[, stride])."""
        return MerkleTree(self, mats, want_root)

    def constraint_eval(self, program, log_height, log_blowup, width, t_lde, pvs, alpha):
        """Quotient values (4 x 2^(log_height+log_blowup) Montgomery words, device) of one AIR on its committed LDE."""
        import torch

        prog = np.ascontiguousarray(program, dtype=np.uint32)
        pv = np.ascontiguousarray(pvs, dtype=np.uint32)
        al = np.ascontiguousarray(alpha, dtype=np.uint32)
        out = torch.empty(4 << (log_height + log_blowup), dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_constraint_eval(self.h, _u32p(prog), prog.size, log_height, log_blowup, width, C.c_void_p(t_lde.data_ptr()),
                                                   _u32p(pv) if pv.size else None, pv.size, _u32p(al), C.c_void_p(out.data_ptr())))
        return out

    def fri_fold(self, t_in, log_n_out, beta):
        import torch

        out = torch.empty(4 << log_n_out, dtype=torch.int32, device=self.device)
        b = np.asarray(beta, dtype=np.uint32)
        self._check(self.lib.zkhip_fri_fold(self.h, C.c_void_p(t_in.data_ptr()), C.c_void_p(out.data_ptr()),
                                            log_n_out, _u32p(b)))
        return out

    def transcript(self):
        return Transcript(self)

    # ---- LogUp / sum-check building blocks (K6, K7) ----------------------------------------------
    def ext_batch_inverse(self, t_in, n):
        import torch

        out = torch.empty(4 * n, dtype=torch.int32, device=self.device)
        self._check(self.lib.zkhip_ext_batch_inverse(self.h, C.c_void_p(t_in.data_ptr()), C.c_void_p(out.data_ptr()), n))
        return out

    def logup_running_sum(self, t_den, t_num, n):
        import torch

        out = torch.empty(4 * n, dtype=torch.int32, device=self.device)
        total = np.zeros(4, dtype=np.uint32)
        self._check(self.lib.zkhip_logup_running_sum(self.h, C.c_void_p(t_den.data_ptr()), C.c_void_p(t_num.data_ptr()),
                                                     n, C.c_void_p(out.data_ptr()), _u32p(total)))
        return out, total

    def mle_fold(self, t_in, n, r):
        import torch

        out = torch.empty(4 * n, dtype=torch.int32, device=self.device)
        rr = np.asarray(r, dtype=np.uint32)
        self._check(self.lib.zkhip_mle_fold(self.h, C.c_void_p(t_in.data_ptr()), C.c_void_p(out.data_ptr()), n, _u32p(rr)))
        return out

    def sumcheck_round(self, tables, n_half):
        arr = (C.c_void_p * len(tables))(*[t.data_ptr() for t in tables])
        out = np.zeros(4 * (len(tables) + 1), dtype=np.uint32)
        self._check(self.lib.zkhip_sumcheck_round(self.h, arr, len(tables), n_half, _u32p(out)))
        return out

    # ---- profiling -----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.zkhip_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._check(self.lib.zkhip_profile_reset(self.h))

    def profile_read(self):
        buf = (_KernelStat * 64)()
        n = self.lib.zkhip_profile_read(self.h, buf, 64)
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(min(n, 64))}


class MerkleTree:
    def __init__(self, ctx, mats, want_root=True):
        self.ctx = ctx
        self._keep = [m[0] for m in mats]
        arr = (_Matrix * len(mats))()
        self.widths, self.log_heights = [], []
        for i, m in enumerate(mats):
            t, lh, w = m[0], m[1], m[2]
            stride = m[3] if len(m) > 3 else (1 << lh)
            arr[i] = _Matrix(t.data_ptr(), stride, lh, w)
            self.widths.append(w)
            self.log_heights.append(lh)
        h = C.c_void_p()
        root = np.zeros(8, dtype=np.uint32)
        ctx._check(ctx.lib.zkhip_merkle_commit(ctx.h, arr, len(mats), C.byref(h), _u32p(root) if want_root else None))
        self.h = h
        self.root = root if want_root else None
        self.log_height = ctx.lib.zkhip_tree_log_height(h)

    def layer(self, l):
        out = np.zeros(8 << (self.log_height - l), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.zkhip_tree_layer(self.ctx.h, self.h, l, _u32p(out)))
        return out.reshape(-1, 8)

    def open(self, indices):
        idx = np.asarray(indices, dtype=np.uint64)
        words = self.ctx.lib.zkhip_merkle_opening_words(self.h)
        out = np.zeros(words * len(idx), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.zkhip_merkle_open(self.ctx.h, self.h, idx.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                       len(idx), _u32p(out), out.size))
        return out.reshape(len(idx), words)

    def close(self):
        if self.h:
            self.ctx.lib.zkhip_tree_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Transcript:
    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx.lib.zkhip_transcript_create(ctx.h, C.byref(h)))
        self.h = h

    def observe(self, vals):
        a = np.ascontiguousarray(vals, dtype=np.uint32)
        self.ctx._check(self.ctx.lib.zkhip_transcript_observe(self.ctx.h, self.h, _u32p(a), a.size))

    def sample(self, n=1):
        out = np.zeros(n, dtype=np.uint32)
        self.ctx._check(self.ctx.lib.zkhip_transcript_sample(self.ctx.h, self.h, _u32p(out), n))
        return out

    def grind(self, bits):
        w = C.c_uint32()
        self.ctx._check(self.ctx.lib.zkhip_transcript_grind(self.ctx.h, self.h, bits, C.byref(w)))
        return w.value

    def close(self):
        if self.h:
            self.ctx.lib.zkhip_transcript_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


DEFAULT_PARAMS = (1, 0, 100, 16, 16)  # crates/circuits/chunk-circuit/openvm.toml:1-6


def _air_structs(airs):
    arr = (_Air * len(airs))()
    keep = []
    for i, a in enumerate(airs):
        prog = np.ascontiguousarray(a["program"], dtype=np.uint32)
        prep = None if a.get("prep") is None else np.ascontiguousarray(a["prep"], dtype=np.uint32)
        pc = None if a.get("prep_commit") is None else np.ascontiguousarray(a["prep_commit"], dtype=np.uint32)
        keep += [prog, prep, pc]
        arr[i] = _Air(_u32p(prog), prog.size, a["log_height"], a["width"], a["n_pvs"],
                      _u32p(prep) if prep is not None else None, _u32p(pc) if pc is not None else None)
    return arr, keep


def _pvs_array(pvs):
    keep = [np.ascontiguousarray(p, dtype=np.uint32) for p in pvs]
    arr = (C.POINTER(C.c_uint32) * len(pvs))(*[_u32p(p) if p.size else None for p in keep])
    return arr, keep


class ProvingKey:
    """zkhip_pk: compiled AIRs + resident workspace + static proof layout for one set of shapes."""

    def __init__(self, ctx, params, airs):
        self.ctx, self.params, self.airs = ctx, tuple(params), airs
        prm = _Params(*self.params)
        arr, keep = _air_structs(airs)
        h = C.c_void_p()
        ctx._check(ctx.lib.zkhip_keygen(ctx.h, C.byref(prm), arr, len(airs), C.byref(h)))
        self.h = h
        self.proof_size = ctx.lib.zkhip_proof_size(h)
        self.workspace_bytes = ctx.lib.zkhip_pk_workspace_bytes(h)

    def prep_commitment(self, air_index):
        """8 canonical words: the verifying-key entry of AIR `air_index`'s preprocessed trace."""
        out = np.zeros(8, dtype=np.uint32)
        self.ctx._check(self.ctx.lib.zkhip_pk_prep_commitment(self.ctx.h, self.h, air_index, _u32p(out)))
        return out

    def verifying_airs(self):
        """The AIR descriptions a verifier needs: programs + shapes + preprocessed commitments (no tables)."""
        out = []
        for i, a in enumerate(self.airs):
            v = {k: a[k] for k in ("program", "log_height", "width", "n_pvs")}
            if a.get("prep") is not None:
                v["prep_commit"] = self.prep_commitment(i)
            out.append(v)
        return out

    def prove_async(self, traces, pvs):
        tp = (C.c_void_p * len(traces))(*[t.data_ptr() for t in traces])
        pa, keep = _pvs_array(pvs)
        self.ctx._check(self.ctx.lib.zkhip_prove_async(self.ctx.h, self.h, tp, pa))

    def fetch(self):
        buf = np.zeros(self.proof_size, dtype=np.uint8)
        n = C.c_size_t()
        self.ctx._check(self.ctx.lib.zkhip_proof_fetch(self.ctx.h, self.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                        buf.size, C.byref(n)))
        return buf[:n.value].tobytes()

    def prove(self, traces, pvs):
        tp = (C.c_void_p * len(traces))(*[t.data_ptr() for t in traces])
        pa, keep = _pvs_array(pvs)
        buf = np.zeros(self.proof_size, dtype=np.uint8)
        n = C.c_size_t()
        self.ctx._check(self.ctx.lib.zkhip_prove(self.ctx.h, self.h, tp, pa, buf.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                  buf.size, C.byref(n)))
        return buf[:n.value].tobytes()

    def close(self):
        if self.h:
            self.ctx.lib.zkhip_pk_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def proof_layout(params, airs):
    """Word offsets of the fields of a proof for these parameters / AIR shapes (dict; needs no GPU)."""
    lib = load_library()
    prm = _Params(*params)
    arr, keep = _air_structs(airs)
    out = _ProofLayout()
    rc = lib.zkhip_proof_layout_of(C.byref(prm), arr, len(airs), C.byref(out))
    if rc != 0:
        raise ZkhipError("zkhip_proof_layout_of returned %d" % rc)
    return {name: int(getattr(out, name)) for name, _ in _ProofLayout._fields_}


def verify(params, airs, pvs, proof_bytes):
    """Host verifier (needs no GPU).  Returns the zkhip status code (0 = accepted)."""
    lib = load_library()
    prm = _Params(*params)
    arr, keep = _air_structs(airs)
    pa, keep2 = _pvs_array(pvs)
    buf = np.frombuffer(proof_bytes, dtype=np.uint8)
    return lib.zkhip_verify(C.byref(prm), arr, len(airs), pa, buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size)


def verify_where(params, airs, pvs, proof_bytes):
    """The host verifier with its diagnosis: (status, line of csrc/verifier.hip whose check refused the proof -- 0 when accepted)."""
    lib = load_library()
    prm = _Params(*params)
    arr, keep = _air_structs(airs)
    pa, keep2 = _pvs_array(pvs)
    buf = np.frombuffer(proof_bytes, dtype=np.uint8)
    where = C.c_int(0)
    rc = lib.zkhip_verify_where(C.byref(prm), arr, len(airs), pa, buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size, C.byref(where))
    return rc, where.value


def keccak_f_air():
    """The Keccak-f[1600] chip's AIR (program, width) from the library; its one definition is include/zkhip_keccak.hpp."""
    lib = load_library()
    a = _Air()
    assert lib.zkhip_keccak_f_air(C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def keccak_f1600_host(lanes):
    st = np.ascontiguousarray(lanes, dtype=np.uint64).copy()
    assert st.size == 25 and load_library().zkhip_keccak_f1600_host(st.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    return st


def _int_words(v, n=8):
    """a non-negative integer below 2^(32 n) as n little-endian 32-bit words"""
    return [(int(v) >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def limb_words(modulus):
    """words per operand of the limb chips for `modulus`: 8 (32 byte limbs) below 2^256, 12 (48 limbs) below 2^384 -- the BLS12-381 base
    field of the reference's batch circuit (crates/circuits/batch-circuit/openvm.toml:18-36)"""
    assert 0 < int(modulus) < 1 << 384
    return 8 if int(modulus) < 1 << 256 else 12


def int256_alu_air(bitwise_bus):
    """The 256-bit ALU chip's AIR: (program, width); its one definition is include/zkhip_int256.hpp."""
    lib = load_library()
    a = _Air()
    assert lib.zkhip_int256_alu_air(bitwise_bus, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def int256_mul_air(bitwise_bus, tuple_bus):
    """The 256-bit multiplication chip's AIR: (program, width); its one definition is include/zkhip_int256.hpp."""
    lib = load_library()
    a = _Air()
    assert lib.zkhip_int256_mul_air(bitwise_bus, tuple_bus, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def int256_cmp_air(bitwise_bus):
    """The 256-bit comparison chip's AIR: (program, width); its one definition is include/zkhip_int256.hpp."""
    lib = load_library()
    a = _Air()
    assert lib.zkhip_int256_cmp_air(bitwise_bus, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def int256_shift_air(bitwise_bus):
    """The 256-bit shift chip's AIR: (program, width); its one definition is include/zkhip_int256.hpp."""
    lib = load_library()
    a = _Air()
    assert lib.zkhip_int256_shift_air(bitwise_bus, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def int256_alu_host(op, b, c):
    """a = b op c modulo 2^256 through the library's host function (integers in, integer out)"""
    wb, wc = (np.array(_int_words(v), dtype=np.uint32) for v in (b, c))
    a = np.zeros(8, np.uint32)
    assert load_library().zkhip_int256_alu_host(op, _u32p(wb), _u32p(wc), _u32p(a)) == 0
    return sum(int(x) << (32 * i) for i, x in enumerate(a))


def modmul_air(modulus, bitwise_bus, tuple_bus):
    """The modular-multiplication chip's AIR for `modulus` (an integer): (program, width); its one definition is include/zkhip_modular.hpp."""
    lib = load_library()
    a = _Air()
    nl = 4 * limb_words(modulus)
    m = (C.c_uint8 * nl)(*int(modulus).to_bytes(nl, "little"))
    assert lib.zkhip_modmul_air_x(m, nl, bitwise_bus, tuple_bus, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def vm_modmul_air(modulus, index, adapter):
    """(program, width) of the VM's multiplication chip (adapter False) or adapter chip (True) of modulus `index`"""
    lib = load_library()
    a = _Air()
    nl = 4 * limb_words(modulus)
    m = (C.c_uint8 * nl)(*int(modulus).to_bytes(nl, "little"))
    assert lib.zkhip_vm_modmul_air_x(m, nl, index, 1 if adapter else 0, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def fp2_air(modulus, bitwise_bus, tuple_bus):
    """The Fp2 chip's AIR for Fp[u] / (u^2 + 1) over `modulus`: (program, width); its one definition is include/zkhip_fp2.hpp."""
    lib = load_library()
    a = _Air()
    nl = 4 * limb_words(modulus)
    m = (C.c_uint8 * nl)(*int(modulus).to_bytes(nl, "little"))
    assert lib.zkhip_fp2_air_x(m, nl, bitwise_bus, tuple_bus, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def vm_fp2_air(modulus, index, adapter):
    """(program, width) of the VM's Fp2 chip (adapter False) or adapter chip (True) of field `index`"""
    lib = load_library()
    a = _Air()
    nl = 4 * limb_words(modulus)
    m = (C.c_uint8 * nl)(*int(modulus).to_bytes(nl, "little"))
    assert lib.zkhip_vm_fp2_air_x(m, nl, index, 1 if adapter else 0, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def fp2_host(op, modulus, a, b):
    """a op b in Fp[u] / (u^2 + 1) through the library's host function ((c0, c1) integer pairs); None if it refuses the call"""
    nw = limb_words(modulus)
    wm = np.array(_int_words(modulus, nw), dtype=np.uint32)
    wa = np.array(_int_words(a[0], nw) + _int_words(a[1], nw), dtype=np.uint32)
    wb = np.array(_int_words(b[0], nw) + _int_words(b[1], nw), dtype=np.uint32)
    r = np.zeros(2 * nw, np.uint32)
    if load_library().zkhip_fp2_host_x(op, nw, _u32p(wm), _u32p(wa), _u32p(wb), _u32p(r)) != 0:
        return None
    return tuple(sum(int(x) << (32 * i) for i, x in enumerate(r[nw * k:nw * k + nw])) for k in range(2))


def ec_air(modulus, coeff_a, bitwise_bus, tuple_bus):
    """The elliptic-curve chip's AIR for the curve y^2 = x^3 + a x + b over `modulus`: (program, width); its one definition is include/zkhip_ecc.hpp."""
    lib = load_library()
    a = _Air()
    nl = 4 * limb_words(modulus)
    m = (C.c_uint8 * nl)(*int(modulus).to_bytes(nl, "little"))
    ca = (C.c_uint8 * nl)(*int(coeff_a).to_bytes(nl, "little"))
    assert lib.zkhip_ec_air_x(m, ca, nl, bitwise_bus, tuple_bus, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def vm_ec_air(modulus, coeff_a, index, adapter):
    """(program, width) of the VM's point chip (adapter False) or adapter chip (True) of curve `index`"""
    lib = load_library()
    a = _Air()
    nl = 4 * limb_words(modulus)
    m = (C.c_uint8 * nl)(*int(modulus).to_bytes(nl, "little"))
    ca = (C.c_uint8 * nl)(*int(coeff_a).to_bytes(nl, "little"))
    assert lib.zkhip_vm_ec_air_x(m, ca, nl, index, 1 if adapter else 0, C.byref(a)) == 0
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width)


def ec_host(op, modulus, coeff_a, p1, p2):
    """(slope, x3, y3) of p1 + p2 (op 0) or 2 p1 (op 1) through the library's host function (integers); None if it refuses the call"""
    nw = limb_words(modulus)
    words = [np.array(_int_words(v, nw), dtype=np.uint32) for v in (modulus, coeff_a, p1[0], p1[1], p2[0], p2[1])]
    out = [np.zeros(nw, np.uint32) for _ in range(3)]
    if load_library().zkhip_ec_host_x(op, nw, *[_u32p(w) for w in words], *[_u32p(o) for o in out]) != 0:
        return None
    return tuple(sum(int(x) << (32 * i) for i, x in enumerate(o)) for o in out)


def modmul_host(a, b, modulus):
    """(q, r) = divmod(a * b, modulus) through the library's host function (integers in, integers out)"""
    nw = limb_words(modulus)
    wa, wb, wm = (np.array(_int_words(v, nw), dtype=np.uint32) for v in (a, b, modulus))
    q, r = np.zeros(nw, np.uint32), np.zeros(nw, np.uint32)
    rc = load_library().zkhip_modular_host_x(0, nw, _u32p(wa), _u32p(wb), _u32p(wm), _u32p(q), _u32p(r))
    assert rc == 0
    return sum(int(x) << (32 * i) for i, x in enumerate(q)), sum(int(x) << (32 * i) for i, x in enumerate(r))


def modular_host(op, a, b, modulus):
    """(q, r) of a * b = q P + r (op 0), a + b = q P + r (1), a - b + q P = r (2) through the library's host function"""
    nw = limb_words(modulus)
    wa, wb, wm = (np.array(_int_words(v, nw), dtype=np.uint32) for v in (a, b, modulus))
    q, r = np.zeros(nw, np.uint32), np.zeros(nw, np.uint32)
    assert load_library().zkhip_modular_host_x(op, nw, _u32p(wa), _u32p(wb), _u32p(wm), _u32p(q), _u32p(r)) == 0
    return sum(int(x) << (32 * i) for i, x in enumerate(q)), sum(int(x) << (32 * i) for i, x in enumerate(r))


def sha256_air(log_height):
    """The SHA-256 compression chip's AIR for 2^log_height rows: (program, width, prep) -- prep = its preprocessed trace (6 columns,
    column-major, canonical); the one definition is include/zkhip_sha256.hpp."""
    lib = load_library()
    a = _Air()
    assert lib.zkhip_sha256_air(log_height, C.byref(a)) == 0
    prep = np.ctypeslib.as_array(a.prep_trace, shape=(6 << log_height,)).copy().reshape(6, 1 << log_height)
    return np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy(), int(a.width), prep


def sha256_compress_host(state, block):
    st = np.ascontiguousarray(state, dtype=np.uint32).copy()
    blk = np.ascontiguousarray(block, dtype=np.uint32)
    assert st.size == 8 and blk.size == 16 and load_library().zkhip_sha256_compress_host(_u32p(st), _u32p(blk)) == 0
    return st


def key_commit(prep_commits):
    """zkhip_recursion_key_commit: the digest of a node key's preprocessed commitments (list of 8-word arrays)."""
    pc = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32).reshape(-1) for x in prep_commits]), dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint32)
    rc = load_library().zkhip_recursion_key_commit(_u32p(pc), pc.size // 8, _u32p(out))
    if rc != 0:
        raise ZkhipError("zkhip_recursion_key_commit returned %d" % rc)
    return out


def vk_digest(params, airs):
    """zkhip_recursion_vk_digest: the digest a leaf circuit built for this verifying key states."""
    prm = _Params(*tuple(params))
    arr, keep = _air_structs(airs)
    out = np.zeros(8, dtype=np.uint32)
    rc = load_library().zkhip_recursion_vk_digest(C.byref(prm), arr, len(airs), _u32p(out))
    if rc != 0:
        raise ZkhipError("zkhip_recursion_vk_digest returned %d" % rc)
    return out


class RecursionCircuit:
    """zkhip_recursion: the verifier circuit of an aggregation node for ONE child verifying key (params + verifying AIRs with
    heights and preprocessed commitments) and up to `max_children` child proofs.  Building it and running the witness need no
    GPU.  stmt: None (no chained state), "node" (the children are node proofs of the level below, per-depth keys), "uniform" (ONE key:
    the children are proofs of the leaf circuit or of this very circuit, their preprocessed commitments are values: witness_uniform) or
    dict(start=[(air, idx)..], end=[(air, idx)..]); uniform=True on a leaf circuit appends the [leaf | internal commitment] words (zero)
    to its public values; min_log_height = (gate chip, Poseidon2 chip) pads; stmt "deferral": the deferral node over root proofs of a child
    app's aggregation key (region_index != 0: over JOIN proofs, include/zkhip.h zkhip_recursion_stmt.region_index)."""

    def __init__(self, params, child_airs, max_children, stmt=None, uniform=False, min_log_height=(0, 0), n_leaf_shapes=1, app_id=None, _fork_of=None,
                 _join_with=None, region_index=0):
        self.lib = load_library()
        self.params, self.child_airs = tuple(params), child_airs
        self.n_leaf_shapes = int(_fork_of.n_leaf_shapes if _fork_of is not None else n_leaf_shapes)
        if _join_with is not None:
            params_b, airs_b = _join_with
            pa, pb = _Params(*self.params), _Params(*tuple(params_b))
            arr_a, keep_a = _air_structs(child_airs)
            arr_b, keep_b = _air_structs(airs_b)
            h = C.c_void_p()
            rc = self.lib.zkhip_recursion_build_join(C.byref(pa), arr_a, len(child_airs), C.byref(pb), arr_b, len(airs_b), C.byref(h))
            if rc != 0:
                raise ZkhipError("zkhip_recursion_build_join returned %d: %s" % (rc, self.lib.zkhip_recursion_last_error(None).decode()))
            self._finish(h, 2)
            return
        if _fork_of is not None:
            h = C.c_void_p()
            assert self.lib.zkhip_recursion_fork(_fork_of.h, C.byref(h)) == 0
            self._finish(h, max_children)
            return
        prm = _Params(*self.params)
        arr, keep = _air_structs(child_airs)
        st = _RecursionStmt()
        st.uniform = 1 if uniform else 0
        st.min_log_height[0], st.min_log_height[1] = int(min_log_height[0]), int(min_log_height[1])
        st.n_leaf_shapes = int(n_leaf_shapes)
        self.n_leaf_shapes = int(n_leaf_shapes)
        aid = None
        if app_id is not None:
            aid = np.ascontiguousarray(app_id, dtype=np.uint32)
            st.app_id = _u32p(aid)
        keep2 = []
        if stmt == "node":
            st.child_is_node = 1
        elif stmt == "uniform":
            st.child_is_node = 2
        elif stmt == "deferral":
            st.child_is_node = 3
            st.region_index = int(region_index)   # != 0: the children are JOIN proofs (a bundle over batches)
        elif stmt:
            cols = [np.ascontiguousarray([x[k] for x in stmt[side]], dtype=np.uint32) for side in ("start", "end") for k in (0, 1)]
            keep2 = cols
            st.n_state = len(stmt["start"])
            st.start_air, st.start_idx, st.end_air, st.end_idx = (_u32p(c) for c in cols)
        h = C.c_void_p()
        rc = self.lib.zkhip_recursion_build(C.byref(prm), arr, len(child_airs), max_children, C.byref(st), C.byref(h))
        if rc != 0:
            raise ZkhipError("zkhip_recursion_build returned %d: %s" % (rc, self.lib.zkhip_recursion_last_error(None).decode()))
        self._finish(h, max_children)

    def _finish(self, h, max_children):
        self.h = h
        self.max_children = max_children
        self.n_pvs = self.lib.zkhip_recursion_n_pvs(h)
        self.n_state = self.lib.zkhip_recursion_n_state(h)
        out = (C.c_size_t * 4)()
        self.lib.zkhip_recursion_stats(h, out)
        self.n_wires, self.n_gates, self.n_perms = int(out[0]), int(out[1]), int(out[2])

    def airs(self):
        """The node circuit as AIR dicts (program, log_height, width, n_pvs, prep) for ProvingKey / the oracle prover."""
        out = []
        for i in range(3):
            a = _Air()
            rc = self.lib.zkhip_recursion_air(self.h, i, C.byref(a))
            assert rc == 0
            prog = np.ctypeslib.as_array(a.program, shape=(a.program_len,)).copy()
            pw = [23, 13, (self.n_pvs + 3) // 4][i]   # (csrc/recursion.hip GATE_PREP, P2W_PREP)
            prep = np.ctypeslib.as_array(a.prep_trace, shape=(pw << a.log_height,)).copy().reshape(pw, -1)
            out.append(dict(program=prog, log_height=int(a.log_height), width=int(a.width), n_pvs=int(a.n_pvs), prep=prep))
        return out

    def child_vk_digest(self):
        out = np.zeros(8, dtype=np.uint32)
        self.lib.zkhip_recursion_child_vk_digest(self.h, _u32p(out))
        return out

    @classmethod
    def join(cls, params_a, airs_a, params_b, airs_b):
        """zkhip_recursion_build_join: child 0 = a guest flow's root under aggregation key A, child 1 = the deferral node's proof (key B)."""
        return cls(params_a, airs_a, 2, _join_with=(params_b, airs_b))

    def fork(self):
        """A second user of the same circuit (shared wiring, own witness)."""
        return RecursionCircuit(self.params, self.child_airs, self.max_children, _fork_of=self)

    def pad(self, log_gate, log_p2):
        hh = (C.c_uint * 2)(int(log_gate), int(log_p2))
        rc = self.lib.zkhip_recursion_pad(self.h, hh)
        if rc != 0:
            raise ZkhipError("zkhip_recursion_pad returned %d" % rc)

    def log_heights(self):
        return [a["log_height"] for a in self.airs()]

    def witness(self, proofs, child_pvs, prep_commits=None, is_leaf=None, leaf_commit=None, internal_commit=None, aux=None, acc_start=None):
        """proofs: list of bytes; child_pvs[c][a]: public values of AIR a of child c.  Returns (status, node public values).
        A uniform circuit also takes prep_commits[c] (the three preprocessed commitments of child c's key, 3 x 8 words), is_leaf[c] and
        the two circuit commitments (key_commit) it states."""
        n = len(proofs)
        bufs = [np.frombuffer(p, dtype=np.uint8) for p in proofs]
        pp = (C.POINTER(C.c_uint8) * n)(*[b.ctypes.data_as(C.POINTER(C.c_uint8)) for b in bufs])
        lens = (C.c_size_t * n)(*[b.size for b in bufs])
        keep, rows = [], []
        for c in range(n):
            pa, k = _pvs_array(child_pvs[c])
            keep += [pa, k]
            rows.append(C.cast(pa, C.POINTER(C.POINTER(C.c_uint32))))
        pv = (C.POINTER(C.POINTER(C.c_uint32)) * n)(*rows)
        out = np.zeros(self.n_pvs, dtype=np.uint32)
        if aux is not None:   # a deferral node: the openings of the children's public values + the chain's value before this node
            ax = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32).reshape(-1) for x in aux]), dtype=np.uint32)
            a0 = np.ascontiguousarray(acc_start if acc_start is not None else np.zeros(8), dtype=np.uint32)
            assert ax.size == n * self.lib.zkhip_recursion_n_aux(self.h) and a0.size == 8
            rc = self.lib.zkhip_recursion_witness_deferral(self.h, pp, lens, pv, _u32p(ax), _u32p(a0), n, _u32p(out))
            return rc, out
        if prep_commits is not None:
            pc = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32).reshape(-1) for x in prep_commits]), dtype=np.uint32)
            kinds = (C.c_int * n)(*[int(x) for x in is_leaf])   # 0 = internal, j + 1 = leaf circuit j
            lc = np.ascontiguousarray(np.asarray(leaf_commit, dtype=np.uint32).reshape(-1), dtype=np.uint32)
            ic = np.ascontiguousarray(internal_commit, dtype=np.uint32)
            assert pc.size == 24 * n and lc.size == 8 * getattr(self, "n_leaf_shapes", 1) and ic.size == 8
            rc = self.lib.zkhip_recursion_witness_uniform(self.h, pp, lens, pv, _u32p(pc), kinds, _u32p(lc), _u32p(ic), n, _u32p(out))
            return rc, out
        rc = self.lib.zkhip_recursion_witness(self.h, pp, lens, pv, n, _u32p(out))
        return rc, out

    def last_error(self):
        return self.lib.zkhip_recursion_last_error(self.h).decode()

    def wires(self):
        n = C.c_size_t()
        self.lib.zkhip_recursion_wires(self.h, None, 0, C.byref(n))
        out = np.zeros(n.value, dtype=np.uint32)
        rc = self.lib.zkhip_recursion_wires(self.h, _u32p(out), out.size, C.byref(n))
        assert rc == 0
        return out.reshape(-1, 4)

    def tracegen(self, ctx):
        """Device traces of the three chips from the last witness (torch int32 tensors, Montgomery)."""
        import torch

        airs = self.airs()
        ts = [torch.empty(a["width"] << a["log_height"], dtype=torch.int32, device=ctx.device) for a in airs]
        ctx._check(self.lib.zkhip_recursion_tracegen(ctx.h, self.h, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr()))
        return ts

    def close(self):
        if self.h:
            self.lib.zkhip_recursion_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the verifier's primitives on the host (need no GPU) ---------------------------------------------
def poseidon2_permute_host(state):
    """Poseidon2 permutation of 16 canonical words on the host (the verifier's copy of the permutation)."""
    s = np.ascontiguousarray(state, dtype=np.uint32).copy()
    rc = load_library().zkhip_poseidon2_permute_host(_u32p(s))
    if rc != 0:
        raise ZkhipError("zkhip_poseidon2_permute_host returned %d" % rc)
    return s


def poseidon2_permute_host_avx512(state):
    """the same through the host's AVX-512 form (the transcript's long absorptions); None on a CPU without AVX-512"""
    s = np.ascontiguousarray(state, dtype=np.uint32).copy()
    rc = load_library().zkhip_poseidon2_permute_host_avx512(_u32p(s))
    if rc == 1:
        return None
    if rc != 0:
        raise ZkhipError("zkhip_poseidon2_permute_host_avx512 returned %d" % rc)
    return s


def mmcs_verify(root, log_heights, widths, index, opening):
    """p3 `Mmcs::verify_batch` for MerkleTreeMmcs<Poseidon2>: status code (0 = the opening leads to `root`)."""
    r = np.ascontiguousarray(root, dtype=np.uint32)
    op = np.ascontiguousarray(opening, dtype=np.uint32)
    lhs = (C.c_uint * len(log_heights))(*log_heights)
    ws = (C.c_size_t * len(widths))(*widths)
    return load_library().zkhip_mmcs_verify(_u32p(r), lhs, ws, len(log_heights), int(index), _u32p(op))


def fri_fold_row(index, log_height, beta, e0, e1):
    """p3-fri `fold_row` (arity 2) on canonical extension elements; returns 4 canonical words."""
    b, a0, a1 = (np.ascontiguousarray(v, dtype=np.uint32) for v in (beta, e0, e1))
    out = np.zeros(4, dtype=np.uint32)
    rc = load_library().zkhip_fri_fold_row(int(index), log_height, _u32p(b), _u32p(a0), _u32p(a1), _u32p(out))
    if rc != 0:
        raise ZkhipError("zkhip_fri_fold_row returned %d" % rc)
    return out


# ---- the reference's stored-proof container (OpenVM-v1 Proof<SC>, bincode): include/zkhip_codec.hpp ----
V1_SINGLE, V1_VEC = 0, 1


def _bytes_ptr(b):
    a = np.frombuffer(b, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def proof_decode_v1(blob, kind=V1_VEC):
    """Parses + validates a v1 container; returns the summary as a dict (raises ZkhipError when malformed)."""
    a, ptr = _bytes_ptr(blob)
    out = _V1Summary()
    rc = load_library().zkhip_proof_decode_v1(ptr, a.size, kind, C.byref(out))
    if rc != 0:
        raise ZkhipError("zkhip_proof_decode_v1 returned %d" % rc)
    d = {n: int(getattr(out, n)) for n, _ in _V1Summary._fields_ if n != "log_degree"}
    d["log_degree"] = [int(out.log_degree[i]) for i in range(min(d["n_airs"], 64))]
    return d


def proof_reencode_v1(blob, kind=V1_VEC):
    a, ptr = _bytes_ptr(blob)
    out = np.zeros(a.size + 64, dtype=np.uint8)
    n = C.c_size_t()
    rc = load_library().zkhip_proof_reencode_v1(ptr, a.size, kind, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size, C.byref(n))
    if rc != 0:
        raise ZkhipError("zkhip_proof_reencode_v1 returned %d" % rc)
    return out[:n.value].tobytes()


def proof_to_v1(params, airs, pvs, proof_bytes):
    """zkhip proof -> bincode(Proof<SC>) bytes."""
    lib = load_library()
    prm = _Params(*params)
    arr, keep = _air_structs(airs)
    pa, keep2 = _pvs_array(pvs)
    a, ptr = _bytes_ptr(proof_bytes)
    n = C.c_size_t()
    rc = lib.zkhip_proof_to_v1(C.byref(prm), arr, len(airs), pa, ptr, a.size, None, 0, C.byref(n))
    if rc != -5:
        raise ZkhipError("zkhip_proof_to_v1 returned %d" % rc)
    out = np.zeros(n.value, dtype=np.uint8)
    rc = lib.zkhip_proof_to_v1(C.byref(prm), arr, len(airs), pa, ptr, a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size,
                               C.byref(n))
    if rc != 0:
        raise ZkhipError("zkhip_proof_to_v1 returned %d" % rc)
    return out.tobytes()


def proof_from_v1(params, airs, v1_bytes):
    """bincode(Proof<SC>) -> (zkhip proof bytes, public values per AIR)."""
    lib = load_library()
    prm = _Params(*params)
    arr, keep = _air_structs(airs)
    a, ptr = _bytes_ptr(v1_bytes)
    size = proof_layout(params, airs)["n_words"] * 4
    out = np.zeros(size, dtype=np.uint8)
    pvs = [np.zeros(max(1, x["n_pvs"]), dtype=np.uint32) for x in airs]
    pv_arr = (C.POINTER(C.c_uint32) * len(airs))(*[_u32p(p) for p in pvs])
    n = C.c_size_t()
    rc = lib.zkhip_proof_from_v1(C.byref(prm), arr, len(airs), ptr, a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size,
                                 C.byref(n), pv_arr)
    if rc != 0:
        raise ZkhipError("zkhip_proof_from_v1 returned %d" % rc)
    return out[:n.value].tobytes(), [p[:x["n_pvs"]] for p, x in zip(pvs, airs)]
