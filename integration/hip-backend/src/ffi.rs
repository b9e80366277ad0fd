//! Raw bindings, one to one with `include/zkhip.h` (every entry point; checked by
//! `tests/test_rust_ffi_matches_header.py`).  All functions return `0` or a negative `ZKHIP_ERR_*`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_uint, c_void};

pub const ZKHIP_OK: c_int = 0;
pub const ZKHIP_ERR_NO_DEVICE: c_int = -1;
pub const ZKHIP_ERR_HIP: c_int = -2;
pub const ZKHIP_ERR_INVALID: c_int = -3;
pub const ZKHIP_ERR_NOMEM: c_int = -4;
pub const ZKHIP_ERR_SMALL_BUFFER: c_int = -5;
pub const ZKHIP_ERR_POW_FAILED: c_int = -6;
pub const ZKHIP_ERR_VERIFY: c_int = -7;
pub const ZKHIP_ERR_CONSTRAINT: c_int = -8;
pub const ZKHIP_DIGEST_WORDS: usize = 8;
pub const ZKHIP_KECCAK_F_WIDTH: usize = 2633;
pub const ZKHIP_INT256_ALU_WIDTH: usize = 101;
pub const ZKHIP_INT256_MUL_WIDTH: usize = 161;
pub const ZKHIP_MODMUL_WIDTH: usize = 325;
pub const ZKHIP_EC_WIDTH: usize = 772;
pub const ZKHIP_FP2_WIDTH: usize = 648;
pub const ZKHIP_FP2_RECORD_WORDS: usize = 33;
pub const ZKHIP_INT256_CMP_WIDTH: usize = 103;
pub const ZKHIP_INT256_SHIFT_WIDTH: usize = 189;
pub const ZKHIP_EC_RECORD_WORDS: usize = 41;
pub const ZKHIP_SHA256_WIDTH: usize = 433;
pub const ZKHIP_SHA256_PREP_WIDTH: usize = 6;
pub const ZKHIP_SHA256_ROWS_PER_BLOCK: usize = 65;
pub const ZKHIP_POSEIDON2_AIR_WIDTH: usize = 298;
pub const ZKHIP_MEMORY_BOUNDARY_WIDTH: usize = 8;
pub const ZKHIP_RV32_ALU_WIDTH: usize = 18;
pub const ZKHIP_RV32_MUL_WIDTH: usize = 13;
pub const ZKHIP_RV32_LT_WIDTH: usize = 18;
pub const ZKHIP_RV32_SHIFT_WIDTH: usize = 32;
pub const ZKHIP_RV32_BRANCH_EQ_WIDTH: usize = 17;
pub const ZKHIP_RV32_BRANCH_LT_WIDTH: usize = 23;
pub const ZKHIP_DUPLEX_WIDTH: usize = 50;
pub const ZKHIP_FRI_FOLD_WIDTH: usize = 19;
pub const ZKHIP_DOMAIN_POINT_BITS: usize = 26;
pub const ZKHIP_DOMAIN_POINT_WIDTH: usize = 54;
pub const ZKHIP_CASTF_WIDTH: usize = 6;
pub const ZKHIP_FIELD_ARITH_WIDTH: usize = 8;
pub const ZKHIP_FIELD_EXT_WIDTH: usize = 20;
pub const ZKHIP_MMCS_PATH_WIDTH: usize = 39;
pub const ZKHIP_RV32_MULH_WIDTH: usize = 21;
pub const ZKHIP_RV32_DIVREM_WIDTH: usize = 41;
pub const ZKHIP_RV32_LOADSTORE_WIDTH: usize = 33;
pub const ZKHIP_RV32_JAL_LUI_WIDTH: usize = 9;
pub const ZKHIP_RV32_AUIPC_WIDTH: usize = 14;
pub const ZKHIP_RV32_JALR_WIDTH: usize = 20;
pub const ZKHIP_MEMORY_ACCESS_WIDTH: usize = 10;
pub const ZKHIP_PROGRAM_FIELDS: usize = 9;
pub const ZKHIP_MAX_LOG_FINAL_POLY: u32 = 8;
pub const ZKHIP_V1_SINGLE: c_int = 0;
pub const ZKHIP_V1_VEC: c_int = 1;
pub const ZKHIP_V1_MAX_AIRS: usize = 64;

#[repr(C)]
pub struct zkhip_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkhip_tree {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkhip_transcript {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkhip_pk {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_matrix {
    pub data: *const u32, // device, column-major, Montgomery
    pub stride: usize,
    pub log_height: c_uint,
    pub width: usize,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_params {
    pub log_blowup: u32,
    pub log_final_poly_len: u32,
    pub num_queries: u32,
    pub commit_pow_bits: u32,
    pub query_pow_bits: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_air {
    pub program: *const u32, // host, AIR constraint bytecode
    pub program_len: usize,  // words
    pub log_height: c_uint,
    pub width: usize,
    pub n_pvs: usize,
    pub prep_trace: *const u32,  // host, canonical, column-major; null when the AIR has no preprocessed trace
    pub prep_commit: *const u32, // 8 canonical words (verifying-key entry); null when none
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct zkhip_proof_layout {
    pub n_words: usize,
    pub root_main: usize,
    pub roots_cached: usize,
    pub n_cached: usize,
    pub root_perm: usize,
    pub exposed: usize,
    pub n_exposed: usize,
    pub root_quot: usize,
    pub opened: usize,
    pub n_opened: usize,
    pub fri_layers: usize,
    pub n_fri_layers: usize,
    pub final_poly: usize,
    pub n_final_poly: usize,
    pub query_pow: usize,
    pub queries: usize,
    pub query_words: usize,
    pub n_queries: usize,
}

/// summary of a decoded OpenVM-v1 `Proof<SC>` container (zkhip_proof_decode_v1)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_v1_summary {
    pub n_proofs: usize,
    pub n_airs: usize,
    pub n_queries: usize,
    pub n_fri_layers: usize,
    pub n_final_poly: usize,
    pub n_main_commits: usize,
    pub n_after_challenge_commits: usize,
    pub n_preprocessed: usize,
    pub n_input_batches: usize,
    pub log_max_height: c_uint,
    pub log_blowup: c_uint,
    pub has_logup_pow: c_int,
    pub log_degree: [c_uint; ZKHIP_V1_MAX_AIRS],
}

#[repr(C)]
pub struct zkhip_recursion {
    _private: [u8; 0],
}

/// include/zkhip.h `zkhip_recursion_stmt`: where the chained state of a child proof lives in its public values.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_recursion_stmt {
    pub n_state: usize,
    pub start_air: *const u32,
    pub start_idx: *const u32,
    pub end_air: *const u32,
    pub end_idx: *const u32,
    /// 0 = leaf (children: proofs of a fixed key), 1 = node of the level below (per-depth keys), 2 = UNIFORM node (one aggregation key)
    pub child_is_node: c_int,
    /// leaf circuit: append the [leaf commitment | internal commitment] words (zero) to the public values
    pub uniform: c_int,
    /// pads the gate / Poseidon2 chip (0 = natural size)
    pub min_log_height: [c_uint; 2],
    /// uniform node: how many leaf circuits (shapes: sets of chips a segment may carry) its leaf children come from (0 = 1)
    pub n_leaf_shapes: usize,
    /// leaf circuits of one app state one app id (8 canonical words or null)
    pub app_id: *const u32,
    /// deferral node (child_is_node = 3) over JOIN proofs: the node of a child's memory tree above its deferral region (0 = plain roots)
    pub region_index: u32,
}

/// include/zkhip.h `zkhip_config`: every behaviour-changing switch of the library; the ZKHIP_* environment variables are overrides read
/// in one place (`zkhip_config_default`).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_config {
    pub host_sponge: c_int,
    pub host_sponge_min_words: u32,
    pub jit: c_int,
    pub jit_min_log_work: u32,
    pub jit_cache_dir: [c_char; 256],
    pub quot_slices: c_int,
    pub grind_sweep_shift: u32,
    pub coop_max_log: u32,
    pub coop_inj_max_log: u32,
    pub top_max_log: u32,
    pub commit_parts: u32,
    pub side_cus: u32,
    pub witness_threads: u32,
    pub pin_witness: c_int,
    pub parallel_queries: c_int,
    pub self_check: c_int,
    pub tree_store_early: c_int,
    pub hash_block: u32,
    pub coop_fused: c_int,
    pub rows_in_bulk: c_int,
    pub rows_coop_max_log: u32,
    pub ntt_log_lanes: u32,
    pub quot_streams: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_kernel_stat {
    pub name: [c_char; 48],
    pub launches: u64,
    pub total_ms: c_double,
}

extern "C" {
    pub fn zkhip_version() -> u32;

    // context
    pub fn zkhip_ctx_create(device: c_int, out: *mut *mut zkhip_ctx) -> c_int;
    pub fn zkhip_ctx_destroy(ctx: *mut zkhip_ctx);
    pub fn zkhip_last_error(ctx: *const zkhip_ctx) -> *const c_char;
    pub fn zkhip_set_stream(ctx: *mut zkhip_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn zkhip_sync(ctx: *mut zkhip_ctx) -> c_int;
    pub fn zkhip_set_commit_pipeline(ctx: *mut zkhip_ctx, parts: c_uint) -> c_int;
    pub fn zkhip_set_cu_partition(ctx: *mut zkhip_ctx, side_cus: c_uint) -> c_int;

    // device memory
    pub fn zkhip_malloc(ctx: *mut zkhip_ctx, bytes: usize, dptr: *mut *mut c_void) -> c_int;
    pub fn zkhip_free(ctx: *mut zkhip_ctx, dptr: *mut c_void) -> c_int;
    pub fn zkhip_h2d(ctx: *mut zkhip_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn zkhip_d2h(ctx: *mut zkhip_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn zkhip_host_alloc(ctx: *mut zkhip_ctx, bytes: usize, hptr: *mut *mut c_void) -> c_int;
    pub fn zkhip_host_free(ctx: *mut zkhip_ctx, hptr: *mut c_void) -> c_int;
    pub fn zkhip_h2d_async(ctx: *mut zkhip_ctx, dst: *mut c_void, src_pinned: *const c_void, bytes: usize) -> c_int;
    pub fn zkhip_zero(ctx: *mut zkhip_ctx, dptr: *mut c_void, bytes: usize) -> c_int;
    pub fn zkhip_to_monty(ctx: *mut zkhip_ctx, d: *mut u32, n: usize) -> c_int;
    pub fn zkhip_from_monty(ctx: *mut zkhip_ctx, d: *mut u32, n: usize) -> c_int;

    // NTT / LDE
    pub fn zkhip_ntt_batch(ctx: *mut zkhip_ctx, d_mat: *mut u32, log_n: c_uint, width: usize, stride: usize,
                           inverse: c_int, bitrev_out: c_int) -> c_int;
    pub fn zkhip_lde_batch(ctx: *mut zkhip_ctx, d_in: *const u32, in_stride: usize, d_out: *mut u32, out_stride: usize,
                           log_n: c_uint, added_bits: c_uint, width: usize, shift: u32) -> c_int;

    // Poseidon2, trace generators
    pub fn zkhip_poseidon2_permute_batch(ctx: *mut zkhip_ctx, d_states: *mut u32, n: usize) -> c_int;
    pub fn zkhip_poseidon2_air_tracegen(ctx: *mut zkhip_ctx, d_inputs: *const u32, n_perms: usize, log_height: c_uint,
                                        d_trace: *mut u32) -> c_int;
    pub fn zkhip_range_counts_tracegen(ctx: *mut zkhip_ctx, d_values: *const u32, n: usize, log_table: c_uint,
                                       d_counts: *mut u32, accumulate: c_int) -> c_int;

    pub fn zkhip_range_tuple_counts_tracegen(ctx: *mut zkhip_ctx, d_x: *const u32, d_y: *const u32, n: usize, size_x: u32, size_y: u32,
                                             d_counts: *mut u32, accumulate: c_int) -> c_int;
    pub fn zkhip_bitwise_lookup_tracegen(ctx: *mut zkhip_ctx, d_x: *const u32, d_y: *const u32, d_op: *const u32, n: usize,
                                         num_bits: c_uint, d_trace: *mut u32, accumulate: c_int) -> c_int;
    pub fn zkhip_rv32_alu_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_b: *const u32, d_c: *const u32, n: usize, log_height: c_uint,
                                   d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_memory_access_tracegen(ctx: *mut zkhip_ctx, d_addr_space: *const u32, d_pointer: *const u32, d_prev_data: *const u32,
                                        d_prev_ts: *const u32, d_data: *const u32, d_ts: *const u32, d_is_read: *const u32, n: usize,
                                        log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_shift_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_b: *const u32, d_c: *const u32, n: usize, log_height: c_uint,
                                     d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_branch_lt_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_a: *const u32, d_b: *const u32, d_imm: *const u32, n: usize,
                                         log_height: c_uint, d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_var_range_counts_tracegen(ctx: *mut zkhip_ctx, d_values: *const u32, d_bits: *const u32, const_bits: u32, n: usize, max_bits: c_uint,
                                           d_counts: *mut u32, accumulate: c_int) -> c_int;
    pub fn zkhip_domain_point_tracegen(ctx: *mut zkhip_ctx, d_k: *const u32, d_mult: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_duplex_tracegen(ctx: *mut zkhip_ctx, d_n_observed: *const u32, d_observed: *const u32, d_n_sampled: *const u32, n: usize,
                                 log_height: c_uint, d_trace: *mut u32, d_hash_inputs: *mut u32) -> c_int;
    pub fn zkhip_fri_fold_chip_tracegen(ctx: *mut zkhip_ctx, d_e0: *const u32, d_e1: *const u32, d_beta: *const u32, d_k: *const u32,
                                        d_log_n_out: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_castf_tracegen(ctx: *mut zkhip_ctx, d_x: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32, d_var_range_counts: *mut u32,
                                max_bits: c_uint) -> c_int;
    pub fn zkhip_field_arith_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_b: *const u32, d_c: *const u32, n: usize, log_height: c_uint,
                                      d_trace: *mut u32) -> c_int;
    pub fn zkhip_field_ext_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_x: *const u32, d_y: *const u32, n: usize, log_height: c_uint,
                                    d_trace: *mut u32) -> c_int;
    pub fn zkhip_mmcs_path_tracegen(ctx: *mut zkhip_ctx, d_leaf: *const u32, d_index: *const u32, d_path_start: *const u32, d_step_kind: *const u32,
                                    d_step_digest: *const u32, n_paths: usize, log_height: c_uint, d_trace: *mut u32, d_hash_inputs: *mut u32) -> c_int;
    pub fn zkhip_rv32_divrem_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_b: *const u32, d_c: *const u32, n: usize, log_height: c_uint,
                                      d_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_mulh_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_b: *const u32, d_c: *const u32, n: usize, log_height: c_uint,
                                    d_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_loadstore_tracegen(ctx: *mut zkhip_ctx, d_case: *const u32, d_read: *const u32, d_prev: *const u32, n: usize, log_height: c_uint,
                                         d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_jal_lui_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_pc: *const u32, d_imm: *const u32, n: usize, log_height: c_uint,
                                       d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_auipc_tracegen(ctx: *mut zkhip_ctx, d_pc: *const u32, d_imm: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                     d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_jalr_tracegen(ctx: *mut zkhip_ctx, d_pc: *const u32, d_rs1: *const u32, d_imm: *const u32, n: usize, log_height: c_uint,
                                    d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_branch_eq_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_a: *const u32, d_b: *const u32, d_imm: *const u32, n: usize,
                                         log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_lt_tracegen(ctx: *mut zkhip_ctx, d_opcode: *const u32, d_b: *const u32, d_c: *const u32, n: usize, log_height: c_uint,
                                  d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_program_freq_tracegen(ctx: *mut zkhip_ctx, d_pc_index: *const u32, n: usize, log_height: c_uint, d_freq: *mut u32) -> c_int;
    pub fn zkhip_exec_frame_tracegen(ctx: *mut zkhip_ctx, d_pc_index: *const u32, n: usize, d_program: *const u32, n_program: usize,
                                     log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_rv32_mul_tracegen(ctx: *mut zkhip_ctx, d_b: *const u32, d_c: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                   d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_memory_boundary_tracegen(ctx: *mut zkhip_ctx, d_addr_space: *const u32, d_pointer: *const u32, d_initial: *const u32,
                                          d_final: *const u32, d_timestamp: *const u32, n: usize, as_bits: c_uint, pointer_bits: c_uint,
                                          log_height: c_uint, d_trace: *mut u32) -> c_int;

    // Merkle
    pub fn zkhip_merkle_commit(ctx: *mut zkhip_ctx, mats: *const zkhip_matrix, n_mats: usize, tree: *mut *mut zkhip_tree,
                               root_out: *mut u32) -> c_int;
    pub fn zkhip_tree_root_device(tree: *const zkhip_tree) -> *const u32;
    pub fn zkhip_tree_log_height(tree: *const zkhip_tree) -> c_uint;
    pub fn zkhip_tree_layer(ctx: *mut zkhip_ctx, tree: *const zkhip_tree, layer: c_uint, out: *mut u32) -> c_int;
    pub fn zkhip_merkle_rebuild(ctx: *mut zkhip_ctx, tree: *mut zkhip_tree) -> c_int;
    pub fn zkhip_tree_check(ctx: *mut zkhip_ctx, tree: *const zkhip_tree, n_bad: *mut u32, first: *mut u32) -> c_int;
    pub fn zkhip_merkle_opening_words(tree: *const zkhip_tree) -> usize;
    pub fn zkhip_merkle_open(ctx: *mut zkhip_ctx, tree: *const zkhip_tree, indices: *const u64, n: usize, out: *mut u32,
                             cap_words: usize) -> c_int;
    pub fn zkhip_tree_destroy(ctx: *mut zkhip_ctx, tree: *mut zkhip_tree);

    // constraint evaluation (K5) as a stage
    pub fn zkhip_constraint_eval(ctx: *mut zkhip_ctx, program: *const u32, program_len: usize, log_height: c_uint, log_blowup: c_uint,
                                 width: usize, d_lde: *const u32, pvs: *const u32, n_pvs: usize, alpha: *const u32, d_q: *mut u32) -> c_int;

    // FRI fold, LogUp / sum-check blocks
    pub fn zkhip_fri_fold(ctx: *mut zkhip_ctx, d_in: *const u32, d_out: *mut u32, log_n_out: c_uint, beta: *const u32) -> c_int;
    pub fn zkhip_ext_batch_inverse(ctx: *mut zkhip_ctx, d_in: *const u32, d_out: *mut u32, n: usize) -> c_int;
    pub fn zkhip_logup_running_sum(ctx: *mut zkhip_ctx, d_den: *const u32, d_num: *const u32, n: usize, d_out: *mut u32,
                                   total_out: *mut u32) -> c_int;
    pub fn zkhip_mle_fold(ctx: *mut zkhip_ctx, d_in: *const u32, d_out: *mut u32, n: usize, r: *const u32) -> c_int;
    pub fn zkhip_sumcheck_round(ctx: *mut zkhip_ctx, d_tables: *const *const u32, k: usize, n_half: usize, out: *mut u32) -> c_int;

    // transcript
    pub fn zkhip_transcript_create(ctx: *mut zkhip_ctx, out: *mut *mut zkhip_transcript) -> c_int;
    pub fn zkhip_transcript_destroy(ctx: *mut zkhip_ctx, t: *mut zkhip_transcript);
    pub fn zkhip_transcript_observe(ctx: *mut zkhip_ctx, t: *mut zkhip_transcript, vals: *const u32, n: usize) -> c_int;
    pub fn zkhip_transcript_sample(ctx: *mut zkhip_ctx, t: *mut zkhip_transcript, out: *mut u32, n: usize) -> c_int;
    pub fn zkhip_transcript_grind(ctx: *mut zkhip_ctx, t: *mut zkhip_transcript, bits: c_uint, witness: *mut u32) -> c_int;

    // STARK
    pub fn zkhip_keygen(ctx: *mut zkhip_ctx, params: *const zkhip_params, airs: *const zkhip_air, n_airs: usize,
                        out: *mut *mut zkhip_pk) -> c_int;
    pub fn zkhip_pk_destroy(ctx: *mut zkhip_ctx, pk: *mut zkhip_pk);
    pub fn zkhip_pk_prep_commitment(ctx: *mut zkhip_ctx, pk: *const zkhip_pk, air_index: usize, out: *mut u32) -> c_int;
    pub fn zkhip_proof_size(pk: *const zkhip_pk) -> usize;
    pub fn zkhip_pk_workspace_bytes(pk: *const zkhip_pk) -> usize;
    pub fn zkhip_prove(ctx: *mut zkhip_ctx, pk: *const zkhip_pk, d_traces: *const *const u32, pvs: *const *const u32,
                       out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
    pub fn zkhip_prove_async(ctx: *mut zkhip_ctx, pk: *const zkhip_pk, d_traces: *const *const u32, pvs: *const *const u32) -> c_int;
    pub fn zkhip_proof_fetch(ctx: *mut zkhip_ctx, pk: *const zkhip_pk, out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
    pub fn zkhip_verify(params: *const zkhip_params, airs: *const zkhip_air, n_airs: usize, pvs: *const *const u32,
                        proof: *const u8, len: usize) -> c_int;
    pub fn zkhip_verify_where(params: *const zkhip_params, airs: *const zkhip_air, n_airs: usize, pvs: *const *const u32,
                        proof: *const u8, len: usize, where_: *mut c_int) -> c_int;

    pub fn zkhip_proof_layout_of(params: *const zkhip_params, airs: *const zkhip_air, n_airs: usize,
                                 out: *mut zkhip_proof_layout) -> c_int;

    // the verifier's primitives on the host (no device)
    pub fn zkhip_poseidon2_permute_host(state: *mut u32) -> c_int;
    pub fn zkhip_poseidon2_permute16_host(states: *mut u32) -> c_int;
    pub fn zkhip_poseidon2_permute_host_avx512(state: *mut u32) -> c_int;
    pub fn zkhip_mmcs_verify(root: *const u32, log_heights: *const c_uint, widths: *const usize, n_mats: usize, index: u64,
                             opening: *const u32) -> c_int;
    pub fn zkhip_fri_fold_row(index: u64, log_height: c_uint, beta: *const u32, e0: *const u32, e1: *const u32, out: *mut u32) -> c_int;
    pub fn zkhip_logup_exposed_check(exposed: *const u32, n: usize) -> c_int;

    // the reference's stored-proof container (OpenVM-v1 Proof<SC>, bincode)
    pub fn zkhip_proof_decode_v1(bytes: *const u8, len: usize, kind: c_int, out: *mut zkhip_v1_summary) -> c_int;
    pub fn zkhip_proof_reencode_v1(bytes: *const u8, len: usize, kind: c_int, out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
    pub fn zkhip_proof_to_v1(params: *const zkhip_params, airs: *const zkhip_air, n_airs: usize, pvs: *const *const u32,
                             proof: *const u8, len: usize, out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
    pub fn zkhip_proof_from_v1(params: *const zkhip_params, airs: *const zkhip_air, n_airs: usize, v1: *const u8, v1_len: usize,
                               out: *mut u8, cap: usize, out_len: *mut usize, pvs_out: *const *mut u32) -> c_int;

    pub fn zkhip_tracegen_defer_checks(ctx: *mut zkhip_ctx, on: c_int) -> c_int;
    pub fn zkhip_tracegen_check(ctx: *mut zkhip_ctx) -> c_int;

    // the one-statement VM circuit: AIR set, decode, program table, trace generators of the adapter-side chips
    pub fn zkhip_vm_n_airs() -> usize;
    pub fn zkhip_vm_air(id: c_uint, out: *mut zkhip_air, prep_width: *mut usize) -> c_int;
    pub fn zkhip_vm_decode(word: u32, pc: u32, out: *mut u32, legal: *mut c_int) -> c_int;
    pub fn zkhip_vm_program_table(words: *const u32, n_words: usize, pc_base: u32, log_program: c_uint, out: *mut u32) -> c_int;
    pub fn zkhip_vm_frame_tracegen(ctx: *mut zkhip_ctx, d_pc_index: *const u32, d_x: *const u32, d_y: *const u32, d_z: *const u32,
                                   d_rd_prev: *const u32, d_pc_inc: *const u32, d_prev_ts_rs1: *const u32, d_prev_ts_rs2: *const u32,
                                   d_prev_ts_rd: *const u32, n: usize, d_program: *const u32, n_program: usize, log_height: c_uint,
                                   d_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_loadstore_tracegen(ctx: *mut zkhip_ctx, d_case: *const u32, d_read: *const u32, d_prev: *const u32, d_ts: *const u32,
                                       d_base: *const u32, d_imm: *const u32, d_prev_ts: *const u32, n: usize, log_height: c_uint,
                                       d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_keccak_tracegen(ctx: *mut zkhip_ctx, d_states: *const u32, d_ts: *const u32, n_perms: usize, log_height: c_uint,
                                    d_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_sha256_tracegen(ctx: *mut zkhip_ctx, d_blocks: *const u32, d_ts: *const u32, n_blocks: usize, log_height: c_uint,
                                    d_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_sha256_prep(log_height: c_uint, out: *mut u32) -> c_int;
    pub fn zkhip_vm_modmul_air(modulus: *const u8, index: c_uint, adapter: c_int, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_vm_modmul_tracegen(ctx: *mut zkhip_ctx, modulus: *const u32, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint,
                                    d_trace: *mut u32, d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_vm_poseidon2_tracegen(ctx: *mut zkhip_ctx, d_inputs: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_rows_tracegen(ctx: *mut zkhip_ctx, d_rows: *const u32, n: usize, width: usize, log_height: c_uint, d_trace: *mut u32,
                               pad_row: *const u32) -> c_int;
    pub fn zkhip_range_counts_scaled_tracegen(ctx: *mut zkhip_ctx, d_values: *const u32, n: usize, scale: u32, log_table: c_uint,
                                              d_counts: *mut u32, accumulate: c_int) -> c_int;

    // Keccak-f[1600] chip
    pub fn zkhip_keccak_f_air(out: *mut zkhip_air) -> c_int;
    pub fn zkhip_keccak_f1600_host(state: *mut u64) -> c_int;
    pub fn zkhip_keccak_f_tracegen(ctx: *mut zkhip_ctx, d_states: *const u32, n_perms: usize, log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_int256_alu_air(bitwise_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_int256_alu_host(op: u32, b: *const u32, c: *const u32, a: *mut u32) -> c_int;
    pub fn zkhip_int256_alu_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                     d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_int256_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                    d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_int256_mul_air(bitwise_bus: u32, tuple_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_int256_mul_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32, d_bitwise_trace: *mut u32,
                                     d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_int256_cmp_air(bitwise_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_int256_cmp_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_cmp256_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                    d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_int256_shift_air(bitwise_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_int256_shift_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32, d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_shift256_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                      d_bitwise_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_native_arith_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_native_ext_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_castf_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32) -> c_int;
    pub fn zkhip_vm_mul256_tracegen(ctx: *mut zkhip_ctx, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                    d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    // the `_x` forms of the limb chips: n_limbs = 32 or 48 (BLS12-381's base field, Fp2 and G1 of the batch circuit), n_words = n_limbs / 4
    pub fn zkhip_vm_modmul_air_x(modulus: *const u8, n_limbs: u32, index: c_uint, adapter: c_int, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_vm_modmul_tracegen_x(ctx: *mut zkhip_ctx, n_words: u32, modulus: *const u32, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint,
                                      d_trace: *mut u32, d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_vm_ec_air_x(modulus: *const u8, a: *const u8, n_limbs: u32, index: c_uint, adapter: c_int, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_vm_ec_tracegen_x(ctx: *mut zkhip_ctx, n_words: u32, modulus: *const u32, a: *const u32, d_records: *const u32, d_ts: *const u32, n: usize,
                                  log_height: c_uint, d_trace: *mut u32, d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_vm_fp2_air_x(modulus: *const u8, n_limbs: u32, index: c_uint, adapter: c_int, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_vm_fp2_tracegen_x(ctx: *mut zkhip_ctx, n_words: u32, modulus: *const u32, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint,
                                   d_trace: *mut u32, d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_modmul_air_x(modulus: *const u8, n_limbs: u32, bitwise_bus: u32, tuple_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_modular_host_x(op: u32, n_words: u32, a: *const u32, b: *const u32, modulus: *const u32, q: *mut u32, r: *mut u32) -> c_int;
    pub fn zkhip_modular_tracegen_x(ctx: *mut zkhip_ctx, n_words: u32, modulus: *const u32, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                    d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_ec_air_x(modulus: *const u8, a: *const u8, n_limbs: u32, bitwise_bus: u32, tuple_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_ec_host_x(op: u32, n_words: u32, modulus: *const u32, a: *const u32, x1: *const u32, y1: *const u32, x2: *const u32, y2: *const u32,
                           slope: *mut u32, x3: *mut u32, y3: *mut u32) -> c_int;
    pub fn zkhip_ec_tracegen_x(ctx: *mut zkhip_ctx, n_words: u32, modulus: *const u32, a: *const u32, d_records: *const u32, n: usize, log_height: c_uint,
                               d_trace: *mut u32, d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_fp2_air_x(modulus: *const u8, n_limbs: u32, bitwise_bus: u32, tuple_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_fp2_host_x(op: u32, n_words: u32, modulus: *const u32, a: *const u32, b: *const u32, r: *mut u32) -> c_int;
    pub fn zkhip_fp2_tracegen_x(ctx: *mut zkhip_ctx, n_words: u32, modulus: *const u32, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_fp2_air(modulus: *const u8, bitwise_bus: u32, tuple_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_fp2_host(op: u32, modulus: *const u32, a: *const u32, b: *const u32, r: *mut u32) -> c_int;
    pub fn zkhip_fp2_tracegen(ctx: *mut zkhip_ctx, modulus: *const u32, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                              d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_vm_fp2_air(modulus: *const u8, index: c_uint, adapter: c_int, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_vm_fp2_tracegen(ctx: *mut zkhip_ctx, modulus: *const u32, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint,
                                 d_trace: *mut u32, d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_ec_air(modulus: *const u8, a: *const u8, bitwise_bus: u32, tuple_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_ec_host(op: u32, modulus: *const u32, a: *const u32, x1: *const u32, y1: *const u32, x2: *const u32, y2: *const u32, slope: *mut u32,
                         x3: *mut u32, y3: *mut u32) -> c_int;
    pub fn zkhip_ec_tracegen(ctx: *mut zkhip_ctx, modulus: *const u32, a: *const u32, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                             d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_vm_ec_air(modulus: *const u8, a: *const u8, index: c_uint, adapter: c_int, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_vm_ec_tracegen(ctx: *mut zkhip_ctx, modulus: *const u32, a: *const u32, d_records: *const u32, d_ts: *const u32, n: usize, log_height: c_uint,
                                d_trace: *mut u32, d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_modmul_air(modulus: *const u8, bitwise_bus: u32, tuple_bus: u32, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_modmul_host(a: *const u32, b: *const u32, modulus: *const u32, q: *mut u32, r: *mut u32) -> c_int;
    pub fn zkhip_modmul_tracegen(ctx: *mut zkhip_ctx, modulus: *const u32, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                 d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_modular_tracegen(ctx: *mut zkhip_ctx, modulus: *const u32, d_records: *const u32, n: usize, log_height: c_uint, d_trace: *mut u32,
                                  d_bitwise_trace: *mut u32, d_tuple_counts: *mut u32, size_x: u32, size_y: u32) -> c_int;
    pub fn zkhip_modular_host(op: u32, a: *const u32, b: *const u32, modulus: *const u32, q: *mut u32, r: *mut u32) -> c_int;
    pub fn zkhip_sha256_air(log_height: c_uint, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_sha256_compress_host(state: *mut u32, block: *const u32) -> c_int;
    pub fn zkhip_sha256_tracegen(ctx: *mut zkhip_ctx, d_blocks: *const u32, n_blocks: usize, log_height: c_uint, d_trace: *mut u32) -> c_int;

    // aggregation layer: the verifier circuit of a node (commit_child_vk / VerifyProver of the reference)
    pub fn zkhip_recursion_build(child_params: *const zkhip_params, child_airs: *const zkhip_air, n_child_airs: usize, max_children: usize,
                                 stmt: *const zkhip_recursion_stmt, out: *mut *mut zkhip_recursion) -> c_int;
    pub fn zkhip_recursion_destroy(r: *mut zkhip_recursion);
    pub fn zkhip_recursion_last_error(r: *const zkhip_recursion) -> *const c_char;
    pub fn zkhip_recursion_n_airs(r: *const zkhip_recursion) -> usize;
    pub fn zkhip_recursion_n_pvs(r: *const zkhip_recursion) -> usize;
    pub fn zkhip_recursion_n_state(r: *const zkhip_recursion) -> usize;
    pub fn zkhip_recursion_max_children(r: *const zkhip_recursion) -> usize;
    pub fn zkhip_recursion_child_proof_bytes(r: *const zkhip_recursion) -> usize;
    pub fn zkhip_recursion_stats(r: *const zkhip_recursion, out: *mut usize) -> c_int;
    pub fn zkhip_recursion_child_vk_digest(r: *const zkhip_recursion, out: *mut u32) -> c_int;
    pub fn zkhip_recursion_air(r: *const zkhip_recursion, i: usize, out: *mut zkhip_air) -> c_int;
    pub fn zkhip_recursion_witness(r: *mut zkhip_recursion, proofs: *const *const u8, proof_lens: *const usize,
                                   child_pvs: *const *const *const u32, n_present: usize, node_pvs_out: *mut u32) -> c_int;
    pub fn zkhip_recursion_witness_uniform(r: *mut zkhip_recursion, proofs: *const *const u8, proof_lens: *const usize,
                                           child_pvs: *const *const *const u32, child_prep_commits: *const u32, child_is_leaf: *const c_int,
                                           leaf_commit: *const u32, internal_commit: *const u32, n_present: usize, node_pvs_out: *mut u32) -> c_int;
    pub fn zkhip_tables_canonical(ctx: *mut zkhip_ctx, on: c_int) -> c_int;
    pub fn zkhip_host_cpus() -> c_uint;
    pub fn zkhip_jit_prewarm(airs: *const zkhip_air, n_airs: usize, log_blowup: c_uint, cache_dir: *const c_char, n_ok: *mut usize) -> c_int;
    pub fn zkhip_config_default(out: *mut zkhip_config);
    pub fn zkhip_ctx_get_config(ctx: *mut zkhip_ctx, out: *mut zkhip_config) -> c_int;
    pub fn zkhip_has_test_kernels() -> c_int;
    pub fn zkhip_ctx_set_config(ctx: *mut zkhip_ctx, cfg: *const zkhip_config) -> c_int;
    pub fn zkhip_set_process_config(cfg: *const zkhip_config) -> c_int;
    pub fn zkhip_recursion_build_join(params_a: *const zkhip_params, airs_a: *const zkhip_air, n_airs_a: usize, params_b: *const zkhip_params,
                                      airs_b: *const zkhip_air, n_airs_b: usize, out: *mut *mut zkhip_recursion) -> c_int;
    pub fn zkhip_recursion_witness_deferral(r: *mut zkhip_recursion, proofs: *const *const u8, proof_lens: *const usize,
                                            child_pvs: *const *const *const u32, child_aux: *const u32, acc_start: *const u32,
                                            n_present: usize, node_pvs_out: *mut u32) -> c_int;
    pub fn zkhip_recursion_n_aux(r: *const zkhip_recursion) -> usize;
    pub fn zkhip_recursion_fork(r: *const zkhip_recursion, out: *mut *mut zkhip_recursion) -> c_int;
    pub fn zkhip_recursion_pad(r: *mut zkhip_recursion, log_height: *const c_uint) -> c_int;
    pub fn zkhip_recursion_vk_digest(params: *const zkhip_params, airs: *const zkhip_air, n_airs: usize, out: *mut u32) -> c_int;
    pub fn zkhip_recursion_key_commit(prep_commits: *const u32, n_commits: usize, out: *mut u32) -> c_int;
    pub fn zkhip_recursion_wires(r: *const zkhip_recursion, out: *mut u32, cap_words: usize, n_words: *mut usize) -> c_int;
    pub fn zkhip_recursion_tracegen(ctx: *mut zkhip_ctx, r: *mut zkhip_recursion, d_gate_trace: *mut u32, d_p2_trace: *mut u32,
                                    d_pv_trace: *mut u32) -> c_int;

    // per-kernel timing
    pub fn zkhip_profile_enable(ctx: *mut zkhip_ctx, on: c_int) -> c_int;
    pub fn zkhip_profile_read(ctx: *mut zkhip_ctx, out: *mut zkhip_kernel_stat, cap: usize) -> c_int;
    pub fn zkhip_profile_reset(ctx: *mut zkhip_ctx) -> c_int;
}
