/*
 * zkhip.h -- C ABI of libzkhip.so, the MI355X (gfx950) STARK proving backend.
 *
 * This is the drop-in boundary for the one path of scroll-tech/zkvm-prover that
 * BASELINE.json names: the work `Prover::gen_proof_stark` hands to
 * `sdk.prove(..)` (crates/prover/src/prover/mod.rs:355-357), i.e. the STARK
 * engine the reference selects with its `cuda` cargo feature
 * (crates/prover/Cargo.toml:41-46, type aliases crates/prover/src/prover/mod.rs:27-39).
 * A Rust `openvm-hip-backend` shim binds these entry points the
 * way the upstream CUDA backend binds its `extern "C"` kernels launchers:
 * plain pointers and sizes, `int` status (0 = ok, negative = error, message via
 * zkhip_last_error), opaque handles with matching _destroy, caller-owned
 * output buffers.  No torch or C++ types cross this boundary.
 *
 * Conventions
 *  - Field elements are BabyBear u32.  DEVICE buffers hold Montgomery form
 *    (R = 2^32), the in-memory form of p3-baby-bear (the reference only
 *    canonicalises at serialization, prover/mod.rs:136-137).  HOST-visible
 *    results (roots, proofs, challenges) are canonical little-endian u32.
 *  - Matrices are column-major: column c occupies [c*stride, c*stride+height).
 *  - Extension elements (degree 4 over BabyBear, x^4 = 11) are 4 consecutive u32.
 *  - Digests are 8 u32 (DIGEST_SIZE, crates/types/src/proof.rs:209).
 *  - One zkhip_ctx per GPU; not thread-safe; all work is issued on the ctx's
 *    HIP stream (zkhip_set_stream) and is asynchronous unless it returns host data.
 *  - There is NO CPU fallback: every entry point needs a gfx950 device and
 *    fails with ZKHIP_ERR_NO_DEVICE otherwise.
 */
#ifndef ZKHIP_H
#define ZKHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKHIP_OK 0
#define ZKHIP_ERR_NO_DEVICE (-1)
#define ZKHIP_ERR_HIP (-2)
#define ZKHIP_ERR_INVALID (-3)
#define ZKHIP_ERR_NOMEM (-4)
#define ZKHIP_ERR_SMALL_BUFFER (-5)
#define ZKHIP_ERR_POW_FAILED (-6)
#define ZKHIP_ERR_VERIFY (-7)
#define ZKHIP_ERR_CONSTRAINT (-8)

#define ZKHIP_DIGEST_WORDS 8

typedef struct zkhip_ctx zkhip_ctx;

/* library / ABI version: (major<<16)|minor */
uint32_t zkhip_version(void);

/* ---- context: replaces `<Engine as StarkEngine>::new(params)` device setup
 *      (crates/prover/src/prover/mod.rs:209) and the CUDA backend's stream/pool ---- */
int zkhip_ctx_create(int device, zkhip_ctx **out);
void zkhip_ctx_destroy(zkhip_ctx *ctx);
const char *zkhip_last_error(const zkhip_ctx *ctx);
/* A context issues all its work on ONE stream: its own (created with the context, non-blocking -- several contexts on one GPU
 * overlap) unless the caller hands in another hipStream_t here (e.g. the stream that produced the trace); NULL = the legacy
 * default stream.  Synchronises with the stream used so far. */
int zkhip_set_stream(zkhip_ctx *ctx, void *hip_stream);
int zkhip_sync(zkhip_ctx *ctx);
/* Trace commit of zkhip_prove as a pipeline of `parts` column blocks (2..8; 0 or 1 = off, the default unless the
 * environment sets ZKHIP_COMMIT_PARTS): the coset LDE of block k+1 runs on a second stream of the context while the row
 * sponge absorbs block k.  Shortens the latency of ONE proof on an otherwise idle GPU (the LDE is memory-bound, the
 * sponge VALU-bound); with several proofs in flight the same overlap already happens across proofs.  Proof bytes are
 * identical either way. */
int zkhip_set_commit_pipeline(zkhip_ctx *ctx, unsigned parts);
/* CU partition of that pipeline: with side_cus > 0 the LDE stream is created with a CU mask of `side_cus` compute units
 * (hipExtStreamCreateWithCUMask; an even slice of every XCD) and the pipeline's row sponge runs on a stream masked to the
 * remaining CUs, so the memory-bound LDE never queues behind hash workgroups.  0 = unmasked streams (the default unless the
 * environment sets ZKHIP_SIDE_CUS).  Measured effect: DESIGN.md section 5.  Proof bytes are identical either way. */
int zkhip_set_cu_partition(zkhip_ctx *ctx, unsigned side_cus);

/* ---- device memory (replaces openvm-cuda-common DeviceBuffer / VPMM pool, AGENTS.md:136) ---- */
int zkhip_malloc(zkhip_ctx *ctx, size_t bytes, void **dptr);
int zkhip_free(zkhip_ctx *ctx, void *dptr);
int zkhip_h2d(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes);
int zkhip_d2h(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes); /* synchronises */
/* Pinned host staging: zkhip_host_alloc returns page-locked memory; zkhip_h2d_async enqueues the copy on the context's stream and
 * returns at once -- the source must be pinned and stay untouched until work enqueued after it has completed (any synchronising
 * call: zkhip_d2h, zkhip_prove, ...).  zkhip_zero clears device memory in stream order.  (A caller that uploads many record
 * arrays per proof -- include/zkhip_vm_prover.hpp -- stages them in one pinned buffer instead of one synchronous pageable copy
 * each.) */
int zkhip_host_alloc(zkhip_ctx *ctx, size_t bytes, void **hptr);
int zkhip_host_free(zkhip_ctx *ctx, void *hptr);
int zkhip_h2d_async(zkhip_ctx *ctx, void *dst, const void *src_pinned, size_t bytes);
int zkhip_zero(zkhip_ctx *ctx, void *dptr, size_t bytes);
/* in-place canonical <-> Montgomery on a device buffer of n words */
int zkhip_to_monty(zkhip_ctx *ctx, uint32_t *d, size_t n);
int zkhip_from_monty(zkhip_ctx *ctx, uint32_t *d, size_t n);

/* ---- NTT / LDE (K1; replaces p3-dft dft_batch / coset_lde_batch as used by the
 *      engine's trace commit) ---- */
/* In-place DFT of every column of a height-2^log_n matrix.  Natural order in;
 * natural order out unless bitrev_out != 0 (then row r holds index bitrev(r)).
 * inverse != 0 computes the inverse DFT (with the 1/n factor). */
int zkhip_ntt_batch(zkhip_ctx *ctx, uint32_t *d_mat, unsigned log_n, size_t width, size_t stride,
                    int inverse, int bitrev_out);
/* Coset low-degree extension: evaluations over H (natural order, height 2^log_n)
 * -> evaluations over shift*K, |K| = 2^(log_n+added_bits), rows in BIT-REVERSED
 * order (the committed-LDE layout).  d_in is preserved.  shift is canonical. */
int zkhip_lde_batch(zkhip_ctx *ctx, const uint32_t *d_in, size_t in_stride, uint32_t *d_out,
                    size_t out_stride, unsigned log_n, unsigned added_bits, size_t width,
                    uint32_t shift);

/* ---- Poseidon2 (K2/K3) ---- */
/* permutes n independent 16-word states, laid out state-major [n][16] */
int zkhip_poseidon2_permute_batch(zkhip_ctx *ctx, uint32_t *d_states, size_t n);

/* ---- device-side trace generation (the reference's GPU backend fills chip traces on the
 *      device, AGENTS.md:183-187) ---- */
/* Trace of the Poseidon2 AIR (one permutation per row; the structure of p3-poseidon2-air 0.4.3 as OpenVM instantiates
 * it: one committed x^3 register per S-box, every constraint of degree <= 3; Cargo.lock p3-poseidon2-air /
 * openvm-poseidon2-air).  Columns: inputs[16] | 4 x {sbox[16], post[16]} | 13 x {sbox, post_sbox} | 4 x {sbox[16],
 * post[16]}; the last 16 columns are the permutation's output.  d_inputs: device, [n_perms][16] Montgomery;
 * d_trace: device, ZKHIP_POSEIDON2_AIR_WIDTH columns with stride 1<<log_height, Montgomery -- the layout zkhip_prove
 * takes.  Rows >= n_perms hold the permutation of the zero state (valid rows).  The matching constraint program is
 * zkvm-prover_amd/air.py poseidon2_air(). */
#define ZKHIP_POSEIDON2_AIR_WIDTH 298
int zkhip_poseidon2_air_tracegen(zkhip_ctx *ctx, const uint32_t *d_inputs, size_t n_perms, unsigned log_height,
                                 uint32_t *d_trace);

/* Multiplicity column of a range-check / lookup table -- the whole trace of OpenVM's VariableRangeChecker-style
 * chips (their tracegen counts requests with atomics while the other chips fill their rows).
 * d_counts[v] = #{ i < n : d_values[i] == v } for v < 2^log_table; d_values: n Montgomery words on the device (e.g. a
 * trace column that is sent to the table's bus); d_counts: 2^log_table Montgomery words = the table chip's trace.
 * accumulate != 0 adds to the counts already there (one call per requesting column).  Synchronises; returns
 * ZKHIP_ERR_INVALID (and leaves the in-range counts) if a value lies outside the table. */
int zkhip_range_counts_tracegen(zkhip_ctx *ctx, const uint32_t *d_values, size_t n, unsigned log_table,
                                uint32_t *d_counts, int accumulate);

/* Multiplicity column of a range-TUPLE table (OpenVM RangeTupleCheckerChip<2>; the chunk circuit's rv32m / bigint
 * extensions use sizes [256, 8192], crates/circuits/chunk-circuit/openvm.toml `range_tuple_checker_sizes`): d_counts[x * size_y
 * + y] = #{ i < n : (d_x[i], d_y[i]) == (x, y) }.  d_x / d_y: n Montgomery words each (requesting trace columns);
 * d_counts: size_x * size_y Montgomery words (a power of two <= 2^27) = the chip's whole trace.  Synchronises;
 * ZKHIP_ERR_INVALID if a request lies outside the table. */
int zkhip_range_tuple_counts_tracegen(zkhip_ctx *ctx, const uint32_t *d_x, const uint32_t *d_y, size_t n, uint32_t size_x,
                                      uint32_t size_y, uint32_t *d_counts, int accumulate);
/* The two multiplicity columns of a bitwise-operation lookup table (OpenVM BitwiseOperationLookupChip<num_bits>, 8 in the
 * reference's circuits): rows = all (x, y) with x, y < 2^num_bits, row (x << num_bits) + y; column 0 counts the range requests
 * (d_op[i] == 0), column 1 the XOR requests (d_op[i] == 1; the requester's z = x ^ y is what the table's preprocessed column holds).
 * d_x, d_y, d_op: n Montgomery words; d_trace: 2 columns of 2^(2 num_bits) Montgomery words. */
int zkhip_bitwise_lookup_tracegen(zkhip_ctx *ctx, const uint32_t *d_x, const uint32_t *d_y, const uint32_t *d_op, size_t n,
                                  unsigned num_bits, uint32_t *d_trace, int accumulate);
/* Trace of a volatile memory boundary chip (OpenVM VolatileBoundaryChip): one row per touched address, SORTED by
 * (address space, pointer) on the device (radix sort), then zero rows up to 2^log_height.  Columns
 * (ZKHIP_MEMORY_BOUNDARY_WIDTH = 8, stride 2^log_height, Montgomery): address space, pointer, initial data, final data,
 * final timestamp, is_valid, gap_lo, gap_hi -- (key_next - key - 1) split at 16 bits with key = as * 2^pointer_bits + pointer,
 * the sortedness witness the range checker receives (0 on the last valid row).  d_addr_space / d_pointer / d_timestamp:
 * n plain integers (addr_space < 2^as_bits, pointer < 2^pointer_bits); d_initial / d_final: n Montgomery words.
 * ZKHIP_ERR_INVALID for out-of-range or duplicate addresses or n > 2^log_height. */
#define ZKHIP_MEMORY_BOUNDARY_WIDTH 8
int zkhip_memory_boundary_tracegen(zkhip_ctx *ctx, const uint32_t *d_addr_space, const uint32_t *d_pointer, const uint32_t *d_initial,
                                   const uint32_t *d_final, const uint32_t *d_timestamp, size_t n, unsigned as_bits,
                                   unsigned pointer_bits, unsigned log_height, uint32_t *d_trace);

/* RV32 less-than core (rv32im LessThanCoreAir<4, 8>: SLT / SLTU).  Record i = (d_opcode[i]: 0 = SLT signed, 1 = SLTU unsigned;
 * d_b[i], d_c[i]) -- plain integers.  Fills d_trace (ZKHIP_RV32_LT_WIDTH = 18 columns, stride 2^log_height, Montgomery:
 * b[4] | c[4] | cmp | is_slt is_sltu | b_msb_f c_msb_f | marker[4] | diff_val; rows >= n zero) and adds the row's two RANGE
 * requests -- (b_msb_f + 128 is_slt, c_msb_f + 128 is_slt) and, when the operands differ, (diff_val - 1, 0) -- to the range column
 * (column 0) of d_bitwise_trace, the 2 x 2^16 trace of the 8-bit bitwise-operation lookup table.  AIR: air.py rv32_lt_core_air(). */
#define ZKHIP_RV32_LT_WIDTH 18
int zkhip_rv32_lt_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_b, const uint32_t *d_c, size_t n,
                           unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);

/* RV32 shift core (rv32im ShiftCoreAir<4, 8>: SLL / SRL / SRA).  Record i = (d_opcode[i]: 0 = SLL, 1 = SRL, 2 = SRA; d_b[i] the
 * value; d_c[i] the shift operand, of which the low five bits count) -- plain integers.  Fills d_trace (ZKHIP_RV32_SHIFT_WIDTH = 32
 * columns, stride 2^log_height, Montgomery: a[4] | b[4] | c0 | is_sll is_srl is_sra | bit_marker[8] | limb_marker[4] | carry[4] |
 * sign | q | mult_left | mult_right; rows >= n zero) and adds the row's lookup requests to d_bitwise_trace (2 x 2^16): seven range
 * pairs -- (carry_i, 2^bit_shift - 1 - carry_i) x 4, (a0, a1), (a2, a3), (q, 32 q) -- and for SRA the XOR request (b3, 128).
 * AIR: air.py rv32_shift_core_air(). */
#define ZKHIP_RV32_SHIFT_WIDTH 32
int zkhip_rv32_shift_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_b, const uint32_t *d_c, size_t n,
                              unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);

/* RV32 branch-equal core (rv32im BranchEqualCoreAir<4>: BEQ / BNE).  Record i = (d_opcode[i]: 0 = BEQ, 1 = BNE; d_a[i], d_b[i] the
 * operands; d_imm[i] the branch offset as a canonical field element, p - |offset| for a backward branch) -- plain integers.  Fills
 * d_trace (ZKHIP_RV32_BRANCH_EQ_WIDTH = 17 columns, stride 2^log_height, Montgomery: a[4] | b[4] | taken | imm | is_beq is_bne |
 * diff_inv_marker[4] | pc_inc; rows >= n zero); the marker of the first differing limb pair holds (a_i - b_i)^-1, inverted on the
 * device.  AIR: air.py rv32_branch_eq_core_air() (no bus interactions). */
#define ZKHIP_RV32_BRANCH_EQ_WIDTH 17
int zkhip_rv32_branch_eq_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_a, const uint32_t *d_b, const uint32_t *d_imm,
                                  size_t n, unsigned log_height, uint32_t *d_trace);

/* RV32 branch-less-than core (rv32im BranchLessThanCoreAir<4, 8>: BLT / BLTU / BGE / BGEU).  Record i = (d_opcode[i]: 0 = BLT, 1 = BLTU,
 * 2 = BGE, 3 = BGEU; operands d_a[i], d_b[i]; d_imm[i] the offset as a canonical field element) -- plain integers.  Fills d_trace
 * (ZKHIP_RV32_BRANCH_LT_WIDTH = 23 columns, stride 2^log_height, Montgomery: a[4] | b[4] | cmp_lt | taken | imm | 4 opcode flags |
 * a_msb_f b_msb_f | marker[4] | diff_val | pc_inc; rows >= n zero) and adds the two range requests of the comparison (as
 * zkhip_rv32_lt_tracegen) to column 0 of d_bitwise_trace.  AIR: air.py rv32_branch_lt_core_air(). */
#define ZKHIP_RV32_BRANCH_LT_WIDTH 23
int zkhip_rv32_branch_lt_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_a, const uint32_t *d_b, const uint32_t *d_imm,
                                  size_t n, unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);

/* RV32 JAL / LUI core (rv32im Rv32JalLuiCoreAir).  Record i = (d_opcode[i]: 0 = JAL, 1 = LUI; d_pc[i] the instruction's pc, < 2^30 - 4 for
 * JAL; d_imm[i]: JAL's offset as a canonical field element, LUI's 20-bit immediate) -- plain integers.  Fills d_trace
 * (ZKHIP_RV32_JAL_LUI_WIDTH = 9 columns, stride 2^log_height, Montgomery: pc | imm | rd[4] | is_jal is_lui | pc_inc; rows >= n zero) and
 * adds the rows' range requests ((rd0, rd1), (rd2, rd3), JAL: (4 rd3, 0)) to column 0 of d_bitwise_trace.  AIR: air.py
 * rv32_jal_lui_core_air().  ZKHIP_ERR_INVALID for a record outside those ranges. */
#define ZKHIP_RV32_JAL_LUI_WIDTH 9
int zkhip_rv32_jal_lui_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_pc, const uint32_t *d_imm, size_t n,
                                unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);

/* RV32 AUIPC core (rv32im Rv32AuipcCoreAir).  Record i = (d_pc[i] < 2^30, d_imm[i] the 20-bit immediate).  ZKHIP_RV32_AUIPC_WIDTH = 14
 * columns: pc | imm | pc_limb[4] | imm_limb[3] (bytes of 16 imm) | rd[4] = pc + (imm << 12) mod 2^32 | is_valid; five range requests
 * per row.  AIR: air.py rv32_auipc_core_air(). */
#define ZKHIP_RV32_AUIPC_WIDTH 14
int zkhip_rv32_auipc_tracegen(zkhip_ctx *ctx, const uint32_t *d_pc, const uint32_t *d_imm, size_t n, unsigned log_height, uint32_t *d_trace,
                              uint32_t *d_bitwise_trace);

/* RV32 JALR core (rv32im Rv32JalrCoreAir).  Record i = (d_pc[i] < 2^30 - 4, d_rs1[i], d_imm[i] the raw 12-bit immediate).
 * ZKHIP_RV32_JALR_WIDTH = 20 columns: pc | imm | imm_limb[2] | imm_sign | rs1[4] | rd[4] = pc + 4 | t[4] = rs1 + sext(imm) mod 2^32 | lsb of
 * t | to_pc = t - lsb (must be < 2^30: the top limb is range-checked as 4 t_3) | is_valid; five range requests per row.  AIR: air.py rv32_jalr_core_air(). */
#define ZKHIP_RV32_JALR_WIDTH 20
int zkhip_rv32_jalr_tracegen(zkhip_ctx *ctx, const uint32_t *d_pc, const uint32_t *d_rs1, const uint32_t *d_imm, size_t n, unsigned log_height,
                             uint32_t *d_trace, uint32_t *d_bitwise_trace);

/* RV32 high-multiplication core (rv32im MulHCoreAir<4, 8>: MULH / MULHSU / MULHU).  Record i = (d_opcode[i]: 0 = MULH, 1 = MULHSU,
 * 2 = MULHU; operands d_b[i], d_c[i]).  Fills d_trace (ZKHIP_RV32_MULH_WIDTH = 21 columns, stride 2^log_height, Montgomery: a[4] = high
 * word | b[4] | c[4] | a_mul[4] = low word | b_sign c_sign | 3 opcode flags; rows >= n zero), adds the eight (limb, carry) requests of
 * every row to d_tuple_counts (the range-tuple table's trace, size_x >= 256, size_y >= 2048) and the sign requests to column 0 of
 * d_bitwise_trace.  AIR: air.py rv32_mulh_core_air(). */
#define ZKHIP_RV32_MULH_WIDTH 21
int zkhip_rv32_mulh_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_b, const uint32_t *d_c, size_t n, unsigned log_height,
                             uint32_t *d_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y, uint32_t *d_bitwise_trace);

/* RV32 load/store cores (rv32im LoadStoreCoreAir<4> and LoadSignExtendCoreAir<4, 8> in one chip).  Record i = (d_case[i] in 0..19:
 * LW, LHU@0 LHU@2, LBU@0..3, SW, SH@0 SH@2, SB@0..3, LH@0 LH@2, LB@0..3 -- @k = byte offset inside the aligned word; d_read[i]: the
 * aligned memory word for loads, the source register for stores; d_prev[i]: what the destination (register / aligned word) held).
 * Fills d_trace (ZKHIP_RV32_LOADSTORE_WIDTH = 33 columns, stride 2^log_height, Montgomery: read[4] | prev[4] | write[4] | case flag[20] |
 * sign; rows >= n zero) and adds the sign request of LH / LB rows to column 0 of d_bitwise_trace.  AIR: air.py
 * rv32_loadstore_core_air(). */
#define ZKHIP_RV32_LOADSTORE_WIDTH 33
int zkhip_rv32_loadstore_tracegen(zkhip_ctx *ctx, const uint32_t *d_case, const uint32_t *d_read, const uint32_t *d_prev, size_t n,
                                  unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);

/* RV32 division core (the job of rv32im DivRemCoreAir<4, 8>: DIV / DIVU / REM / REMU).  Record i = (d_opcode[i]: 0 = DIV, 1 = DIVU, 2 = REM,
 * 3 = REMU; dividend d_b[i]; divisor d_c[i]).  Fills d_trace (ZKHIP_RV32_DIVREM_WIDTH = 41 columns, stride 2^log_height, Montgomery:
 * b[4] | c[4] | q[4] | r[4] | c_abs[4] | r_abs[4] | b_sign c_sign q_sign r_sign | k_c k_r | zero_divisor c_sum_inv | marker[4] | diff | 4 opcode
 * flags; rows >= n zero), adds the eight (limb, carry) requests of every row to d_tuple_counts (size_x >= 256, size_y >= 2048) and
 * its range requests to column 0 of d_bitwise_trace.  RISC-V's exceptional results (division by zero, -2^31 / -1) included.
 * AIR: air.py rv32_divrem_core_air(). */
#define ZKHIP_RV32_DIVREM_WIDTH 41
int zkhip_rv32_divrem_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_b, const uint32_t *d_c, size_t n, unsigned log_height,
                               uint32_t *d_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y, uint32_t *d_bitwise_trace);

/* MMCS path chip: in-circuit verification of mixed-height Merkle openings, a piece of the recursion circuit the reference's
 * aggregation nodes run.  Path p = (leaf digest d_leaf[8 p ..], index d_index[p], steps d_path_start[p] ..
 * d_path_start[p + 1] bottom-up: d_step_kind 0 = sibling digest (consumes the next index bit, least significant first), 1 = injected
 * row digest of the shorter matrices joining there; d_step_digest[8 s ..]) -- canonical words.  Fills d_trace
 * (ZKHIP_MMCS_PATH_WIDTH = 39 columns, stride 2^log_height, Montgomery: root[8] | parent[8] | a[8] | b[8] | bit | is_inj | is_first | is_last |
 * is_real | idx | lvl; path p occupies rows d_path_start[p] .., written from the root down; other rows zero) and d_hash_inputs
 * ([2^log_height][16] Montgomery: a || b of every row, the input of zkhip_poseidon2_air_tracegen for the Poseidon2 chip that
 * serves the chip's hash bus).  AIR: air.py mmcs_path_air().  ZKHIP_ERR_INVALID if a path does not fit or does not end in a sibling
 * step. */
#define ZKHIP_MMCS_PATH_WIDTH 39
int zkhip_mmcs_path_tracegen(zkhip_ctx *ctx, const uint32_t *d_leaf, const uint32_t *d_index, const uint32_t *d_path_start,
                             const uint32_t *d_step_kind, const uint32_t *d_step_digest, size_t n_paths, unsigned log_height, uint32_t *d_trace,
                             uint32_t *d_hash_inputs);

/* NATIVE chips -- the arithmetic of the recursion programs the aggregation circuits run (native FieldArithmeticCoreAir,
 * FieldExtensionCoreAir).  Records are canonical field elements: (d_opcode[i]: 0 = ADD, 1 = SUB, 2 = MUL, 3 = DIV; operands).
 * zkhip_field_arith_tracegen: ZKHIP_FIELD_ARITH_WIDTH = 8 columns a = b op c | b | c | 4 opcode flags | divisor_inv.
 * zkhip_field_ext_tracegen: operands are quartic-extension elements, four words per record in d_x / d_y (16-byte aligned);
 * ZKHIP_FIELD_EXT_WIDTH = 20 columns x[4] | y[4] | z[4] = x op y | 4 opcode flags | divisor_inv[4].  Rows >= n zero; Montgomery;
 * ZKHIP_ERR_INVALID on an opcode > 3, an operand >= p or a division by zero.  AIRs: air.py field_arith_air(), field_ext_air(). */
#define ZKHIP_FIELD_ARITH_WIDTH 8
#define ZKHIP_FIELD_EXT_WIDTH 20
int zkhip_field_arith_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_b, const uint32_t *d_c, size_t n, unsigned log_height,
                               uint32_t *d_trace);
int zkhip_field_ext_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_x, const uint32_t *d_y, size_t n, unsigned log_height,
                             uint32_t *d_trace);

/* Variable range checker (OpenVM VariableRangeCheckerChip): the multiplicity column of the table that serves every check
 * value < 2^bits with bits <= max_bits (2^(max_bits + 1) rows: row 2^bits - 1 + value; air.py var_range_table_air / var_range_prep).
 * d_values: a requesting trace column (Montgomery); d_bits: the column of the bit counts, or NULL to use const_bits for every
 * request.  accumulate as in zkhip_range_counts_tracegen.  ZKHIP_ERR_INVALID for a request outside the table. */
int zkhip_var_range_counts_tracegen(zkhip_ctx *ctx, const uint32_t *d_values, const uint32_t *d_bits, uint32_t const_bits, size_t n, unsigned max_bits,
                                    uint32_t *d_counts, int accumulate);

/* Native CASTF core (native CastFCoreAir): record i = d_x[i] < 2^30 (plain integer).  Fills d_trace (ZKHIP_CASTF_WIDTH = 6 columns,
 * stride 2^log_height, Montgomery: x | limb[4] of 8, 8, 8, 6 bits | is_valid; rows >= n zero) and adds the four limb checks of
 * every row to d_var_range_counts (the variable range checker's trace for max_bits >= 8).  AIR: air.py castf_air(). */
#define ZKHIP_CASTF_WIDTH 6
int zkhip_castf_tracegen(zkhip_ctx *ctx, const uint32_t *d_x, size_t n, unsigned log_height, uint32_t *d_trace, uint32_t *d_var_range_counts,
                         unsigned max_bits);

/* FRI fold chip (a piece of the recursion circuit, air.py fri_fold_air): record i = (d_e0[4 i ..], d_e1[4 i ..], d_beta[4 i ..] canonical
 * extension elements; pair index d_k[i] < 2^d_log_n_out[i]; d_log_n_out[i] <= 26).  Fills d_trace (ZKHIP_FRI_FOLD_WIDTH = 19 columns,
 * stride 2^log_height, Montgomery: e0[4] | e1[4] | beta[4] | x_inv | folded[4] | is_real | k) with x = g^bitrev(k) in the subgroup of order
 * 2^(log_n_out + 1) and folded = (e0 + e1) / 2 + beta (e0 - e1) x_inv / 2 (p3 TwoAdicFriFolding::fold_row, arity 2). */
#define ZKHIP_FRI_FOLD_WIDTH 19
int zkhip_fri_fold_chip_tracegen(zkhip_ctx *ctx, const uint32_t *d_e0, const uint32_t *d_e1, const uint32_t *d_beta, const uint32_t *d_k,
                                 const uint32_t *d_log_n_out, size_t n, unsigned log_height, uint32_t *d_trace);

/* Transcript chip (a piece of the recursion circuit, air.py duplex_air: the DuplexChallenger in-circuit).  Record r = one
 * duplexing: d_n_observed[r] <= 8 values d_observed[8 r ..] (canonical) overwrite the first rate lanes of the running state, the
 * state is permuted, d_n_sampled[r] <= 8 output lanes are popped from the end.  Fills d_trace (ZKHIP_DUPLEX_WIDTH = 50 columns, stride
 * 2^log_height, Montgomery: st_in[16] | st_out[16] | f[8] | s[8] | seq | is_real; rows >= n zero) and d_hash_inputs
 * ([2^log_height][16]: st_in of every row, the input of zkhip_poseidon2_air_tracegen for the chip serving the 32-field hash bus). */
#define ZKHIP_DUPLEX_WIDTH 50
int zkhip_duplex_tracegen(zkhip_ctx *ctx, const uint32_t *d_n_observed, const uint32_t *d_observed, const uint32_t *d_n_sampled, size_t n,
                          unsigned log_height, uint32_t *d_trace, uint32_t *d_hash_inputs);

/* Domain-point chip (air.py domain_point_air): row i = pair index d_k[i] (< 2^26), its bits, the running product that ends in
 * x^-1 for x = g^bitrev(k) (the same per-bit constants for a layer of any size), and the multiplicity d_mult[i] with which rows of
 * the FRI fold chip use the pair.  ZKHIP_DOMAIN_POINT_WIDTH = 54 columns: k | bit[26] | acc[26] | mult; rows >= n: k = 0, mult = 0. */
#define ZKHIP_DOMAIN_POINT_BITS 26
#define ZKHIP_DOMAIN_POINT_WIDTH 54
int zkhip_domain_point_tracegen(zkhip_ctx *ctx, const uint32_t *d_k, const uint32_t *d_mult, size_t n, unsigned log_height, uint32_t *d_trace);

/* System chips: the PROGRAM chip and the execution frames that look instructions up in it.  OpenVM's ProgramAir keeps the program
 * (ZKHIP_PROGRAM_FIELDS = 9 fields per instruction: pc, opcode, operands a..g) as a CACHED main partition and one common column,
 * the execution frequency of each instruction; it receives every instruction that often on the program bus (the first AIR of the
 * reference's stored proofs has exactly this shape: cached width 9, common width 1).  From the list of executed instruction
 * indices (d_pc_index: n plain integers, row numbers of the program):
 *   zkhip_program_freq_tracegen  the frequency column: 2^log_height Montgomery words, the chip's whole common trace;
 *   zkhip_exec_frame_tracegen    one row per executed instruction, [9 program fields | is_valid] (ZKHIP_PROGRAM_FIELDS + 1 columns,
 *                                stride 2^log_height, rows >= n zero), gathered from d_program (9 columns of n_program Montgomery
 *                                words, column-major: the program chip's cached partition as uploaded) -- the rows that SEND the
 *                                instruction on the program bus.
 * ZKHIP_ERR_INVALID if an index is not a row of the program.  AIRs: air.py program_air() / exec_frame_air(). */
#define ZKHIP_PROGRAM_FIELDS 9
int zkhip_program_freq_tracegen(zkhip_ctx *ctx, const uint32_t *d_pc_index, size_t n, unsigned log_height, uint32_t *d_freq);
int zkhip_exec_frame_tracegen(zkhip_ctx *ctx, const uint32_t *d_pc_index, size_t n, const uint32_t *d_program, size_t n_program,
                              unsigned log_height, uint32_t *d_trace);

/* Per-access rows of the offline memory-checking argument (OpenVM's memory bus; the access adapters of its chips): record i =
 * one access of a 16-bit memory cell -- (d_addr_space[i], d_pointer[i]) the cell, (d_prev_data[i], d_prev_ts[i]) the state the
 * previous access left (the initial state with timestamp 0 for the first), (d_data[i], d_ts[i]) the state this access leaves,
 * d_is_read[i] = 1 for reads (data == prev_data) -- plain integers, as include/zkhip_vm.hpp ExecRecords::acc_* holds them.  Fills
 * d_trace (ZKHIP_MEMORY_ACCESS_WIDTH = 10 columns, stride 2^log_height, Montgomery): as | ptr | prev_data | prev_ts | data | ts |
 * is_read | is_valid | gap_lo | gap_hi with ts - prev_ts - 1 = gap_lo + 2^16 gap_hi; rows >= n zero.  With the boundary chip
 * (zkhip_memory_boundary_tracegen over the touched cells) the memory bus balances exactly when the log is a consistent history.
 * ZKHIP_ERR_INVALID for a value above 16 bits, ts <= prev_ts, or a read that changes its cell.  AIR: air.py memory_access_air(). */
#define ZKHIP_MEMORY_ACCESS_WIDTH 10
int zkhip_memory_access_tracegen(zkhip_ctx *ctx, const uint32_t *d_addr_space, const uint32_t *d_pointer, const uint32_t *d_prev_data,
                                 const uint32_t *d_prev_ts, const uint32_t *d_data, const uint32_t *d_ts, const uint32_t *d_is_read, size_t n,
                                 unsigned log_height, uint32_t *d_trace);

/* Trace of an INSTRUCTION chip from execution records: the core of OpenVM's RV32 base ALU chip (rv32im BaseAluCoreAir: ADD, SUB,
 * XOR, OR, AND on 4 x 8-bit limbs; the `rv32i` extension of crates/circuits/chunk-circuit/openvm.toml).  Record i = (d_opcode[i] in
 * 0..4 = add, sub, xor, or, and; d_b[i], d_c[i]: the 32-bit operands) -- plain integers.  Fills d_trace (ZKHIP_RV32_ALU_WIDTH = 18
 * columns, stride 2^log_height, Montgomery: a[4] | b[4] | c[4] | 5 opcode flags | is_valid; rows >= n zero) and, in the same pass,
 * adds the bitwise-lookup requests of every row (4 per record: (b_i, c_i) for the bitwise opcodes, (a_i, a_i) for ADD / SUB) to the
 * XOR multiplicity column of d_bitwise_trace, the 2 x 2^16 trace of the 8-bit bitwise-operation lookup table
 * (zkhip_bitwise_lookup_tracegen's layout; zero it first or pass a table other chips already counted into). */
#define ZKHIP_RV32_ALU_WIDTH 18
int zkhip_rv32_alu_tracegen(zkhip_ctx *ctx, const uint32_t *d_opcode, const uint32_t *d_b, const uint32_t *d_c, size_t n,
                            unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);

/* RV32 multiplication core (OpenVM rv32im MultiplicationCoreAir; the `rv32m` extension of the chunk circuit): record i = operands
 * (d_b[i], d_c[i]), plain integers.  d_trace: ZKHIP_RV32_MUL_WIDTH = 13 columns (a[4] = low word of b * c | b[4] | c[4] | is_valid),
 * stride 2^log_height, Montgomery, rows >= n zero.  The four (result limb, carry) pairs of every record are added to
 * d_tuple_counts, the multiplicity column of the range-tuple table (size_x x size_y Montgomery words, row limb * size_y + carry;
 * the reference's sizes are [256, 8192]). */
#define ZKHIP_RV32_MUL_WIDTH 13
int zkhip_rv32_mul_tracegen(zkhip_ctx *ctx, const uint32_t *d_b, const uint32_t *d_c, size_t n, unsigned log_height, uint32_t *d_trace,
                            uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);

typedef struct {
    const uint32_t *data; /* device, column-major, Montgomery */
    size_t stride;
    unsigned log_height;
    size_t width;
} zkhip_matrix;

typedef struct zkhip_tree zkhip_tree;
/* Merkle-Poseidon2 commitment of matrices of (possibly) mixed power-of-two
 * heights; the matrices must stay alive as long as the tree.  root_out is a
 * HOST buffer of 8 canonical words (synchronises); pass NULL to stay asynchronous. */
int zkhip_merkle_commit(zkhip_ctx *ctx, const zkhip_matrix *mats, size_t n_mats,
                        zkhip_tree **tree, uint32_t *root_out);
/* Rebuilds a committed tree in place from its matrices' current contents: the commit's launches on the same digest store; no allocation,
 * asynchronous on the context's stream.  (What a prover does proof after proof on its key's workspace; tools/merkle_stress.cpp.) */
int zkhip_merkle_rebuild(zkhip_ctx *ctx, zkhip_tree *tree);
/* Diagnosis: every plain layer (no injected matrices) against the compression of its stored children, recomputed on the device through
 * the plain one-lane-per-node permutation.  *n_bad = differing nodes, *first (may be NULL) = layer << 24 | index of the first one.
 * Synchronises.  The check zkhip_config.self_check runs after every proof. */
int zkhip_tree_check(zkhip_ctx *ctx, const zkhip_tree *tree, uint32_t *n_bad, uint32_t *first);
/* device pointer to the Montgomery root digest (8 words) */
const uint32_t *zkhip_tree_root_device(const zkhip_tree *tree);
unsigned zkhip_tree_log_height(const zkhip_tree *tree);
/* copies digest layer `layer` (0 = leaves) to host, canonical; out needs 8<<(log_height-layer) words */
int zkhip_tree_layer(zkhip_ctx *ctx, const zkhip_tree *tree, unsigned layer, uint32_t *out);
/* words one opening occupies: sum(widths) + 8*log_height */
size_t zkhip_merkle_opening_words(const zkhip_tree *tree);
/* K4: opens n leaf indices; out (HOST, canonical) receives n openings of
 * zkhip_merkle_opening_words() each: opened rows (matrix order) then sibling digests bottom-up */
int zkhip_merkle_open(zkhip_ctx *ctx, const zkhip_tree *tree, const uint64_t *indices, size_t n,
                      uint32_t *out, size_t cap_words);
void zkhip_tree_destroy(zkhip_ctx *ctx, zkhip_tree *tree);

/* ---- constraint evaluation (K5) on its own: quotient values q(x) = (sum_i alpha^(n-1-i) c_i(x)) / Z_H(x) of ONE AIR on its LDE
 *      domain (what the engine's quotient / constraint prover computes per chip).  program: AIR bytecode (host);
 *      d_lde: the committed LDE of the main trace, `width` columns of 2^(log_height+log_blowup) rows in the committed (bit-reversed,
 *      coset shift 31) layout, Montgomery; pvs: host canonical; alpha: 4 canonical words; d_q: 4 columns (extension coordinates) of
 *      2^(log_height+log_blowup) Montgomery words, same row order.  AIRs with buses / preprocessed traces go through zkhip_prove. ---- */
int zkhip_constraint_eval(zkhip_ctx *ctx, const uint32_t *program, size_t program_len, unsigned log_height, unsigned log_blowup,
                          size_t width, const uint32_t *d_lde, const uint32_t *pvs, size_t n_pvs, const uint32_t alpha[4], uint32_t *d_q);

/* ---- FRI fold (K8): d_in = 2n extension elements in bit-reversed domain order,
 *      d_out = n; beta = 4 canonical words (host) ---- */
int zkhip_fri_fold(zkhip_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, unsigned log_n_out,
                   const uint32_t beta[4]);

/* ---- LogUp and sum-check building blocks (K6, K7 of SURVEY.md 2.3; a7.4 of 8(a)): the
 *      protocol-independent kernels under the pinned backend's LogUp / zero-check / stacking
 *      sum-checks.  All buffers are device, Montgomery; extension elements are 4 words. ---- */
/* out[i] = 1 / in[i] for n extension elements (Montgomery batch inversion); a zero element gives 0 (= 0^(p^4-2), what
 * an element-wise Fermat inversion returns) and does not disturb its neighbours */
int zkhip_ext_batch_inverse(zkhip_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, size_t n);
/* LogUp running sum: out[i] = sum_{j<=i} num[j] / den[j]; den = n extension elements, num = n
 * base-field multiplicities; total (the cumulative sum, 4 canonical words) goes to the HOST
 * buffer total_out when it is not NULL (synchronises) */
int zkhip_logup_running_sum(zkhip_ctx *ctx, const uint32_t *d_den, const uint32_t *d_num, size_t n,
                            uint32_t *d_out, uint32_t *total_out);
/* Fix the LOWEST variable of a multilinear table of 2n extension evaluations to r:
 * out[i] = in[2i] + r * (in[2i+1] - in[2i]); r = 4 canonical words (host) */
int zkhip_mle_fold(zkhip_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, size_t n, const uint32_t r[4]);
/* One sum-check round for the product of k multilinear tables (each 2*n_half extension
 * evaluations, lowest variable = adjacent pairs): out[t] = sum_i prod_j (f_j[2i] + t*(f_j[2i+1]-f_j[2i]))
 * for t = 0..k.  out = (k+1)*4 canonical words on the HOST (synchronises).  1 <= k <= 4. */
int zkhip_sumcheck_round(zkhip_ctx *ctx, const uint32_t *const *d_tables, size_t k, size_t n_half,
                         uint32_t *out);

/* ---- transcript (T4) and proof-of-work (K9), device resident ---- */
typedef struct zkhip_transcript zkhip_transcript;
int zkhip_transcript_create(zkhip_ctx *ctx, zkhip_transcript **out);
void zkhip_transcript_destroy(zkhip_ctx *ctx, zkhip_transcript *t);
/* observe n canonical words from host */
int zkhip_transcript_observe(zkhip_ctx *ctx, zkhip_transcript *t, const uint32_t *vals, size_t n);
/* sample n base-field elements (canonical) to host; synchronises */
int zkhip_transcript_sample(zkhip_ctx *ctx, zkhip_transcript *t, uint32_t *out, size_t n);
/* smallest witness such that observe(w); sample_bits(bits) == 0; advances the transcript */
int zkhip_transcript_grind(zkhip_ctx *ctx, zkhip_transcript *t, unsigned bits, uint32_t *witness);

/* ---- STARK proving / verification (a7 of SURVEY.md 8(a)): the `StarkEngine::prove`
 *      replacement.  See DESIGN.md for protocol, AIR bytecode and proof layout. ---- */
#define ZKHIP_MAX_LOG_FINAL_POLY 8
typedef struct {
    uint32_t log_blowup;         /* openvm.toml:2  */
    uint32_t log_final_poly_len; /* openvm.toml:3; <= ZKHIP_MAX_LOG_FINAL_POLY, and no trace shorter than 2^log_final_poly_len rows */
    uint32_t num_queries;        /* openvm.toml:4  */
    uint32_t commit_pow_bits;    /* openvm.toml:5  */
    uint32_t query_pow_bits;     /* openvm.toml:6  */
} zkhip_params;

typedef struct {
    const uint32_t *program; /* host, AIR constraint bytecode */
    size_t program_len;      /* words */
    unsigned log_height;
    size_t width;
    size_t n_pvs;
    /* Preprocessed trace of the AIR (programs with a PREP section; the engine's `preprocessed` columns, fixed at
     * keygen): zkhip_keygen reads `prep_trace` (HOST, canonical u32, column-major, stride 1<<log_height), extends
     * and commits it once and keeps it resident; zkhip_verify reads `prep_commit` (8 canonical words, the
     * verifying-key entry obtained from zkhip_pk_prep_commitment).  NULL when the AIR has none. */
    const uint32_t *prep_trace;
    const uint32_t *prep_commit;
} zkhip_air;

typedef struct zkhip_pk zkhip_pk;
/* keygen: validates the programs, uploads them, plans kernels/workspace/proof layout */
int zkhip_keygen(zkhip_ctx *ctx, const zkhip_params *params, const zkhip_air *airs, size_t n_airs,
                 zkhip_pk **out);
void zkhip_pk_destroy(zkhip_ctx *ctx, zkhip_pk *pk);
/* commitment (8 canonical words) of AIR `air_index`'s preprocessed trace; ZKHIP_ERR_INVALID if it has none */
int zkhip_pk_prep_commitment(zkhip_ctx *ctx, const zkhip_pk *pk, size_t air_index, uint32_t out[8]);
/* exact proof size in bytes for this key (FRI proofs are shape-static) */
size_t zkhip_proof_size(const zkhip_pk *pk);
/* bytes of device memory the key keeps resident (LDEs, trees, FRI layers, tables): capacity planning for several
 * keys / several proofs in flight on one 288 GB device */
size_t zkhip_pk_workspace_bytes(const zkhip_pk *pk);
/* d_traces[i]: device, column-major Montgomery trace of AIR i with stride 1<<log_height;
 * pvs[i]: host canonical public values.  Proof bytes (canonical LE u32) are written to the
 * HOST buffer `out`.  Runs entirely on the ctx stream; returns after the final D2H. */
int zkhip_prove(zkhip_ctx *ctx, const zkhip_pk *pk, const uint32_t *const *d_traces,
                const uint32_t *const *pvs, uint8_t *out, size_t cap, size_t *out_len);
/* Asynchronous variant for benchmarking / pipelining: leaves the proof in device memory
 * (retrieve with zkhip_proof_fetch after zkhip_sync) */
int zkhip_prove_async(zkhip_ctx *ctx, const zkhip_pk *pk, const uint32_t *const *d_traces,
                      const uint32_t *const *pvs);
int zkhip_proof_fetch(zkhip_ctx *ctx, const zkhip_pk *pk, uint8_t *out, size_t cap, size_t *out_len);
/* Host verifier (replaces Sdk::verify_proof as called at crates/verifier/src/verifier.rs:82
 * for this backend's proofs).  Needs no device. */
int zkhip_verify(const zkhip_params *params, const zkhip_air *airs, size_t n_airs,
                 const uint32_t *const *pvs, const uint8_t *proof, size_t len);
/* the same; on ZKHIP_ERR_VERIFY `*where` (optional) is the line of csrc/verifier.hip whose check refused the proof -- a diagnosis
 * for the prover's own failure messages (which commitment, query or fold), not an interface */
int zkhip_verify_where(const zkhip_params *params, const zkhip_air *airs, size_t n_airs,
                       const uint32_t *const *pvs, const uint8_t *proof, size_t len, int *where);

/* The verifier's primitives on the HOST (no device, canonical words in and out).  They are what p3's permutation,
 * `Mmcs::verify_batch` (MerkleTreeMmcs over PaddingFreeSponge / TruncatedPermutation) and p3-fri's `fold_row` are to
 * `Sdk::verify_proof` (crates/verifier/src/verifier.rs:82), usable on any BabyBear-Poseidon2 v1 proof -- including the
 * reference's own stored proofs (tests/test_ref_vectors_cpu.py pins them to those).
 * zkhip_mmcs_verify: `opening` = the opened rows in matrix order (widths[m] words each) followed by max(log_heights)
 * sibling digests bottom-up; matrix m's row index is index >> (max - log_heights[m]).  ZKHIP_OK / ZKHIP_ERR_VERIFY.
 * zkhip_fri_fold_row: pair `index` of a layer that folds 2^(log_height+1) values (bit-reversed order) to 2^log_height:
 * out = e0 + (beta - x)(e1 - e0)/(-2x), x = two_adic_generator(log_height+1)^bitrev(index, log_height). */
int zkhip_poseidon2_permute_host(uint32_t state[16]);
/* the same permutation through the host's AVX-512 form (zkvm-prover_amd/csrc/poseidon2_avx512.cpp: what the prover's transcript uses for
 * long absorptions); returns 1 -- state untouched -- on a CPU without AVX-512 */
int zkhip_poseidon2_permute_host_avx512(uint32_t state[16]);
/* sixteen independent permutations side by side (states[16 k + w] = word w of state k): one state word of all sixteen per 512-bit
 * register where the CPU has AVX-512 (what the aggregation witness generator advances its queries with), else sixteen scalar calls */
int zkhip_poseidon2_permute16_host(uint32_t states[256]);
int zkhip_mmcs_verify(const uint32_t root[8], const unsigned *log_heights, const size_t *widths, size_t n_mats,
                      uint64_t index, const uint32_t *opening);
int zkhip_fri_fold_row(uint64_t index, unsigned log_height, const uint32_t beta[4], const uint32_t e0[4],
                       const uint32_t e1[4], uint32_t out[4]);
/* the LogUp bus check: n exposed cumulative sums (4 canonical words each, one per AIR with interactions) must add up to zero in
 * F_p[X]/(X^4 - 11); ZKHIP_OK = balanced, ZKHIP_ERR_VERIFY = not */
int zkhip_logup_exposed_check(const uint32_t *exposed, size_t n);

/* Field offsets of a proof, in 32-bit words from its start (the layout is static for given parameters and AIR shapes:
 * DESIGN.md section 4).  The counterpart of `Proof::<SC>::decode_from_bytes` (crates/verifier/src/verifier.rs:62) for this
 * backend's proofs: commitments, opened values, FRI layers and query openings are read in place.  Needs no device. */
typedef struct {
    size_t n_words;      /* whole proof; n_words * 4 == zkhip_proof_size */
    size_t root_main;    /* 8 words: the COMMON main commitment */
    size_t roots_cached; /* n_cached x 8 words (0 = none): the cached main partitions' commitments, AIR order */
    size_t n_cached;
    size_t root_perm;    /* 8 words; 0 = no AIR has bus interactions */
    size_t exposed;      /* n_exposed x 4 words: the LogUp sums of the AIRs with interactions, in AIR order */
    size_t n_exposed;
    size_t root_quot;    /* 8 words */
    size_t opened;       /* n_opened extension elements (4 words each): main, preprocessed, permutation, quotient */
    size_t n_opened;
    size_t fri_layers;   /* n_fri_layers x { root(8), proof-of-work witness(1) } */
    size_t n_fri_layers;
    size_t final_poly;   /* n_final_poly = 2^log_final_poly_len coefficients, 4 words each */
    size_t n_final_poly;
    size_t query_pow;    /* 1 word */
    size_t queries;      /* query i occupies [queries + i * query_words, +query_words) */
    size_t query_words;
    size_t n_queries;
} zkhip_proof_layout;
int zkhip_proof_layout_of(const zkhip_params *params, const zkhip_air *airs, size_t n_airs, zkhip_proof_layout *out);

/* ---- the reference's stored-proof container: OpenVM-v1 `Proof<SC>`, bincode (C++ codec in
 *      include/zkhip_codec.hpp).  Replaces `Proof::<SC>::decode_from_bytes` (crates/verifier/src/verifier.rs:62) /
 *      `encode_to_vec` (crates/prover/src/prover/mod.rs:375-378) for v1-format proofs such as the ones under
 *      crates/verifier/testdata/proofs/.  Host only.  Words inside the container are Montgomery (p3's in-memory form). ---- */
#define ZKHIP_V1_SINGLE 0 /* bincode(Proof<SC>) */
#define ZKHIP_V1_VEC 1    /* bincode(Vec<Proof<SC>>): the `proofs` field of VmInternalStarkProof (crates/types/src/proof.rs:69-74) */
#define ZKHIP_V1_MAX_AIRS 64
typedef struct {
    size_t n_proofs; /* the fields below describe proof 0 */
    size_t n_airs, n_queries, n_fri_layers, n_final_poly;
    size_t n_main_commits, n_after_challenge_commits, n_preprocessed, n_input_batches;
    unsigned log_max_height; /* longest input-batch Merkle path = log2 of the tallest LDE */
    unsigned log_blowup;     /* log_max_height - max log2(degree) */
    int has_logup_pow;
    unsigned log_degree[ZKHIP_V1_MAX_AIRS];
} zkhip_v1_summary;
/* parses and structurally validates (lengths, field elements < p, uniform query shapes); ZKHIP_ERR_VERIFY if malformed */
int zkhip_proof_decode_v1(const uint8_t *bytes, size_t len, int kind, zkhip_v1_summary *out);
/* decode -> encode; for a well-formed input the output equals the input byte for byte */
int zkhip_proof_reencode_v1(const uint8_t *bytes, size_t len, int kind, uint8_t *out, size_t cap, size_t *out_len);
/* a proof of zkhip_prove (params, AIR shapes and public values as given to it) as bincode(Proof<SC>): commitments
 * {main_trace: [main], after_challenge: [permutation]?, quotient}, FRI proof (input batches in the reference's order:
 * preprocessed, main, after-challenge, quotient), opened values, per-AIR data.  Needs commit_pow_bits == 0: the v1 container
 * has no field for per-layer commit-phase witnesses.  *out_len is set even when the buffer is too small. */
int zkhip_proof_to_v1(const zkhip_params *params, const zkhip_air *airs, size_t n_airs, const uint32_t *const *pvs,
                      const uint8_t *proof, size_t len, uint8_t *out, size_t cap, size_t *out_len);
/* the inverse: checks the container against the key's shapes and writes this backend's layout (zkhip_verify's input);
 * pvs_out (may be NULL): per AIR a buffer of n_pvs canonical words */
int zkhip_proof_from_v1(const zkhip_params *params, const zkhip_air *airs, size_t n_airs, const uint8_t *v1, size_t v1_len,
                        uint8_t *out, size_t cap, size_t *out_len, uint32_t *const *pvs_out);

/* Trace generators check their records on the device and report at once: one stream synchronisation per call.  A caller that
 * issues dozens of generators per proof turns the immediate reports off (on = 1): bad-record counts are then summed on the device,
 * and zkhip_tracegen_check synchronises once, returns ZKHIP_ERR_INVALID if anything was refused since the last check, and resets. */
int zkhip_tracegen_defer_checks(zkhip_ctx *ctx, int on);
int zkhip_tracegen_check(zkhip_ctx *ctx);

/* ---- the ONE-STATEMENT VM circuit (include/zkhip_vm_circuit.hpp; SURVEY.md 8(f) f3 adapters / execution bus / connector /
 *      persistent memory -- OpenVM's openvm-circuit and openvm-rv32im-circuit crates, un-vendored; reached by the reference through
 *      sdk.prove, crates/prover/src/prover/mod.rs:355-357).  The AIR set of a segment is defined ONCE, in C++; these entry points
 *      hand its programs, the instruction decode and the program table to any host language. ---- */
size_t zkhip_vm_n_airs(void);
/* AIR `id` < zkhip_vm_n_airs() (order of zkhip::vmc::AirId; the modular extension's chips come from zkhip_vm_modmul_air): program words, width, public-value count; log_height and the preprocessed fields are
 * left zero (heights come from the segment configuration, tables from zkhip_vm_program_table / the lookup tables); *prep_width
 * receives the width of its preprocessed trace.  Pointers stay valid for the life of the process. */
int zkhip_vm_air(unsigned id, zkhip_air *out, size_t *prep_width);
int zkhip_vm_decode(uint32_t word, uint32_t pc, uint32_t out[17], int *legal);
/* preprocessed trace of the program chip: 17 columns x 2^log_program rows, column-major, canonical */
int zkhip_vm_program_table(const uint32_t *words, size_t n_words, uint32_t pc_base, unsigned log_program, uint32_t *out);
/* Frame chip (43 columns): record r = (program row d_pc_index[r], operands d_x / d_y, result d_z, previous value of rd d_rd_prev --
 * plain 32-bit words --, pc step d_pc_inc as a canonical field element, and the timestamps of the previous accesses of the rs1 / rs2 /
 * rd words -- its register adapter; ignored where the instruction skips the access); its timestamp is 1 + 16 r.  d_program: the program
 * table on the device, Montgomery, stride n_program. */
int zkhip_vm_frame_tracegen(zkhip_ctx *ctx, const uint32_t *d_pc_index, const uint32_t *d_x, const uint32_t *d_y, const uint32_t *d_z,
                            const uint32_t *d_rd_prev, const uint32_t *d_pc_inc, const uint32_t *d_prev_ts_rs1, const uint32_t *d_prev_ts_rs2,
                            const uint32_t *d_prev_ts_rd, size_t n, const uint32_t *d_program, size_t n_program, unsigned log_height, uint32_t *d_trace);
/* Load/store chip (48 columns): the core's records (zkhip_rv32_loadstore_tracegen) + per record the instruction's timestamp, base
 * register value, 32-bit immediate and the timestamp of the previous access of the memory word (its memory adapter). */
int zkhip_vm_loadstore_tracegen(zkhip_ctx *ctx, const uint32_t *d_case, const uint32_t *d_read, const uint32_t *d_prev, const uint32_t *d_ts,
                                const uint32_t *d_base, const uint32_t *d_imm, const uint32_t *d_prev_ts, size_t n, unsigned log_height, uint32_t *d_trace,
                                uint32_t *d_bitwise_trace);
/* Keccak-f chip inside the VM (2634 columns): zkhip_keccak_f_tracegen + the timestamp of call p on its 24 rows (d_ts: n_perms plain
 * integers).  Its adapter's rows (vmc::KECCAK_IO_WIDTH) are written by the executor and transposed with zkhip_rows_tracegen. */
int zkhip_vm_keccak_tracegen(zkhip_ctx *ctx, const uint32_t *d_states, const uint32_t *d_ts, size_t n_perms, unsigned log_height, uint32_t *d_trace);
/* SHA-256 compression chip inside the VM (434 columns, 9 preprocessed): zkhip_sha256_tracegen + the timestamp of call b on its 65 rows;
 * zkhip_vm_sha256_prep writes its preprocessed trace (the standalone chip's six columns + input / digest / round-index). */
int zkhip_vm_sha256_tracegen(zkhip_ctx *ctx, const uint32_t *d_blocks, const uint32_t *d_ts, size_t n_blocks, unsigned log_height, uint32_t *d_trace);
int zkhip_vm_sha256_prep(unsigned log_height, uint32_t *out);
/* The two chips of modulus `index` (< 8) of an app with the modular extension: adapter == 0 the multiplication chip inside the VM (326
 * columns: zkhip_modmul_air + a timestamp column + the 24 word receives), adapter != 0 its adapter; zkhip_vm_modmul_tracegen =
 * zkhip_modular_tracegen (17-word records: op | a | b) + the timestamp of call i on row i. */
int zkhip_vm_modmul_air(const uint8_t modulus[32], unsigned index, int adapter, zkhip_air *out);
int zkhip_vm_modmul_tracegen(zkhip_ctx *ctx, const uint32_t modulus[8], const uint32_t *d_records, const uint32_t *d_ts, size_t n, unsigned log_height,
                             uint32_t *d_trace, uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
/* The two chips of curve `index` (< 4) of an app with the ecc extension: adapter == 0 the point chip inside the VM (773 columns:
 * zkhip_ec_air + a timestamp column + the 48 word receives), adapter != 0 its adapter; zkhip_vm_ec_tracegen = zkhip_ec_tracegen + the
 * timestamp of call i on row i. */
int zkhip_vm_ec_air(const uint8_t modulus[32], const uint8_t a[32], unsigned index, int adapter, zkhip_air *out);
int zkhip_vm_ec_tracegen(zkhip_ctx *ctx, const uint32_t modulus[8], const uint32_t a[8], const uint32_t *d_records, const uint32_t *d_ts, size_t n,
                         unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
/* The two chips of field `index` (< 2) of an app with the fp2 extension: adapter == 0 the Fp2 chip inside the VM (649 columns: zkhip_fp2_air
 * + a timestamp column + the 48 word receives), adapter != 0 its adapter; zkhip_vm_fp2_tracegen = zkhip_fp2_tracegen + the timestamps. */
int zkhip_vm_fp2_air(const uint8_t modulus[32], unsigned index, int adapter, zkhip_air *out);
int zkhip_vm_fp2_tracegen(zkhip_ctx *ctx, const uint32_t modulus[8], const uint32_t *d_records, const uint32_t *d_ts, size_t n, unsigned log_height,
                          uint32_t *d_trace, uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
/* The `_x` forms of the limb chips take the modulus's width: n_limbs = 32 (a modulus below 2^256; the functions above) or 48 (below 2^384:
 * BLS12-381's base field, its Fp2 and G1 -- crates/circuits/batch-circuit/openvm.toml:18-36; OpenVM instantiates its chips with 32 or 48
 * limbs likewise); n_words = n_limbs / 4.  Operands, records and results are n_words words each; the chips have 485 / 1156 / 968 columns
 * (+ the timestamp inside the VM), the adapters 3 n_words + 11 (modular) and 6 n_words + 11 (ecc, fp2). */
int zkhip_vm_modmul_air_x(const uint8_t *modulus, uint32_t n_limbs, unsigned index, int adapter, zkhip_air *out);
int zkhip_vm_modmul_tracegen_x(zkhip_ctx *ctx, uint32_t n_words, const uint32_t *modulus, const uint32_t *d_records, const uint32_t *d_ts, size_t n,
                               unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
int zkhip_vm_ec_air_x(const uint8_t *modulus, const uint8_t *a, uint32_t n_limbs, unsigned index, int adapter, zkhip_air *out);
int zkhip_vm_ec_tracegen_x(zkhip_ctx *ctx, uint32_t n_words, const uint32_t *modulus, const uint32_t *a, const uint32_t *d_records, const uint32_t *d_ts, size_t n,
                           unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
int zkhip_vm_fp2_air_x(const uint8_t *modulus, uint32_t n_limbs, unsigned index, int adapter, zkhip_air *out);
int zkhip_vm_fp2_tracegen_x(zkhip_ctx *ctx, uint32_t n_words, const uint32_t *modulus, const uint32_t *d_records, const uint32_t *d_ts, size_t n,
                            unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
/* the standalone chips and the host arithmetic (records: op | a | b = 2 n_words + 1 words; op | x1 y1 x2 y2 | slope = 5 n_words + 1;
 * op | a0 a1 | b0 b1 = 4 n_words + 1) */
int zkhip_modmul_air_x(const uint8_t *modulus, uint32_t n_limbs, uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air *out);
int zkhip_modular_host_x(uint32_t op, uint32_t n_words, const uint32_t *a, const uint32_t *b, const uint32_t *modulus, uint32_t *q, uint32_t *r);
int zkhip_modular_tracegen_x(zkhip_ctx *ctx, uint32_t n_words, const uint32_t *modulus, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace,
                             uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
int zkhip_ec_air_x(const uint8_t *modulus, const uint8_t *a, uint32_t n_limbs, uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air *out);
int zkhip_ec_host_x(uint32_t op, uint32_t n_words, const uint32_t *modulus, const uint32_t *a, const uint32_t *x1, const uint32_t *y1, const uint32_t *x2,
                    const uint32_t *y2, uint32_t *slope, uint32_t *x3, uint32_t *y3);
int zkhip_ec_tracegen_x(zkhip_ctx *ctx, uint32_t n_words, const uint32_t *modulus, const uint32_t *a, const uint32_t *d_records, size_t n, unsigned log_height,
                        uint32_t *d_trace, uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
int zkhip_fp2_air_x(const uint8_t *modulus, uint32_t n_limbs, uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air *out);
int zkhip_fp2_host_x(uint32_t op, uint32_t n_words, const uint32_t *modulus, const uint32_t *a, const uint32_t *b, uint32_t *r);
int zkhip_fp2_tracegen_x(zkhip_ctx *ctx, uint32_t n_words, const uint32_t *modulus, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace,
                         uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);

/* Poseidon2 chip serving the hash bus (299 columns): zkhip_poseidon2_air_tracegen + multiplicity 1 on the first n rows. */
int zkhip_vm_poseidon2_tracegen(zkhip_ctx *ctx, const uint32_t *d_inputs, size_t n, unsigned log_height, uint32_t *d_trace);
/* n rows of `width` canonical words (row-major, device) -> column-major Montgomery trace of 2^log_height rows; the other rows get
 * pad_row (HOST, `width` words; NULL = zeros).  For chips whose rows the executor writes itself (ecall, leaf, merkle, connector). */
int zkhip_rows_tracegen(zkhip_ctx *ctx, const uint32_t *d_rows, size_t n, size_t width, unsigned log_height, uint32_t *d_trace, const uint32_t *pad_row);
/* multiplicities of the requests scale * value (value = Montgomery words of a trace column) in a range table of 2^log_table entries */
int zkhip_range_counts_scaled_tracegen(zkhip_ctx *ctx, const uint32_t *d_values, size_t n, uint32_t scale, unsigned log_table, uint32_t *d_counts,
                                       int accumulate);

/* ---- Keccak-f[1600] chip (include/zkhip_keccak.hpp: one round per row, 24 rows per permutation, 2633 columns, degree 3 -- the
 *      shape of Plonky3's p3-keccak-air that OpenVM's keccak extension wraps; crates/circuits/chunk-circuit/openvm.toml:8-59 lists
 *      `keccak` among the chunk circuit's extensions).  Pinned to FIPS 202 / hashlib (tests/golden/keccak_kat.json). ---- */
#define ZKHIP_KECCAK_F_WIDTH 2633
int zkhip_keccak_f_air(zkhip_air *out);            /* the AIR program (pointer valid for the life of the process), width, no public values */
int zkhip_keccak_f1600_host(uint64_t state[25]);   /* the permutation on lanes state[x + 5 y] (host) */
/* d_states: n_perms x 50 words (lane x + 5 y = words 2 i, 2 i + 1, low word first; plain integers); fills 2^log_height rows: rows
 * 24 p .. 24 p + 23 = the rounds of permutation p, the rest permutations of the zero state (valid padding) */
int zkhip_keccak_f_tracegen(zkhip_ctx *ctx, const uint32_t *d_states, size_t n_perms, unsigned log_height, uint32_t *d_trace);

/* ---- SHA-256 compression chip (include/zkhip_sha256.hpp: one round per row, 65 rows per block, 433 main + 6 preprocessed columns,
 *      degree 3, no lookups; crates/circuits/chunk-circuit/openvm.toml:8-59 lists `sha256` among the chunk circuit's extensions;
 *      OpenVM's chip is openvm-sha256-circuit, un-vendored).  Pinned to FIPS 180-4 / hashlib (tests/golden/sha256_kat.json). ---- */
#define ZKHIP_SHA256_WIDTH 433
#define ZKHIP_SHA256_PREP_WIDTH 6
#define ZKHIP_SHA256_ROWS_PER_BLOCK 65
/* the AIR for a trace of 2^log_height rows (7 <= log_height <= 24): program, width, and `prep_trace` = the round constants and the
 * round / final / first / schedule gates of floor(2^log_height / 65) blocks (HOST, column-major; pointers valid for the life of the process) */
int zkhip_sha256_air(unsigned log_height, zkhip_air *out);
/* the compression function (host): state[8] <- compress(state, block), block = sixteen big-endian message words (FIPS 180-4 6.2.2) */
int zkhip_sha256_compress_host(uint32_t state[8], const uint32_t block[16]);
/* d_blocks: n_blocks x 24 words (H_in[8], then the sixteen message words; plain integers); rows 65 b .. 65 b + 63 = the rounds of block
 * b, row 65 b + 64 = H_out (state columns), further whole blocks = compressions of the zero record with real = 0, the tail rows zero */
int zkhip_sha256_tracegen(zkhip_ctx *ctx, const uint32_t *d_blocks, size_t n_blocks, unsigned log_height, uint32_t *d_trace);

/* ---- modular multiplication chip (include/zkhip_modular.hpp: r = a b mod P for a 256-bit modulus on byte limbs, one multiplication per
 *      row, 325 columns (flags is_add, is_sub: the same columns state a + b = q P + r and a - b + q P = r; is_div: the row (x / y, y, x) with the quotient canonical as well; is_eq on top of a subtraction row: the bit [a = b mod P]), degree 3; byte pairs looked up in the 8-bit bitwise table, carries in the range-tuple table;
 *      crates/circuits/chunk-circuit/openvm.toml:8-59 lists `modular` with the secp256k1 / bn254 / bls12-381 moduli; OpenVM's chip is
 *      openvm-algebra-circuit's ModularMulDiv, un-vendored).  Pinned to Python's integers (tests/golden/modular_kat.json). ---- */
#define ZKHIP_MODMUL_WIDTH 325
/* the AIR for one modulus (32 little-endian bytes, non-zero) sending on the given buses; pointers valid for the life of the process */
int zkhip_modmul_air(const uint8_t modulus[32], uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air *out);
/* (q, r) = divmod(a b, modulus) on little-endian 32-bit words (host); ZKHIP_ERR_INVALID if q does not fit 256 bits */
int zkhip_modmul_host(const uint32_t a[8], const uint32_t b[8], const uint32_t modulus[8], uint32_t q[8], uint32_t r[8]);
/* d_records: n x 16 words (a[8] | b[8], little-endian words, plain integers); fills 2^log_height rows and adds the rows' lookups to
 * the bitwise table's range column (d_bitwise_trace, 8-bit table) and to the range-tuple table d_tuple_counts (size_x >= 256) */
int zkhip_modmul_tracegen(zkhip_ctx *ctx, const uint32_t modulus[8], const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace,
                          uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
/* the same chip with an operation per record: d_records n x 17 words (op | a[8] | b[8]; op 0 mul, 1 add, 2 sub -- OpenVM's ModularAddSub in the
 * multiplication chip's columns; a subtraction needs |a - b| < P --, 3 div: the record holds (x / y, y), the row is the product (x / y) y = x);
 * zkhip_modular_host: (q, r) of one operation on the host (op 3: r = a / b mod P, q = 0; a < P, b invertible; op 4: r = [a = b mod P] for |a - b| < P) */
int zkhip_modular_tracegen(zkhip_ctx *ctx, const uint32_t modulus[8], const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace,
                           uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
int zkhip_modular_host(uint32_t op, const uint32_t a[8], const uint32_t b[8], const uint32_t modulus[8], uint32_t q[8], uint32_t r[8]);

/* ---- elliptic-curve chip (include/zkhip_ecc.hpp: short-Weierstrass chord addition (x1 != x2) and tangent doubling over a 256-bit prime on
 *      byte limbs, one point operation per row, 772 columns, degree 3: three limb identities with signed 33-byte quotients, byte pairs
 *      looked up in the 8-bit bitwise table, carries in the range-tuple table (256 x 2048), x3 and y3 canonical;
 *      crates/circuits/chunk-circuit/openvm.toml:38-59 lists `ecc.supported_curves` secp256k1, P-256, bn254 G1; OpenVM's chips are
 *      openvm-ecc-circuit's EcAddNe / EcDouble, un-vendored).  Pinned to Python's integers and the published multiples of the
 *      secp256k1 generator (tests/golden/ecc_kat.json). ---- */
#define ZKHIP_EC_WIDTH 772
#define ZKHIP_EC_RECORD_WORDS 41
/* the AIR for one curve (modulus and coefficient a as 32 little-endian bytes; modulus odd, top byte non-zero) sending on the given
 * buses; pointers valid for the life of the process */
int zkhip_ec_air(const uint8_t modulus[32], const uint8_t a[32], uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air *out);
/* one operation on the host (op 0: (x3, y3) = (x1, y1) + (x2, y2) with x1 != x2; op 1: the double of (x1, y1), y1 != 0): the slope and
 * the result, little-endian words; ZKHIP_ERR_INVALID if an operand is not reduced or the slope does not exist */
int zkhip_ec_host(uint32_t op, const uint32_t modulus[8], const uint32_t a[8], const uint32_t x1[8], const uint32_t y1[8], const uint32_t x2[8],
                  const uint32_t y2[8], uint32_t slope[8], uint32_t x3[8], uint32_t y3[8]);
/* d_records: n x 41 words (op | x1 y1 x2 y2 | slope, plain integers); fills 2^log_height rows and adds the rows' lookups to the bitwise
 * table's range column (d_bitwise_trace, 8-bit table) and to the range-tuple table d_tuple_counts (size_x >= 256, size_y >= 2048);
 * an error if a record's slope does not solve its identity */
int zkhip_ec_tracegen(zkhip_ctx *ctx, const uint32_t modulus[8], const uint32_t a[8], const uint32_t *d_records, size_t n, unsigned log_height,
                      uint32_t *d_trace, uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);

/* ---- Fp2 chip (include/zkhip_fp2.hpp: multiplication, division, addition, subtraction in Fp[u] / (u^2 + 1) over a 256-bit prime on byte
 *      limbs, one operation per row, 648 columns, degree 3: two component identities with signed 33-byte quotients, results canonical (and
 *      the quotient of a division); crates/circuits/chunk-circuit/openvm.toml:30-33 lists `fp2.supported_moduli` with bn254's Fp2;
 *      OpenVM's chips are openvm-algebra-circuit's Fp2AddSub / Fp2MulDiv, un-vendored).  Pinned to Python's integers
 *      (tests/golden/fp2_kat.json). ---- */
#define ZKHIP_FP2_WIDTH 648
#define ZKHIP_FP2_RECORD_WORDS 33
int zkhip_fp2_air(const uint8_t modulus[32], uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air *out);
/* r = a op b on the host (op 0 mul, 1 add, 2 sub, 3 div; elements as c0[8] | c1[8] little-endian words, components below the modulus);
 * ZKHIP_ERR_INVALID for an operand that is not reduced or a division by zero */
int zkhip_fp2_host(uint32_t op, const uint32_t modulus[8], const uint32_t a[16], const uint32_t b[16], uint32_t r[16]);
/* d_records: n x 33 words (op | a0 a1 | b0 b1; a division's record holds the quotient in the a slot and the divisor in b: the row is
 * their product); fills 2^log_height rows and adds the rows' lookups to the bitwise table's range column and the range-tuple table
 * (size_x >= 256, size_y >= 2048) */
int zkhip_fp2_tracegen(zkhip_ctx *ctx, const uint32_t modulus[8], const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace,
                       uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);

/* ---- 256-bit ALU chip (include/zkhip_int256.hpp: a = b op c on 32 byte limbs, op = 0 add, 1 sub, 2 xor, 3 or, 4 and, one operation per
 *      row, 101 columns, degree 3; 32 lookups per row in the 8-bit bitwise table; crates/circuits/chunk-circuit/openvm.toml:16-17 enables
 *      `bigint`; OpenVM's chip is openvm-bigint-circuit's Rv32BaseAlu256, un-vendored).  Pinned to Python's integers
 *      (tests/golden/int256_kat.json). ---- */
#define ZKHIP_INT256_ALU_WIDTH 101
int zkhip_int256_alu_air(uint32_t bitwise_bus, zkhip_air *out);   /* pointers valid for the life of the process */
int zkhip_int256_alu_host(uint32_t op, const uint32_t b[8], const uint32_t c[8], uint32_t a[8]);   /* little-endian words (host) */
/* d_records: n x 17 words (op | b[8] | c[8], plain integers); fills 2^log_height rows and adds the rows' lookups to the XOR column of
 * the 8-bit bitwise table (d_bitwise_trace: 2 x 65536 Montgomery words) */
int zkhip_int256_alu_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);
/* 256-bit multiplication chip (the low 256 bits of b c; 161 columns; byte pairs in the bitwise table, carries in the range-tuple table):
 * d_records: n x 16 words (b[8] | c[8]); zkhip_int256_alu_host accepts op 5 for it */
#define ZKHIP_INT256_MUL_WIDTH 161
int zkhip_int256_mul_air(uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air *out);
int zkhip_int256_mul_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace,
                              uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
/* inside the VM (162 columns; records = the int256 calls' 17 words with op 5) */
int zkhip_vm_mul256_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, const uint32_t *d_ts, size_t n, unsigned log_height, uint32_t *d_trace,
                             uint32_t *d_bitwise_trace, uint32_t *d_tuple_counts, uint32_t size_x, uint32_t size_y);
/* 256-bit comparison chip (b < c unsigned / signed, b == c; 103 columns; 34 lookups per row in the bitwise table's range column): d_records:
 * n x 17 words (op | b[8] | c[8], op 6 sltu, 7 slt, 8 eq); zkhip_int256_alu_host accepts ops 6..8 for it (the result word is 0 or 1);
 * inside the VM 108 columns: + the timestamp and the 256-BIT BRANCH columns is_br | neg | taken | opcode -- there d_records may also carry
 * the branch opcodes 12 beq, 13 bne, 14 bltu, 15 blt, 16 bgeu, 17 bge (OpenVM's Rv32BranchEqual256 / Rv32BranchLessThan256,
 * /root/reference/crates/circuits/chunk-circuit/openvm.toml:17-18): the row is the comparison's they rest on, `taken` = its result XOR neg goes to
 * the ecall chip, whose row steps the pc by 4 or by a2 (include/zkhip_int256.hpp cmp256_vm_air, include/zkhip_vm_circuit.hpp ecall_air) */
#define ZKHIP_INT256_CMP_WIDTH 103
int zkhip_int256_cmp_air(uint32_t bitwise_bus, zkhip_air *out);
int zkhip_int256_cmp_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);
int zkhip_vm_cmp256_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, const uint32_t *d_ts, size_t n, unsigned log_height, uint32_t *d_trace,
                             uint32_t *d_bitwise_trace);
/* 256-bit shift chip (b << / >> / >>arithmetic (c mod 256); 189 columns; 66 lookups per row in the 8-bit bitwise table, SRA's sign bit in
 * its XOR column: d_bitwise_trace = both multiplicity columns, 2 x 65536): d_records: n x 17 words (op | b[8] | c[8], op 9 sll, 10 srl,
 * 11 sra); zkhip_int256_alu_host accepts ops 9..11 for it; inside the VM 190 columns (+ the timestamp) */
#define ZKHIP_INT256_SHIFT_WIDTH 189
int zkhip_int256_shift_air(uint32_t bitwise_bus, zkhip_air *out);
int zkhip_int256_shift_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace, uint32_t *d_bitwise_trace);
int zkhip_vm_shift256_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, const uint32_t *d_ts, size_t n, unsigned log_height, uint32_t *d_trace,
                               uint32_t *d_bitwise_trace);
/* ---- NATIVE field / extension / castf chips of the one-statement circuit (`[app_vm_config.native]`, `[app_vm_config.castf]`: the chips
 *      of openvm-native-circuit the reference's batch and bundle circuits enable, crates/circuits/batch-circuit/openvm.toml:16,24 and
 *      bundle-circuit/openvm.toml:16,18; include/zkhip_native.hpp, include/zkhip_vm_circuit.hpp native_arith_air / native_ext_air /
 *      castf_vm_air; AIRs through zkhip_vm_air).  ONE row per call, core and memory adapter together; the rows are made on the device
 *      from the executor's call records (canonical words):
 *        native arith (27 columns), 9 words per call:  op | b | c | the result word before | word pointer | ts | previous timestamps of
 *                                                       the three words          (op 0 add, 1 sub, 2 mul, 3 div)
 *        native ext (90 columns), 27 words per call:    op | x[4] | y[4] | the result words before [4] | word pointer | ts | previous
 *                                                       timestamps of the twelve words
 *        castf (16 columns), 6 words per call:          x (< 2^30) | the output word before | word pointer | ts | previous timestamps of
 *                                                       the two words
 *      The chips' lookups (range, range-tuple, bitwise) are counted from the trace columns by the caller (include/zkhip_vm_flow.hpp).
 *      ZKHIP_ERR_INVALID for an unknown operation, a division by zero, a castf operand >= 2^30 or a timestamp gap out of range. ---- */
int zkhip_vm_native_arith_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace);
int zkhip_vm_native_ext_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace);
int zkhip_vm_castf_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, size_t n, unsigned log_height, uint32_t *d_trace);
/* the chip inside the VM (102 columns: + the timestamp of call i on row i) */
int zkhip_vm_int256_tracegen(zkhip_ctx *ctx, const uint32_t *d_records, const uint32_t *d_ts, size_t n, unsigned log_height, uint32_t *d_trace,
                             uint32_t *d_bitwise_trace);

/* ---- aggregation layer: the verifier circuit.  Replaces, for this backend's proofs, the leaf /
 *      internal verifier programs the reference's SDK proves at the nodes of its aggregation tree and `commit_child_vk`
 *      (crates/prover/src/prover/mod.rs:47-60, 200-282; crates/integration/src/lib.rs:461-514).  A circuit is built for ONE
 *      child verifying key -- FRI parameters, AIR programs, trace heights, preprocessed commitments -- and up to `max_children`
 *      proofs of it; it is an AIR set of three chips (gate chip, Poseidon2 chip, public-value chip; zkvm-prover_amd/csrc/
 *      recursion.hip) whose preprocessed traces hold the whole wiring, so the node's preprocessed commitments are the committed
 *      child vk.  A satisfying witness exists exactly when zkhip_verify accepts every present child.
 *      Public values of a node: [child-vk digest at the leaf level (8) | start state (K) | end state (K) | accumulator (8)]:
 *      the children's states are chained (end of child i = start of child i + 1), the accumulator is a Poseidon2 chain over the
 *      children's public values (leaf level) or accumulators (above). ---- */
typedef struct zkhip_recursion zkhip_recursion;
typedef struct {
    /* leaf level (child_is_node = 0): state word k of a child's START state is public value start_idx[k] of its AIR
     * start_air[k], of its END state public value end_idx[k] of AIR end_air[k]; n_state <= 16 (0 = no chained state).
     * child_is_node = 1: the children are proofs of a node circuit (their last AIR is its public-value chip); the layout
     * above is read from there and the arrays are ignored. */
    size_t n_state;
    const uint32_t *start_air, *start_idx, *end_air, *end_idx;
    int child_is_node;
    /* ONE AGGREGATION KEY (crates/prover/src/prover/mod.rs:147-170 one agg_vk; crates/verifier/src/verifier.rs:96-111): leaf and internal
     * circuits share the AIR programs and -- padded -- the heights, so the only thing that tells their proofs apart is the three
     * preprocessed commitments.  child_is_node = 2 builds the UNIFORM internal circuit: `child_airs` = the node AIR set at the common
     * heights (prep_commit ignored), the children's preprocessed commitments are VALUES of the circuit, every child is a proof of the
     * leaf circuit or of the internal circuit itself, and the node's public values end with [leaf commitment (8) | internal
     * commitment (8)] = zkhip_recursion_key_commit of the two keys: each child's commitments hash to one of the two, an internal
     * child states the same pair.  The verifier holds (internal key, leaf commitment) whatever the depth of the tree.
     * uniform = 1 on a leaf circuit (child_is_node = 0) appends the 16 words (zero) so that its proofs have the internal circuit's
     * public-value layout.  min_log_height pads the gate / Poseidon2 chip (0 = natural size).
     * DEFERRAL (crates/prover/src/prover/mod.rs:200-282 `enable_deferral`: child agg vk -> VerifyProver; crates/types/circuit/src/lib.rs:137-154
     * `verify_stark(input_commit, expected{exe commit, vm commit, public values})`): child_is_node = 3 builds the DEFERRAL NODE for the
     * aggregation key of a child app (`child_airs` = its internal circuit's key, commitments included): every child is a ROOT proof of a
     * guest flow; per child the circuit derives the claim a parent guest makes about it -- input commitment = sponge of the root's
     * statement, exe commitment = compress(initial memory root, entry pc), vm commitment = compress(app-vk digest, leaf commitment), the
     * 32 public-value bytes opened in the final memory root -- requires exit code 0 and the key's own internal commitment, and chains
     * acc <- compress(acc, chunk) over the claim's five 8-element chunks.  Public values [acc before (8) | acc after (8)]. */
    int uniform;
    unsigned min_log_height[2];
    /* PER-PROOF CHIP PRESENCE (the reference proves only the chips a segment used: AGENTS.md:183-185): an app has one segment key -- hence
     * one leaf circuit -- per SHAPE (set of chips a segment may carry: the base chips; + the hash intrinsics; + every extension), all
     * padded to one height set.  A uniform internal circuit built with n_leaf_shapes = S (<= 8; 0 = 1) takes the S leaf commitments as
     * values, a one-hot selector per leaf child, and states as its leaf commitment the sponge of the list (S = 1: the commitment itself).
     * app_id (leaf circuits; 8 canonical words or NULL): the statement's first 8 words, instead of the digest of the child key -- the
     * leaf circuits of one app state ONE app id (zkhip_recursion_vk_digest of the full shape). */
    size_t n_leaf_shapes;
    const uint32_t *app_id;
    /* BUNDLE OVER BATCHES (crates/integration/src/testers: chunk -> batch -> bundle, each layer deferring the verification of the one
     * below): a deferral node (child_is_node = 3) with region_index != 0 takes JOIN proofs as children -- proofs of a guest that itself
     * deferred.  `child_airs` = the join circuit's key; a child's statement is a root's 50 words followed by the chain (8) its own
     * deferral node verified.  What the verifier of a join does on the host happens in the circuit: the child guest's deferral region
     * (8 KiB = 512 blocks = one subtree of its memory tree, whose root is node `region_index` of the level 19 below the tree's root:
     * ((2 << 26) | base / 16) >> 9) opens in the child's final memory root, its n claims chain to the stated value.  Auxiliary words
     * per child then: 232 (public values) + 4096 cells + 19 x 8 siblings (bottom-up) + 63 flags (flag k = [k < n]). */
    uint32_t region_index;
} zkhip_recursion_stmt;
/* `airs[a].log_height` and `airs[a].prep_commit` are part of the child verifying key; program pointers are copied. */
int zkhip_recursion_build(const zkhip_params *child_params, const zkhip_air *child_airs, size_t n_child_airs, size_t max_children,
                          const zkhip_recursion_stmt *stmt, zkhip_recursion **out);
void zkhip_recursion_destroy(zkhip_recursion *r);
/* The JOIN of a guest that defers verification: child 0 = the guest's root proof under aggregation key A (its internal circuit's key,
 * commitments included), child 1 = the deferral node's proof (key B) whose chain starts at zero.  Public values: the root's statement
 * followed by the deferral accumulator (8): the verifier opens the guest's claims in the final memory root and hashes them.  Witness:
 * zkhip_recursion_witness with the two proofs.  Child 1 may also be a FOLD of deferral nodes (a task with more children than one
 * deferral node takes: up to 45 chunks in a batch, crates/types/batch/src/payload/v6.rs:10): a node circuit over deferral-node proofs
 * (child_is_node = 0, the chain as its chained state: start = public values 0..7 of the deferral node, end = 8..15), whose 32 public
 * values are [key digest | chain before | chain after | accumulator]. */
int zkhip_recursion_build_join(const zkhip_params *params_a, const zkhip_air *airs_a, size_t n_airs_a, const zkhip_params *params_b,
                               const zkhip_air *airs_b, size_t n_airs_b, zkhip_recursion **out);
/* a second user of the same circuit: wiring, programs and preprocessed traces shared, witness and device buffers its own (the levels of
 * an aggregation tree above the leaves run ONE internal circuit side by side); destroy each with zkhip_recursion_destroy */
int zkhip_recursion_fork(const zkhip_recursion *r, zkhip_recursion **out);
/* lays the preprocessed traces out again at heights >= 2^log_height[0] (gate chip) / 2^log_height[1] (Poseidon2 chip); not on a forked circuit */
int zkhip_recursion_pad(zkhip_recursion *r, const unsigned log_height[2]);
/* digest of a verifying key (parameters, AIR programs, heights, preprocessed commitments): what a leaf circuit built for it states */
int zkhip_recursion_vk_digest(const zkhip_params *params, const zkhip_air *airs, size_t n_airs, uint32_t out[8]);
/* digest of a node key's preprocessed commitments (n_commits x 8 canonical words, AIR order): the leaf / internal commitment of a uniform node */
int zkhip_recursion_key_commit(const uint32_t *prep_commits, size_t n_commits, uint32_t out[8]);
const char *zkhip_recursion_last_error(const zkhip_recursion *r); /* r = NULL: why the last zkhip_recursion_build of this thread failed */
size_t zkhip_recursion_n_airs(const zkhip_recursion *r);            /* 3 */
size_t zkhip_recursion_n_pvs(const zkhip_recursion *r);             /* 16 + 2 K (32 + 2 K under one key) */
size_t zkhip_recursion_n_state(const zkhip_recursion *r);           /* K */
size_t zkhip_recursion_max_children(const zkhip_recursion *r);
size_t zkhip_recursion_child_proof_bytes(const zkhip_recursion *r);
int zkhip_recursion_stats(const zkhip_recursion *r, size_t out[4]); /* wires, gate rows, permutations, public values */
int zkhip_recursion_child_vk_digest(const zkhip_recursion *r, uint32_t out[8]);
/* chip i of the node circuit as a zkhip_air (program, height, width, preprocessed trace; pointers stay owned by `r`):
 * 0 = gate chip, 1 = Poseidon2 chip, 2 = public-value chip (the node's public values are its public values) */
int zkhip_recursion_air(const zkhip_recursion *r, size_t i, zkhip_air *out);
/* Runs the circuit on n_present child proofs (proofs[c] with child_pvs[c][a] = the public values of AIR a of child c, canonical);
 * absent slots repeat child 0 without entering the statement.  node_pvs_out (zkhip_recursion_n_pvs words, may be NULL) receives
 * the node's public values.  ZKHIP_ERR_VERIFY: an assertion of the circuit fails (some child does not verify) -- the wire
 * values are kept, so the traces can still be generated (and their proof will not verify). */
int zkhip_recursion_witness(zkhip_recursion *r, const uint8_t *const *proofs, const size_t *proof_lens, const uint32_t *const *const *child_pvs,
                            size_t n_present, uint32_t *node_pvs_out);
/* the witness of a UNIFORM node: child c is a proof under the key with preprocessed commitments child_prep_commits[c] (3 x 8 canonical
 * words), of the internal circuit (child_is_leaf[c] = 0) or of leaf circuit j (child_is_leaf[c] = j + 1); leaf_commit = the S leaf
 * commitments (8 S words), internal_commit: see zkhip_recursion_stmt */
int zkhip_recursion_witness_uniform(zkhip_recursion *r, const uint8_t *const *proofs, const size_t *proof_lens, const uint32_t *const *const *child_pvs,
                                    const uint32_t *child_prep_commits, const int *child_is_leaf, const uint32_t leaf_commit[8],
                                    const uint32_t internal_commit[8], size_t n_present, uint32_t *node_pvs_out);
/* the witness of a DEFERRAL node: child_aux[c] = zkhip_recursion_n_aux words (232): the 16 cells of the child's two public-value blocks, then
 * the 27 sibling digests above the block pair in its final memory root, bottom-up (JOIN children: + the region opening, see
 * zkhip_recursion_stmt.region_index: 4543 words); acc_start = the claim chain before this node */
int zkhip_recursion_witness_deferral(zkhip_recursion *r, const uint8_t *const *proofs, const size_t *proof_lens, const uint32_t *const *const *child_pvs,
                                     const uint32_t *child_aux, const uint32_t acc_start[8], size_t n_present, uint32_t *node_pvs_out);
size_t zkhip_recursion_n_aux(const zkhip_recursion *r);
/* wire values of the last witness, canonical [n_wires + 1][4]; out = NULL: only the size */
int zkhip_recursion_wires(const zkhip_recursion *r, uint32_t *out, size_t cap_words, size_t *n_words);
/* device traces of the three chips from the last witness (Montgomery, column-major, stride = 2^log_height of the chip) */
int zkhip_recursion_tracegen(zkhip_ctx *ctx, zkhip_recursion *r, uint32_t *d_gate_trace, uint32_t *d_p2_trace, uint32_t *d_pv_trace);

/* ---- configuration.  Every behaviour-changing switch of the library is a FIELD here; the ZKHIP_* environment variables are overrides
 *      read in ONE place (zkhip_config_default) when a context is created -- or when the caller asks for the defaults -- and nowhere else.
 *      (What stays a bare environment variable is listed in README.md: measurement / A-B switches of single kernels and
 *      ZKHIP_KEYGEN_TIMING, none of which changes what the library computes or when it blocks.) ---- */
typedef struct {
    /* transcript: absorptions of >= host_sponge_min_words words (the opened values of a big proof) run on the host's 512-bit vector unit
     * -- ~0.25 us per permutation against ~1.5 us for the dependent chain on the device.  NOTE: this makes zkhip_prove_async BLOCK the
     * calling thread (two stream synchronisations) in the middle of such a proof; a host thread that drives several pipelines should
     * set host_sponge = 0.  [ZKHIP_NO_HOST_SPONGE=1 -> 0; ZKHIP_HOST_SPONGE_MIN_WORDS] */
    int host_sponge;
    uint32_t host_sponge_min_words;
    /* constraint kernels: 0 = interpreter only, 1 = compiled (hipRTC, at keygen) where 2^jit_min_log_work row-instructions repay the
     * compile, 2 = always compiled.  [ZKHIP_NO_JIT=1 -> 0, ZKHIP_FORCE_JIT=1 -> 2; ZKHIP_JIT_MIN_LOG_WORK] */
    int jit;
    uint32_t jit_min_log_work;
    /* directory of compiled constraint kernels kept across processes ("" = none).  Default: `jit_cache` next to libzkhip.so when it
     * exists (populated by __graft_entry__.build() -- hipRTC needs no GPU), else none.  [ZKHIP_JIT_CACHE_DIR] */
    char jit_cache_dir[256];
    int quot_slices;               /* short, wide chips run as up to 16 constraint slices side by side [ZKHIP_NO_QUOT_SLICES=1 -> 0] */
    uint32_t grind_sweep_shift;    /* a proof-of-work sweep covers 2^(bits + shift) candidates [ZKHIP_GRIND_SWEEP_SHIFT] */
    uint32_t coop_max_log, coop_inj_max_log; /* largest tree layer (plain / with injected rows) in the cooperative form [ZKHIP_COOP_MAX_LOG, ZKHIP_COOP_INJ_MAX_LOG] */
    uint32_t top_max_log;                    /* the layers from 2^top_max_log nodes down to the root are ONE launch of one workgroup (<= 8) [ZKHIP_TOP_MAX_LOG] */
    uint32_t commit_parts, side_cus;         /* pipelined trace commit (zkhip_set_commit_pipeline / zkhip_set_cu_partition) [ZKHIP_COMMIT_PARTS, ZKHIP_SIDE_CUS] */
    /* the verifier circuit's witness: threads of the query parts (0 = all cores) and queries of a child side by side are read from the
     * PROCESS configuration (the witness generator has no context: zkhip_set_process_config); page-locking of the wire values for the copy
     * to the device is a property of the CONTEXT that generates the node's traces (zkhip_recursion_tracegen reads its ctx's field)
     * [ZKHIP_WITNESS_THREADS, ZKHIP_NO_PIN_WITNESS=1 -> 0, ZKHIP_RECURSION_SERIAL_QUERIES=1 -> 0] */
    uint32_t witness_threads;
    int pin_witness, parallel_queries;
    /* diagnosis: after every proof the prover recomputes every plain layer of the proof's Merkle trees (and the FRI leaves) on the device
     * and fetches the proof twice; what differs is reported on stderr and the proof fails with ZKHIP_ERR_HIP [ZKHIP_SELF_CHECK=1] */
    int self_check;
    /* TEST ONLY: the round-4 bodies of the two fused tree kernels ("store a layer's word, then compute on"), the form under which the guest
     * flow stored a wrong tree node in ~3 % of runs -- kept for the A/B of docs/stale_node.md and the test that must go red on it.  The
     * kernels exist only in libzkhip_test.so (csrc/Makefile, -DZKHIP_TEST_KERNELS: zkhip_has_test_kernels() == 1); libzkhip.so refuses a
     * non-zero value in zkhip_ctx_set_config and ignores the variable [ZKHIP_TREE_STORE_EARLY=1, test library only] */
    int tree_store_early;
    /* lanes per workgroup of the row-sponge kernel of trees of >= 2^20 rows (a multiple of 64, 64..768).  256: eight workgroups fill every wave
     * slot of a CU.  768: two workgroups of 12 waves hold 6 of a SIMD's 8 slots and a third does not fit -- two slots, 176 VGPRs and the LDS
     * of every CU stay free for the memory-bound kernels (NTT passes, constraint kernel) of the other proofs in flight [ZKHIP_HASH_BLOCK] */
    uint32_t hash_block;
    /* small tree layers (<= 2^min(coop_max_log, coop_inj_max_log) nodes) in groups of up to five per launch, injected rows included
     * (k_compress_coop_fused); 0 = one launch per layer with injected rows, as round 4 [ZKHIP_NO_COOP_FUSED=1 -> 0] */
    int coop_fused;
    /* the rows of EVERY level of a mixed-height tree that has matrices are hashed by one launch beside the leaf level's (the layer kernels then
     * find a row's digest parked in its node's slot); 0 = only the levels above 2^coop_inj_max_log rows, at most eight, the others' rows inside
     * the cooperative layer kernels (rounds 4 - 5).  A level of at most 2^rows_coop_max_log rows whose rows are so wide that one lane's chain of
     * permutations would outlast the launch is hashed with 16 lanes per row [ZKHIP_NO_ROWS_IN_BULK=1 -> 0; ZKHIP_ROWS_COOP_MAX_LOG] */
    int rows_in_bulk;
    uint32_t rows_coop_max_log;
    /* log2 of the lanes of a transform-pass workgroup for passes of at most 2^10 rows (8, 9 or 10): 10 = tiles of 2^14 words in 1024-lane
     * workgroups with 70 KiB of LDS (two per CU), 9 / 8 = tiles of 2^13 / 2^12 words (four / eight per CU) -- the short transforms of a proof
     * with many small chips get scheduled beside the other streams' kernels instead of waiting for half a CU [ZKHIP_NTT_LOG_LANES] */
    uint32_t ntt_log_lanes;
    /* the compiled constraint kernels of one proof (one per chip, independent of each other) go round-robin over so many further streams of the
     * context, forked from and joined to the proof's stream by events -- the short chips' kernels (a few workgroups, a long program per row) run
     * beside the large ones instead of one after the other; 0 (the default) = all on the proof's stream; <= 4.  Off while the per-kernel timing of
     * zkhip_profile_enable is on (it times the proof's stream) [ZKHIP_QUOT_STREAMS] */
    uint32_t quot_streams;
} zkhip_config;
/* While `on`, the trace generators treat the shared lookup-count tables handed to them (the 8-bit bitwise table, the range-tuple table, the
 * range table) as canonical counts and leave them canonical -- none converts a table from Montgomery form and back around its increments.
 * The caller zeroes the tables, switches this on, runs a segment's generators, switches it off and converts each table once
 * (zkhip_to_monty).  A property of the context; off by default (every generator then leaves Montgomery tables, as its standalone tests expect). */
int zkhip_tables_canonical(zkhip_ctx *ctx, int on);
/* Compiles the constraint kernels of `airs` for blow-up 2^log_blowup into `cache_dir` (no GPU needed: hipRTC), where key generation finds
 * them when zkhip_config.jit_cache_dir names the directory; *n_ok (may be NULL) = kernels present afterwards. */
int zkhip_jit_prewarm(const zkhip_air *airs, size_t n_airs, unsigned log_blowup, const char *cache_dir, size_t *n_ok);
/* the built-in defaults with the environment's overrides applied */
void zkhip_config_default(zkhip_config *out);
/* CPUs this process may actually use: min(scheduler affinity, the cgroup's CPU quota -- /sys/fs/cgroup/cpu.max, or cpu.cfs_quota_us of
 * cgroup v1).  A container can SEE every host thread (256 on the MI355X boxes of this pool) while its cgroup grants 16: one thread per
 * visible CPU then spends its time throttled.  What the host-side thread counts of the library and of include/ *.hpp default to (the
 * aggregation witness generator, the executor's memory close, the host verification of segment proofs). */
unsigned zkhip_host_cpus(void);
/* a context's configuration (set at zkhip_ctx_create from zkhip_config_default); the witness fields are process-wide:
 * zkhip_set_process_config stores them (and the rest as the default of contexts created later).  zkhip_ctx_set_config applies every
 * field or refuses the call (ZKHIP_ERR_INVALID: jit 0..2, coop_* <= 27, rows_coop_max_log <= 27, ntt_log_lanes 8..10, quot_streams <= 4, jit_min_log_work <= 62, top_max_log <= 8, grind_sweep_shift <= 8,
 * commit_parts <= 8, side_cus < the device's CUs, hash_block a multiple of 64 in 64..768); a changed side_cus re-partitions as zkhip_set_cu_partition does. */
int zkhip_ctx_get_config(zkhip_ctx *ctx, zkhip_config *out);
/* 1 in libzkhip_test.so (the A/B bodies behind zkhip_config.tree_store_early are compiled in), 0 in the library that ships */
int zkhip_has_test_kernels(void);
int zkhip_ctx_set_config(zkhip_ctx *ctx, const zkhip_config *cfg);
int zkhip_set_process_config(const zkhip_config *cfg);

/* ---- per-kernel timing (HIP events on the ctx stream), for bench.py's roofline ---- */
int zkhip_profile_enable(zkhip_ctx *ctx, int on);
/* copies up to cap entries; returns number of distinct kernel names recorded */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
} zkhip_kernel_stat;
int zkhip_profile_read(zkhip_ctx *ctx, zkhip_kernel_stat *out, size_t cap);
int zkhip_profile_reset(zkhip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
