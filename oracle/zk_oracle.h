/*
 * zk_oracle.h -- CPU restatement ("oracle") of the STARK hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product (zkvm-prover_amd/, libzkhip.so).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * only as the checker / the timed CPU baseline.
 *
 * PARITY: the arithmetic of this path is not in /root/reference.  It
 * lives in un-vendored dependencies pinned by the reference's Cargo.lock:
 *   p3-baby-bear / p3-monty-31 / p3-field / p3-dft / p3-poseidon2 /
 *   p3-symmetric / p3-challenger 0.4.3   (Cargo.lock:5535-5750)
 *   zkhash-axiom 0.2.0                    (Cargo.lock:10231)
 *   openvm-stark-backend / -sdk 2.0.0 @16d60de7 (Cargo.lock:4634-4699,5337-5398)
 * none of which can be built or fetched here (no Rust, no network).  This file
 * restates their *published* algorithms (BabyBear, x^4-11 extension, radix-2
 * DFT / coset LDE, Poseidon2-BabyBear-16 with Grain-LFSR constants,
 * PaddingFreeSponge / TruncatedPermutation, MerkleTreeMmcs, DuplexChallenger,
 * and the uni-stark + two-adic-FRI pipeline named by BASELINE.json
 * north_star).
 * PINNED to the reference's own data (tests/golden/ref_v1_vectors.json, taken
 * by tests/golden/gen_ref_vectors.py from the eight OpenVM-v1 proofs under
 * /root/reference/crates/{verifier/testdata/proofs,prover/testdata};
 * tests/test_ref_vectors_cpu.py): the Poseidon2 permutation and
 * ora_compress (193 commitments), ora_hash_slice + ora_mmcs_verify incl. mixed
 * heights (72 batch openings, 20 FRI-layer leaves), ora_fri_fold and the
 * extension arithmetic (55 fold triples).
 * PARITY UNPINNED (not extractable without the verifying key's pre-hash, which
 * the tree does not hold): the transcript order (challenger.c's use in
 * stark.c), proof-of-work, the quotient identity and the reduced openings.
 * Those rest on (i) the constants the reference tree holds (p = 2013265921,
 * scripts/compress_bn254.py:10; DIGEST_SIZE = 8, crates/types/src/proof.rs:209;
 * FRI parameters, crates/circuits/chunk-circuit/openvm.toml:1-6), (ii) the
 * round-constant anchors of SURVEY.md A.3 and (iii) the independent big-int
 * Python model (tests/pymodel.py, tests/golden/gen_golden.py).
 *
 * All values crossing this API are CANONICAL u32 in [0, p).  Internally the
 * oracle computes with 64-bit integers and `% p` (deliberately not the
 * Montgomery form the HIP product uses).
 */
#ifndef ZK_ORACLE_H
#define ZK_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_P 2013265921u /* 2^31 - 2^27 + 1, scripts/compress_bn254.py:10 */
#define ORA_DIGEST 8      /* crates/types/src/proof.rs:209 DIGEST_SIZE */
#define ORA_WIDTH 16
#define ORA_RATE 8

/* ---- base field (p3-baby-bear semantics, canonical representation) ---- */
uint32_t ora_add(uint32_t a, uint32_t b);
uint32_t ora_sub(uint32_t a, uint32_t b);
uint32_t ora_mul(uint32_t a, uint32_t b);
uint32_t ora_pow(uint32_t a, uint64_t e);
uint32_t ora_inv(uint32_t a);
uint32_t ora_two_adic_generator(unsigned bits); /* 0x1a427a41^(2^(27-bits)) */

/* ---- quartic extension F[x]/(x^4 - 11), element = 4 coefficients ---- */
void ora_ext_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]);
void ora_ext_inv(const uint32_t a[4], uint32_t out[4]);

/* ---- DFT (p3-dft Radix2Dit semantics: natural order in, natural out) ---- */
/* column-major matrix: column c occupies mat[c*stride .. c*stride+n) */
void ora_dft_batch(uint32_t *mat, unsigned log_n, size_t width, size_t stride, int inverse);
/* naive O(n^2) definition, for pinning the fast one */
void ora_dft_naive(const uint32_t *in, uint32_t *out, unsigned log_n, int inverse);
/* coset LDE: evals over H (natural) -> evals over shift*K, |K| = n<<added_bits.
 * Output natural order unless bitrev_out, then row r holds point index bitrev(r). */
void ora_coset_lde_batch(const uint32_t *in, size_t in_stride, uint32_t *out, size_t out_stride,
                         unsigned log_n, unsigned added_bits, size_t width, uint32_t shift,
                         int bitrev_out);

/* ---- Poseidon2-BabyBear width 16 (p3-poseidon2 layers + zkhash RC16) ---- */
const uint32_t *ora_poseidon2_round_constants(void); /* 141 values: 4x16 ext, 13 int, 4x16 ext */
void ora_poseidon2_permute(uint32_t state[16]);
/* PaddingFreeSponge<_,16,8,8>: absorb `len` elements, squeeze 8 */
void ora_hash_slice(const uint32_t *in, size_t len, uint32_t out[8]);
/* TruncatedPermutation<_,2,8,16> */
void ora_compress(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]);
/* trace of the Poseidon2 AIR (298 columns, one permutation per row; p3-poseidon2-air layout with one S-box register):
 * inputs [n_perms][16] canonical, trace column-major with stride 2^log_height, rows >= n_perms permute the zero state */
#define ORA_POSEIDON2_AIR_WIDTH 298
void ora_poseidon2_air_trace(const uint32_t *inputs, size_t n_perms, unsigned log_height, uint32_t *trace);
/* multiplicity column of a range-check table: counts[v] (+)= #{i : values[i] == v}, canonical; returns #values out of range */
size_t ora_range_counts(const uint32_t *values, size_t n, unsigned log_table, uint32_t *counts, int accumulate);

/* three more periphery chips' traces (oracle/tracegen.c): range-tuple table, bitwise-operation lookup, volatile memory boundary */
size_t ora_range_tuple_counts(const uint32_t *xs, const uint32_t *ys, size_t n, uint32_t size_x, uint32_t size_y, uint32_t *counts,
                              int accumulate);
size_t ora_bitwise_lookup_counts(const uint32_t *xs, const uint32_t *ys, const uint32_t *ops, size_t n, unsigned bits, uint32_t *trace,
                                 int accumulate);
size_t ora_rv32_alu_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                          uint32_t *xor_counts);
size_t ora_memory_access_trace(const uint32_t *as, const uint32_t *ptr, const uint32_t *prev_data, const uint32_t *prev_ts, const uint32_t *data,
                               const uint32_t *ts, const uint32_t *is_read, size_t n, unsigned log_height, uint32_t *trace);
size_t ora_rv32_shift_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                            uint32_t *range_counts, uint32_t *xor_counts);
size_t ora_rv32_branch_eq_trace(const uint32_t *opc, const uint32_t *as, const uint32_t *bs, const uint32_t *imms, size_t n, unsigned log_height,
                                uint32_t *trace);
size_t ora_rv32_branch_lt_trace(const uint32_t *opc, const uint32_t *as, const uint32_t *bs, const uint32_t *imms, size_t n, unsigned log_height,
                                uint32_t *trace, uint32_t *range_counts);
size_t ora_rv32_jal_lui_trace(const uint32_t *opc, const uint32_t *pcs, const uint32_t *imms, size_t n, unsigned log_height, uint32_t *trace,
                              uint32_t *range_counts);
size_t ora_rv32_auipc_trace(const uint32_t *pcs, const uint32_t *imms, size_t n, unsigned log_height, uint32_t *trace, uint32_t *range_counts);
size_t ora_rv32_jalr_trace(const uint32_t *pcs, const uint32_t *rs1s, const uint32_t *imms, size_t n, unsigned log_height, uint32_t *trace,
                           uint32_t *range_counts);
size_t ora_rv32_mulh_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                           uint32_t *tuple_counts, uint32_t size_y, uint32_t *range_counts);
size_t ora_rv32_loadstore_trace(const uint32_t *cases, const uint32_t *reads, const uint32_t *prevs, size_t n, unsigned log_height, uint32_t *trace,
                                uint32_t *range_counts);
size_t ora_rv32_divrem_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                             uint32_t *tuple_counts, uint32_t size_y, uint32_t *range_counts);
size_t ora_mmcs_path_trace(const uint32_t *leaf, const uint32_t *index, const uint32_t *path_start, const uint32_t *step_kind,
                           const uint32_t *step_digest, size_t n_paths, unsigned log_height, uint32_t *trace, uint32_t *hash_inputs,
                           uint32_t *claims, size_t *n_claims);
size_t ora_field_arith_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace);
size_t ora_field_ext_trace(const uint32_t *opc, const uint32_t *xs, const uint32_t *ys, size_t n, unsigned log_height, uint32_t *trace);
size_t ora_var_range_counts(const uint32_t *values, const uint32_t *bits, uint32_t const_bits, size_t n, unsigned max_bits, uint32_t *counts);
size_t ora_castf_trace(const uint32_t *xs, size_t n, unsigned log_height, uint32_t *trace, uint32_t *var_range_counts);
size_t ora_rv32_lt_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                         uint32_t *range_counts);
size_t ora_program_freq_trace(const uint32_t *idx, size_t n, unsigned log_height, uint32_t *freq);
size_t ora_exec_frame_trace(const uint32_t *idx, size_t n, const uint32_t *program, size_t n_program, unsigned log_height, uint32_t *trace);
void ora_rv32_mul_trace(const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace, uint32_t *tuple_counts,
                        uint32_t size_y);
size_t ora_memory_boundary_trace(const uint32_t *as, const uint32_t *ptr, const uint32_t *init, const uint32_t *fin, const uint32_t *ts,
                                 size_t n, unsigned as_bits, unsigned ptr_bits, unsigned log_height, uint32_t *trace);

/* ---- MerkleTreeMmcs over column-major matrices of (possibly) mixed heights ---- */
typedef struct {
    const uint32_t *data; /* column-major */
    size_t stride;        /* elements between columns */
    unsigned log_height;
    size_t width;
} ora_matrix;

typedef struct ora_tree ora_tree;
ora_tree *ora_mmcs_commit(const ora_matrix *mats, size_t n_mats, uint32_t root[8]);
unsigned ora_tree_log_height(const ora_tree *t);
/* digest layer l (0 = leaves), returns pointer to 8*(2^(log_height-l)) words */
const uint32_t *ora_tree_layer(const ora_tree *t, unsigned layer);
/* open at leaf `index` of the tallest matrix: writes opened rows (matrices in
 * given order, each `width` words, row = index >> (log_height - mat.log_height))
 * then log_height sibling digests; returns number of words written. */
size_t ora_mmcs_open(const ora_tree *t, size_t index, uint32_t *out);
/* returns 1 if the opening verifies against root */
int ora_mmcs_verify(const uint32_t root[8], const unsigned *log_heights, const size_t *widths,
                    size_t n_mats, size_t index, const uint32_t *opening);
void ora_tree_free(ora_tree *t);
/* test utility (fixture generation): number of leaf indices at which a single-matrix opening verifies; first -> *index_out */
size_t ora_mmcs_find_index(const uint32_t root[8], unsigned log_height, size_t width, const uint32_t *opening,
                           size_t *index_out);

/* ---- DuplexChallenger<BabyBear, Perm, 16, 8> ---- */
typedef struct {
    uint32_t state[16];
    uint32_t in_buf[8];
    unsigned n_in;
    uint32_t out_buf[8];
    unsigned n_out;
} ora_challenger;
void ora_ch_init(ora_challenger *c);
void ora_ch_observe(ora_challenger *c, const uint32_t *vals, size_t n);
uint32_t ora_ch_sample(ora_challenger *c);
void ora_ch_sample_ext(ora_challenger *c, uint32_t out[4]);
uint32_t ora_ch_sample_bits(ora_challenger *c, unsigned bits);
/* smallest witness w in [0,p) with check_witness(bits, w); also advances c */
uint32_t ora_ch_grind(ora_challenger *c, unsigned bits);
int ora_ch_check_witness(ora_challenger *c, unsigned bits, uint32_t witness);

/* ---- FRI fold of one layer (p3-fri fold_row semantics, arity 2) ---- */
/* in: 2*n ext elements (bit-reversed domain order), out: n ext elements */
void ora_fri_fold(const uint32_t *in, uint32_t *out, unsigned log_n_out, const uint32_t beta[4]);

/* ---- Keccak (oracle/keccak.c): the permutation, the FIPS 202 sponge at rate 136, the Keccak-f chip's trace ---- */
void ora_keccak_f1600(uint64_t st[25]);
void ora_sha3_256(const uint8_t *msg, size_t len, uint8_t out[32], int keccak_padding);
void ora_keccak_f_trace(const uint64_t *inputs, size_t n_perms, unsigned log_height, uint32_t *trace);
/* ---- SHA-256 (oracle/sha256.c): compression function, padded hash, the compression chip's trace ---- */
void ora_sha256_compress(uint32_t h[8], const uint32_t m[16]);
void ora_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
void ora_sha256_trace(const uint32_t *blocks, size_t n_blocks, unsigned log_height, uint32_t *trace);
/* ---- 256-bit ALU (oracle/int256.c): a = b op c modulo 2^256 on bytes, the chip's trace ---- */
void ora_int256_alu(uint32_t op, const uint8_t b[32], const uint8_t c[32], uint8_t a[32]);
size_t ora_int256_alu_trace(const uint32_t *records, size_t n, unsigned log_height, uint32_t *trace, uint32_t *xor_counts);
size_t ora_mul256_trace(const uint32_t *records, size_t stride, size_t off, size_t n, unsigned log_height, uint32_t *trace, uint32_t *bitwise_range, uint32_t *tuple,
                        uint32_t size_y);
/* ---- modular multiplication (oracle/modular.c): r = a b mod P on bytes, the chip's trace ---- */
int ora_modmul(const uint8_t a[32], const uint8_t b[32], const uint8_t p[32], uint8_t q[32], uint8_t r[32]);
int ora_modaddsub(unsigned op, const uint8_t a[32], const uint8_t b[32], const uint8_t p[32], uint8_t q[32], uint8_t r[32]);
size_t ora_modular_trace(const uint8_t *records, const uint32_t *ops, size_t n, const uint8_t p[32], unsigned log_height, uint32_t *trace, uint32_t *bitwise_range,
                         uint32_t *tuple, uint32_t size_y);
size_t ora_modmul_trace(const uint8_t *records, size_t n, const uint8_t p[32], unsigned log_height, uint32_t *trace, uint32_t *bitwise_range, uint32_t *tuple,
                        uint32_t size_y);

/* the verifier's bus check: n exposed cumulative sums (4 canonical words each) must add up to zero; 0 = balanced */
int ora_logup_exposed_check(const uint32_t *exposed, size_t n);

/* ---- LogUp / sum-check building blocks (oracle/sumcheck.c) ---- */
void ora_ext_batch_inverse(const uint32_t *in, uint32_t *out, size_t n);
void ora_logup_running_sum(const uint32_t *den, const uint32_t *num, size_t n, uint32_t *out);
void ora_mle_fold(const uint32_t *in, uint32_t *out, size_t n, const uint32_t r[4]);
void ora_sumcheck_round(const uint32_t *const *tables, size_t k, size_t n_half, uint32_t *out);

/* ---- full STARK (see oracle/stark.c) ---- */
typedef struct {
    unsigned log_blowup;        /* 1  (openvm.toml:2) */
    unsigned log_final_poly_len;/* 0  (openvm.toml:3) */
    unsigned num_queries;       /* 100 (openvm.toml:4) */
    unsigned commit_pow_bits;   /* 16 (openvm.toml:5) */
    unsigned query_pow_bits;    /* 16 (openvm.toml:6) */
} ora_params;

typedef struct {
    const uint32_t *program; /* constraint bytecode, see DESIGN.md "AIR bytecode" */
    size_t program_len;      /* words */
    unsigned log_height;
    size_t width;
    const uint32_t *trace;   /* column-major, stride = 1<<log_height, canonical */
    const uint32_t *pvs;
    size_t n_pvs;
    const uint32_t *prep;        /* preprocessed trace (column-major, stride 1<<log_height) or NULL */
    const uint32_t *prep_commit; /* its 8-word commitment (verifier; NULL -> recomputed from prep) */
} ora_air_instance;

/* K5 on its own: quotient values (4 extension-coordinate columns of 2^(lh+b) rows) of one AIR over its committed LDE */
int ora_constraint_eval(const uint32_t *program, size_t program_len, unsigned log_height, unsigned log_blowup, size_t width,
                        const uint32_t *lde, const uint32_t *pvs, size_t n_pvs, const uint32_t alpha[4], uint32_t *q);
/* Proves; writes proof words (canonical u32 LE) into out (cap words); returns
 * number of words, or 0 on failure (e.g. constraints unsatisfied when checked). */
size_t ora_stark_prove(const ora_params *prm, const ora_air_instance *airs, size_t n_airs,
                       uint32_t *out, size_t cap);
/* Commitment of one AIR's preprocessed trace (keygen output, part of the verifying key). 0 = ok. */
int ora_prep_commit(const ora_params *prm, const ora_air_instance *air, uint32_t root[8]);
/* Verifies a proof against the AIR programs/public values. 0 = ok, <0 = error code. */
int ora_stark_verify(const ora_params *prm, const ora_air_instance *airs /* trace ignored */,
                     size_t n_airs, const uint32_t *proof, size_t n_words);

#ifdef __cplusplus
}
#endif
#endif
