/*
 * oracle/poseidon2.c -- Poseidon2-BabyBear, width 16, and the two hash
 * constructions built on it.  TEST INFRASTRUCTURE.  PINNED to the reference: 193 commitments of its stored proofs
 * are reproduced by the permutation / compress, 72 openings by the sponge (tests/test_ref_vectors_cpu.py).
 *
 * Restates p3-poseidon2 0.4.3 (external/internal layer structure, MDSMat4),
 * p3-baby-bear 0.4.3 (internal diagonal for width 16) and zkhash-axiom 0.2.0
 * (round constants RC16, produced by the Poseidon Grain-LFSR generator);
 * pins: Cargo.lock:5708,5736,10231.  Round-constant anchors and the
 * permutation self-consistency vectors are those of SURVEY.md A.3.
 * Sponge = p3-symmetric PaddingFreeSponge<Perm,16,8,8>; compression =
 * TruncatedPermutation<Perm,2,8,16>; digest = 8 words
 * (crates/types/src/proof.rs:209, crates/build-guest/src/main.rs:72).
 */
#include <string.h>
#include "zk_oracle.h"

#define RF_HALF 4
#define RP 13
#define NCONST (2 * RF_HALF * 16 + RP) /* 141 */

static uint32_t g_rc[NCONST];
static int g_rc_ready = 0;

/* ---- Grain LFSR (Poseidon reference generator) ---- */
typedef struct {
    uint8_t b[80];
} grain;

static int grain_step(grain *g) {
    int nb = g->b[62] ^ g->b[51] ^ g->b[38] ^ g->b[23] ^ g->b[13] ^ g->b[0];
    memmove(g->b, g->b + 1, 79);
    g->b[79] = (uint8_t)nb;
    return nb;
}
static void put_bits(uint8_t *dst, unsigned v, int n) {
    for (int i = 0; i < n; i++) dst[i] = (uint8_t)((v >> (n - 1 - i)) & 1);
}
static int grain_next_bit(grain *g) {
    /* take a bit; while it is 0 discard the next and retry; output the bit following a 1 */
    for (;;) {
        int b = grain_step(g);
        int c = grain_step(g);
        if (b) return c;
    }
}

static void gen_constants(void) {
    grain g;
    int o = 0;
    put_bits(g.b + o, 1, 2);   o += 2;   /* field = prime */
    put_bits(g.b + o, 0, 4);   o += 4;   /* sbox = x^alpha */
    put_bits(g.b + o, 31, 12); o += 12;  /* n bits */
    put_bits(g.b + o, 16, 12); o += 12;  /* t */
    put_bits(g.b + o, 8, 10);  o += 10;  /* R_F */
    put_bits(g.b + o, 13, 10); o += 10;  /* R_P */
    for (; o < 80; o++) g.b[o] = 1;
    for (int i = 0; i < 160; i++) grain_step(&g);
    int n = 0;
    while (n < NCONST) {
        uint32_t v = 0;
        for (int i = 0; i < 31; i++) v = (v << 1) | (uint32_t)grain_next_bit(&g);
        if (v < ORA_P) g_rc[n++] = v;
    }
    g_rc_ready = 1;
}

const uint32_t *ora_poseidon2_round_constants(void) {
    if (!g_rc_ready) gen_constants();
    return g_rc;
}

static uint32_t sbox7(uint32_t x) {
    uint32_t x2 = ora_mul(x, x), x3 = ora_mul(x2, x), x4 = ora_mul(x2, x2);
    return ora_mul(x3, x4);
}

/* p3-poseidon2 mds_light_permutation, width 16: M4 on each 4-block then add
 * the column sums: circ(2*M4, M4, M4, M4). M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] */
static void external_linear(uint32_t s[16]) {
    for (int b = 0; b < 16; b += 4) {
        uint64_t x0 = s[b], x1 = s[b + 1], x2 = s[b + 2], x3 = s[b + 3];
        s[b + 0] = (uint32_t)((2 * x0 + 3 * x1 + x2 + x3) % ORA_P);
        s[b + 1] = (uint32_t)((x0 + 2 * x1 + 3 * x2 + x3) % ORA_P);
        s[b + 2] = (uint32_t)((x0 + x1 + 2 * x2 + 3 * x3) % ORA_P);
        s[b + 3] = (uint32_t)((3 * x0 + x1 + x2 + 2 * x3) % ORA_P);
    }
    for (int k = 0; k < 4; k++) {
        uint64_t sum = (uint64_t)s[k] + s[4 + k] + s[8 + k] + s[12 + k];
        uint32_t sm = (uint32_t)(sum % ORA_P);
        for (int b = 0; b < 16; b += 4) s[b + k] = ora_add(s[b + k], sm);
    }
}

/* p3-baby-bear internal diagonal, width 16:
 * V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27] */
static uint32_t g_diag[16];
static int g_diag_ready = 0;
static void gen_diag(void) {
    uint32_t i2 = ora_inv(2);
    uint32_t inv2_8 = ora_pow(i2, 8), inv2_27 = ora_pow(i2, 27);
    uint32_t v[16] = {ORA_P - 2, 1, 2, i2, 3, 4, ORA_P - i2, ORA_P - 3, ORA_P - 4, inv2_8,
                      ora_pow(i2, 2), ora_pow(i2, 3), inv2_27, ORA_P - inv2_8,
                      ORA_P - ora_pow(i2, 4), ORA_P - inv2_27};
    memcpy(g_diag, v, sizeof v);
    g_diag_ready = 1;
}
static void internal_linear(uint32_t s[16]) {
    uint64_t sum = 0;
    for (int i = 0; i < 16; i++) sum += s[i];
    uint32_t sm = (uint32_t)(sum % ORA_P);
    for (int i = 0; i < 16; i++) s[i] = ora_add(ora_mul(s[i], g_diag[i]), sm);
}

void ora_poseidon2_permute(uint32_t s[16]) {
    const uint32_t *rc = ora_poseidon2_round_constants();
    if (!g_diag_ready) gen_diag();
    external_linear(s);
    for (int r = 0; r < RF_HALF; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox7(ora_add(s[i], rc[r * 16 + i]));
        external_linear(s);
    }
    for (int r = 0; r < RP; r++) {
        s[0] = sbox7(ora_add(s[0], rc[RF_HALF * 16 + r]));
        internal_linear(s);
    }
    for (int r = 0; r < RF_HALF; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox7(ora_add(s[i], rc[RF_HALF * 16 + RP + r * 16 + i]));
        external_linear(s);
    }
}

/* PaddingFreeSponge: overwrite the first <=RATE lanes per chunk, permute after
 * every (possibly partial) chunk, output lanes 0..8.  Empty input -> zeros. */
void ora_hash_slice(const uint32_t *in, size_t len, uint32_t out[8]) {
    uint32_t s[16];
    memset(s, 0, sizeof s);
    size_t i = 0;
    while (i < len) {
        size_t n = len - i < ORA_RATE ? len - i : ORA_RATE;
        for (size_t k = 0; k < n; k++) s[k] = in[i + k];
        ora_poseidon2_permute(s);
        i += n;
    }
    memcpy(out, s, 8 * sizeof(uint32_t));
}

void ora_compress(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]) {
    uint32_t s[16];
    memcpy(s, l, 32);
    memcpy(s + 8, r, 32);
    ora_poseidon2_permute(s);
    memcpy(out, s, 32);
}

/* Trace of the Poseidon2 AIR (one permutation per row; the structure of p3-poseidon2-air 0.4.3 with one S-box register,
 * Cargo.lock: p3-poseidon2-air / openvm-poseidon2-air): columns
 *   inputs[16] | 4 x { sbox[16] = (s+rc)^3, post[16] } | 13 x { sbox, post_sbox } | 4 x { sbox[16], post[16] }
 * column-major, stride N = 2^log_height, canonical.  Rows >= n_perms hold the permutation of the zero state (valid rows).
 * The last 16 columns are the permutation's output. */
void ora_poseidon2_air_trace(const uint32_t *inputs, size_t n_perms, unsigned log_height, uint32_t *trace) {
    const uint32_t *rc = ora_poseidon2_round_constants();
    if (!g_diag_ready) gen_diag();
    const size_t N = (size_t)1 << log_height;
    for (size_t r = 0; r < N; r++) {
        uint32_t s[16];
        size_t col = 0;
        for (int i = 0; i < 16; i++) {
            s[i] = r < n_perms ? inputs[r * 16 + i] : 0;
            trace[(col++) * N + r] = s[i];
        }
        external_linear(s);
        for (int half = 0; half < 2; half++) {
            for (int rd = 0; rd < RF_HALF; rd++) {
                const uint32_t *k = rc + (half ? RF_HALF * 16 + RP : 0) + rd * 16;
                for (int i = 0; i < 16; i++) {
                    uint32_t y = ora_add(s[i], k[i]);
                    uint32_t y3 = ora_mul(ora_mul(y, y), y);
                    trace[(col + i) * N + r] = y3;
                    s[i] = ora_mul(ora_mul(y3, y3), y);
                }
                external_linear(s);
                for (int i = 0; i < 16; i++) trace[(col + 16 + i) * N + r] = s[i];
                col += 32;
            }
            if (half == 0) {
                for (int rd = 0; rd < RP; rd++) {
                    uint32_t y = ora_add(s[0], rc[RF_HALF * 16 + rd]);
                    uint32_t y3 = ora_mul(ora_mul(y, y), y);
                    s[0] = ora_mul(ora_mul(y3, y3), y);
                    trace[(col++) * N + r] = y3;
                    trace[(col++) * N + r] = s[0];
                    internal_linear(s);
                }
            }
        }
    }
}
