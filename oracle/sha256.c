/* sha256.c -- CPU restatement (TEST INFRASTRUCTURE) of SHA-256 (FIPS 180-4: compression function 6.2.2, padding 5.1.1) and of the trace
 * of the SHA-256 compression chip (include/zkhip_sha256.hpp; zkhip_sha256_tracegen).  The reference's chip is un-vendored
 * (openvm-sha256-circuit, Cargo.lock); parity is anchored OUTSIDE this repository: the standard's vectors and Python's hashlib
 * (tests/golden/sha256_kat.json, tests/test_sha256_cpu.py).  Written from the standard; shares no code with the product. */
#include <stdint.h>
#include <string.h>

#include "zk_oracle.h"

static const uint32_t SK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
    0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
    0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
    0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
    0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const uint32_t SIV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

static uint32_t rr(uint32_t v, unsigned r) { return (v >> r) | (v << (32 - r)); }
static uint32_t bsig0(uint32_t x) { return rr(x, 2) ^ rr(x, 13) ^ rr(x, 22); }
static uint32_t bsig1(uint32_t x) { return rr(x, 6) ^ rr(x, 11) ^ rr(x, 25); }
static uint32_t ssig0(uint32_t x) { return rr(x, 7) ^ rr(x, 18) ^ (x >> 3); }
static uint32_t ssig1(uint32_t x) { return rr(x, 17) ^ rr(x, 19) ^ (x >> 10); }
static uint32_t fch(uint32_t x, uint32_t y, uint32_t z) { return (x & y) ^ (~x & z); }
static uint32_t fmaj(uint32_t x, uint32_t y, uint32_t z) { return (x & y) ^ (x & z) ^ (y & z); }

static void schedule(const uint32_t m[16], uint32_t w[64]) {
    for (int t = 0; t < 16; t++) w[t] = m[t];
    for (int t = 16; t < 64; t++) w[t] = ssig1(w[t - 2]) + w[t - 7] + ssig0(w[t - 15]) + w[t - 16];
}
/* one round: working variables v[0..7] = a..h */
static void sround(uint32_t v[8], uint32_t k, uint32_t w) {
    const uint32_t t1 = v[7] + bsig1(v[4]) + fch(v[4], v[5], v[6]) + k + w, t2 = bsig0(v[0]) + fmaj(v[0], v[1], v[2]);
    v[7] = v[6], v[6] = v[5], v[5] = v[4], v[4] = v[3] + t1, v[3] = v[2], v[2] = v[1], v[1] = v[0], v[0] = t1 + t2;
}

void ora_sha256_compress(uint32_t h[8], const uint32_t m[16]) {
    uint32_t w[64], v[8];
    schedule(m, w);
    memcpy(v, h, sizeof v);
    for (int t = 0; t < 64; t++) sround(v, SK[t], w[t]);
    for (int i = 0; i < 8; i++) h[i] += v[i];
}

void ora_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8];
    memcpy(h, SIV, sizeof h);
    const size_t total = ((len + 9 + 63) / 64) * 64;
    for (size_t off = 0; off < total; off += 64) {
        uint8_t blk[64];
        for (size_t i = 0; i < 64; i++) {
            const size_t p = off + i;
            blk[i] = p < len ? msg[p] : (p == len ? 0x80 : 0);
            if (p >= total - 8) blk[i] = (uint8_t)(((uint64_t)len * 8) >> (8 * (total - 1 - p)));
        }
        uint32_t m[16];
        for (int i = 0; i < 16; i++) m[i] = (uint32_t)blk[4 * i] << 24 | (uint32_t)blk[4 * i + 1] << 16 | (uint32_t)blk[4 * i + 2] << 8 | blk[4 * i + 3];
        ora_sha256_compress(h, m);
    }
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(h[i / 4] >> (8 * (3 - i % 4)));
}

/* trace of the compression chip: 433 columns x 2^log_height rows, column-major, canonical; blocks[24 b] = H_in[8] | M[16] */
void ora_sha256_trace(const uint32_t *blocks, size_t n_blocks, unsigned log_height, uint32_t *trace) {
    const size_t N = (size_t)1 << log_height, whole = N / 65;
    enum { STATE = 0, CARRY_A = 256, CARRY_E = 262, CARRY_SHIFT = 268, SIGMA0 = 280, SIGMA1 = 282, MAJ = 284, HIN = 286, W15 = 302, W14 = 334, W1 = 366,
           W0 = 398, W2 = 400, SIG0 = 424, SIG1 = 426, CARRY_W = 428, REAL = 432, WIDTH = 433 };
    memset(trace, 0, (size_t)WIDTH * N * sizeof(uint32_t));
#define PUT(col, v) trace[(size_t)(col) * N + row] = (uint32_t)(v)
#define LIMBS(col, v) PUT(col, (v) & 0xffff), PUT((col) + 1, (v) >> 16)
    for (size_t blk = 0; blk < whole; blk++) {
        uint32_t hin[8] = {0}, m[16] = {0}, w[64], v[8];
        const int real = blk < n_blocks;
        if (real) memcpy(hin, blocks + 24 * blk, sizeof hin), memcpy(m, blocks + 24 * blk + 8, sizeof m);
        schedule(m, w);
        memcpy(v, hin, sizeof v);
        for (unsigned t = 0; t <= 64; t++) {
            const size_t row = 65 * blk + t;
            uint32_t cur[8];
            for (int i = 0; i < 8; i++) cur[i] = t == 64 ? v[i] + hin[i] : v[i];
            for (int i = 0; i < 8; i++)
                for (int j = 0; j < 32; j++) PUT(STATE + 32 * i + j, (cur[i] >> j) & 1);
            const uint32_t s0 = bsig0(cur[0]), s1 = bsig1(cur[4]), mj = fmaj(cur[0], cur[1], cur[2]), ch = fch(cur[4], cur[5], cur[6]);
            LIMBS(SIGMA0, s0), LIMBS(SIGMA1, s1), LIMBS(MAJ, mj);
            for (int i = 0; i < 8; i++) LIMBS(HIN + 2 * i, hin[i]);
            PUT(REAL, real);
            if (t == 64) continue; /* the digest row's window and carries are zero */
            uint32_t win[16];
            for (unsigned k = 0; k < 16; k++) win[k] = t + k >= 15 ? w[t + k - 15] : 0;
            for (int j = 0; j < 32; j++) PUT(W15 + j, (win[15] >> j) & 1), PUT(W14 + j, (win[14] >> j) & 1), PUT(W1 + j, (win[1] >> j) & 1);
            LIMBS(W0, win[0]);
            for (unsigned k = 2; k < 14; k++) LIMBS(W2 + 2 * (k - 2), win[k]);
            const uint32_t g0 = ssig0(win[1]), g1 = ssig1(win[14]);
            LIMBS(SIG0, g0), LIMBS(SIG1, g1);
            /* carries: the sums the constraints state, limb by limb */
            const int last = t == 63;
            const uint32_t terms_a[8] = {cur[7], s1, ch, SK[t], w[t], s0, mj, last ? hin[0] : 0};
            const uint32_t terms_e[7] = {cur[3], cur[7], s1, ch, SK[t], w[t], last ? hin[4] : 0};
            uint32_t lo = 0, hi = 0;
            for (int i = 0; i < 8; i++) lo += terms_a[i] & 0xffff, hi += terms_a[i] >> 16;
            hi += lo >> 16;
            for (int k = 0; k < 3; k++) PUT(CARRY_A + k, (lo >> (16 + k)) & 1), PUT(CARRY_A + 3 + k, (hi >> (16 + k)) & 1);
            lo = hi = 0;
            for (int i = 0; i < 7; i++) lo += terms_e[i] & 0xffff, hi += terms_e[i] >> 16;
            hi += lo >> 16;
            for (int k = 0; k < 3; k++) PUT(CARRY_E + k, (lo >> (16 + k)) & 1), PUT(CARRY_E + 3 + k, (hi >> (16 + k)) & 1);
            static const int shifted[6] = {1, 2, 3, 5, 6, 7};
            for (int i = 0; i < 6; i++) {
                const uint32_t add = last ? hin[shifted[i]] : 0, src = cur[shifted[i] - 1];
                const uint32_t c0 = ((src & 0xffff) + (add & 0xffff)) >> 16, c1 = ((src >> 16) + (add >> 16) + c0) >> 16;
                PUT(CARRY_SHIFT + 2 * i, c0), PUT(CARRY_SHIFT + 2 * i + 1, c1);
            }
            if (t >= 15 && t < 63) {
                const uint32_t c0 = ((g1 & 0xffff) + (win[9] & 0xffff) + (g0 & 0xffff) + (win[0] & 0xffff)) >> 16;
                const uint32_t c1 = ((g1 >> 16) + (win[9] >> 16) + (g0 >> 16) + (win[0] >> 16) + c0) >> 16;
                PUT(CARRY_W, c0 & 1), PUT(CARRY_W + 1, c0 >> 1), PUT(CARRY_W + 2, c1 & 1), PUT(CARRY_W + 3, c1 >> 1);
            }
            sround(v, SK[t], w[t]);
        }
    }
#undef LIMBS
#undef PUT
}
