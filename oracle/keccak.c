/* keccak.c -- CPU restatement (TEST INFRASTRUCTURE) of Keccak-f[1600], the FIPS 202 sponge on top of it and the trace of the
 * Keccak-f chip (include/zkhip_keccak.hpp; zkhip_keccak_f_tracegen).  The reference's keccak chip is un-vendored (openvm-keccak256-circuit
 * over p3-keccak-air, Cargo.lock); parity is anchored OUTSIDE this repository: FIPS 202 / the Keccak team's vectors and Python's hashlib
 * (tests/golden/keccak_kat.json, tests/test_keccak_cpu.py).  Written from the specification (theta, rho, pi, chi, iota on lanes
 * A[x][y] = st[x + 5 y]); shares no code with the product. */
#include <stdint.h>
#include <string.h>

#include "zk_oracle.h"

static const uint64_t KRC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
                                 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
                                 0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
                                 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                                 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
/* rotation offsets r[x][y] of rho */
static const unsigned KROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};

static uint64_t rol(uint64_t v, unsigned r) { return r ? (v << r) | (v >> (64 - r)) : v; }

/* one round; the intermediate values the chip's row holds are returned when the pointers are given */
static void kround(uint64_t st[25], unsigned round, uint64_t *c_out, uint64_t *cp_out, uint64_t *ap_out, uint64_t *app_out) {
    uint64_t c[5], d[5], b[25], ap[25];
    for (int x = 0; x < 5; x++) c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) ap[x + 5 * y] = st[x + 5 * y] ^ d[x];
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(ap[x + 5 * y], KROT[x][y]);
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) st[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    if (c_out) memcpy(c_out, c, sizeof c);
    if (cp_out)
        for (int x = 0; x < 5; x++) cp_out[x] = c[x] ^ d[x];
    if (ap_out) memcpy(ap_out, ap, sizeof ap);
    if (app_out) memcpy(app_out, st, 25 * sizeof(uint64_t));
    st[0] ^= KRC[round];
}

void ora_keccak_f1600(uint64_t st[25]) {
    for (unsigned r = 0; r < 24; r++) kround(st, r, 0, 0, 0, 0);
}

/* sponge with rate 136 (SHA3-256 / Keccak-256): domain byte 0x06 (FIPS 202) or 0x01 (the original Keccak padding Ethereum uses) */
void ora_sha3_256(const uint8_t *msg, size_t len, uint8_t out[32], int keccak_padding) {
    uint64_t st[25] = {0};
    uint8_t block[136];
    size_t off = 0;
    for (;;) {
        const size_t n = len - off < 136 ? len - off : 136;
        memset(block, 0, sizeof block);
        if (n) memcpy(block, msg + off, n);
        const int last = n < 136;
        if (last) block[n] ^= keccak_padding ? 0x01 : 0x06, block[135] ^= 0x80;
        for (int i = 0; i < 17; i++) {
            uint64_t w = 0;
            for (int k = 0; k < 8; k++) w |= (uint64_t)block[8 * i + k] << (8 * k);
            st[i] ^= w;
        }
        ora_keccak_f1600(st);
        off += n;
        if (last) break;
    }
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(st[i / 8] >> (8 * (i % 8)));
}

/* trace of the Keccak-f chip: 2633 columns x 2^log_height rows, column-major, canonical; inputs[25 p + (x + 5 y)] */
void ora_keccak_f_trace(const uint64_t *inputs, size_t n_perms, unsigned log_height, uint32_t *trace) {
    const size_t N = (size_t)1 << log_height;
    enum { FLAGS = 0, EXPORT = 24, PRE = 25, A = 125, C = 225, CP = 545, AP = 865, APP = 2465, BITS = 2565, APPP = 2629 };
    for (size_t row = 0; row < N; row++) {
        const size_t p = row / 24;
        const unsigned r = (unsigned)(row % 24);
        uint64_t pre[25] = {0}, st[25], c[5], cp[5], ap[25], app[25], a[25];
        if (p < n_perms) memcpy(pre, inputs + 25 * p, sizeof pre);
        memcpy(st, pre, sizeof st);
        for (unsigned q = 0; q < r; q++) kround(st, q, 0, 0, 0, 0);
        memcpy(a, st, sizeof a);
        kround(st, r, c, cp, ap, app);
#define PUT(col, v) trace[(size_t)(col) * N + row] = (uint32_t)(v)
        for (unsigned i = 0; i < 24; i++) PUT(FLAGS + i, i == r);
        PUT(EXPORT, r == 23 && p < n_perms);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++)
                for (int l = 0; l < 4; l++) {
                    PUT(PRE + (y * 5 + x) * 4 + l, (pre[x + 5 * y] >> (16 * l)) & 0xffff);
                    PUT(A + (y * 5 + x) * 4 + l, (a[x + 5 * y] >> (16 * l)) & 0xffff);
                    PUT(APP + (y * 5 + x) * 4 + l, (app[x + 5 * y] >> (16 * l)) & 0xffff);
                }
        for (int x = 0; x < 5; x++)
            for (int z = 0; z < 64; z++) {
                PUT(C + x * 64 + z, (c[x] >> z) & 1);
                PUT(CP + x * 64 + z, (cp[x] >> z) & 1);
            }
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++)
                for (int z = 0; z < 64; z++) PUT(AP + (y * 5 + x) * 64 + z, (ap[x + 5 * y] >> z) & 1);
        for (int z = 0; z < 64; z++) PUT(BITS + z, (app[0] >> z) & 1);
        for (int l = 0; l < 4; l++) PUT(APPP + l, (st[0] >> (16 * l)) & 0xffff);
#undef PUT
    }
}
