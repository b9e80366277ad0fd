/* modular.c -- CPU restatement (TEST INFRASTRUCTURE) of the modular-multiplication chip's trace (include/zkhip_modular.hpp;
 * zkhip_modmul_tracegen): r = a b mod P on byte limbs.  The reference's chip is un-vendored (openvm-algebra-circuit ModularMulDiv,
 * Cargo.lock); parity is anchored OUTSIDE this repository: Python's integers (tests/golden/modular_kat.json, tests/test_modular_cpu.py).
 * Written on BYTES (schoolbook product, byte-wise long division with trial subtraction): shares no code with the product. */
#include <stdint.h>
#include <string.h>

#include "zk_oracle.h"

/* out[0..63] = a[0..31] * b[0..31], little-endian bytes */
static void mul_bytes(const uint8_t *a, const uint8_t *b, uint8_t *out) {
    uint32_t acc[64] = {0};
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) acc[i + j] += (uint32_t)a[i] * b[j];
    uint32_t c = 0;
    for (int k = 0; k < 64; k++) {
        c += acc[k];
        out[k] = (uint8_t)c, c >>= 8;
    }
}
/* does rem (33 bytes) >= p (32 bytes)? */
static int ge33(const uint8_t *rem, const uint8_t *p) {
    if (rem[32]) return 1;
    for (int k = 31; k >= 0; k--)
        if (rem[k] != p[k]) return rem[k] > p[k];
    return 1;
}
/* (q[64], r[32]) = divmod(x[64], p[32]): one byte of the dividend at a time, the quotient byte by repeated subtraction */
static void divmod_bytes(const uint8_t *x, const uint8_t *p, uint8_t *q, uint8_t *r) {
    uint8_t rem[33] = {0};
    for (int k = 63; k >= 0; k--) {
        memmove(rem + 1, rem, 32);
        rem[0] = x[k];
        unsigned d = 0;
        while (ge33(rem, p)) {
            int br = 0;
            for (int i = 0; i < 33; i++) {
                const int v = (int)rem[i] - (i < 32 ? p[i] : 0) - br;
                rem[i] = (uint8_t)(v & 255), br = v < 0;
            }
            d++;
        }
        q[k] = (uint8_t)d;
    }
    memcpy(r, rem, 32);
}

/* q, r of one product (bytes); returns 0 if the quotient fits 32 bytes */
int ora_modmul(const uint8_t a[32], const uint8_t b[32], const uint8_t p[32], uint8_t q[32], uint8_t r[32]) {
    uint8_t x[64], qq[64];
    mul_bytes(a, b, x);
    divmod_bytes(x, p, qq, r);
    memcpy(q, qq, 32);
    for (int k = 32; k < 64; k++)
        if (qq[k]) return 1;
    return 0;
}

/* a + b = q p + r (op 1) or a - b + q p = r (op 2; q is 0 or 1), r < p, on bytes; returns 0 if such q, r exist */
int ora_modaddsub(unsigned op, const uint8_t a[32], const uint8_t b[32], const uint8_t p[32], uint8_t q[32], uint8_t r[32]) {
    uint8_t num[33] = {0};
    memset(q, 0, 32);
    int c = 0;
    if (op == 1) {
        for (int i = 0; i < 32; i++) c += a[i] + b[i], num[i] = (uint8_t)c, c >>= 8;
        num[32] = (uint8_t)c;
        while (ge33(num, p)) {
            int br = 0;
            for (int i = 0; i < 33; i++) {
                const int v = (int)num[i] - (i < 32 ? p[i] : 0) - br;
                num[i] = (uint8_t)(v & 255), br = v < 0;
            }
            q[0]++;
        }
    } else {
        for (int i = 0; i < 32; i++) {
            const int v = (int)a[i] - b[i] - c;
            num[i] = (uint8_t)(v & 255), c = v < 0;
        }
        if (c) { /* negative: one p must bring it back */
            int cc = 0;
            for (int i = 0; i < 32; i++) cc += num[i] + p[i], num[i] = (uint8_t)cc, cc >>= 8;
            if (!cc) return 1;
            q[0] = 1;
        }
        if (ge33(num, p)) return 1;
    }
    memcpy(r, num, 32);
    return 0;
}

/* trace: 325 columns x 2^log_height rows, column-major, canonical; records[64 i] = a bytes | b bytes, ops[i] = 0 mul, 1 add, 2 sub, 3 div
 * (a = the quotient x / y, b = y), 4 equality test (NULL: all mul); adds the rows' lookups to bitwise_range[65536] (index 256 x + y) and tuple[256 * size_y] (index x * size_y + y); returns
 * the number of bad records */
size_t ora_modular_trace(const uint8_t *records, const uint32_t *ops, size_t n, const uint8_t p[32], unsigned log_height, uint32_t *trace, uint32_t *bitwise_range,
                         uint32_t *tuple, uint32_t size_y) {
    const size_t N = (size_t)1 << log_height;
    enum { A = 0, B = 32, Q = 64, R = 96, CX = 128, CY = 190, MARK = 252, DIFF = 284, REAL = 285, IS_ADD = 286, IS_SUB = 287, IS_DIV = 288, MARK2 = 289, DIFF2 = 321, IS_EQ = 322, EQ = 323, INV = 324, WIDTH = 325 };
    memset(trace, 0, (size_t)WIDTH * N * sizeof(uint32_t));
    size_t bad = 0;
    for (size_t row = 0; row < n && row < N; row++) {
        const uint8_t *a = records + 64 * row, *b = a + 32;
        /* a division row = the product (x / y) y; an equality test (4) = a subtraction row with the bit on top */
        const unsigned op_in = ops ? ops[row] : 0, is_div = op_in == 3, is_eq = op_in == 4, op = is_div ? 0 : is_eq ? 2 : op_in;
        uint8_t q[32], r[32];
        if (op_in > 4 || (op == 0 ? ora_modmul(a, b, p, q, r) : ora_modaddsub(op, a, b, p, q, r))) bad++;
#define PUT(col, v) trace[(size_t)(col) * N + row] = (uint32_t)(v)
        for (int i = 0; i < 32; i++) PUT(A + i, a[i]), PUT(B + i, b[i]), PUT(Q + i, q[i]), PUT(R + i, r[i]);
        for (int i = 0; i < 32; i += 2)
            bitwise_range[256 * a[i] + a[i + 1]]++, bitwise_range[256 * b[i] + b[i + 1]]++, bitwise_range[256 * q[i] + q[i + 1]]++, bitwise_range[256 * r[i] + r[i + 1]]++;
        long long c = 0;
        for (int k = 0; k <= 62; k++) {
            long long s = c;
            for (int i = 0; i < 32; i++) {
                const int j = k - i;
                if (j < 0 || j >= 32) continue;
                if (op == 0) s += (long long)a[i] * b[j];
                s -= (op == 2 ? -1 : 1) * (long long)q[i] * p[j];
            }
            if (k < 32 && op == 1) s += a[k] + b[k];
            if (k < 32 && op == 2) s += a[k] - b[k];
            if (k < 32) s -= r[k];
            c = s >> 8; /* exact: a b = q p + r */
            if (k < 62) {
                const long long v = c + (1 << 14);
                PUT(CX + k, v & 255), PUT(CY + k, v >> 8);
                tuple[(size_t)(v & 255) * size_y + (size_t)(v >> 8)]++;
            }
        }
        int mark = -1;
        for (int i = 31; i >= 0; i--)
            if (r[i] != p[i]) {
                mark = i;
                break;
            }
        if (mark >= 0) PUT(MARK + mark, 1);
        const unsigned diff = mark >= 0 ? (unsigned)(p[mark] - r[mark]) : 0;
        PUT(DIFF, diff), PUT(REAL, 1), PUT(IS_ADD, op == 1), PUT(IS_SUB, op == 2);
        bitwise_range[256 * ((diff - 1) & 255)]++;
        if (is_div) { /* the quotient in the a columns is below p */
            int m2 = -1;
            for (int i = 31; i >= 0; i--)
                if (a[i] != p[i]) {
                    m2 = a[i] < p[i] ? i : -2;
                    break;
                }
            if (m2 < 0) {
                bad++;
            } else {
                const unsigned d2 = (unsigned)(p[m2] - a[m2]);
                PUT(MARK2 + m2, 1), PUT(DIFF2, d2);
                bitwise_range[256 * ((d2 - 1) & 255)]++;
            }
            PUT(IS_DIV, 1);
        }
        if (is_eq) { /* eq = [r = 0] through the sum of r's limbs; its inverse modulo the BabyBear prime where it is not zero */
            unsigned long long sum = 0, inv = 1, base, e = 2013265921ull - 2;
            for (int i = 0; i < 32; i++) sum += r[i];
            for (base = sum; e; e >>= 1, base = base * base % 2013265921ull)
                if (e & 1) inv = inv * base % 2013265921ull;
            PUT(IS_EQ, 1), PUT(EQ, sum == 0), PUT(INV, sum ? inv : 0);
        }
#undef PUT
    }
    return bad;
}
size_t ora_modmul_trace(const uint8_t *records, size_t n, const uint8_t p[32], unsigned log_height, uint32_t *trace, uint32_t *bitwise_range, uint32_t *tuple,
                        uint32_t size_y) {
    return ora_modular_trace(records, 0, n, p, log_height, trace, bitwise_range, tuple, size_y);
}
