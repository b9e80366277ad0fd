/* int256.c -- CPU restatement (TEST INFRASTRUCTURE) of the 256-bit ALU chip's trace (include/zkhip_int256.hpp; zkhip_int256_alu_tracegen):
 * a = b op c modulo 2^256 on byte limbs, op = 0 add, 1 sub, 2 xor, 3 or, 4 and.  The reference's chip is un-vendored
 * (openvm-bigint-circuit Rv32BaseAlu256, Cargo.lock); parity is anchored OUTSIDE this repository: Python's integers
 * (tests/golden/int256_kat.json, tests/test_int256_cpu.py).  Written on BYTES; shares no code with the product. */
#include <stdint.h>
#include <string.h>

#include "zk_oracle.h"

void ora_int256_alu(uint32_t op, const uint8_t b[32], const uint8_t c[32], uint8_t a[32]) {
    int carry = 0;
    for (int i = 0; i < 32; i++) {
        int v;
        switch (op) {
            case 0: v = b[i] + c[i] + carry, carry = v >> 8; break;
            case 1: v = b[i] - c[i] - carry, carry = v < 0; break;
            case 2: v = b[i] ^ c[i]; break;
            case 3: v = b[i] | c[i]; break;
            default: v = b[i] & c[i]; break;
        }
        a[i] = (uint8_t)(v & 255);
    }
}

/* trace: 101 columns x 2^log_height rows, column-major, canonical; records[17 i] = op | b words | c words (little-endian); adds the
 * rows' 32 lookups to xor_counts[65536] (index 256 x + y); returns the number of bad records */
size_t ora_int256_alu_trace(const uint32_t *records, size_t n, unsigned log_height, uint32_t *trace, uint32_t *xor_counts) {
    const size_t N = (size_t)1 << log_height;
    enum { A = 0, B = 32, C = 64, FLAGS = 96, WIDTH = 101 };
    memset(trace, 0, (size_t)WIDTH * N * sizeof(uint32_t));
    size_t bad = 0;
    for (size_t row = 0; row < n && row < N; row++) {
        const uint32_t op = records[17 * row];
        if (op > 4) {
            bad++;
            continue;
        }
        uint8_t b[32], c[32], a[32];
        for (int i = 0; i < 32; i++) b[i] = (uint8_t)(records[17 * row + 1 + i / 4] >> (8 * (i % 4))), c[i] = (uint8_t)(records[17 * row + 9 + i / 4] >> (8 * (i % 4)));
        ora_int256_alu(op, b, c, a);
        for (int i = 0; i < 32; i++) {
            trace[(size_t)(A + i) * N + row] = a[i], trace[(size_t)(B + i) * N + row] = b[i], trace[(size_t)(C + i) * N + row] = c[i];
            xor_counts[op >= 2 ? 256 * b[i] + c[i] : 256 * a[i] + a[i]]++;
        }
        trace[(size_t)(FLAGS + op) * N + row] = 1;
    }
    return bad;
}

/* 256-bit multiplication chip: trace 161 columns; records[stride i + off ..] = b words | c words; adds the rows' lookups to
 * bitwise_range[65536] (byte pairs) and tuple[256 * size_y] (carries); returns the number of bad records */
size_t ora_mul256_trace(const uint32_t *records, size_t stride, size_t off, size_t n, unsigned log_height, uint32_t *trace, uint32_t *bitwise_range, uint32_t *tuple,
                        uint32_t size_y) {
    const size_t N = (size_t)1 << log_height;
    enum { A = 0, B = 32, C = 64, CX = 96, CY = 128, REAL = 160, WIDTH = 161 };
    memset(trace, 0, (size_t)WIDTH * N * sizeof(uint32_t));
    size_t bad = 0;
    for (size_t row = 0; row < n && row < N; row++) {
        uint8_t b[32], c[32], a[32];
        for (int i = 0; i < 32; i++) b[i] = (uint8_t)(records[stride * row + off + i / 4] >> (8 * (i % 4))), c[i] = (uint8_t)(records[stride * row + off + 8 + i / 4] >> (8 * (i % 4)));
        unsigned long carry = 0;
        for (int k = 0; k < 32; k++) {
            unsigned long s = carry;
            for (int i = 0; i <= k; i++) s += (unsigned long)b[i] * c[k - i];
            a[k] = (uint8_t)(s & 255), carry = s >> 8;
            if ((carry >> 8) >= size_y) bad++;
            trace[(size_t)(CX + k) * N + row] = (uint32_t)(carry & 255), trace[(size_t)(CY + k) * N + row] = (uint32_t)(carry >> 8);
            tuple[(size_t)(carry & 255) * size_y + (size_t)(carry >> 8)]++;
        }
        for (int i = 0; i < 32; i++) trace[(size_t)(A + i) * N + row] = a[i], trace[(size_t)(B + i) * N + row] = b[i], trace[(size_t)(C + i) * N + row] = c[i];
        for (int i = 0; i < 32; i += 2) bitwise_range[256 * a[i] + a[i + 1]]++, bitwise_range[256 * b[i] + b[i + 1]]++, bitwise_range[256 * c[i] + c[i + 1]]++;
        trace[(size_t)REAL * N + row] = 1;
    }
    return bad;
}
