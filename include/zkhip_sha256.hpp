// zkhip_sha256.hpp -- the SHA-256 compression-function AIR (crates/circuits/chunk-circuit/openvm.toml:8-59 lists
// `sha256` among the chunk circuit's extensions).  OpenVM's chip (openvm-sha256-circuit / openvm-sha256-air, un-vendored, Cargo.lock) packs
// four rounds into a row; this is the same statement in this repository's own columns: ONE ROUND PER ROW, 65 rows per block, 433 main
// columns + 6 preprocessed, every constraint of degree <= 3, no lookups (every word that needs a range is held in bits):
//   rows 0..63 of a block hold the working state (a..h)_t BEFORE round t and the message-schedule window W_{t-15} .. W_t;
//   row 64 holds H_out = H_in + (a..h)_64 (the feed-forward is folded into round 63's transition);
//   rows beyond the last whole block of the trace are zero (the preprocessed gates are zero there).
// Main columns:
//   a b c d e f g h bits [8][32] | carries of a, e (3 bits per limb) | carries of the six shifted words (1 bit per limb) |
//   Sigma0(a), Sigma1(e), Maj(a,b,c) as 16-bit limbs | H_in [8][2] carried through the block | W_t bits | W_{t-1} bits | W_{t-14} bits |
//   W_{t-15}, W_{t-13} .. W_{t-2} as limbs | sigma0(W_{t-14}), sigma1(W_{t-1}) limbs | schedule carries (2 bits per limb) | real
// Preprocessed (period 65): K_t limbs | round (t < 63) | final (t = 63) | first (t = 0) | sched (15 <= t < 63)
//   round t:   a' = h + Sigma1(e) + Ch(e,f,g) + K_t + W_t + Sigma0(a) + Maj(a,b,c),  e' = d + h + Sigma1(e) + Ch + K_t + W_t,  b' = a, ... (FIPS 180-4 6.2.2)
//   schedule:  W_{t+1} = sigma1(W_{t-1}) + W_{t-6} + sigma0(W_{t-14}) + W_{t-15}   for t + 1 >= 16; the first sixteen words are the block.
// Additions are mod 2^32 on 16-bit limbs with explicit carries; the xor3 / Maj sums are degree-3 DEFINITIONS of limb columns that hold on
// every row, so the gated transitions stay at degree 3.
// External parity: tests pin the compression function, the padded hash built on it and the trace's H_out rows to FIPS 180-4 through
// hashlib's SHA-256 and the standard's "abc" / empty / two-block vectors (tests/golden/sha256_kat.json).  Header-only; device generator:
// csrc/sha256.hip.
#pragma once
#include <cstdint>
#include <vector>

#include "zkhip_air.hpp"

namespace zkhip {
namespace sha256 {
using air::AirBuilder;
using air::Expr;

constexpr size_t ROWS_PER_BLOCK = 65;
constexpr size_t COL_STATE = 0;                       // word w (a = 0 .. h = 7), bit j: COL_STATE + 32 w + j
constexpr size_t COL_CARRY_A = 256, COL_CARRY_E = 262;   // 3 bits for the low limb's carry, 3 for the high limb's
constexpr size_t COL_CARRY_SHIFT = 268;               // words b, c, d, f, g, h: (lo, hi) carry bits, 12 columns
constexpr size_t COL_SIGMA0 = 280, COL_SIGMA1 = 282, COL_MAJ = 284;
constexpr size_t COL_HIN = 286;                       // word k limbs at COL_HIN + 2 k
constexpr size_t COL_W15_BITS = 302, COL_W14_BITS = 334, COL_W1_BITS = 366;
constexpr size_t COL_W0 = 398;                        // limbs of W_{t-15}
constexpr size_t COL_W2 = 400;                        // limbs of window positions 2 .. 13 (W_{t-13} .. W_{t-2}): COL_W2 + 2 (k - 2)
constexpr size_t COL_SIG0 = 424, COL_SIG1 = 426;
constexpr size_t COL_CARRY_W = 428;                   // 2 bits per limb
constexpr size_t COL_REAL = 432, WIDTH = 433;
constexpr size_t PREP_K = 0, PREP_ROUND = 2, PREP_FINAL = 3, PREP_FIRST = 4, PREP_SCHED = 5, PREP_WIDTH = 6;

constexpr uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
    0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
    0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
    0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
    0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
constexpr uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

inline uint32_t rotr(uint32_t v, unsigned r) { return (v >> r) | (v << (32 - r)); }

// the compression function on big-endian message words (FIPS 180-4 6.2.2)
inline void compress(uint32_t h[8], const uint32_t m[16]) {
    uint32_t w[64], s[8];
    for (int t = 0; t < 16; t++) w[t] = m[t];
    for (int t = 16; t < 64; t++)
        w[t] = (rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> 10)) + w[t - 7] + (rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> 3)) + w[t - 16];
    for (int i = 0; i < 8; i++) s[i] = h[i];
    for (int t = 0; t < 64; t++) {
        const uint32_t t1 = s[7] + (rotr(s[4], 6) ^ rotr(s[4], 11) ^ rotr(s[4], 25)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + K[t] + w[t];
        const uint32_t t2 = (rotr(s[0], 2) ^ rotr(s[0], 13) ^ rotr(s[0], 22)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
        for (int i = 7; i > 0; i--) s[i] = s[i - 1];
        s[4] += t1, s[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += s[i];
}

// preprocessed trace: PREP_WIDTH columns x 2^log_height rows, column-major, canonical
inline std::vector<uint32_t> prep_trace(unsigned log_height) {
    const size_t N = (size_t)1 << log_height, blocks = N / ROWS_PER_BLOCK;
    std::vector<uint32_t> p(PREP_WIDTH * N, 0);
    for (size_t b = 0; b < blocks; b++)
        for (size_t t = 0; t < ROWS_PER_BLOCK; t++) {
            const size_t row = b * ROWS_PER_BLOCK + t;
            if (t < 64) p[(PREP_K + 0) * N + row] = K[t] & 0xffffu, p[(PREP_K + 1) * N + row] = K[t] >> 16;
            p[PREP_ROUND * N + row] = t < 63, p[PREP_FINAL * N + row] = t == 63, p[PREP_FIRST * N + row] = t == 0;
            p[PREP_SCHED * N + row] = t >= 15 && t < 63;
        }
    return p;
}

// AirBuilder(WIDTH, 0, PREP_WIDTH)
inline void compress_air(AirBuilder& b) {
    auto xor2 = [](Expr p, Expr q) { return p + q - p * q * 2; };
    auto xor3 = [&](Expr p, Expr q, Expr r) { return xor2(xor2(p, q), r); };
    auto bit = [&](size_t base, size_t j) { return b.var(base + j); };
    auto limb_of_bits = [&](size_t base, size_t l, bool next) {
        Expr s = b.constant(0);
        for (size_t k = 0; k < 16; k++) s = s + (next ? b.next(base + 16 * l + k) : b.var(base + 16 * l + k)) * (int64_t)(1u << k);
        return s;
    };
    auto small = [&](size_t base, size_t n_bits) {
        Expr s = b.constant(0);
        for (size_t k = 0; k < n_bits; k++) s = s + b.var(base + k) * (int64_t)(1u << k);
        return s;
    };
    auto boolean = [&](size_t col) { b.assert_zero(b.var(col) * (b.var(col) - 1)); };
    for (size_t c = COL_STATE; c < COL_SIGMA0; c++) boolean(c);
    for (size_t c = COL_W15_BITS; c < COL_W0; c++) boolean(c);
    for (size_t c = COL_CARRY_W; c < WIDTH; c++) boolean(c);
    const Expr round = b.prep(PREP_ROUND), fin = b.prep(PREP_FINAL), first = b.prep(PREP_FIRST), sched = b.prep(PREP_SCHED), gate = round + fin;
    const size_t A = COL_STATE, Bw = COL_STATE + 32, Cw = COL_STATE + 64, E = COL_STATE + 128, F = COL_STATE + 160, G = COL_STATE + 192;
    // definitions that hold on every row
    for (size_t l = 0; l < 2; l++) {
        Expr s0 = b.constant(0), s1 = b.constant(0), mj = b.constant(0), g0 = b.constant(0), g1 = b.constant(0);
        for (size_t k = 0; k < 16; k++) {
            const size_t j = 16 * l + k;
            const int64_t w = (int64_t)(1u << k);
            s0 = s0 + xor3(bit(A, (j + 2) % 32), bit(A, (j + 13) % 32), bit(A, (j + 22) % 32)) * w;
            s1 = s1 + xor3(bit(E, (j + 6) % 32), bit(E, (j + 11) % 32), bit(E, (j + 25) % 32)) * w;
            const Expr x = bit(A, j), y = bit(Bw, j), z = bit(Cw, j);
            mj = mj + (x * y + x * z + y * z - x * y * z * 2) * w;
            const Expr r7 = bit(COL_W1_BITS, (j + 7) % 32), r18 = bit(COL_W1_BITS, (j + 18) % 32);
            g0 = g0 + (j + 3 < 32 ? xor3(r7, r18, bit(COL_W1_BITS, j + 3)) : xor2(r7, r18)) * w;
            const Expr r17 = bit(COL_W14_BITS, (j + 17) % 32), r19 = bit(COL_W14_BITS, (j + 19) % 32);
            g1 = g1 + (j + 10 < 32 ? xor3(r17, r19, bit(COL_W14_BITS, j + 10)) : xor2(r17, r19)) * w;
        }
        b.assert_zero(b.var(COL_SIGMA0 + l) - s0);
        b.assert_zero(b.var(COL_SIGMA1 + l) - s1);
        b.assert_zero(b.var(COL_MAJ + l) - mj);
        b.assert_zero(b.var(COL_SIG0 + l) - g0);
        b.assert_zero(b.var(COL_SIG1 + l) - g1);
    }
    // the round (rows 0..63; the final round adds H_in)
    for (size_t l = 0; l < 2; l++) {
        Expr ch = b.constant(0);
        for (size_t k = 0; k < 16; k++) {
            const size_t j = 16 * l + k;
            ch = ch + (bit(E, j) * (bit(F, j) - bit(G, j)) + bit(G, j)) * (int64_t)(1u << k);
        }
        const Expr t1 = limb_of_bits(COL_STATE + 32 * 7, l, false) + b.var(COL_SIGMA1 + l) + ch + b.prep(PREP_K + l) + limb_of_bits(COL_W15_BITS, l, false);
        const Expr t2 = b.var(COL_SIGMA0 + l) + b.var(COL_MAJ + l);
        const Expr cin_a = l ? small(COL_CARRY_A, 3) : b.constant(0), cin_e = l ? small(COL_CARRY_E, 3) : b.constant(0);
        b.when_transition(gate * (limb_of_bits(A, l, true) + small(COL_CARRY_A + 3 * l, 3) * 65536 - t1 - t2 - cin_a) - fin * b.var(COL_HIN + 0 + l));
        b.when_transition(gate * (limb_of_bits(E, l, true) + small(COL_CARRY_E + 3 * l, 3) * 65536 - limb_of_bits(COL_STATE + 32 * 3, l, false) - t1 - cin_e) -
                          fin * b.var(COL_HIN + 8 + l));
    }
    const size_t shifted[6] = {1, 2, 3, 5, 6, 7};   // b' = a, c' = b, d' = c, f' = e, g' = f, h' = g
    for (size_t i = 0; i < 6; i++) {
        const size_t w = shifted[i];
        for (size_t l = 0; l < 2; l++) {
            const Expr cin = l ? b.var(COL_CARRY_SHIFT + 2 * i) : b.constant(0);
            b.when_transition(gate * (limb_of_bits(COL_STATE + 32 * w, l, true) + b.var(COL_CARRY_SHIFT + 2 * i + l) * 65536 - limb_of_bits(COL_STATE + 32 * (w - 1), l, false) - cin) -
                              fin * b.var(COL_HIN + 2 * w + l));
        }
    }
    // H_in: the state of the block's first row, carried to the digest row
    for (size_t w = 0; w < 8; w++)
        for (size_t l = 0; l < 2; l++) {
            b.assert_zero(first * (b.var(COL_HIN + 2 * w + l) - limb_of_bits(COL_STATE + 32 * w, l, false)));
            b.when_transition(gate * (b.next(COL_HIN + 2 * w + l) - b.var(COL_HIN + 2 * w + l)));
        }
    b.when_transition(gate * (b.next(COL_REAL) - b.var(COL_REAL)));
    // the window moves one word per round
    for (size_t j = 0; j < 32; j++) b.when_transition(round * (b.next(COL_W14_BITS + j) - b.var(COL_W15_BITS + j)));
    for (size_t l = 0; l < 2; l++) {
        b.when_transition(round * (b.next(COL_W2 + 2 * 11 + l) - limb_of_bits(COL_W14_BITS, l, false)));            // position 13 <- 14
        for (size_t k = 2; k < 13; k++) b.when_transition(round * (b.next(COL_W2 + 2 * (k - 2) + l) - b.var(COL_W2 + 2 * (k - 1) + l)));
        b.when_transition(round * (limb_of_bits(COL_W1_BITS, l, true) - b.var(COL_W2 + l)));                         // position 1 <- 2
        b.when_transition(round * (b.next(COL_W0 + l) - limb_of_bits(COL_W1_BITS, l, false)));                       // position 0 <- 1
        // W_{t+1} = sigma1(W_{t-1}) + W_{t-6} + sigma0(W_{t-14}) + W_{t-15}: window positions 14, 9, 1, 0
        const Expr cin = l ? small(COL_CARRY_W, 2) : b.constant(0);
        b.when_transition(sched * (limb_of_bits(COL_W15_BITS, l, true) + small(COL_CARRY_W + 2 * l, 2) * 65536 - b.var(COL_SIG1 + l) - b.var(COL_W2 + 2 * (9 - 2) + l) -
                                   b.var(COL_SIG0 + l) - b.var(COL_W0 + l) - cin));
    }
}

// The chip inside the VM (include/zkhip_vm_circuit.hpp): the same AIR plus a timestamp column (constant through a block) and three
// more preprocessed columns (input: t < 16, digest: t = 64, tidx: t).  The sixteen message words are received from the SHA-256
// adapter on `msg_bus` as (timestamp, t, W_t limbs) on the rows that hold them, the eight state words as (timestamp, k, H_in[k]
// limbs, H_out[k] limbs) on the digest row -- which ties the block to the 24 memory words the adapter reads and the 8 it rewrites.
// AirBuilder(VM_WIDTH, 0, VM_PREP_WIDTH)
constexpr size_t COL_TS = WIDTH, VM_WIDTH = WIDTH + 1;
constexpr size_t PREP_INPUT = PREP_WIDTH, PREP_DIGEST = PREP_WIDTH + 1, PREP_TIDX = PREP_WIDTH + 2, VM_PREP_WIDTH = PREP_WIDTH + 3;
inline std::vector<uint32_t> prep_trace_vm(unsigned log_height) {
    const size_t N = (size_t)1 << log_height, blocks = N / ROWS_PER_BLOCK;
    const std::vector<uint32_t> base = prep_trace(log_height);
    std::vector<uint32_t> p(VM_PREP_WIDTH * N, 0);
    std::copy(base.begin(), base.end(), p.begin());
    for (size_t b = 0; b < blocks; b++)
        for (size_t t = 0; t < ROWS_PER_BLOCK; t++) {
            const size_t row = b * ROWS_PER_BLOCK + t;
            p[PREP_INPUT * N + row] = t < 16, p[PREP_DIGEST * N + row] = t == 64, p[PREP_TIDX * N + row] = (uint32_t)t;
        }
    return p;
}
inline void compress_vm_air(AirBuilder& b, uint32_t msg_bus, uint32_t state_bus) {
    compress_air(b);
    const Expr ts = b.var(COL_TS), real = b.var(COL_REAL), gate = b.prep(PREP_ROUND) + b.prep(PREP_FINAL);
    b.when_transition(gate * (b.next(COL_TS) - ts));
    auto limb = [&](size_t base, size_t l) {
        Expr s = b.constant(0);
        for (size_t k = 0; k < 16; k++) s = s + b.var(base + 16 * l + k) * (int64_t)(1u << k);
        return s;
    };
    b.push_interaction(msg_bus, {ts, b.prep(PREP_TIDX), limb(COL_W15_BITS, 0), limb(COL_W15_BITS, 1)}, real * b.prep(PREP_INPUT), air::Kind::Receive);
    for (size_t k = 0; k < 8; k++)
        b.push_interaction(state_bus, {ts, b.constant((uint32_t)k), b.var(COL_HIN + 2 * k), b.var(COL_HIN + 2 * k + 1), limb(COL_STATE + 32 * k, 0), limb(COL_STATE + 32 * k, 1)},
                           real * b.prep(PREP_DIGEST), air::Kind::Receive);
}

}  // namespace sha256
}  // namespace zkhip
