/*
 * oracle/challenger.c -- Fiat-Shamir duplex challenger, PoW grinding and the
 * arity-2 FRI fold.  TEST INFRASTRUCTURE.  The fold is PINNED to the reference (55 fold triples of its stored proofs,
 * tests/test_ref_vectors_cpu.py); the challenger (observe / sample order, grinding) is PARITY UNPINNED vs p3 0.4.3:
 * a transcript cannot be replayed without the verifying key pre-hash.
 *
 * Restates p3-challenger 0.4.3 DuplexChallenger<F, Perm, 16, 8>
 * (Cargo.lock:5576): observe() buffers up to RATE inputs and duplexes when
 * full; duplexing overwrites state[0..n_in], permutes, and refills the output
 * buffer with state[0..8]; sample() duplexes if inputs are pending or the
 * output buffer is empty and pops from the END of the output buffer;
 * sample_bits = low bits of a sampled canonical element; grind(bits) = the
 * first witness w (ascending from 0) such that, after observe(w),
 * sample_bits(bits) == 0.  PoW widths: openvm.toml:5-6.
 * FRI fold = p3-fri TwoAdicFriFolding::fold_row, arity 2.
 */
#include <string.h>
#include "zk_oracle.h"

void ora_ch_init(ora_challenger *c) { memset(c, 0, sizeof *c); }

static void duplex(ora_challenger *c) {
    for (unsigned i = 0; i < c->n_in; i++) c->state[i] = c->in_buf[i];
    c->n_in = 0;
    ora_poseidon2_permute(c->state);
    memcpy(c->out_buf, c->state, 8 * sizeof(uint32_t));
    c->n_out = 8;
}

void ora_ch_observe(ora_challenger *c, const uint32_t *vals, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->n_out = 0; /* any buffered output is invalidated */
        c->in_buf[c->n_in++] = vals[i];
        if (c->n_in == ORA_RATE) duplex(c);
    }
}

uint32_t ora_ch_sample(ora_challenger *c) {
    if (c->n_in != 0 || c->n_out == 0) duplex(c);
    return c->out_buf[--c->n_out];
}

void ora_ch_sample_ext(ora_challenger *c, uint32_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = ora_ch_sample(c);
}

uint32_t ora_ch_sample_bits(ora_challenger *c, unsigned bits) {
    return ora_ch_sample(c) & (uint32_t)(((uint64_t)1 << bits) - 1);
}

int ora_ch_check_witness(ora_challenger *c, unsigned bits, uint32_t witness) {
    ora_ch_observe(c, &witness, 1);
    return ora_ch_sample_bits(c, bits) == 0;
}

uint32_t ora_ch_grind(ora_challenger *c, unsigned bits) {
    for (uint32_t w = 0; w < ORA_P; w++) {
        ora_challenger t = *c;
        if (ora_ch_check_witness(&t, bits, w)) {
            *c = t;
            return w;
        }
    }
    return 0xffffffffu;
}

static size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

/* out[i] = e0 + (beta - x)(e1 - e0)/(-2x), x = w_{h+1}^{bitrev_h(i)}, (e0,e1) = in[2i], in[2i+1] */
void ora_fri_fold(const uint32_t *in, uint32_t *out, unsigned log_n_out, const uint32_t beta[4]) {
    size_t n = (size_t)1 << log_n_out;
    uint32_t g = ora_two_adic_generator(log_n_out + 1);
    uint32_t inv2 = ora_inv(2);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint32_t x = ora_pow(g, bitrev(i, log_n_out));
        uint32_t c = ora_mul(ora_inv(x), inv2);
        c = ora_sub(0, c); /* 1/(-2x) */
        const uint32_t *e0 = in + 8 * i, *e1 = in + 8 * i + 4;
        uint32_t d[4], bx[4], t[4];
        for (int k = 0; k < 4; k++) d[k] = ora_mul(ora_sub(e1[k], e0[k]), c);
        memcpy(bx, beta, 16);
        bx[0] = ora_sub(bx[0], x);
        ora_ext_mul(bx, d, t);
        for (int k = 0; k < 4; k++) out[4 * i + k] = ora_add(e0[k], t[k]);
    }
}
