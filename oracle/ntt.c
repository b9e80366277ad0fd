/*
 * oracle/ntt.c -- radix-2 DFT over BabyBear two-adic subgroups and coset LDE.
 * TEST INFRASTRUCTURE.  PARITY UNPINNED vs p3-dft 0.4.3 (Cargo.lock:5590).
 *
 * Definition restated from p3-dft's TwoAdicSubgroupDft:
 *   dft_batch:  out[i] = sum_j in[j] * g^(i*j),  g = two_adic_generator(log n),
 *               natural order in and out, every column independently;
 *   coset_lde_batch(evals, added_bits, shift): coeffs = idft(evals);
 *               coeffs[i] *= shift^i; zero-pad to n << added_bits; dft.
 * p3 stores committed LDEs with rows in bit-reversed order
 * (TwoAdicFriPcs::commit); `bitrev_out` selects that layout.
 * Field results are unique, so "bit-exact vs p3-dft" = this definition.
 */
#include <stdlib.h>
#include <string.h>
#include "zk_oracle.h"

static size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

void ora_dft_naive(const uint32_t *in, uint32_t *out, unsigned log_n, int inverse) {
    size_t n = (size_t)1 << log_n;
    uint32_t g = ora_two_adic_generator(log_n);
    if (inverse) g = ora_inv(g);
    uint32_t ninv = inverse ? ora_inv((uint32_t)(n % ORA_P)) : 1;
    for (size_t i = 0; i < n; i++) {
        uint32_t gi = ora_pow(g, i), acc = 0, w = 1;
        for (size_t j = 0; j < n; j++) {
            acc = ora_add(acc, ora_mul(in[j], w));
            w = ora_mul(w, gi);
        }
        out[i] = ora_mul(acc, ninv);
    }
}

/* in-place iterative decimation-in-time on one column, natural in / natural out */
static void dft_col(uint32_t *a, unsigned log_n, const uint32_t *tw /* g^k, k < n/2 */) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, log_n);
        if (i < j) {
            uint32_t t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    }
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, half = m >> 1, step = n >> s;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                uint32_t w = tw[j * step];
                uint32_t u = a[k + j], v = ora_mul(a[k + j + half], w);
                a[k + j] = ora_add(u, v);
                a[k + j + half] = ora_sub(u, v);
            }
    }
}

static uint32_t *make_twiddles(unsigned log_n, int inverse) {
    size_t n = (size_t)1 << log_n, h = n > 1 ? n / 2 : 1;
    uint32_t *tw = (uint32_t *)malloc(h * sizeof(uint32_t));
    uint32_t g = ora_two_adic_generator(log_n);
    if (inverse) g = ora_inv(g);
    uint32_t w = 1;
    for (size_t i = 0; i < h; i++) {
        tw[i] = w;
        w = ora_mul(w, g);
    }
    return tw;
}

void ora_dft_batch(uint32_t *mat, unsigned log_n, size_t width, size_t stride, int inverse) {
    size_t n = (size_t)1 << log_n;
    uint32_t *tw = make_twiddles(log_n, inverse);
    uint32_t ninv = ora_inv((uint32_t)(n % ORA_P));
#pragma omp parallel for schedule(dynamic) if (width > 1)
    for (size_t c = 0; c < width; c++) {
        uint32_t *col = mat + c * stride;
        dft_col(col, log_n, tw);
        if (inverse)
            for (size_t i = 0; i < n; i++) col[i] = ora_mul(col[i], ninv);
    }
    free(tw);
}

void ora_coset_lde_batch(const uint32_t *in, size_t in_stride, uint32_t *out, size_t out_stride,
                         unsigned log_n, unsigned added_bits, size_t width, uint32_t shift,
                         int bitrev_out) {
    size_t n = (size_t)1 << log_n, m = n << added_bits;
    unsigned log_m = log_n + added_bits;
    uint32_t *twi = make_twiddles(log_n, 1), *twf = make_twiddles(log_m, 0);
    uint32_t ninv = ora_inv((uint32_t)(n % ORA_P));
#pragma omp parallel for schedule(dynamic) if (width > 1)
    for (size_t c = 0; c < width; c++) {
        uint32_t *buf = (uint32_t *)malloc(m * sizeof(uint32_t));
        memcpy(buf, in + c * in_stride, n * sizeof(uint32_t));
        dft_col(buf, log_n, twi);
        uint32_t sp = ninv;
        for (size_t i = 0; i < n; i++) {
            buf[i] = ora_mul(buf[i], sp);
            sp = ora_mul(sp, shift);
        }
        memset(buf + n, 0, (m - n) * sizeof(uint32_t));
        dft_col(buf, log_m, twf);
        uint32_t *o = out + c * out_stride;
        if (bitrev_out)
            for (size_t r = 0; r < m; r++) o[r] = buf[bitrev(r, log_m)];
        else
            memcpy(o, buf, m * sizeof(uint32_t));
        free(buf);
    }
    free(twi);
    free(twf);
}
