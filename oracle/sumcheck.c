/*
 * oracle/sumcheck.c -- CPU restatement of the LogUp / sum-check building blocks (K6, K7).
 * TEST INFRASTRUCTURE.  PARITY UNPINNED: these are the textbook operations under the pinned
 * backend's LogUp-GKR and sum-check provers (openvm-stark-backend 2.0.0, Cargo.lock:5337; SURVEY.md
 * Appendix C); the variable order (lowest variable = adjacent pairs) is this repository's choice.
 *   batch inverse : out[i] = in[i]^-1 (each computed independently here)
 *   running sum   : out[i] = sum_{j<=i} num[j]/den[j]   (LogUp fractional sum)
 *   mle fold      : out[i] = in[2i] + r (in[2i+1] - in[2i])
 *   sumcheck round: s(t) = sum_i prod_j (f_j[2i] + t (f_j[2i+1] - f_j[2i])), t = 0..k
 */
#include <string.h>
#include "zk_oracle.h"

void ora_ext_batch_inverse(const uint32_t *in, uint32_t *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) ora_ext_inv(in + 4 * i, out + 4 * i);
}

void ora_logup_running_sum(const uint32_t *den, const uint32_t *num, size_t n, uint32_t *out) {
    uint32_t acc[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        uint32_t inv[4];
        ora_ext_inv(den + 4 * i, inv);
        for (int k = 0; k < 4; k++) {
            acc[k] = ora_add(acc[k], ora_mul(inv[k], num[i]));
            out[4 * i + k] = acc[k];
        }
    }
}

void ora_mle_fold(const uint32_t *in, uint32_t *out, size_t n, const uint32_t r[4]) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint32_t d[4], t[4];
        for (int k = 0; k < 4; k++) d[k] = ora_sub(in[8 * i + 4 + k], in[8 * i + k]);
        ora_ext_mul(r, d, t);
        for (int k = 0; k < 4; k++) out[4 * i + k] = ora_add(in[8 * i + k], t[k]);
    }
}

void ora_sumcheck_round(const uint32_t *const *tables, size_t k, size_t n_half, uint32_t *out) {
    memset(out, 0, (k + 1) * 16);
    for (size_t i = 0; i < n_half; i++)
        for (size_t t = 0; t <= k; t++) {
            uint32_t prod[4] = {1, 0, 0, 0};
            for (size_t j = 0; j < k; j++) {
                const uint32_t *a = tables[j] + 8 * i, *b = a + 4;
                uint32_t v[4];
                for (int q = 0; q < 4; q++) v[q] = ora_add(a[q], ora_mul(ora_sub(b[q], a[q]), (uint32_t)t));
                ora_ext_mul(prod, v, prod);
            }
            for (int q = 0; q < 4; q++) out[4 * t + q] = ora_add(out[4 * t + q], prod[q]);
        }
}
