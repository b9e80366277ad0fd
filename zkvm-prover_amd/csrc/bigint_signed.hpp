// bigint_signed.hpp -- 576-bit signed accumulators for the limb chips' device generators (csrc/ecc.hip, csrc/fp2.hip): sums of 256 x 256-bit
// products with signs, and their division by a 256-bit modulus into a signed quotient and the canonical residue.  One lane per row.
#pragma once
#include <stdint.h>

namespace zk {

constexpr int SW = 18;   // words of a signed accumulator (two's complement)

// v += sign * scale * x * y  (x, y eight words; scale small)
__device__ inline void acc_product(uint32_t* v, const uint32_t* x, const uint32_t* y, int sign, uint32_t scale) {
    uint32_t prod[SW];
    for (int i = 0; i < SW; i++) prod[i] = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)x[i] * y[j] + prod[i + j];
            prod[i + j] = (uint32_t)c, c >>= 32;
        }
        prod[i + 8] = (uint32_t)c;
    }
    if (scale != 1) {
        uint64_t c = 0;
        for (int i = 0; i < SW; i++) c += (uint64_t)prod[i] * scale, prod[i] = (uint32_t)c, c >>= 32;
    }
    if (sign > 0) {
        uint64_t c = 0;
        for (int i = 0; i < SW; i++) c += (uint64_t)v[i] + prod[i], v[i] = (uint32_t)c, c >>= 32;
    } else {
        uint32_t br = 0;
        for (int i = 0; i < SW; i++) {
            const uint64_t d = (uint64_t)v[i] - prod[i] - br;
            v[i] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
        }
    }
}
// v += sign * x  (x eight words)
__device__ inline void acc_word(uint32_t* v, const uint32_t* x, int sign) {
    if (sign > 0) {
        uint64_t c = 0;
        for (int i = 0; i < SW; i++) c += (uint64_t)v[i] + (i < 8 ? x[i] : 0u), v[i] = (uint32_t)c, c >>= 32;
    } else {
        uint32_t br = 0;
        for (int i = 0; i < SW; i++) {
            const uint64_t d = (uint64_t)v[i] - (i < 8 ? x[i] : 0u) - br;
            v[i] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
        }
    }
}
// v (signed) = sgn * (quo * p + rem) with 0 <= rem < p as a magnitude split; then the canonical residue of v and the signed quotient
// with v - res = (neg ? -1 : 1) * q * p.  Returns false if q does not fit 33 bytes.
__device__ inline bool signed_divmod(uint32_t* v, const uint32_t* p, uint32_t* q /*9*/, uint32_t* res /*8*/, bool* neg) {
    *neg = (v[SW - 1] >> 31) != 0;
    if (*neg) {   // magnitude
        uint64_t c = 1;
        for (int i = 0; i < SW; i++) c += (uint64_t)(~v[i]), v[i] = (uint32_t)c, c >>= 32;
    }
    uint32_t rem[9], quo[SW];
    for (int i = 0; i < 9; i++) rem[i] = 0;
    for (int i = 0; i < SW; i++) quo[i] = 0;
    for (int bit = 32 * SW - 1; bit >= 0; bit--) {
        for (int k = 8; k > 0; k--) rem[k] = (rem[k] << 1) | (rem[k - 1] >> 31);
        rem[0] = (rem[0] << 1) | ((v[bit >> 5] >> (bit & 31)) & 1u);
        bool ge = rem[8] != 0;
        if (!ge) {
            ge = true;
            for (int k = 7; k >= 0; k--)
                if (rem[k] != p[k]) {
                    ge = rem[k] > p[k];
                    break;
                }
        }
        if (ge) {
            uint32_t br = 0;
            for (int k = 0; k < 9; k++) {
                const uint64_t d = (uint64_t)rem[k] - (k < 8 ? p[k] : 0u) - br;
                rem[k] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
            }
            quo[bit >> 5] |= 1u << (bit & 31);
        }
    }
    bool rem_zero = true;
    for (int k = 0; k < 8; k++) rem_zero = rem_zero && rem[k] == 0;
    if (*neg && !rem_zero) {   // -(quo p + rem) = -(quo + 1) p + (p - rem)
        uint32_t br = 0;
        for (int k = 0; k < 8; k++) {
            const uint64_t d = (uint64_t)p[k] - rem[k] - br;
            rem[k] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
        }
        uint64_t c = 1;
        for (int i = 0; i < SW; i++) c += quo[i], quo[i] = (uint32_t)c, c >>= 32;
    }
    bool fits = quo[8] < 256u, q_zero = true;
    for (int i = 9; i < SW; i++) fits = fits && quo[i] == 0;
    for (int i = 0; i < 9; i++) q[i] = quo[i], q_zero = q_zero && quo[i] == 0;
    for (int k = 0; k < 8; k++) res[k] = rem[k];
    if (q_zero) *neg = false;   // one representation of zero
    return fits;
}

}  // namespace zk
