// bigint_signed.hpp -- signed accumulators for the limb chips' device generators (csrc/ecc.hip, csrc/fp2.hip): sums of products of two
// NW-word operands with signs (2 NW + 2 words, two's complement: 576 bits for 256-bit operands, 832 for 384-bit ones), and their division
// by an NW-word modulus into a signed quotient and the canonical residue.  One lane per row.
#pragma once
#include <stdint.h>

namespace zk {

template <int NW>
struct Signed {
    static constexpr int SW = 2 * NW + 2;   // words of a signed accumulator

    // v += sign * scale * x * y  (x, y NW words; scale small)
    __device__ static inline void acc_product(uint32_t* v, const uint32_t* x, const uint32_t* y, int sign, uint32_t scale) {
        uint32_t prod[SW];
        for (int i = 0; i < SW; i++) prod[i] = 0;
        for (int i = 0; i < NW; i++) {
            uint64_t c = 0;
            for (int j = 0; j < NW; j++) {
                c += (uint64_t)x[i] * y[j] + prod[i + j];
                prod[i + j] = (uint32_t)c, c >>= 32;
            }
            prod[i + NW] = (uint32_t)c;
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
    // v += sign * x  (x NW words)
    __device__ static inline void acc_word(uint32_t* v, const uint32_t* x, int sign) {
        if (sign > 0) {
            uint64_t c = 0;
            for (int i = 0; i < SW; i++) c += (uint64_t)v[i] + (i < NW ? x[i] : 0u), v[i] = (uint32_t)c, c >>= 32;
        } else {
            uint32_t br = 0;
            for (int i = 0; i < SW; i++) {
                const uint64_t d = (uint64_t)v[i] - (i < NW ? x[i] : 0u) - br;
                v[i] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
            }
        }
    }
    // v (signed) = sgn * (quo * p + rem) with 0 <= rem < p as a magnitude split; then the canonical residue of v and the signed quotient
    // with v - res = (neg ? -1 : 1) * q * p.  Returns false if q does not fit 4 NW + 1 bytes.
    __device__ static inline bool signed_divmod(uint32_t* v, const uint32_t* p, uint32_t* q /*NW + 1*/, uint32_t* res /*NW*/, bool* neg) {
        *neg = (v[SW - 1] >> 31) != 0;
        if (*neg) {   // magnitude
            uint64_t c = 1;
            for (int i = 0; i < SW; i++) c += (uint64_t)(~v[i]), v[i] = (uint32_t)c, c >>= 32;
        }
        uint32_t rem[NW + 1], quo[SW];
        for (int i = 0; i <= NW; i++) rem[i] = 0;
        for (int i = 0; i < SW; i++) quo[i] = 0;
        for (int bit = 32 * SW - 1; bit >= 0; bit--) {
            for (int k = NW; k > 0; k--) rem[k] = (rem[k] << 1) | (rem[k - 1] >> 31);
            rem[0] = (rem[0] << 1) | ((v[bit >> 5] >> (bit & 31)) & 1u);
            bool ge = rem[NW] != 0;
            if (!ge) {
                ge = true;
                for (int k = NW - 1; k >= 0; k--)
                    if (rem[k] != p[k]) {
                        ge = rem[k] > p[k];
                        break;
                    }
            }
            if (ge) {
                uint32_t br = 0;
                for (int k = 0; k <= NW; k++) {
                    const uint64_t d = (uint64_t)rem[k] - (k < NW ? p[k] : 0u) - br;
                    rem[k] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
                }
                quo[bit >> 5] |= 1u << (bit & 31);
            }
        }
        bool rem_zero = true;
        for (int k = 0; k < NW; k++) rem_zero = rem_zero && rem[k] == 0;
        if (*neg && !rem_zero) {   // -(quo p + rem) = -(quo + 1) p + (p - rem)
            uint32_t br = 0;
            for (int k = 0; k < NW; k++) {
                const uint64_t d = (uint64_t)p[k] - rem[k] - br;
                rem[k] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
            }
            uint64_t c = 1;
            for (int i = 0; i < SW; i++) c += quo[i], quo[i] = (uint32_t)c, c >>= 32;
        }
        bool fits = quo[NW] < 256u, q_zero = true;
        for (int i = NW + 1; i < SW; i++) fits = fits && quo[i] == 0;
        for (int i = 0; i <= NW; i++) q[i] = quo[i], q_zero = q_zero && quo[i] == 0;
        for (int k = 0; k < NW; k++) res[k] = rem[k];
        if (q_zero) *neg = false;   // one representation of zero
        return fits;
    }
};

}  // namespace zk
