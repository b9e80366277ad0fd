// poseidon2_avx512.cpp -- the Poseidon2-BabyBear permutation on the HOST with the whole state in one 512-bit register (plain C++, built
// with -mavx512f and called only after a run-time CPU check).  A duplex sponge is a chain of dependent permutations: absorbing the
// opened values of a proof (4 k words for the base segment circuit -- those stay on the device --, 110 k for the reference's full
// chunk-circuit configuration) takes
// one permutation per 8 words, ~1.5 us each for a cooperative wave on the device -- latency no occupancy can hide -- and ~0.25 us here.
// csrc/transcript.hip hands long absorptions to this file (the words cross PCIe anyway: they are part of the proof) and keeps the short
// ones on the device.  Same algorithm and constants as csrc/poseidon2.hpp (its scalar code is this file's checker:
// tests/test_abi_cpu.py via zkhip_poseidon2_permute_host_avx512); every lane stays canonical in [0, p), so results are bit-identical.
#include <immintrin.h>
#include <stdint.h>

#include "poseidon2.hpp"

namespace zk {
namespace {

struct Tables {
    __m512i rc_ext[8];   // the external rounds' constants, one vector per round
    __m512i diag;        // the internal layer's diagonal, Montgomery form
    uint32_t m15;        // 15, Montgomery form
};

inline __m512i vadd(__m512i a, __m512i b, __m512i vp) {
    const __m512i t = _mm512_add_epi32(a, b);
    return _mm512_min_epu32(t, _mm512_sub_epi32(t, vp));
}
// Montgomery product of 16 lanes: even and odd lanes through the 32 x 32 -> 64 multiplier, q = lo(ab) p^-1, (ab - q p) / 2^32 in (-p, p)
inline __m512i vmul(__m512i a, __m512i b, __m512i vp, __m512i vmu) {
    const __m512i pe = _mm512_mul_epu32(a, b), po = _mm512_mul_epu32(_mm512_srli_epi64(a, 32), _mm512_srli_epi64(b, 32));
    const __m512i me = _mm512_mul_epu32(_mm512_mul_epu32(pe, vmu), vp), mo = _mm512_mul_epu32(_mm512_mul_epu32(po, vmu), vp);
    const __m512i de = _mm512_sub_epi64(pe, me), dd = _mm512_sub_epi64(po, mo);   // low words cancel; the high words hold the signed result
    const __m512i r = _mm512_mask_blend_epi32(0xAAAA, _mm512_srli_epi64(de, 32), dd);
    return _mm512_min_epu32(r, _mm512_add_epi32(r, vp));
}
inline __m512i external_linear(__m512i s, __m512i vp) {
    // per 4-block: 2 x_i + 3 x_{i+1} + x_{i+2} + x_{i+3} (indices mod 4), then every lane gains the sum of its column over the blocks
    const __m512i r1 = _mm512_shuffle_epi32(s, (_MM_PERM_ENUM)_MM_SHUFFLE(0, 3, 2, 1)), r2 = _mm512_shuffle_epi32(s, (_MM_PERM_ENUM)_MM_SHUFFLE(1, 0, 3, 2)),
                  r3 = _mm512_shuffle_epi32(s, (_MM_PERM_ENUM)_MM_SHUFFLE(2, 1, 0, 3));
    const __m512i t = vadd(vadd(s, r1, vp), vadd(r2, r3, vp), vp);
    const __m512i m = vadd(vadd(t, s, vp), vadd(r1, r1, vp), vp);
    const __m512i c1 = _mm512_shuffle_i32x4(m, m, _MM_SHUFFLE(0, 3, 2, 1)), c2 = _mm512_shuffle_i32x4(m, m, _MM_SHUFFLE(1, 0, 3, 2)),
                  c3 = _mm512_shuffle_i32x4(m, m, _MM_SHUFFLE(2, 1, 0, 3));
    return vadd(m, vadd(vadd(m, c1, vp), vadd(c2, c3, vp), vp), vp);
}
inline __m512i lane_sum(__m512i s, __m512i vp) {
    const __m512i r1 = _mm512_shuffle_epi32(s, (_MM_PERM_ENUM)_MM_SHUFFLE(0, 3, 2, 1)), r2 = _mm512_shuffle_epi32(s, (_MM_PERM_ENUM)_MM_SHUFFLE(1, 0, 3, 2)),
                  r3 = _mm512_shuffle_epi32(s, (_MM_PERM_ENUM)_MM_SHUFFLE(2, 1, 0, 3));
    const __m512i a = vadd(vadd(s, r1, vp), vadd(r2, r3, vp), vp);
    const __m512i c1 = _mm512_shuffle_i32x4(a, a, _MM_SHUFFLE(0, 3, 2, 1)), c2 = _mm512_shuffle_i32x4(a, a, _MM_SHUFFLE(1, 0, 3, 2)),
                  c3 = _mm512_shuffle_i32x4(a, a, _MM_SHUFFLE(2, 1, 0, 3));
    return vadd(vadd(a, c1, vp), vadd(c2, c3, vp), vp);
}

const Tables& tables() {
    static const Tables t = [] {
        Tables x;
        alignas(64) uint32_t w[16];
        for (int r = 0; r < 8; r++) {
            for (int i = 0; i < 16; i++) w[i] = Poseidon2Consts::RC[(r < 4 ? 0 : 77 - 64) + r * 16 + i];
            x.rc_ext[r] = _mm512_load_si512(w);
        }
        // diag = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]
        const uint32_t one = MONTY_ONE, two = madd(one, one), half = minv(two);
        auto pw = [&](uint32_t base, unsigned k) {
            uint32_t r = one;
            for (unsigned i = 0; i < k; i++) r = mmul(r, base);
            return r;
        };
        const uint32_t d[16] = {mneg(two), one, two, half, madd(two, one), madd(two, two), mneg(half), mneg(madd(two, one)), mneg(madd(two, two)), pw(half, 8),
                                pw(half, 2), pw(half, 3), pw(half, 27), mneg(pw(half, 8)), mneg(pw(half, 4)), mneg(pw(half, 27))};
        for (int i = 0; i < 16; i++) w[i] = d[i];
        x.diag = _mm512_load_si512(w);
        x.m15 = to_monty(15u);
        return x;
    }();
    return t;
}

// SIXTEEN independent permutations, one per lane: t[16 w + k] = word w of instance k (the states transposed), every register holds one
// state word of all sixteen instances, every operation is lane-wise -- no shuffles, and sixteen results for roughly the instructions of
// two single permutations.  The aggregation witness generator replays a hundred Merkle / FRI queries per child, each a chain of
// permutations of its own: it advances sixteen queries side by side (csrc/recursion.hip).  Same field arithmetic, canonical lanes:
// bit-identical to sixteen calls of the scalar code.
struct Tables16 {
    __m512i rc[141];     // every round constant broadcast
    __m512i diag[16];    // the internal layer's diagonal, broadcast
};
const Tables16& tables16() {
    static const Tables16 t = [] {
        Tables16 x;
        for (int i = 0; i < 141; i++) x.rc[i] = _mm512_set1_epi32((int)Poseidon2Consts::RC[i]);
        const uint32_t one = MONTY_ONE, two = madd(one, one), half = minv(two);
        auto pw = [&](uint32_t base, unsigned k) {
            uint32_t r = one;
            for (unsigned i = 0; i < k; i++) r = mmul(r, base);
            return r;
        };
        const uint32_t d[16] = {mneg(two), one, two, half, madd(two, one), madd(two, two), mneg(half), mneg(madd(two, one)), mneg(madd(two, two)), pw(half, 8),
                                pw(half, 2), pw(half, 3), pw(half, 27), mneg(pw(half, 8)), mneg(pw(half, 4)), mneg(pw(half, 27))};
        for (int i = 0; i < 16; i++) x.diag[i] = _mm512_set1_epi32((int)d[i]);
        return x;
    }();
    return t;
}
inline __m512i vsbox7(__m512i x, __m512i rc, __m512i vp, __m512i vmu) {
    x = vadd(x, rc, vp);
    const __m512i x2 = vmul(x, x, vp, vmu), x3 = vmul(x2, x, vp, vmu), x4 = vmul(x2, x2, vp, vmu);
    return vmul(x3, x4, vp, vmu);
}
inline void external_linear16(__m512i (&s)[16], __m512i vp) {
    for (int b = 0; b < 16; b += 4) {   // M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on each 4-block (poseidon2.hpp p2_external_linear)
        const __m512i x0 = s[b], x1 = s[b + 1], x2 = s[b + 2], x3 = s[b + 3];
        const __m512i t01 = vadd(x0, x1, vp), t23 = vadd(x2, x3, vp), t0123 = vadd(t01, t23, vp);
        const __m512i t01123 = vadd(t0123, x1, vp), t01233 = vadd(t0123, x3, vp);
        s[b + 3] = vadd(vadd(t01233, x0, vp), x0, vp);
        s[b + 1] = vadd(vadd(t01123, x2, vp), x2, vp);
        s[b + 0] = vadd(t01123, t01, vp);
        s[b + 2] = vadd(t01233, t23, vp);
    }
    for (int k = 0; k < 4; k++) {
        const __m512i sum = vadd(vadd(s[k], s[4 + k], vp), vadd(s[8 + k], s[12 + k], vp), vp);
        for (int b = 0; b < 16; b += 4) s[b + k] = vadd(s[b + k], sum, vp);
    }
}

}  // namespace

void poseidon2_permute16_avx512(uint32_t* t) {
    const Tables16& T = tables16();
    const __m512i vp = _mm512_set1_epi32((int)P), vmu = _mm512_set1_epi32((int)MONTY_MU);
    __m512i s[16];
    for (int i = 0; i < 16; i++) s[i] = _mm512_loadu_si512(t + 16 * i);
    external_linear16(s, vp);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = vsbox7(s[i], T.rc[r * 16 + i], vp, vmu);
        external_linear16(s, vp);
    }
    for (int r = 0; r < 13; r++) {
        s[0] = vsbox7(s[0], T.rc[64 + r], vp, vmu);
        __m512i sum = vadd(vadd(vadd(s[0], s[1], vp), vadd(s[2], s[3], vp), vp), vadd(vadd(s[4], s[5], vp), vadd(s[6], s[7], vp), vp), vp);
        sum = vadd(sum, vadd(vadd(vadd(s[8], s[9], vp), vadd(s[10], s[11], vp), vp), vadd(vadd(s[12], s[13], vp), vadd(s[14], s[15], vp), vp), vp), vp);
        for (int i = 0; i < 16; i++) s[i] = vadd(vmul(s[i], T.diag[i], vp, vmu), sum, vp);
    }
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = vsbox7(s[i], T.rc[77 + r * 16 + i], vp, vmu);
        external_linear16(s, vp);
    }
    for (int i = 0; i < 16; i++) _mm512_storeu_si512(t + 16 * i, s[i]);
}

// s: 16 Montgomery words in [0, p), 64-byte alignment not required
void poseidon2_permute_avx512(uint32_t s[16]) {
    const Tables& T = tables();
    const __m512i vp = _mm512_set1_epi32((int)P), vmu = _mm512_set1_epi32((int)MONTY_MU);
    __m512i v = _mm512_loadu_si512(s);
    v = external_linear(v, vp);
    for (int half = 0; half < 2; half++) {
        for (int r = 0; r < 4; r++) {
            const __m512i x = vadd(v, T.rc_ext[4 * half + r], vp);
            const __m512i x2 = vmul(x, x, vp, vmu), x3 = vmul(x2, x, vp, vmu), x4 = vmul(x2, x2, vp, vmu);
            v = external_linear(vmul(x3, x4, vp, vmu), vp);
        }
        if (half == 1) break;
        // internal rounds: lane 0 (the only one with an S-box) lives in a scalar, lanes 1..15 in the vector with lane 0 held at zero.
        // The other lanes' diagonal products and their sum do not wait for the S-box; the sum of lanes 1..15 of the NEXT state is
        // 15 total + sum(d_i s_i), so the dependent chain of a round is the scalar S-box and three additions.
        uint32_t s0 = (uint32_t)_mm_cvtsi128_si32(_mm512_castsi512_si128(v));
        __m512i w = _mm512_maskz_mov_epi32(0xFFFE, v);
        uint32_t rest = (uint32_t)_mm_cvtsi128_si32(_mm512_castsi512_si128(lane_sum(w, vp)));
        for (int r = 0; r < 13; r++) {
            const __m512i dm = vmul(w, T.diag, vp, vmu);   // lane 0 stays zero
            const uint32_t dsum = (uint32_t)_mm_cvtsi128_si32(_mm512_castsi512_si128(lane_sum(dm, vp)));
            const uint32_t y = sbox7_rc(s0, Poseidon2Consts::RC[64 + r]);
            const uint32_t total = madd(rest, y);
            s0 = msub(total, mdouble(y));                    // total + (-2) y
            w = _mm512_maskz_mov_epi32(0xFFFE, vadd(dm, _mm512_set1_epi32((int)total), vp));
            rest = madd(mmul(total, T.m15), dsum);
        }
        v = _mm512_mask_set1_epi32(w, 1, (int)s0);
    }
    _mm512_storeu_si512(s, v);
}

}  // namespace zk
