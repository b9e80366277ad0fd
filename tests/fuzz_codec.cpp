// Sanitizer-built (ASan + UBSan) driver for the v1 proof-container codec (include/zkhip_codec.hpp through
// zkvm-prover_amd/csrc/codec.hip compiled as plain C++): the reference's stored proof blob and thousands of mutated /
// truncated / extended variants go through zkhip_proof_decode_v1 and zkhip_proof_reencode_v1; malformed input must be
// rejected (or, if it still parses, re-encode to exactly the mutated input) without a memory error.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "zkhip.h"

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const int iters = atoi(argv[2]);
    zkhip_v1_summary s;
    if (zkhip_proof_decode_v1(blob.data(), blob.size(), ZKHIP_V1_VEC, &s) != ZKHIP_OK) {
        std::printf("the unmodified container does not decode\n");
        return 1;
    }
    std::vector<uint8_t> out(blob.size() + 4096);
    size_t n = 0;
    if (zkhip_proof_reencode_v1(blob.data(), blob.size(), ZKHIP_V1_VEC, out.data(), out.size(), &n) != ZKHIP_OK || n != blob.size() ||
        memcmp(out.data(), blob.data(), n)) {
        std::printf("round trip of the unmodified container failed\n");
        return 1;
    }
    int accepted = 0, roundtrip_bad = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> m = blob;
        switch (rnd() % 5) {
            case 0: m.resize(rnd() % m.size()); break;                                      // truncate
            case 1: m.insert(m.end(), (size_t)(rnd() % 64) + 1, (uint8_t)rnd()); break;      // extend
            case 2: {                                                                        // corrupt a length prefix region
                size_t pos = (size_t)(rnd() % 4096);
                if (pos < m.size()) m[pos] ^= (uint8_t)(1u << (rnd() % 8));
            } break;
            case 3: {                                                                        // random byte anywhere
                size_t pos = (size_t)(rnd() % m.size());
                m[pos] = (uint8_t)rnd();
            } break;
            default: {                                                                       // huge length
                size_t pos = (size_t)(rnd() % (m.size() - 8)) & ~(size_t)3;
                uint64_t big = rnd();
                memcpy(&m[pos], &big, 8);
            } break;
        }
        const int kind = (rnd() & 7) == 0 ? ZKHIP_V1_SINGLE : ZKHIP_V1_VEC;
        int rc = zkhip_proof_decode_v1(m.data(), m.size(), kind, &s);
        if (rc == ZKHIP_OK) {
            accepted++;
            std::vector<uint8_t> o2(m.size() + 64);
            size_t n2 = 0;
            int r2 = zkhip_proof_reencode_v1(m.data(), m.size(), kind, o2.data(), o2.size(), &n2);
            if (r2 != ZKHIP_OK || n2 != m.size() || memcmp(o2.data(), m.data(), n2)) roundtrip_bad++;
        }
    }
    std::printf("%d variants, %d still well-formed, %d round-trip mismatches\n", iters, accepted, roundtrip_bad);
    return roundtrip_bad ? 1 : 0;
}
