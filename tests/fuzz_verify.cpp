// Sanitizer-built (ASan + UBSan) driver for the HOST verifier: feeds zkhip_verify a valid proof and
// thousands of mutated / truncated / extended variants; every variant must be rejected without a
// memory error.  Built and run by tests/test_verifier_fuzz_cpu.py from the product sources
// (zkvm-prover_amd/csrc/verifier.hip compiled as plain C++; it contains no device code).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "zkhip.h"

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    // case file: [5 params][n_airs]{[log_height][width][n_pvs][prog_len][prog...][pvs...][has_commit][commit(8) if has_commit]}
    std::vector<uint8_t> cb = slurp(argv[1]), proof = slurp(argv[2]);
    int iters = atoi(argv[3]);
    const uint32_t* w = (const uint32_t*)cb.data();
    zkhip_params prm{w[0], w[1], w[2], w[3], w[4]};
    size_t n_airs = w[5], p = 6;
    std::vector<zkhip_air> airs(n_airs);
    std::vector<const uint32_t*> pvs(n_airs);
    for (size_t a = 0; a < n_airs; a++) {
        airs[a].log_height = w[p], airs[a].width = w[p + 1], airs[a].n_pvs = w[p + 2], airs[a].program_len = w[p + 3];
        airs[a].program = w + p + 4;
        p += 4 + airs[a].program_len;
        pvs[a] = w + p;
        p += airs[a].n_pvs;
        airs[a].prep_trace = nullptr;
        airs[a].prep_commit = w[p] ? w + p + 1 : nullptr;
        p += w[p] ? 9 : 1;
    }
    if (zkhip_verify(&prm, airs.data(), n_airs, pvs.data(), proof.data(), proof.size()) != 0) {
        std::printf("valid proof rejected\n");
        return 1;
    }
    int accepted = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> m = proof;
        switch (rnd() % 6) {
            case 0: m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;                 // bit flip
            case 1: m.resize(rnd() % m.size()); break;                                          // truncate
            case 2: m.resize(m.size() + 4 * (1 + rnd() % 8), (uint8_t)rnd()); break;            // extend
            case 3: { size_t i = (rnd() % (m.size() / 4)) * 4; uint32_t v = (uint32_t)rnd(); memcpy(&m[i], &v, 4); } break;
            case 4: { size_t i = (rnd() % (m.size() / 4)) * 4; uint32_t v = 0xffffffffu; memcpy(&m[i], &v, 4); } break;  // non-canonical
            default: { size_t a = rnd() % m.size(), b = rnd() % m.size(); std::swap(m[a], m[b]); } break;
        }
        if (m == proof) continue;
        if (zkhip_verify(&prm, airs.data(), n_airs, pvs.data(), m.data(), m.size()) == 0) accepted++;
    }
    // mutated AIR programs (the parser sees caller-supplied words too): must neither crash nor accept
    std::vector<uint32_t> cw(w, w + cb.size() / 4);
    for (int it = 0; it < iters / 4; it++) {
        std::vector<uint32_t> m = cw;
        size_t a = rnd() % n_airs;
        size_t off = (size_t)(airs[a].program - w), len = airs[a].program_len;
        size_t i = off + rnd() % len;
        switch (rnd() % 3) {
            case 0: m[i] ^= 1u << (rnd() % 31); break;
            case 1: m[i] = (uint32_t)rnd(); break;
            default: m[i] = (uint32_t)(rnd() % 64); break;
        }
        if (m[i] == cw[i]) continue;
        std::vector<zkhip_air> airs2 = airs;
        std::vector<const uint32_t*> pvs2(n_airs);
        for (size_t k = 0; k < n_airs; k++) {
            airs2[k].program = m.data() + (airs[k].program - w);
            pvs2[k] = m.data() + (pvs[k] - w);
            if (airs[k].prep_commit) airs2[k].prep_commit = m.data() + (airs[k].prep_commit - w);
        }
        size_t plen = airs2[a].program_len;
        if (rnd() % 8 == 0) airs2[a].program_len = plen - 1 - rnd() % (plen < 9 ? 1 : 8);  // truncated program
        if (zkhip_verify(&prm, airs2.data(), n_airs, pvs2.data(), proof.data(), proof.size()) == 0) accepted++;
    }
    // header words that are not bound by the transcript could in principle be mutated harmlessly; none exist
    std::printf("fuzz done: %d iterations, %d mutated proofs accepted\n", iters, accepted);
    return accepted == 0 ? 0 : 1;
}
