// hasher_demo -- the whole host side in C++ over the C ABI, no Python: a requester chip asks for 2-to-1 Poseidon2
// compressions over a 24-field bus, the Poseidon2 chip serves them.  AIRs from include/zkhip_air.hpp, the Poseidon2
// chip's trace generated on the device (zkhip_poseidon2_air_tracegen), the digests read back from that trace to fill
// the requester's trace, then zkhip_keygen / zkhip_prove / zkhip_verify.
//   hasher_demo [log_rows]        (default 12; both chips get 2^log_rows rows, all but 5 of them real requests)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "zkhip.h"
#include "zkhip_air.hpp"

using namespace zkhip::air;

#define CK(expr)                                                                              \
    do {                                                                                      \
        int _rc = (expr);                                                                     \
        if (_rc != 0) {                                                                       \
            std::fprintf(stderr, "%s failed (%d): %s\n", #expr, _rc, zkhip_last_error(ctx));  \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

int main(int argc, char** argv) {
    const unsigned lh = argc > 1 ? (unsigned)atoi(argv[1]) : 12;
    const size_t N = (size_t)1 << lh, n_req = N > 5 ? N - 5 : 1;
    const uint32_t BUS = 9;
    zkhip_ctx* ctx = nullptr;
    if (zkhip_ctx_create(0, &ctx) != 0) {
        std::fprintf(stderr, "no gfx950 device\n");
        return 2;
    }
    // AIR descriptions
    AirBuilder user(25, 0), chip(POSEIDON2_AIR_WIDTH + 1, 0);
    {
        const Expr real = user.var(24);
        user.assert_zero(real * (real - 1));
        std::vector<Expr> msg;
        for (int i = 0; i < 24; i++) msg.push_back(user.var(i));
        user.push_interaction(BUS, msg, real, Kind::Send);
    }
    poseidon2_air(chip, (int)BUS);
    const std::vector<uint32_t> p_user = user.program(), p_chip = chip.program();
    const zkhip_params prm{1, 0, 100, 16, 16};
    const zkhip_air airs[2] = {{p_user.data(), p_user.size(), lh, 25, 0, nullptr, nullptr},
                               {p_chip.data(), p_chip.size(), lh, POSEIDON2_AIR_WIDTH + 1, 0, nullptr, nullptr}};
    auto t0 = std::chrono::steady_clock::now();
    zkhip_pk* pk = nullptr;
    CK(zkhip_keygen(ctx, &prm, airs, 2, &pk));
    auto t1 = std::chrono::steady_clock::now();

    // requests: left ++ right per row, canonical, pseudo-random
    std::vector<uint32_t> in(n_req * 16);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (uint32_t& x : in) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        x = (uint32_t)(s % P);
    }
    uint32_t *d_in = nullptr, *d_chip = nullptr, *d_user = nullptr;
    CK(zkhip_malloc(ctx, in.size() * 4, (void**)&d_in));
    CK(zkhip_malloc(ctx, (POSEIDON2_AIR_WIDTH + 1) * N * 4, (void**)&d_chip));
    CK(zkhip_malloc(ctx, 25 * N * 4, (void**)&d_user));
    CK(zkhip_h2d(ctx, d_in, in.data(), in.size() * 4));
    CK(zkhip_to_monty(ctx, d_in, in.size()));
    auto t2 = std::chrono::steady_clock::now();
    CK(zkhip_poseidon2_air_tracegen(ctx, d_in, n_req, lh, d_chip));
    CK(zkhip_sync(ctx));
    auto t3 = std::chrono::steady_clock::now();
    // multiplicity column of the chip: 1 for served rows
    std::vector<uint32_t> mult(N, 0);
    for (size_t r = 0; r < n_req; r++) mult[r] = 1;
    CK(zkhip_h2d(ctx, d_chip + POSEIDON2_AIR_WIDTH * N, mult.data(), N * 4));
    CK(zkhip_to_monty(ctx, d_chip + POSEIDON2_AIR_WIDTH * N, N));
    // requester trace, column-major: left[8] right[8] = the chip's input columns, out[8] = its first 8 output columns
    // (device-to-device would do; the demo goes through the host to stay within the C ABI's copy calls)
    std::vector<uint32_t> col(N);
    for (int c = 0; c < 24; c++) {
        const size_t src = c < 16 ? (size_t)c : POSEIDON2_AIR_WIDTH - 16 + (size_t)(c - 16);
        CK(zkhip_d2h(ctx, col.data(), d_chip + src * N, N * 4));
        for (size_t r = n_req; r < N; r++) col[r] = 0;  // padding rows send nothing
        CK(zkhip_h2d(ctx, d_user + (size_t)c * N, col.data(), N * 4));
    }
    CK(zkhip_h2d(ctx, d_user + 24 * N, mult.data(), N * 4));
    CK(zkhip_to_monty(ctx, d_user + 24 * N, N));

    const uint32_t* traces[2] = {d_user, d_chip};
    const uint32_t* pvs[2] = {nullptr, nullptr};
    std::vector<uint8_t> proof(zkhip_proof_size(pk));
    size_t len = 0;
    CK(zkhip_prove(ctx, pk, traces, pvs, proof.data(), proof.size(), &len));  // first proof: code objects, scratch
    auto t4 = std::chrono::steady_clock::now();
    CK(zkhip_prove(ctx, pk, traces, pvs, proof.data(), proof.size(), &len));
    auto t5 = std::chrono::steady_clock::now();
    const int v = zkhip_verify(&prm, airs, 2, pvs, proof.data(), len);
    auto t6 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::printf("hasher_demo 2^%u rows: keygen %.0f ms, tracegen %.3f ms, prove %.2f ms (first %.0f ms), verify %.1f ms, %zu proof bytes: %s\n",
                lh, ms(t0, t1), ms(t2, t3), ms(t4, t5), ms(t3, t4), ms(t5, t6), len, v == 0 ? "verified" : "REJECTED");
    // a tampered request must not verify: flip one digest word of the requester and prove again
    uint32_t w = 0;
    CK(zkhip_d2h(ctx, &w, d_user + 16 * N, 4));
    w ^= 1;
    CK(zkhip_h2d(ctx, d_user + 16 * N, &w, 4));
    CK(zkhip_prove(ctx, pk, traces, pvs, proof.data(), proof.size(), &len));
    const int v2 = zkhip_verify(&prm, airs, 2, pvs, proof.data(), len);
    std::printf("tampered digest: %s\n", v2 != 0 ? "rejected" : "ACCEPTED");
    zkhip_free(ctx, d_in), zkhip_free(ctx, d_chip), zkhip_free(ctx, d_user);
    zkhip_pk_destroy(ctx, pk);
    zkhip_ctx_destroy(ctx);
    return v == 0 && v2 != 0 ? 0 : 1;
}
