// tracegen.hip -- device-side trace generation (the step immediately before the proving path; the
// reference's GPU backend fills chip traces on the device so that the multi-GB trace never crosses PCIe,
// AGENTS.md:183-187).
//
// First chip: the Poseidon2 AIR (one permutation per row, the structure of p3-poseidon2-air 0.4.3 as OpenVM
// instantiates it for BabyBear: width 16, x^7 with one committed register x^3 per S-box; Cargo.lock p3-poseidon2-air /
// openvm-poseidon2-air).  Column layout = zkvm-prover_amd/air.py poseidon2_air() = oracle/poseidon2.c
// ora_poseidon2_air_trace:
//   inputs[16] | 4 x { sbox[16], post[16] } | 13 x { sbox, post_sbox } | 4 x { sbox[16], post[16] }      (298 columns)
//
// One row per lane: the 298 stores of a wave are 298 contiguous 256-byte runs (column-major trace), the 16 input words
// of a row are four 16-byte loads.  HBM-bound on the writes: 1192 B per row written, 64 B read.
#include "poseidon2.hpp"
#include "poseidon2_coop.hpp"
#include "zkhip_internal.hpp"

namespace zk {

static_assert(ZKHIP_POSEIDON2_AIR_WIDTH == 16 + 8 * 32 + 13 * 2, "column layout");

__global__ __launch_bounds__(256) void k_poseidon2_air_trace(const uint32_t* __restrict__ inputs, size_t n_perms,
                                                             size_t N, uint32_t* __restrict__ trace) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t s[16];
    if (r < n_perms) {
        const uint4* p = reinterpret_cast<const uint4*>(inputs + r * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = p[q];
            s[4 * q] = v.x, s[4 * q + 1] = v.y, s[4 * q + 2] = v.z, s[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = 0;
    }
    typedef const __attribute__((address_space(4))) uint32_t* cptr;
    const cptr rc = (cptr)POSEIDON2_RC_CONST;  // scalar loads
    uint32_t* out = trace + r;  // column c lives at out[c * N]
#pragma unroll
    for (int i = 0; i < 16; i++) out[(size_t)i * N] = s[i];
    out += 16 * N;
    p2_external_linear(s);
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll 1
        for (int rd = 0; rd < 4; rd++) {
            const cptr k = rc + (half ? 77 : 0) + rd * 16;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t y = madd(s[i], k[i]);
                const uint32_t y3 = mmul(mmul(y, y), y);
                out[(size_t)i * N] = y3;
                s[i] = mmul(mmul(y3, y3), y);
            }
            p2_external_linear(s);
#pragma unroll
            for (int i = 0; i < 16; i++) out[(size_t)(16 + i) * N] = s[i];
            out += 32 * N;
        }
        if (half == 0) {
#pragma unroll 1
            for (int rd = 0; rd < 13; rd++) {
                const uint32_t y = madd(s[0], rc[64 + rd]);
                const uint32_t y3 = mmul(mmul(y, y), y);
                s[0] = mmul(mmul(y3, y3), y);
                out[0] = y3;
                out[N] = s[0];
                out += 2 * N;
                p2_internal_linear(s);
            }
        }
    }
}

int poseidon2_air_tracegen(zkhip_ctx* ctx, const uint32_t* d_inputs, size_t n_perms, unsigned log_height, uint32_t* d_trace) {
    if (log_height > 27) return set_error(ctx, ZKHIP_ERR_INVALID, "poseidon2_air_tracegen: log_height > 27");
    const size_t N = (size_t)1 << log_height;
    if (n_perms > N) return set_error(ctx, ZKHIP_ERR_INVALID, "poseidon2_air_tracegen: more permutations than rows");
    if (n_perms && !d_inputs) return set_error(ctx, ZKHIP_ERR_INVALID, "poseidon2_air_tracegen: null inputs");
    KernelScope ks(ctx, "poseidon2_air_tracegen");
    hipLaunchKernelGGL(k_poseidon2_air_trace, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_inputs, n_perms, N,
                       d_trace);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace zk

using namespace zk;
extern "C" int zkhip_poseidon2_air_tracegen(zkhip_ctx* ctx, const uint32_t* d_inputs, size_t n_perms, unsigned log_height,
                                            uint32_t* d_trace) {
    if (!ctx || !d_trace) return ZKHIP_ERR_INVALID;
    return poseidon2_air_tracegen(ctx, d_inputs, n_perms, log_height, d_trace);
}
