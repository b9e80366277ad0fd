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
#include "hist.hpp"
#include "poseidon2_coop.hpp"
#include "zkhip_internal.hpp"

#include <algorithm>

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

// Second chip: the multiplicity column of a range-check / lookup table (the whole trace of OpenVM's
// VariableRangeChecker-style chips: how often each table entry was requested; the reference's CPU tracegen bumps
// atomic counters while the other chips fill their rows, its GPU backend does the same with device atomics).
// counts[v] (+)= #{ i : values[i] == v } for v < 2^log_table; values and counts are Montgomery residues.
// Tables of <= 2^13 entries are counted in a per-workgroup LDS histogram (32 KiB) and merged with one atomic per
// non-zero bin; larger tables count straight into HBM.  `bad` receives the number of out-of-range values.
constexpr unsigned RC_LDS_LOG = 13;
__global__ __launch_bounds__(256) void k_range_counts(const uint32_t* __restrict__ values, size_t n, unsigned log_table,
                                                      uint32_t* __restrict__ hist, uint32_t* __restrict__ bad) {
    extern __shared__ uint32_t bins[];
    const uint32_t T = 1u << log_table;
    const bool use_lds = log_table <= RC_LDS_LOG;
    if (use_lds) {
        for (uint32_t i = threadIdx.x; i < T; i += 256) bins[i] = 0;
        __syncthreads();
    }
    uint32_t n_bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t v = from_monty(values[i]);
        if (v >= T) {
            n_bad++;
            continue;
        }
        hist_add(use_lds ? bins : hist, v);
    }
    if (n_bad) atomicAdd(bad, n_bad);
    if (use_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < T; i += 256)
            if (bins[i]) atomicAdd(&hist[i], bins[i]);
    }
}
// integer histogram <-> Montgomery counts (counts stay far below p: at most 2^27 rows x a few hundred columns per call,
// and the sum is reduced mod p, which is what the bus argument needs anyway)
__global__ void k_counts_repr(uint32_t* c, size_t n, int to_m) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = to_m ? to_monty(c[i] % P) : from_monty(c[i]);
}

int range_counts_tracegen(zkhip_ctx* ctx, const uint32_t* d_values, size_t n, unsigned log_table, uint32_t* d_counts, int accumulate) {
    if (log_table > 27) return set_error(ctx, ZKHIP_ERR_INVALID, "range_counts_tracegen: log_table > 27");
    const size_t T = (size_t)1 << log_table;
    void* flag = nullptr;
    ZK_TRY(get_scratch(ctx, 2, 16, &flag));
    KernelScope ks(ctx, "range_counts_tracegen");
    ZK_HIP_CHECK(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (accumulate) hipLaunchKernelGGL(k_counts_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 0);
    else ZK_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, T * 4, ctx->stream));
    if (n) {
        const unsigned blocks = (unsigned)std::min<size_t>((n + 256 * 16 - 1) / (256 * 16), 4096);
        hipLaunchKernelGGL(k_range_counts, dim3(blocks), dim3(256), log_table <= RC_LDS_LOG ? T * 4 : 0, ctx->stream, d_values, n,
                           log_table, d_counts, (uint32_t*)flag);
    }
    hipLaunchKernelGGL(k_counts_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    uint32_t h_bad = 0;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(&h_bad, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_bad) return set_error(ctx, ZKHIP_ERR_INVALID, "range_counts_tracegen: " + std::to_string(h_bad) + " values outside the table");
    return ZKHIP_OK;
}

}  // namespace zk

using namespace zk;
extern "C" int zkhip_range_counts_tracegen(zkhip_ctx* ctx, const uint32_t* d_values, size_t n, unsigned log_table,
                                           uint32_t* d_counts, int accumulate) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_counts || (n && !d_values)) return ZKHIP_ERR_INVALID;
    return range_counts_tracegen(ctx, d_values, n, log_table, d_counts, accumulate);
}
extern "C" int zkhip_poseidon2_air_tracegen(zkhip_ctx* ctx, const uint32_t* d_inputs, size_t n_perms, unsigned log_height,
                                            uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace) return ZKHIP_ERR_INVALID;
    return poseidon2_air_tracegen(ctx, d_inputs, n_perms, log_height, d_trace);
}
