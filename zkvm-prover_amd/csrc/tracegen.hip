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
#include "lds_barrier.hpp"
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
// Every workgroup counts the values below 2^13 in an LDS histogram (32 KiB) and merges it with one atomic per non-zero bin;
// larger values go straight to HBM through the wave-aggregated increment.  The split is by VALUE, not by table size: what
// makes a global histogram slow is a hot bin, and the hot bins of range-check columns are the small values (timestamp gaps,
// carries, high limbs) -- a 2^16-entry table fed with a guest's memory log took 0.83 ms per 2.8 M values while all 11 k waves
// sent their aggregated increments to the same few addresses.
__global__ __launch_bounds__(256) void k_range_counts(const uint32_t* __restrict__ values, size_t n, unsigned log_table,
                                                      uint32_t* __restrict__ hist, uint32_t* __restrict__ bad) {
    __shared__ uint32_t bins[1u << RC_LDS_LOG];
    const uint32_t T = 1u << log_table, L = T < (1u << RC_LDS_LOG) ? T : (1u << RC_LDS_LOG);
    for (uint32_t i = threadIdx.x; i < L; i += 256) bins[i] = 0;
    zk_syncthreads();
    uint32_t n_bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t v = from_monty(values[i]);
        if (v >= T) {
            n_bad++;
            continue;
        }
        if (v < L) atomicAdd(&bins[v], 1u);
        else hist_add(hist, v);
    }
    if (n_bad) atomicAdd(bad, n_bad);
    zk_syncthreads();
    for (uint32_t i = threadIdx.x; i < L; i += 256)
        if (bins[i]) atomicAdd(&hist[i], bins[i]);
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
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "range_counts_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!accumulate) ZK_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, T * 4, ctx->stream));
    else if (!ctx->tables_canonical) hipLaunchKernelGGL(k_counts_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 0);
    if (n) {
        const unsigned blocks = (unsigned)std::min<size_t>((n + 256 * 32 - 1) / (256 * 32), 1024);
        hipLaunchKernelGGL(k_range_counts, dim3(blocks), dim3(256), 0, ctx->stream, d_values, n, log_table, d_counts, (uint32_t*)flag);
    }
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_counts_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return tracegen_finish(ctx, flag, "range_counts_tracegen (values outside the table)");
}

// ---- MMCS path chip: in-circuit verification of mixed-height Merkle openings (air.py mmcs_path_air) ------------------------------
// One thread per path.  Pass 1 climbs from the leaf (the compressions are the hashes this chip asks the Poseidon2 chip to
// prove, so the generator computes them once and hands a || b of every row to zkhip_poseidon2_air_tracegen) and writes each
// row at its final place -- rows run from the root DOWN --; pass 2 descends again with the root and the position counters.
__global__ __launch_bounds__(64) void k_mmcs_path(const uint32_t* __restrict__ leaf, const uint32_t* __restrict__ index,
                                                  const uint32_t* __restrict__ path_start, const uint32_t* __restrict__ step_kind,
                                                  const uint32_t* __restrict__ step_digest, size_t n_paths, size_t N, uint32_t* __restrict__ trace,
                                                  uint32_t* __restrict__ hash_inputs, uint32_t* __restrict__ bad) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= n_paths) return;
    const size_t s0 = path_start[p], s1 = path_start[p + 1], ns = s1 - s0;
    if (s1 > N || s1 <= s0 || step_kind[s0] != 0) {
        atomicAdd(bad, 1u);
        return;
    }
    const uint32_t q = index[p];
    uint32_t st[16], node[8];
#pragma unroll
    for (int i = 0; i < 8; i++) node[i] = to_monty(leaf[8 * p + i]);
    unsigned sibs = 0;
    for (size_t j = 0; j < ns; j++) {
        const bool inj = step_kind[s0 + j] != 0;
        const uint32_t bit = inj ? 0u : (q >> sibs) & 1u;
        sibs += inj ? 0 : 1;
        const size_t r = s0 + (ns - 1 - j);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t d = to_monty(step_digest[8 * (s0 + j) + i]);
            st[i] = bit ? d : node[i], st[8 + i] = bit ? node[i] : d;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) trace[(size_t)(16 + i) * N + r] = st[i], hash_inputs[16 * r + i] = st[i];
        poseidon2_permute_rolled(st);
#pragma unroll
        for (int i = 0; i < 8; i++) node[i] = st[i], trace[(size_t)(8 + i) * N + r] = st[i];
    }
    uint32_t idx = 0, lvl = 0;
    for (size_t k = 0; k < ns; k++) {
        const size_t j = ns - 1 - k, r = s0 + k;
        const bool inj = step_kind[s0 + j] != 0;
        sibs -= inj ? 0 : 1;
        const uint32_t bit = inj ? 0u : (q >> sibs) & 1u;
        idx = idx * (inj ? 1u : 2u) + bit, lvl += inj ? 0u : 1u;
#pragma unroll
        for (int i = 0; i < 8; i++) trace[(size_t)i * N + r] = node[i];
        trace[(size_t)32 * N + r] = bit ? MONTY_ONE : 0u, trace[(size_t)33 * N + r] = inj ? MONTY_ONE : 0u;
        trace[(size_t)34 * N + r] = k == 0 ? MONTY_ONE : 0u, trace[(size_t)35 * N + r] = j == 0 ? MONTY_ONE : 0u;
        trace[(size_t)36 * N + r] = MONTY_ONE, trace[(size_t)37 * N + r] = to_monty(idx), trace[(size_t)38 * N + r] = to_monty(lvl);
    }
}

// ---- transcript chip (air.py duplex_air): the rows of a DuplexChallenger run -----------------------------------------------------
// Record r = one duplexing: n_observed[r] values (observed[8 r ..], canonical) overwrite the first rate lanes, then the state is
// permuted; n_sampled[r] output lanes are popped from the end afterwards.  The sponge is a chain, so one lane walks it (a rolled
// permutation per row); the inputs of every permutation are handed to zkhip_poseidon2_air_tracegen, which is where the width is.
__global__ void k_duplex_rows(const uint32_t* __restrict__ n_observed, const uint32_t* __restrict__ observed, const uint32_t* __restrict__ n_sampled,
                              size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ hash_inputs, uint32_t* __restrict__ bad) {
    if (blockIdx.x || threadIdx.x) return;
    uint32_t st[16];
#pragma unroll
    for (int i = 0; i < 16; i++) st[i] = 0;
    for (size_t r = 0; r < n; r++) {
        const uint32_t k = n_observed[r], ns = n_sampled[r];
        if (k > 8 || ns > 8) {
            atomicAdd(bad, 1u);
            return;
        }
        for (uint32_t j = 0; j < k; j++) {
            const uint32_t v = observed[8 * r + j];
            if (v >= P) {
                atomicAdd(bad, 1u);
                return;
            }
            st[j] = to_monty(v);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) trace[(size_t)i * N + r] = st[i], hash_inputs[16 * r + i] = st[i];
        poseidon2_permute_rolled(st);
#pragma unroll
        for (int i = 0; i < 16; i++) trace[(size_t)(16 + i) * N + r] = st[i];
        for (uint32_t j = 0; j < 8; j++) {
            trace[(size_t)(32 + j) * N + r] = j < k ? MONTY_ONE : 0u;
            trace[(size_t)(40 + j) * N + r] = j >= 8 - ns ? MONTY_ONE : 0u;
        }
        trace[(size_t)48 * N + r] = to_monty((uint32_t)r), trace[(size_t)49 * N + r] = MONTY_ONE;
    }
}

}  // namespace zk

using namespace zk;
extern "C" int zkhip_duplex_tracegen(zkhip_ctx* ctx, const uint32_t* d_n_observed, const uint32_t* d_observed, const uint32_t* d_n_sampled, size_t n,
                                     unsigned log_height, uint32_t* d_trace, uint32_t* d_hash_inputs) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_hash_inputs || log_height > 27 || (n && (!d_n_observed || !d_observed || !d_n_sampled))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "duplex_tracegen: more duplexings than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "duplex_tracegen");
    ZK_HIP_CHECK(ctx, hipMemsetAsync(d_trace, 0, (size_t)ZKHIP_DUPLEX_WIDTH * N * 4, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemsetAsync(d_hash_inputs, 0, 16 * N * 4, ctx->stream));
    if (n) hipLaunchKernelGGL(k_duplex_rows, dim3(1), dim3(64), 0, ctx->stream, d_n_observed, d_observed, d_n_sampled, n, N, d_trace, d_hash_inputs, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    uint32_t h_bad = 0;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(&h_bad, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_bad) return set_error(ctx, ZKHIP_ERR_INVALID, "duplex_tracegen: a record with more than 8 observed / sampled lanes or a value that is not a field element");
    return ZKHIP_OK;
}
extern "C" int zkhip_mmcs_path_tracegen(zkhip_ctx* ctx, const uint32_t* d_leaf, const uint32_t* d_index, const uint32_t* d_path_start,
                                        const uint32_t* d_step_kind, const uint32_t* d_step_digest, size_t n_paths, unsigned log_height,
                                        uint32_t* d_trace, uint32_t* d_hash_inputs) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_hash_inputs || log_height > 27 || (n_paths && (!d_leaf || !d_index || !d_path_start || !d_step_kind || !d_step_digest)))
        return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "mmcs_path_tracegen");
    ZK_HIP_CHECK(ctx, hipMemsetAsync(d_trace, 0, (size_t)ZKHIP_MMCS_PATH_WIDTH * N * 4, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemsetAsync(d_hash_inputs, 0, 16 * N * 4, ctx->stream));
    if (n_paths)
        hipLaunchKernelGGL(k_mmcs_path, dim3((unsigned)((n_paths + 63) / 64)), dim3(64), 0, ctx->stream, d_leaf, d_index, d_path_start, d_step_kind,
                           d_step_digest, n_paths, N, d_trace, d_hash_inputs, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    uint32_t h_bad = 0;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(&h_bad, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_bad) return set_error(ctx, ZKHIP_ERR_INVALID, "mmcs_path_tracegen: " + std::to_string(h_bad) + " paths do not fit the trace or do not end in a sibling step");
    return ZKHIP_OK;
}
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
