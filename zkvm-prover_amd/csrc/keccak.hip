// keccak.hip -- Keccak-f[1600] chip on the device (include/zkhip_keccak.hpp: the AIR, one round per row, 2633 columns).
// Record = the 25 input lanes of a permutation (as 50 32-bit words: lane x + 5 y = words 2 (x + 5 y), 2 (x + 5 y) + 1, low word first).
// One lane per ROW: row 24 p + r replays rounds 0 .. r - 1 of permutation p in registers (12 rounds on average: cheap next to the 2633
// stores) and writes round r's columns; consecutive lanes write consecutive rows, so every column store is coalesced.  Rows beyond
// 24 n_perms continue with permutations of the zero state (the AIR's padding: valid rows, export = 0).
// Replaces the trace generation of OpenVM's keccak chip (openvm-keccak256-circuit over p3-keccak-air, un-vendored).
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_keccak.hpp"
#include "babybear.hpp"
#include "zkhip_internal.hpp"

namespace zk {
namespace {
namespace kk = zkhip::keccak;

__constant__ uint64_t d_RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                                  0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                                  0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                                  0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
__constant__ unsigned d_R[25] = {0, 36, 3, 41, 18, 1, 44, 10, 45, 2, 62, 6, 43, 15, 61, 28, 55, 25, 21, 56, 27, 20, 39, 8, 14};   // r[x][y] at 5 x + y

__device__ __forceinline__ uint64_t rol64(uint64_t v, unsigned r) { return r ? (v << r) | (v >> (64 - r)) : v; }

// one round on st[x + 5 y]; leaves theta's c / c', the post-theta lanes ap and the post-chi lanes app (before iota) for the row
__device__ __forceinline__ void round_parts(const uint64_t st[25], uint64_t c[5], uint64_t cp[5], uint64_t ap[25], uint64_t app[25]) {
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
#pragma unroll
    for (int x = 0; x < 5; x++) cp[x] = c[x] ^ c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
#pragma unroll
    for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) ap[x + 5 * y] = st[x + 5 * y] ^ c[x] ^ cp[x];
    uint64_t b[25];
#pragma unroll
    for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(ap[x + 5 * y], d_R[5 * x + y]);
#pragma unroll
    for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) app[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
}

__global__ __launch_bounds__(256) void k_keccak_trace(const uint32_t* __restrict__ states, size_t n_perms, size_t N, uint32_t* __restrict__ trace) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const size_t p = row / 24;
    const unsigned r = (unsigned)(row % 24);
    uint64_t pre[25], st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) pre[i] = p < n_perms ? ((uint64_t)states[50 * p + 2 * i] | ((uint64_t)states[50 * p + 2 * i + 1] << 32)) : 0ull;
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = pre[i];
    uint64_t c[5], cp[5], ap[25], app[25];
    for (unsigned q = 0; q < r; q++) {
        round_parts(st, c, cp, ap, app);
#pragma unroll
        for (int i = 0; i < 25; i++) st[i] = app[i];
        st[0] ^= d_RC[q];
    }
    round_parts(st, c, cp, ap, app);
    auto put = [&](size_t col, uint32_t v) { trace[col * N + row] = to_monty(v); };
    for (unsigned i = 0; i < 24; i++) put(kk::COL_FLAGS + i, i == r ? 1u : 0u);
    put(kk::COL_EXPORT, (r == 23 && p < n_perms) ? 1u : 0u);
    for (int y = 0; y < 5; y++)
        for (int x = 0; x < 5; x++)
            for (int l = 0; l < 4; l++) {
                put(kk::COL_PREIMAGE + (y * 5 + x) * 4 + l, (uint32_t)(pre[x + 5 * y] >> (16 * l)) & 0xffffu);
                put(kk::COL_A + (y * 5 + x) * 4 + l, (uint32_t)(st[x + 5 * y] >> (16 * l)) & 0xffffu);
                put(kk::COL_A_PP + (y * 5 + x) * 4 + l, (uint32_t)(app[x + 5 * y] >> (16 * l)) & 0xffffu);
            }
    for (int x = 0; x < 5; x++)
        for (int z = 0; z < 64; z++) {
            put(kk::COL_C + x * 64 + z, (uint32_t)(c[x] >> z) & 1u);
            put(kk::COL_C_PRIME + x * 64 + z, (uint32_t)(cp[x] >> z) & 1u);
        }
    for (int y = 0; y < 5; y++)
        for (int x = 0; x < 5; x++)
            for (int z = 0; z < 64; z++) put(kk::COL_A_PRIME + (y * 5 + x) * 64 + z, (uint32_t)(ap[x + 5 * y] >> z) & 1u);
    const uint64_t out00 = app[0] ^ d_RC[r];
    for (int z = 0; z < 64; z++) put(kk::COL_A_PP_00_BITS + z, (uint32_t)(app[0] >> z) & 1u);
    for (int l = 0; l < 4; l++) put(kk::COL_A_PPP_00 + l, (uint32_t)(out00 >> (16 * l)) & 0xffffu);
}

// the VM chip's timestamp column: every row of call p carries the call's timestamp, the padding permutations zero
__global__ __launch_bounds__(256) void k_keccak_ts(const uint32_t* __restrict__ ts, size_t n_perms, size_t N, uint32_t* __restrict__ col) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const size_t p = row / 24;
    col[row] = p < n_perms ? to_monty(ts[p] % P) : 0u;
}

std::once_flag g_once;
std::vector<uint32_t> g_program;

}  // namespace
}  // namespace zk

using namespace zk;

extern "C" {

int zkhip_keccak_f_air(zkhip_air* out) {
    if (!out) return ZKHIP_ERR_INVALID;
    try {
        std::call_once(g_once, [] {
            zkhip::air::AirBuilder b(kk::WIDTH, 0);
            kk::keccak_f_air(b);
            g_program = b.program();
        });
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    out->program = g_program.data(), out->program_len = g_program.size(), out->log_height = 0, out->width = kk::WIDTH, out->n_pvs = 0;
    out->prep_trace = nullptr, out->prep_commit = nullptr;
    return ZKHIP_OK;
}

int zkhip_keccak_f1600_host(uint64_t state[25]) {
    if (!state) return ZKHIP_ERR_INVALID;
    kk::keccak_f1600(state);
    return ZKHIP_OK;
}

int zkhip_keccak_f_tracegen(zkhip_ctx* ctx, const uint32_t* d_states, size_t n_perms, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 24 || (n_perms && !d_states)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (24 * n_perms > N) return set_error(ctx, ZKHIP_ERR_INVALID, "keccak_f_tracegen: 24 rows per permutation do not fit the trace");
    KernelScope ks(ctx, "keccak_f_tracegen");
    hipLaunchKernelGGL(k_keccak_trace, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_states, n_perms, N, d_trace);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int zkhip_vm_keccak_tracegen(zkhip_ctx* ctx, const uint32_t* d_states, const uint32_t* d_ts, size_t n_perms, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || (n_perms && !d_ts)) return ZKHIP_ERR_INVALID;
    ZK_TRY(zkhip_keccak_f_tracegen(ctx, d_states, n_perms, log_height, d_trace));
    const size_t N = (size_t)1 << log_height;
    KernelScope ks(ctx, "vm_keccak_timestamps");
    hipLaunchKernelGGL(k_keccak_ts, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_ts, n_perms, N, d_trace + (size_t)kk::COL_TS * N);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // extern "C"
