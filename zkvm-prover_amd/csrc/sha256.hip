// sha256.hip -- SHA-256 compression chip on the device (include/zkhip_sha256.hpp: the AIR, one round per row, 65 rows per block).
// Record = 24 words per block: H_in[8] then the sixteen big-endian message words.  One lane per ROW: row 65 b + t recomputes the message
// schedule and replays rounds 0 .. t - 1 of block b in registers (a few hundred integer operations next to 433 stores) and writes the
// columns of round t; consecutive lanes write consecutive rows, so every column store is coalesced.  Whole blocks beyond n_blocks are
// compressions of the zero record with real = 0; the rows after the last whole block are zero.
// Replaces the trace generation of OpenVM's SHA-256 chip (openvm-sha256-circuit, un-vendored).
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_sha256.hpp"
#include "babybear.hpp"
#include "zkhip_internal.hpp"

namespace zk {
namespace {
namespace sh = zkhip::sha256;

__constant__ uint32_t d_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
    0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
    0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
    0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
    0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t ror32(uint32_t v, unsigned r) { return (v >> r) | (v << (32 - r)); }
__device__ __forceinline__ uint32_t lo16(uint32_t v) { return v & 0xffffu; }
__device__ __forceinline__ uint32_t hi16(uint32_t v) { return v >> 16; }

__global__ __launch_bounds__(256) void k_sha256_trace(const uint32_t* __restrict__ blocks, size_t n_blocks, size_t N, uint32_t* __restrict__ trace) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const size_t blk = row / sh::ROWS_PER_BLOCK;
    const unsigned t = (unsigned)(row % sh::ROWS_PER_BLOCK);
    auto put = [&](size_t col, uint32_t v) { trace[col * N + row] = v; };   // 0 and 1 are written in Montgomery form by the callers below
    auto put_c = [&](size_t col, uint32_t v) { trace[col * N + row] = to_monty(v); };
    const uint32_t ONE = to_monty(1u);
    if (blk >= N / sh::ROWS_PER_BLOCK) {
        for (size_t c = 0; c < sh::WIDTH; c++) put(c, 0u);
        return;
    }
    const bool real = blk < n_blocks;
    uint32_t hin[8], w[64], s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) hin[i] = real ? blocks[24 * blk + i] : 0u;
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = real ? blocks[24 * blk + 8 + i] : 0u;
    for (int i = 16; i < 64; i++)
        w[i] = (ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10)) + w[i - 7] + (ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 16];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = hin[i];
    const unsigned rounds_done = t < 64 ? t : 64;
    for (unsigned r = 0; r < rounds_done; r++) {
        const uint32_t t1 = s[7] + (ror32(s[4], 6) ^ ror32(s[4], 11) ^ ror32(s[4], 25)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + d_K[r] + w[r];
        const uint32_t t2 = (ror32(s[0], 2) ^ ror32(s[0], 13) ^ ror32(s[0], 22)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
#pragma unroll
        for (int i = 7; i > 0; i--) s[i] = s[i - 1];
        s[4] += t1, s[0] = t1 + t2;
    }
    if (t == 64) {
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] += hin[i];
    }
#pragma unroll
    for (int wd = 0; wd < 8; wd++)
        for (int j = 0; j < 32; j++) put(sh::COL_STATE + 32 * wd + j, ((s[wd] >> j) & 1u) ? ONE : 0u);
    const uint32_t sg0 = ror32(s[0], 2) ^ ror32(s[0], 13) ^ ror32(s[0], 22), sg1 = ror32(s[4], 6) ^ ror32(s[4], 11) ^ ror32(s[4], 25);
    const uint32_t mj = (s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]), ch = (s[4] & s[5]) ^ (~s[4] & s[6]);
    put_c(sh::COL_SIGMA0, lo16(sg0)), put_c(sh::COL_SIGMA0 + 1, hi16(sg0)), put_c(sh::COL_SIGMA1, lo16(sg1)), put_c(sh::COL_SIGMA1 + 1, hi16(sg1));
    put_c(sh::COL_MAJ, lo16(mj)), put_c(sh::COL_MAJ + 1, hi16(mj));
#pragma unroll
    for (int wd = 0; wd < 8; wd++) put_c(sh::COL_HIN + 2 * wd, lo16(hin[wd])), put_c(sh::COL_HIN + 2 * wd + 1, hi16(hin[wd]));
    put(sh::COL_REAL, real ? ONE : 0u);
    // window: position k holds W_{t-15+k}; words before the block and the digest row's window are zero
    auto win = [&](unsigned k) -> uint32_t { return (t < 64 && t + k >= 15) ? w[t + k - 15] : 0u; };
    const uint32_t w15 = win(15), w14 = win(14), w1 = win(1), w0 = win(0), w9 = win(9);
    for (int j = 0; j < 32; j++) {
        put(sh::COL_W15_BITS + j, ((w15 >> j) & 1u) ? ONE : 0u);
        put(sh::COL_W14_BITS + j, ((w14 >> j) & 1u) ? ONE : 0u);
        put(sh::COL_W1_BITS + j, ((w1 >> j) & 1u) ? ONE : 0u);
    }
    put_c(sh::COL_W0, lo16(w0)), put_c(sh::COL_W0 + 1, hi16(w0));
    for (unsigned k = 2; k < 14; k++) {
        const uint32_t v = win(k);
        put_c(sh::COL_W2 + 2 * (k - 2), lo16(v)), put_c(sh::COL_W2 + 2 * (k - 2) + 1, hi16(v));
    }
    const uint32_t g0 = ror32(w1, 7) ^ ror32(w1, 18) ^ (w1 >> 3), g1 = ror32(w14, 17) ^ ror32(w14, 19) ^ (w14 >> 10);
    put_c(sh::COL_SIG0, lo16(g0)), put_c(sh::COL_SIG0 + 1, hi16(g0)), put_c(sh::COL_SIG1, lo16(g1)), put_c(sh::COL_SIG1 + 1, hi16(g1));
    // carries of the transition out of this row (zero on the digest row)
    uint32_t ca = 0, cah = 0, ce = 0, ceh = 0, cs[12] = {0}, cw = 0, cwh = 0;
    if (t < 64) {
        const uint32_t kt = d_K[t], wt = w[t];
        const bool last = t == 63;
        ca = (lo16(s[7]) + lo16(sg1) + lo16(ch) + lo16(kt) + lo16(wt) + lo16(sg0) + lo16(mj) + (last ? lo16(hin[0]) : 0u)) >> 16;
        cah = (hi16(s[7]) + hi16(sg1) + hi16(ch) + hi16(kt) + hi16(wt) + hi16(sg0) + hi16(mj) + (last ? hi16(hin[0]) : 0u) + ca) >> 16;
        ce = (lo16(s[3]) + lo16(s[7]) + lo16(sg1) + lo16(ch) + lo16(kt) + lo16(wt) + (last ? lo16(hin[4]) : 0u)) >> 16;
        ceh = (hi16(s[3]) + hi16(s[7]) + hi16(sg1) + hi16(ch) + hi16(kt) + hi16(wt) + (last ? hi16(hin[4]) : 0u) + ce) >> 16;
        const int shifted[6] = {1, 2, 3, 5, 6, 7};
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const uint32_t add = last ? hin[shifted[i]] : 0u;
            cs[2 * i] = (lo16(s[shifted[i] - 1]) + lo16(add)) >> 16;
            cs[2 * i + 1] = (hi16(s[shifted[i] - 1]) + hi16(add) + cs[2 * i]) >> 16;
        }
        if (t >= 15 && t < 63) {
            cw = (lo16(g1) + lo16(w9) + lo16(g0) + lo16(w0)) >> 16;
            cwh = (hi16(g1) + hi16(w9) + hi16(g0) + hi16(w0) + cw) >> 16;
        }
    }
    for (int k = 0; k < 3; k++) {
        put(sh::COL_CARRY_A + k, ((ca >> k) & 1u) ? ONE : 0u), put(sh::COL_CARRY_A + 3 + k, ((cah >> k) & 1u) ? ONE : 0u);
        put(sh::COL_CARRY_E + k, ((ce >> k) & 1u) ? ONE : 0u), put(sh::COL_CARRY_E + 3 + k, ((ceh >> k) & 1u) ? ONE : 0u);
    }
    for (int i = 0; i < 12; i++) put(sh::COL_CARRY_SHIFT + i, cs[i] ? ONE : 0u);
    put(sh::COL_CARRY_W, (cw & 1u) ? ONE : 0u), put(sh::COL_CARRY_W + 1, (cw & 2u) ? ONE : 0u);
    put(sh::COL_CARRY_W + 2, (cwh & 1u) ? ONE : 0u), put(sh::COL_CARRY_W + 3, (cwh & 2u) ? ONE : 0u);
}

// the VM chip's timestamp column: every row of block b carries the call's timestamp, the padding blocks and the tail zero
__global__ __launch_bounds__(256) void k_sha256_ts(const uint32_t* __restrict__ ts, size_t n_blocks, size_t N, uint32_t* __restrict__ col) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const size_t blk = row / sh::ROWS_PER_BLOCK;
    col[row] = blk < n_blocks ? to_monty(ts[blk] % P) : 0u;
}

std::mutex g_mu;
std::vector<uint32_t> g_program;
std::map<unsigned, std::vector<uint32_t>> g_prep;   // log_height -> preprocessed trace (kept for the life of the process)

}  // namespace
}  // namespace zk

using namespace zk;

extern "C" {

int zkhip_sha256_air(unsigned log_height, zkhip_air* out) {
    if (!out || log_height < 7 || log_height > 24) return ZKHIP_ERR_INVALID;
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_program.empty()) {
            zkhip::air::AirBuilder b(sh::WIDTH, 0, sh::PREP_WIDTH);
            sh::compress_air(b);
            g_program = b.program();
        }
        auto it = g_prep.find(log_height);
        if (it == g_prep.end()) it = g_prep.emplace(log_height, sh::prep_trace(log_height)).first;
        out->program = g_program.data(), out->program_len = g_program.size(), out->log_height = log_height, out->width = sh::WIDTH, out->n_pvs = 0;
        out->prep_trace = it->second.data(), out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}

int zkhip_sha256_compress_host(uint32_t state[8], const uint32_t block[16]) {
    if (!state || !block) return ZKHIP_ERR_INVALID;
    sh::compress(state, block);
    return ZKHIP_OK;
}

int zkhip_sha256_tracegen(zkhip_ctx* ctx, const uint32_t* d_blocks, size_t n_blocks, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height < 7 || log_height > 24 || (n_blocks && !d_blocks)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n_blocks > N / sh::ROWS_PER_BLOCK) return set_error(ctx, ZKHIP_ERR_INVALID, "sha256_tracegen: 65 rows per block do not fit the trace");
    KernelScope ks(ctx, "sha256_tracegen");
    hipLaunchKernelGGL(k_sha256_trace, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_blocks, n_blocks, N, d_trace);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int zkhip_vm_sha256_tracegen(zkhip_ctx* ctx, const uint32_t* d_blocks, const uint32_t* d_ts, size_t n_blocks, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || (n_blocks && !d_ts)) return ZKHIP_ERR_INVALID;
    ZK_TRY(zkhip_sha256_tracegen(ctx, d_blocks, n_blocks, log_height, d_trace));
    const size_t N = (size_t)1 << log_height;
    KernelScope ks(ctx, "vm_sha256_timestamps");
    hipLaunchKernelGGL(k_sha256_ts, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_ts, n_blocks, N, d_trace + (size_t)sh::COL_TS * N);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

/* preprocessed trace of the VM chip (9 columns x 2^log_height rows, column-major, canonical) */
int zkhip_vm_sha256_prep(unsigned log_height, uint32_t* out) {
    if (!out || log_height < 7 || log_height > 24) return ZKHIP_ERR_INVALID;
    const std::vector<uint32_t> p = sh::prep_trace_vm(log_height);
    memcpy(out, p.data(), p.size() * 4);
    return ZKHIP_OK;
}

}  // extern "C"
