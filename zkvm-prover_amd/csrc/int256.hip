// int256.hip -- the 256-bit ALU chip on the device (include/zkhip_int256.hpp: a = b op c on byte limbs, one operation per row, 101
// columns).  Record = op | b[8] | c[8] (17 words).  One lane per row; the row's 32 lookups go into the bitwise table's XOR column in
// the same pass (wave-merged atomics).  Replaces the trace generation of OpenVM's Rv32BaseAlu256 chip (openvm-bigint-circuit,
// un-vendored).
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_int256.hpp"
#include "babybear.hpp"
#include "hist.hpp"
#include "zkhip_internal.hpp"

namespace zk {
namespace {
namespace i2 = zkhip::int256;

__global__ void k_i256_repr(uint32_t* c, size_t n, int to_m) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = to_m ? to_monty(c[i] % P) : from_monty(c[i]);
}

__global__ __launch_bounds__(256) void k_alu256_trace(const uint32_t* __restrict__ recs, const uint32_t* __restrict__ ts, size_t n, size_t N, size_t width,
                                                      uint32_t* __restrict__ trace, uint32_t* __restrict__ bitwise_xor, uint32_t* __restrict__ bad) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    if (row >= n) {
        for (size_t c = 0; c < width; c++) trace[c * N + row] = 0u;
        return;
    }
    const uint32_t op = recs[17 * row];
    if (op >= i2::N_OPS) atomicAdd(bad, 1u);
    uint32_t b[8], c[8], a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = recs[17 * row + 1 + i], c[i] = recs[17 * row + 9 + i];
    uint64_t carry = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (op == i2::OP_ADD) {
            carry += (uint64_t)b[i] + c[i], a[i] = (uint32_t)carry, carry >>= 32;
        } else if (op == i2::OP_SUB) {
            const uint64_t d = (uint64_t)b[i] - c[i] - borrow;
            a[i] = (uint32_t)d, borrow = (d >> 32) & 1u;
        } else {
            a[i] = op == i2::OP_XOR ? b[i] ^ c[i] : op == i2::OP_OR ? b[i] | c[i] : b[i] & c[i];
        }
    }
    auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
    auto put = [&](size_t col, uint32_t v) { trace[col * N + row] = to_monty(v); };
    const bool bitwise = op >= i2::OP_XOR;
    for (int i = 0; i < 32; i++) {
        const uint32_t ai = byte_of(a, i), bi = byte_of(b, i), ci = byte_of(c, i);
        put(i2::COL_A + i, ai), put(i2::COL_B + i, bi), put(i2::COL_C + i, ci);
        hist_add(bitwise_xor, bitwise ? bi * 256 + ci : ai * 256 + ai);
    }
    for (uint32_t f = 0; f < i2::N_OPS; f++) put(i2::COL_FLAGS + f, f == op ? 1u : 0u);
    if (width > i2::WIDTH) put(i2::COL_TS, ts ? ts[row] % P : 0u);
}

// 256-bit multiplication chip: record = b[8] | c[8]; the row's 48 byte pairs and 32 carry tuples are counted in the same pass
__global__ __launch_bounds__(256) void k_mul256_trace(const uint32_t* __restrict__ recs, size_t rec_stride, size_t rec_off, const uint32_t* __restrict__ ts, size_t n,
                                                      size_t N, size_t width, uint32_t* __restrict__ trace, uint32_t* __restrict__ bitwise_range,
                                                      uint32_t* __restrict__ tuple, uint32_t tuple_y, uint32_t* __restrict__ bad) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    if (row >= n) {
        for (size_t c = 0; c < width; c++) trace[c * N + row] = 0u;
        return;
    }
    uint32_t b[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = recs[rec_stride * row + rec_off + i], c[i] = recs[rec_stride * row + rec_off + 8 + i];
    auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
    auto put = [&](size_t col, uint32_t v) { trace[col * N + row] = to_monty(v); };
    uint32_t carry = 0, a_bytes[32];
    for (int k = 0; k < 32; k++) {
        uint32_t s = carry;
        for (int i = 0; i <= k; i++) s += byte_of(b, i) * byte_of(c, k - i);
        a_bytes[k] = s & 255u, carry = s >> 8;
        if ((carry >> 8) >= tuple_y) atomicAdd(bad, 1u);
        put(i2::COL_A + k, a_bytes[k]), put(i2::COL_B + k, byte_of(b, k)), put(i2::COL_C + k, byte_of(c, k));
        put(i2::MUL_COL_CX + k, carry & 255u), put(i2::MUL_COL_CY + k, carry >> 8);
        hist_add(tuple, (carry & 255u) * tuple_y + (carry >> 8));
    }
    for (int i = 0; i < 32; i += 2) {
        hist_add(bitwise_range, a_bytes[i] * 256 + a_bytes[i + 1]);
        hist_add(bitwise_range, byte_of(b, i) * 256 + byte_of(b, i + 1)), hist_add(bitwise_range, byte_of(c, i) * 256 + byte_of(c, i + 1));
    }
    put(i2::MUL_COL_REAL, 1u);
    if (width > i2::MUL_WIDTH) put(i2::MUL_COL_TS, ts ? ts[row] % P : 0u);
}

// 256-bit comparison chip: record = op | b[8] | c[8] (op 6 sltu, 7 slt, 8 eq); the row's 34 lookups are counted in the same pass
__global__ __launch_bounds__(256) void k_cmp256_trace(const uint32_t* __restrict__ recs, const uint32_t* __restrict__ ts, size_t n, size_t N, size_t width,
                                                      uint32_t* __restrict__ trace, uint32_t* __restrict__ bitwise_range, uint32_t* __restrict__ bad) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    if (row >= n) {
        for (size_t c = 0; c < width; c++) trace[c * N + row] = 0u;
        return;
    }
    const uint32_t op_rec = recs[17 * row];
    // (the VM form also takes the branch opcodes 12 .. 17: the comparison they rest on + the branch columns)
    const bool is_br = op_rec >= i2::OP_BEQ && op_rec <= i2::OP_BGE && width > i2::CMP_COL_OPC;
    const uint32_t op = is_br ? (op_rec == i2::OP_BEQ || op_rec == i2::OP_BNE ? i2::OP_EQ : op_rec == i2::OP_BLTU || op_rec == i2::OP_BGEU ? i2::OP_SLTU : i2::OP_SLT) : op_rec;
    if (op < i2::OP_SLTU || op > i2::OP_EQ) atomicAdd(bad, 1u);
    uint32_t b[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = recs[17 * row + 1 + i], c[i] = recs[17 * row + 9 + i];
    auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
    auto put = [&](size_t col, uint32_t v) { trace[col * N + row] = to_monty(v); };
    auto put_signed = [&](size_t col, int v) { trace[col * N + row] = to_monty(v < 0 ? P - (uint32_t)(-v) : (uint32_t)v); };
    const bool is_signed = op == i2::OP_SLT;
    auto limb = [&](const uint32_t* w, int i) -> int {   // the top limb as a signed byte for SLT
        const int v = (int)byte_of(w, i);
        return i == 31 && is_signed && v >= 128 ? v - 256 : v;
    };
    int mark = -1;
    for (int i = 31; i >= 0; i--)
        if (limb(b, i) != limb(c, i)) {
            mark = i;
            break;
        }
    const bool lt = mark >= 0 && limb(b, mark) < limb(c, mark);
    const uint32_t diff = mark < 0 ? 0u : (uint32_t)(lt ? limb(c, mark) - limb(b, mark) : limb(b, mark) - limb(c, mark));
    for (int i = 0; i < 32; i++) {
        put(i2::CMP_COL_B + i, byte_of(b, i)), put(i2::CMP_COL_C + i, byte_of(c, i)), put(i2::CMP_COL_MARK + i, i == mark ? 1u : 0u);
        if (!(i & 1)) hist_add(bitwise_range, byte_of(b, i) * 256 + byte_of(b, i + 1)), hist_add(bitwise_range, byte_of(c, i) * 256 + byte_of(c, i + 1));
    }
    put(i2::CMP_COL_T, lt ? 1u : 0u), put(i2::CMP_COL_DIFF, diff);
    put_signed(i2::CMP_COL_BMSB, limb(b, 31)), put_signed(i2::CMP_COL_CMSB, limb(c, 31));
    const uint32_t shift = is_signed ? 128u : 0u;
    hist_add(bitwise_range, ((uint32_t)(limb(b, 31) + (int)shift) & 255u) * 256 + ((uint32_t)(limb(c, 31) + (int)shift) & 255u));
    if (mark >= 0) hist_add(bitwise_range, ((diff - 1u) & 255u) * 256);
    put(i2::CMP_COL_FLAGS, op == i2::OP_SLTU ? 1u : 0u), put(i2::CMP_COL_FLAGS + 1, op == i2::OP_SLT ? 1u : 0u), put(i2::CMP_COL_FLAGS + 2, op == i2::OP_EQ ? 1u : 0u);
    if (width > i2::CMP_WIDTH) put(i2::CMP_COL_TS, ts ? ts[row] % P : 0u);
    if (width > i2::CMP_COL_OPC) {   // the VM form's branch columns: is_br | neg | taken | the opcode the adapter announces
        const bool neg = is_br && (op_rec == i2::OP_BNE || op_rec == i2::OP_BGEU || op_rec == i2::OP_BGE);
        const bool out = op == i2::OP_EQ ? mark < 0 : lt;
        put(i2::CMP_COL_BR, is_br ? 1u : 0u), put(i2::CMP_COL_NEG, neg ? 1u : 0u), put(i2::CMP_COL_TAKEN, is_br && (out != neg) ? 1u : 0u);
        put(i2::CMP_COL_OPC, op_rec);
    }
}

// 256-bit shift chip: record = op | b[8] | c[8] (op 9 sll, 10 srl, 11 sra; the amount is c mod 256); the row's 66 lookups (carries,
// byte pairs, the amount's bytes, SRA's sign bit in the XOR column) are counted in the same pass
__global__ __launch_bounds__(256) void k_shift256_trace(const uint32_t* __restrict__ recs, const uint32_t* __restrict__ ts, size_t n, size_t N, size_t width,
                                                        uint32_t* __restrict__ trace, uint32_t* __restrict__ bitwise_range, uint32_t* __restrict__ bitwise_xor,
                                                        uint32_t* __restrict__ bad) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    if (row >= n) {
        for (size_t c = 0; c < width; c++) trace[c * N + row] = 0u;
        return;
    }
    const uint32_t op = recs[17 * row];
    if (op < i2::OP_SLL || op > i2::OP_SRA) atomicAdd(bad, 1u);
    uint32_t b[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = recs[17 * row + 1 + i], c[i] = recs[17 * row + 9 + i];
    auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
    auto put = [&](size_t col, uint32_t v) { trace[col * N + row] = to_monty(v); };
    const uint32_t amount = c[0] & 255u, bs = amount & 7u, ls = amount >> 3, mult = 1u << bs;
    const bool left = op == i2::OP_SLL;
    const uint32_t sign = op == i2::OP_SRA ? byte_of(b, 31) >> 7 : 0u;
    uint32_t t[32], cy[32];
    if (left) {
        uint32_t carry = 0;
        for (int k = 0; k < 32; k++) {
            const uint32_t v = byte_of(b, k) * mult + carry;
            t[k] = v & 255u, carry = v >> 8, cy[k] = carry;
        }
    } else {   // t stored reversed: column k holds limb 31 - k of b >> bit_shift
        uint32_t in = sign * (mult - 1u);
        for (int m = 31; m >= 0; m--) {
            const uint32_t v = byte_of(b, m) + 256u * in;
            t[31 - m] = v >> bs, cy[m] = v & (mult - 1u), in = cy[m];
        }
    }
    for (int i = 0; i < 32; i++) {
        const uint32_t sel = (uint32_t)i >= ls ? t[i - (int)ls] : (left ? 0u : 255u * sign);
        put(i2::SH_COL_A + (left ? i : 31 - i), sel);
        put(i2::SH_COL_B + i, byte_of(b, i)), put(i2::SH_COL_T + i, t[i]), put(i2::SH_COL_CY + i, cy[i]), put(i2::SH_COL_LM + i, (uint32_t)i == ls ? 1u : 0u);
        hist_add(bitwise_range, cy[i] * 256 + (mult - 1u - cy[i]));
    }
    for (int i = 0; i < 32; i += 2) hist_add(bitwise_range, t[i] * 256 + t[i + 1]), hist_add(bitwise_range, byte_of(b, i) * 256 + byte_of(b, i + 1));
    put(i2::SH_COL_C0, byte_of(c, 0)), put(i2::SH_COL_C1, byte_of(c, 1)), put(i2::SH_COL_CHI0, c[0] >> 16);
    hist_add(bitwise_range, byte_of(c, 0) * 256 + byte_of(c, 1));
    for (int k = 1; k < 8; k++) put(i2::SH_COL_CW + 2 * (k - 1), c[k] & 0xffffu), put(i2::SH_COL_CW + 2 * (k - 1) + 1, c[k] >> 16);
    for (uint32_t i = 0; i < 8; i++) put(i2::SH_COL_BM + i, i == bs ? 1u : 0u);
    put(i2::SH_COL_SIGN, sign);
    if (op == i2::OP_SRA) hist_add(bitwise_xor, byte_of(b, 31) * 256 + 128u);
    put(i2::SH_COL_FLAGS, op == i2::OP_SLL ? 1u : 0u), put(i2::SH_COL_FLAGS + 1, op == i2::OP_SRL ? 1u : 0u), put(i2::SH_COL_FLAGS + 2, op == i2::OP_SRA ? 1u : 0u);
    if (width > i2::SH_WIDTH) put(i2::SH_COL_TS, ts ? ts[row] % P : 0u);
}

std::mutex g_mu;
std::map<uint32_t, std::vector<uint32_t>> g_programs;   // bitwise bus -> program

int tracegen(zkhip_ctx* ctx, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height, size_t width, uint32_t* d_trace,
             uint32_t* d_bitwise_trace, const char* what) {
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, std::string(what) + ": more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, what);
    uint32_t* xor_col = d_bitwise_trace + ((size_t)1 << 16);   // the XOR multiplicities of the 8-bit table
    const unsigned bb = (unsigned)(((size_t)1 << 16) + 255) / 256;
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(bb), dim3(256), 0, ctx->stream, xor_col, (size_t)1 << 16, 0);
    hipLaunchKernelGGL(k_alu256_trace, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_records, d_ts, n, N, width, d_trace, xor_col, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(bb), dim3(256), 0, ctx->stream, xor_col, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return tracegen_finish(ctx, flag, std::string(what) + " (an opcode above 4)");
}

int air_of(uint32_t bitwise_bus, zkhip_air* out) {
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_programs.find(bitwise_bus);
        if (it == g_programs.end()) {
            zkhip::air::AirBuilder b(i2::WIDTH, 0);
            i2::alu256_air(b, bitwise_bus);
            it = g_programs.emplace(bitwise_bus, b.program()).first;
        }
        out->program = it->second.data(), out->program_len = it->second.size(), out->log_height = 0, out->width = i2::WIDTH, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}

int mul_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, size_t rec_stride, size_t rec_off, const uint32_t* d_ts, size_t n, unsigned log_height, size_t width,
                 uint32_t* d_trace, uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y, const char* what) {
    const size_t N = (size_t)1 << log_height, T = (size_t)size_x * size_y;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, std::string(what) + ": more records than rows");
    if (size_x < 256 || size_y < 32 || T > ((size_t)1 << 27)) return set_error(ctx, ZKHIP_ERR_INVALID, std::string(what) + ": the tuple table must cover (x < 256, y < 32)");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, what);
    const unsigned tb = (unsigned)((T + 255) / 256), bb = (unsigned)(((size_t)1 << 16) + 255) / 256;
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 0);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);
    hipLaunchKernelGGL(k_mul256_trace, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_records, rec_stride, rec_off, d_ts, n, N, width, d_trace,
                       d_bitwise_trace, d_tuple_counts, size_y, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 1);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return tracegen_finish(ctx, flag, std::string(what) + " (a carry outside the tuple table)");
}

std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> g_mul_programs;   // (bitwise bus, tuple bus) -> program
std::map<uint32_t, std::vector<uint32_t>> g_cmp_programs;   // bitwise bus -> program
std::map<uint32_t, std::vector<uint32_t>> g_shift_programs;   // bitwise bus -> program

int shift_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height, size_t width, uint32_t* d_trace,
                   uint32_t* d_bitwise_trace, const char* what) {
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, std::string(what) + ": more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, what);
    const unsigned bb = (unsigned)(((size_t)2 << 16) + 255) / 256;   // both multiplicity columns of the 8-bit table
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)2 << 16, 0);
    hipLaunchKernelGGL(k_shift256_trace, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_records, d_ts, n, N, width, d_trace, d_bitwise_trace,
                       d_bitwise_trace + ((size_t)1 << 16), (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)2 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return tracegen_finish(ctx, flag, std::string(what) + " (an opcode that is not a shift)");
}

int cmp_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height, size_t width, uint32_t* d_trace,
                 uint32_t* d_bitwise_trace, const char* what) {
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, std::string(what) + ": more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, what);
    const unsigned bb = (unsigned)(((size_t)1 << 16) + 255) / 256;
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);   // the range column of the 8-bit table
    hipLaunchKernelGGL(k_cmp256_trace, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_records, d_ts, n, N, width, d_trace, d_bitwise_trace, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_i256_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return tracegen_finish(ctx, flag, std::string(what) + " (an opcode that is not a comparison)");
}

}  // namespace
}  // namespace zk

using namespace zk;

extern "C" {

int zkhip_int256_alu_air(uint32_t bitwise_bus, zkhip_air* out) { return out ? air_of(bitwise_bus, out) : ZKHIP_ERR_INVALID; }

int zkhip_int256_alu_host(uint32_t op, const uint32_t b[8], const uint32_t c[8], uint32_t a[8]) {
    if (!a || !b || !c || op >= i2::N_INT256_OPS) return ZKHIP_ERR_INVALID;
    if (op >= i2::OP_SLL && !i2::is_branch_op(op)) {
        i2::shift256(op, b, c, a);
    } else if (op >= i2::OP_SLTU) {   // (a branch opcode answers with the comparison it rests on)
        for (int i = 1; i < 8; i++) a[i] = 0;
        a[0] = i2::cmp256(op, b, c);
    } else if (op == i2::OP_MUL) i2::mul256(b, c, a);
    else i2::alu256(op, b, c, a);
    return ZKHIP_OK;
}

int zkhip_int256_mul_air(uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air* out) {
    if (!out) return ZKHIP_ERR_INVALID;
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        const auto key = std::make_pair(bitwise_bus, tuple_bus);
        auto it = g_mul_programs.find(key);
        if (it == g_mul_programs.end()) {
            zkhip::air::AirBuilder b(i2::MUL_WIDTH, 0);
            i2::mul256_air(b, bitwise_bus, tuple_bus);
            it = g_mul_programs.emplace(key, b.program()).first;
        }
        out->program = it->second.data(), out->program_len = it->second.size(), out->log_height = 0, out->width = i2::MUL_WIDTH, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}

int zkhip_int256_mul_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace,
                              uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || !d_tuple_counts || log_height > 24 || (n && !d_records)) return ZKHIP_ERR_INVALID;
    return mul_tracegen(ctx, d_records, 16, 0, nullptr, n, log_height, i2::MUL_WIDTH, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y, "int256_mul_tracegen");
}

int zkhip_vm_mul256_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height, uint32_t* d_trace,
                             uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || !d_tuple_counts || log_height > 24 || (n && (!d_records || !d_ts))) return ZKHIP_ERR_INVALID;
    return mul_tracegen(ctx, d_records, 17, 1, d_ts, n, log_height, i2::MUL_VM_WIDTH, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y, "vm_mul256_tracegen");
}

int zkhip_int256_cmp_air(uint32_t bitwise_bus, zkhip_air* out) {
    if (!out) return ZKHIP_ERR_INVALID;
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_cmp_programs.find(bitwise_bus);
        if (it == g_cmp_programs.end()) {
            zkhip::air::AirBuilder b(i2::CMP_WIDTH, 0);
            (void)i2::cmp256_air(b, bitwise_bus);
            it = g_cmp_programs.emplace(bitwise_bus, b.program()).first;
        }
        out->program = it->second.data(), out->program_len = it->second.size(), out->log_height = 0, out->width = i2::CMP_WIDTH, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}

int zkhip_int256_cmp_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 24 || (n && !d_records)) return ZKHIP_ERR_INVALID;
    return cmp_tracegen(ctx, d_records, nullptr, n, log_height, i2::CMP_WIDTH, d_trace, d_bitwise_trace, "int256_cmp_tracegen");
}

int zkhip_vm_cmp256_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height, uint32_t* d_trace,
                             uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 24 || (n && (!d_records || !d_ts))) return ZKHIP_ERR_INVALID;
    return cmp_tracegen(ctx, d_records, d_ts, n, log_height, i2::CMP_VM_WIDTH, d_trace, d_bitwise_trace, "vm_cmp256_tracegen");
}

int zkhip_int256_shift_air(uint32_t bitwise_bus, zkhip_air* out) {
    if (!out) return ZKHIP_ERR_INVALID;
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_shift_programs.find(bitwise_bus);
        if (it == g_shift_programs.end()) {
            zkhip::air::AirBuilder b(i2::SH_WIDTH, 0);
            i2::shift256_air(b, bitwise_bus);
            it = g_shift_programs.emplace(bitwise_bus, b.program()).first;
        }
        out->program = it->second.data(), out->program_len = it->second.size(), out->log_height = 0, out->width = i2::SH_WIDTH, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}

int zkhip_int256_shift_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 24 || (n && !d_records)) return ZKHIP_ERR_INVALID;
    return shift_tracegen(ctx, d_records, nullptr, n, log_height, i2::SH_WIDTH, d_trace, d_bitwise_trace, "int256_shift_tracegen");
}

int zkhip_vm_shift256_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height, uint32_t* d_trace,
                               uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 24 || (n && (!d_records || !d_ts))) return ZKHIP_ERR_INVALID;
    return shift_tracegen(ctx, d_records, d_ts, n, log_height, i2::SH_VM_WIDTH, d_trace, d_bitwise_trace, "vm_shift256_tracegen");
}

int zkhip_int256_alu_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 24 || (n && !d_records)) return ZKHIP_ERR_INVALID;
    return tracegen(ctx, d_records, nullptr, n, log_height, i2::WIDTH, d_trace, d_bitwise_trace, "int256_alu_tracegen");
}

int zkhip_vm_int256_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height, uint32_t* d_trace,
                             uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 24 || (n && (!d_records || !d_ts))) return ZKHIP_ERR_INVALID;
    return tracegen(ctx, d_records, d_ts, n, log_height, i2::VM_WIDTH, d_trace, d_bitwise_trace, "vm_int256_tracegen");
}

}  // extern "C"
