// modular.hip -- the modular-multiplication chip on the device (include/zkhip_modular.hpp: r = a b mod P on byte limbs, one operation
// per row, 325 columns).  Record = a[8] | b[8] (little-endian 32-bit words; with an operation word in front for the form that also adds and subtracts).  One lane per row: schoolbook product, binary long division
// by the modulus (512 shift-compare-subtract steps on nine words: ~20 k integer operations next to 325 stores and 127 table increments),
// then the carries of the limb equations; the row's lookups (65 byte pairs, 62 carry tuples) are counted into the bitwise and range-tuple
// tables in the same pass (wave-merged atomics, csrc/hist.hpp).  Replaces the trace generation of OpenVM's ModularMulDiv chip
// (openvm-algebra-circuit, un-vendored).
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_modular.hpp"
#include "babybear.hpp"
#include "hist.hpp"
#include "zkhip_internal.hpp"

namespace zk {
namespace {
namespace md = zkhip::modular;

struct ModWords {
    uint32_t w[12];
};

__global__ void k_mod_repr(uint32_t* c, size_t n, int to_m) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = to_m ? to_monty(c[i] % P) : from_monty(c[i]);
}

// NW = words of the modulus: 8 (32 limbs) or 12 (48 limbs: the BLS12-381 base field, crates/circuits/batch-circuit/openvm.toml:18-21)
template <int NW>
__global__ __launch_bounds__(256) void k_modmul_trace(const uint32_t* __restrict__ recs, size_t rec_stride, size_t n, size_t N, ModWords pm, uint32_t* __restrict__ trace,
                                                      uint32_t* __restrict__ bitwise_range, uint32_t* __restrict__ tuple, uint32_t tuple_y,
                                                      uint32_t* __restrict__ bad) {
    constexpr md::Cols C(4 * NW);
    constexpr int L = 4 * NW;
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    if (row >= n) {
        for (size_t c = 0; c < C.WIDTH; c++) trace[c * N + row] = 0u;
        return;
    }
    uint32_t a[NW], b[NW], prod[2 * NW], rem[NW + 1], quo[NW];
    const size_t ro = rec_stride * row + (rec_stride - 2 * NW);
    const uint32_t op_in = rec_stride == 2 * NW + 1 ? recs[rec_stride * row] : 0u;   // 0 mul, 1 add, 2 sub, 3 div (the record then holds the quotient x / y and y)
    const bool is_div = op_in == md::OP_DIV, is_eq = op_in == md::OP_IS_EQ;   // (an equality test is a subtraction row with the bit on top)
    const uint32_t op = is_div ? (uint32_t)md::OP_MUL : is_eq ? (uint32_t)md::OP_SUB : op_in;
    if (op_in >= md::N_OPS) atomicAdd(bad, 1u);
#pragma unroll
    for (int i = 0; i < NW; i++) a[i] = recs[ro + i], b[i] = recs[ro + NW + i];
#pragma unroll
    for (int i = 0; i < 2 * NW; i++) prod[i] = 0;
    bool sub_wrapped = false;
    if (op == md::OP_ADD) {
        uint64_t c = 0;
        for (int i = 0; i < NW; i++) c += (uint64_t)a[i] + b[i], prod[i] = (uint32_t)c, c >>= 32;
        prod[NW] = (uint32_t)c;
    } else if (op == md::OP_SUB) {   // a - b, plus one P if that is negative: the numerator is then below P (or the record is refused)
        uint32_t br = 0;
        for (int i = 0; i < NW; i++) {
            const uint64_t d = (uint64_t)a[i] - b[i] - br;
            prod[i] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
        }
        if (br) {
            uint64_t c = 0;
            for (int i = 0; i < NW; i++) c += (uint64_t)prod[i] + pm.w[i], prod[i] = (uint32_t)c, c >>= 32;
            if (!c) atomicAdd(bad, 1u);
            sub_wrapped = true;
        }
    } else {
        for (int i = 0; i < NW; i++) {
            uint64_t c = 0;
            for (int j = 0; j < NW; j++) {
                c += (uint64_t)a[i] * b[j] + prod[i + j];
                prod[i + j] = (uint32_t)c, c >>= 32;
            }
            prod[i + NW] = (uint32_t)c;
        }
    }
#pragma unroll
    for (int i = 0; i <= NW; i++) rem[i] = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) quo[i] = 0;
    bool overflow = false;
    for (int bit = 64 * NW - 1; bit >= 0; bit--) {
        for (int k = NW; k > 0; k--) rem[k] = (rem[k] << 1) | (rem[k - 1] >> 31);
        rem[0] = (rem[0] << 1) | ((prod[bit >> 5] >> (bit & 31)) & 1u);
        bool ge = rem[NW] != 0;
        if (!ge) {
            ge = true;
            for (int k = NW - 1; k >= 0; k--)
                if (rem[k] != pm.w[k]) {
                    ge = rem[k] > pm.w[k];
                    break;
                }
        }
        if (ge) {
            uint32_t br = 0;
            for (int k = 0; k <= NW; k++) {
                const uint64_t d = (uint64_t)rem[k] - (k < NW ? pm.w[k] : 0u) - br;
                rem[k] = (uint32_t)d, br = (uint32_t)(d >> 32) & 1u;
            }
            if (bit >= 32 * NW) overflow = true;
            else quo[bit >> 5] |= 1u << (bit & 31);
        }
    }
    if (overflow) atomicAdd(bad, 1u);   // the quotient does not fit L limbs: operands far above the modulus
    if (op == md::OP_SUB) {             // a - b + q P = r: q is 0 or 1 and the numerator was already reduced
        bool any = false;
        for (int i = 0; i < NW; i++) any = any || quo[i];
        if (any) atomicAdd(bad, 1u);    // |a - b| >= P
        quo[0] = sub_wrapped ? 1u : 0u;
    }
    auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
    auto put = [&](size_t col, uint32_t v) { trace[col * N + row] = to_monty(v); };
    for (int i = 0; i < L; i++) {
        put(C.A + i, byte_of(a, i)), put(C.B + i, byte_of(b, i)), put(C.Q + i, byte_of(quo, i)), put(C.R + i, byte_of(rem, i));
    }
    for (int i = 0; i < L; i += 2) {
        hist_add(bitwise_range, byte_of(a, i) * 256 + byte_of(a, i + 1)), hist_add(bitwise_range, byte_of(b, i) * 256 + byte_of(b, i + 1));
        hist_add(bitwise_range, byte_of(quo, i) * 256 + byte_of(quo, i + 1)), hist_add(bitwise_range, byte_of(rem, i) * 256 + byte_of(rem, i + 1));
    }
    // carries of the limb equations: position k of a b - q P - r, plus the carry in, is 256 times the carry out
    int64_t c = 0;
    const int64_t q_sign = op == md::OP_SUB ? -1 : 1;
    for (int k = 0; k <= (int)C.N_CARRY; k++) {
        int64_t s = c;
        for (int i = 0; i < L; i++) {
            const int j = k - i;
            if (j < 0 || j >= L) continue;
            if (op == md::OP_MUL) s += (int64_t)byte_of(a, i) * byte_of(b, j);
            s -= q_sign * (int64_t)byte_of(quo, i) * byte_of(pm.w, j);
        }
        if (k < L && op == md::OP_ADD) s += (int64_t)byte_of(a, k) + byte_of(b, k);
        if (k < L && op == md::OP_SUB) s += (int64_t)byte_of(a, k) - byte_of(b, k);
        if (k < L) s -= byte_of(rem, k);
        if ((s & 255) != 0 && !overflow) atomicAdd(bad, 1u);   // (cannot happen: a b = q P + r)
        c = s >> 8;
        if (k < (int)C.N_CARRY) {
            const int64_t shifted = c + md::CARRY_OFFSET;
            const uint32_t v = shifted < 0 || shifted >= (int64_t)256 * tuple_y ? 0u : (uint32_t)shifted;
            if ((int64_t)v != shifted) atomicAdd(bad, 1u);
            put(C.CX + k, v & 255u), put(C.CY + k, v >> 8);
            hist_add(tuple, (v & 255u) * tuple_y + (v >> 8));
        } else if (c != 0 && !overflow) {
            atomicAdd(bad, 1u);
        }
    }
    // r < P: the most significant differing limb
    int mark = -1;
    for (int i = L - 1; i >= 0; i--)
        if (byte_of(rem, i) != byte_of(pm.w, i)) {
            mark = i;
            break;
        }
    for (int i = 0; i < L; i++) put(C.MARK + i, i == mark ? 1u : 0u);
    const uint32_t diff = mark >= 0 ? byte_of(pm.w, mark) - byte_of(rem, mark) : 0u;
    put(C.DIFF, diff), put(C.REAL, 1u), put(C.IS_ADD, op == md::OP_ADD ? 1u : 0u), put(C.IS_SUB, op == md::OP_SUB ? 1u : 0u);   // (set on equality tests too)
    hist_add(bitwise_range, ((diff - 1u) & 255u) * 256);
    // a division's quotient (the a columns) is canonical as well
    int mark2 = -1;
    if (is_div)
        for (int i = L - 1; i >= 0; i--)
            if (byte_of(a, i) != byte_of(pm.w, i)) {
                mark2 = byte_of(a, i) < byte_of(pm.w, i) ? i : -2;
                break;
            }
    if (is_div && mark2 < 0) atomicAdd(bad, 1u);   // the record's quotient is not below the modulus
    for (int i = 0; i < L; i++) put(C.MARK2 + i, i == mark2 ? 1u : 0u);
    const uint32_t diff2 = mark2 >= 0 ? byte_of(pm.w, mark2) - byte_of(a, mark2) : 0u;
    put(C.IS_DIV, is_div ? 1u : 0u), put(C.DIFF2, diff2);
    uint32_t limb_sum = 0;
    for (int i = 0; i < L; i++) limb_sum += byte_of(rem, i);
    const bool eq = is_eq && limb_sum == 0;
    put(C.IS_EQ, is_eq ? 1u : 0u), put(C.EQ, eq ? 1u : 0u);
    trace[(size_t)C.INV * N + row] = is_eq && !eq ? minv(to_monty(limb_sum)) : 0u;   // (Montgomery form, as every cell)
    if (is_div) hist_add(bitwise_range, ((diff2 - 1u) & 255u) * 256);
}

// the VM chip's timestamp column: row i carries the timestamp of call i
__global__ __launch_bounds__(256) void k_modmul_ts(const uint32_t* __restrict__ ts, size_t n, size_t N, uint32_t* __restrict__ col) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row < N) col[row] = row < n ? to_monty(ts[row] % P) : 0u;
}

std::mutex g_mu;
std::map<std::pair<md::Modulus, std::pair<uint32_t, uint32_t>>, std::vector<uint32_t>> g_programs;   // per (modulus, buses), kept for the life of the process

}  // namespace
}  // namespace zk

using namespace zk;

extern "C" {

int zkhip_modmul_air_x(const uint8_t* modulus, uint32_t n_limbs, uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air* out) {
    if (!modulus || !out || (n_limbs != 32 && n_limbs != 48)) return ZKHIP_ERR_INVALID;
    md::Modulus m;
    m.limbs = n_limbs;
    memcpy(m.data(), modulus, n_limbs);
    bool zero = true;
    for (uint8_t v : m) zero = zero && v == 0;
    if (zero) return ZKHIP_ERR_INVALID;
    const md::Cols C(n_limbs);
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        const auto key = std::make_pair(m, std::make_pair(bitwise_bus, tuple_bus));
        auto it = g_programs.find(key);
        if (it == g_programs.end()) {
            zkhip::air::AirBuilder b(C.WIDTH, 0);
            md::modmul_air(b, m, bitwise_bus, tuple_bus);
            it = g_programs.emplace(key, b.program()).first;
        }
        out->program = it->second.data(), out->program_len = it->second.size(), out->log_height = 0, out->width = C.WIDTH, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}
int zkhip_modmul_air(const uint8_t modulus[32], uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air* out) { return zkhip_modmul_air_x(modulus, 32, bitwise_bus, tuple_bus, out); }

// r = a op b mod modulus on n_words-word operands (op 0 mul, 1 add, 2 sub, 3 div, 4 is_eq); q = the quotient of the limb identity
int zkhip_modular_host_x(uint32_t op, uint32_t n_words, const uint32_t* a, const uint32_t* b, const uint32_t* modulus, uint32_t* q, uint32_t* r) {
    if (!a || !b || !modulus || !q || !r || op >= md::N_OPS || (n_words != 8 && n_words != 12)) return ZKHIP_ERR_INVALID;
    const md::UInt A = md::load_words(a, n_words), B = md::load_words(b, n_words), M = md::load_words(modulus, n_words);
    md::UInt Q{}, R{};
    bool zero = true;
    for (uint32_t v : M.w) zero = zero && v == 0;
    if (zero) return ZKHIP_ERR_INVALID;
    bool ok = true;
    if (op == md::OP_MUL) {
        ok = md::mulmod(A, B, M, &Q, &R);
        // (for a modulus of 8 words handed in with n_words = 12 the quotient's width is the modulus's: mulmod checks words_of(M))
    } else if (op == md::OP_IS_EQ) {   // r = [a = b (mod P)] for |a - b| < P
        if (!md::addsubmod(md::OP_SUB, A, B, M, &Q, &R)) return ZKHIP_ERR_INVALID;
        bool zero_r = true;
        for (uint32_t w : R.w) zero_r = zero_r && w == 0;
        R = md::UInt{}, Q = md::UInt{};
        R.w[0] = zero_r ? 1u : 0u;
    } else if (op == md::OP_DIV) {
        ok = md::divmod_p(A, B, M, &R);
    } else {
        ok = md::addsubmod(op, A, B, M, &Q, &R);
    }
    memcpy(q, Q.w, 4 * n_words), memcpy(r, R.w, 4 * n_words);
    return ok ? ZKHIP_OK : ZKHIP_ERR_INVALID;
}
int zkhip_modmul_host(const uint32_t a[8], const uint32_t b[8], const uint32_t modulus[8], uint32_t q[8], uint32_t r[8]) {
    return zkhip_modular_host_x(md::OP_MUL, 8, a, b, modulus, q, r);
}
int zkhip_modular_host(uint32_t op, const uint32_t a[8], const uint32_t b[8], const uint32_t modulus[8], uint32_t q[8], uint32_t r[8]) {
    return zkhip_modular_host_x(op, 8, a, b, modulus, q, r);
}

static int modular_tracegen(zkhip_ctx* ctx, uint32_t n_words, const uint32_t* modulus, const uint32_t* d_records, bool with_op, size_t n, unsigned log_height, uint32_t* d_trace,
                            uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !modulus || !d_trace || !d_bitwise_trace || !d_tuple_counts || log_height > 24 || (n && !d_records) || (n_words != 8 && n_words != 12)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height, T = (size_t)size_x * size_y;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "modmul_tracegen: more records than rows");
    if (size_x < 256 || size_y < 128 || T > ((size_t)1 << 27)) return set_error(ctx, ZKHIP_ERR_INVALID, "modmul_tracegen: the tuple table must cover (x < 256, y < 128)");
    ModWords pm{};
    memcpy(pm.w, modulus, 4 * n_words);
    const size_t rec_stride = 2 * n_words + (with_op ? 1 : 0);
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "modmul_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256), bb = (unsigned)(((size_t)1 << 16) + 255) / 256;
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_mod_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 0);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_mod_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);   // the range column of the 8-bit table
    if (n_words == 8)
        hipLaunchKernelGGL(k_modmul_trace<8>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_records, rec_stride, n, N, pm, d_trace, d_bitwise_trace,
                           d_tuple_counts, size_y, (uint32_t*)flag);
    else
        hipLaunchKernelGGL(k_modmul_trace<12>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_records, rec_stride, n, N, pm, d_trace, d_bitwise_trace,
                           d_tuple_counts, size_y, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_mod_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 1);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_mod_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return tracegen_finish(ctx, flag, "modular tracegen (a quotient beyond the modulus's limbs, operands of a subtraction further apart than the modulus, or an unknown operation)");
}

int zkhip_modmul_tracegen(zkhip_ctx* ctx, const uint32_t modulus[8], const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace,
                          uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    return modular_tracegen(ctx, 8, modulus, d_records, false, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y);
}

int zkhip_modular_tracegen_x(zkhip_ctx* ctx, uint32_t n_words, const uint32_t* modulus, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace,
                             uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    return modular_tracegen(ctx, n_words, modulus, d_records, true, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y);
}
int zkhip_modular_tracegen(zkhip_ctx* ctx, const uint32_t modulus[8], const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace,
                           uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    return zkhip_modular_tracegen_x(ctx, 8, modulus, d_records, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y);
}

int zkhip_vm_modmul_tracegen_x(zkhip_ctx* ctx, uint32_t n_words, const uint32_t* modulus, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height,
                               uint32_t* d_trace, uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || (n && !d_ts) || (n_words != 8 && n_words != 12)) return ZKHIP_ERR_INVALID;
    ZK_TRY(zkhip_modular_tracegen_x(ctx, n_words, modulus, d_records, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y));   // (op | a | b) records
    const size_t N = (size_t)1 << log_height;
    KernelScope ks(ctx, "vm_modmul_timestamps");
    hipLaunchKernelGGL(k_modmul_ts, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_ts, n, N, d_trace + (size_t)md::Cols(4 * n_words).TS * N);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}
int zkhip_vm_modmul_tracegen(zkhip_ctx* ctx, const uint32_t modulus[8], const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height,
                             uint32_t* d_trace, uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    return zkhip_vm_modmul_tracegen_x(ctx, 8, modulus, d_records, d_ts, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y);
}

}  // extern "C"
