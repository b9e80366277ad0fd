// fp2.hip -- the Fp2 chip on the device (include/zkhip_fp2.hpp: multiplication, division, addition, subtraction in Fp[u] / (u^2 + 1) on byte
// limbs, one operation per row, 648 columns).  Record = op | a0[8] a1[8] | b0[8] b1[8] (little-endian 32-bit words; a division's record
// holds the quotient x / y in the a slot -- the executor owns the inversion -- and its row is the product (x / y) y = x).  One lane per
// row: the two component identities' left sides as 576-bit signed integers, binary long division by the modulus for the signed quotients
// and the canonical r0, r1, then the carries of the 2 x 64 limb equations; the row's lookups (130 byte pairs, 126 carry tuples, 2 - 4
// comparisons) are counted into the bitwise and range-tuple tables in the same pass.  Replaces the trace generation of OpenVM's
// Fp2AddSub / Fp2MulDiv chips (openvm-algebra-circuit, un-vendored).
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_fp2.hpp"
#include "babybear.hpp"
#include "bigint_signed.hpp"
#include "hist.hpp"
#include "zkhip_internal.hpp"

namespace zk {
namespace {
namespace f2 = zkhip::fp2;

struct Fp2Words {
    uint32_t p[12];
};
__global__ void k_fp2_repr(uint32_t* c, size_t n, int to_m) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = to_m ? to_monty(c[i] % P) : from_monty(c[i]);
}

// NW = words of the modulus: 8 (32 limbs) or 12 (48 limbs: BLS12-381's Fp2, crates/circuits/batch-circuit/openvm.toml)
template <int NW>
__global__ __launch_bounds__(64) void k_fp2_trace(const uint32_t* __restrict__ recs, size_t n, size_t N, Fp2Words cw, uint32_t* __restrict__ trace,
                                                  uint32_t* __restrict__ bitwise_range, uint32_t* __restrict__ tuple, uint32_t tuple_y, uint32_t* __restrict__ bad) {
    constexpr f2::Cols C(4 * NW);
    constexpr int L = 4 * NW, SW = Signed<NW>::SW;
    const size_t row = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= N) return;
    if (row >= n) {
        for (size_t c = 0; c < C.WIDTH; c++) trace[c * N + row] = 0u;
        return;
    }
    const uint32_t* rec = recs + C.RECORD_WORDS * row;
    const uint32_t op_in = rec[0];
    if (op_in >= f2::N_OPS) atomicAdd(bad, 1u);
    const bool is_div = op_in == f2::OP_DIV;
    const uint32_t op = is_div ? (uint32_t)f2::OP_MUL : op_in;
    uint32_t a[2][NW], b[2][NW], r[2][NW], q[2][NW + 1], v[SW];
    bool neg[2];
    for (int i = 0; i < NW; i++) a[0][i] = rec[1 + i], a[1][i] = rec[1 + NW + i], b[0][i] = rec[1 + 2 * NW + i], b[1][i] = rec[1 + 3 * NW + i];
    bool fits = true;
    for (int e = 0; e < 2; e++) {
        for (int i = 0; i < SW; i++) v[i] = 0;
        if (op == f2::OP_MUL) {
            if (e == 0) Signed<NW>::acc_product(v, a[0], b[0], +1, 1), Signed<NW>::acc_product(v, a[1], b[1], -1, 1);
            else Signed<NW>::acc_product(v, a[0], b[1], +1, 1), Signed<NW>::acc_product(v, a[1], b[0], +1, 1);
        } else {
            Signed<NW>::acc_word(v, a[e], +1), Signed<NW>::acc_word(v, b[e], op == f2::OP_ADD ? +1 : -1);
        }
        fits = Signed<NW>::signed_divmod(v, cw.p, q[e], r[e], &neg[e]) && fits;
    }
    if (!fits) atomicAdd(bad, 1u);
    auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
    auto put = [&](size_t col, uint32_t val) { trace[col * N + row] = to_monty(val); };
    const uint32_t* vars[6] = {a[0], a[1], b[0], b[1], r[0], r[1]};
    for (int o = 0; o < 6; o++)
        for (int i = 0; i < L; i++) {
            put(L * o + i, byte_of(vars[o], i));
            if (!(i & 1)) hist_add(bitwise_range, byte_of(vars[o], i) * 256 + byte_of(vars[o], i + 1));
        }
    for (int e = 0; e < 2; e++) {
        for (int i = 0; i < (int)C.Q_LIMBS; i++) {
            put(C.Q + e * C.Q_LIMBS + i, byte_of(q[e], i));
            if (!(i & 1)) hist_add(bitwise_range, byte_of(q[e], i) * 256 + (i + 1 < (int)C.Q_LIMBS ? byte_of(q[e], i + 1) : 0u));
        }
        put(C.QS + e, neg[e] ? 1u : 0u);
    }
    for (int e = 0; e < 2; e++) {
        int64_t c = 0;
        const int64_t q_sign = neg[e] ? -1 : 1;
        for (int k = 0; k < (int)C.N_POS; k++) {
            int64_t s = c;
            for (int i = 0; i < (int)C.Q_LIMBS; i++) {
                const int j = k - i;
                if (j < 0 || j >= L) continue;
                s -= q_sign * (int64_t)byte_of(q[e], i) * byte_of(cw.p, j);
                if (i >= L || op != f2::OP_MUL) continue;
                if (e == 0) s += (int64_t)byte_of(a[0], i) * byte_of(b[0], j) - (int64_t)byte_of(a[1], i) * byte_of(b[1], j);
                else s += (int64_t)byte_of(a[0], i) * byte_of(b[1], j) + (int64_t)byte_of(a[1], i) * byte_of(b[0], j);
            }
            if (k < L) {
                if (op == f2::OP_ADD) s += (int64_t)byte_of(a[e], k) + byte_of(b[e], k);
                if (op == f2::OP_SUB) s += (int64_t)byte_of(a[e], k) - byte_of(b[e], k);
                s -= byte_of(r[e], k);
            }
            if ((s & 255) != 0 && fits) atomicAdd(bad, 1u);   // (cannot happen: the identities hold)
            c = s >> 8;
            if (k < (int)C.N_CARRY) {
                const int64_t shifted = c + f2::CARRY_OFFSET;
                const uint32_t val = shifted < 0 || shifted >= (int64_t)256 * tuple_y ? 0u : (uint32_t)shifted;
                if ((int64_t)val != shifted) atomicAdd(bad, 1u);
                put(C.CX + e * C.N_CARRY + k, val & 255u), put(C.CY + e * C.N_CARRY + k, val >> 8);
                hist_add(tuple, (val & 255u) * tuple_y + (val >> 8));
            } else if (c != 0 && fits) {
                atomicAdd(bad, 1u);
            }
        }
    }
    // r0, r1 < P; on a division row also a0, a1 < P (the quotient): the most significant differing limb
    for (int set = 0; set < 2; set++)
        for (int e = 0; e < 2; e++) {
            const uint32_t* x = set == 0 ? r[e] : a[e];
            const bool on = set == 0 || is_div;
            int mark = -1;
            if (on)
                for (int i = L - 1; i >= 0; i--)
                    if (byte_of(x, i) != byte_of(cw.p, i)) {
                        mark = byte_of(x, i) < byte_of(cw.p, i) ? i : -2;
                        break;
                    }
            if (on && mark < 0) atomicAdd(bad, 1u);   // a quotient (or result) that is not below the modulus
            const size_t mcol = (set == 0 ? C.MARK : C.MARK2) + L * e, dcol = (set == 0 ? C.DIFF : C.DIFF2) + e;
            for (int i = 0; i < L; i++) put(mcol + i, i == mark ? 1u : 0u);
            const uint32_t diff = mark >= 0 ? byte_of(cw.p, mark) - byte_of(x, mark) : 0u;
            put(dcol, diff);
            if (on) hist_add(bitwise_range, ((diff - 1u) & 255u) * 256);
        }
    put(C.REAL, 1u), put(C.IS_ADD, op_in == f2::OP_ADD ? 1u : 0u), put(C.IS_SUB, op_in == f2::OP_SUB ? 1u : 0u), put(C.IS_DIV, is_div ? 1u : 0u);
}

__global__ __launch_bounds__(256) void k_fp2_ts(const uint32_t* __restrict__ ts, size_t n, size_t N, uint32_t* __restrict__ col) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row < N) col[row] = row < n ? to_monty(ts[row] % P) : 0u;
}

std::mutex g_mu;
std::map<std::pair<f2::Modulus, std::pair<uint32_t, uint32_t>>, std::vector<uint32_t>> g_programs;   // (modulus, buses) -> program

// an odd modulus that fills its top two words (room for the (L + 1)-byte quotients)
bool modulus_ok(const uint32_t* m, uint32_t nw) { return (nw == 8 || nw == 12) && (m[0] & 1u) && m[nw - 1] != 0 && m[nw - 2] != 0; }

}  // namespace
}  // namespace zk

using namespace zk;

extern "C" {

int zkhip_fp2_air_x(const uint8_t* modulus, uint32_t n_limbs, uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air* out) {
    if (!modulus || !out || (n_limbs != 32 && n_limbs != 48) || !(modulus[0] & 1u) || !modulus[n_limbs - 1]) return ZKHIP_ERR_INVALID;
    f2::Modulus m;
    m.limbs = n_limbs;
    memcpy(m.data(), modulus, n_limbs);
    const f2::Cols C(n_limbs);
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        const auto key = std::make_pair(m, std::make_pair(bitwise_bus, tuple_bus));
        auto it = g_programs.find(key);
        if (it == g_programs.end()) {
            zkhip::air::AirBuilder b(C.WIDTH, 0);
            f2::fp2_air(b, m, bitwise_bus, tuple_bus);
            it = g_programs.emplace(key, b.program()).first;
        }
        out->program = it->second.data(), out->program_len = it->second.size(), out->log_height = 0, out->width = C.WIDTH, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}
int zkhip_fp2_air(const uint8_t modulus[32], uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air* out) { return zkhip_fp2_air_x(modulus, 32, bitwise_bus, tuple_bus, out); }

int zkhip_fp2_host_x(uint32_t op, uint32_t n_words, const uint32_t* modulus, const uint32_t* a, const uint32_t* b, uint32_t* r) {
    if (!modulus || !a || !b || !r || !modulus_ok(modulus, n_words)) return ZKHIP_ERR_INVALID;
    const f2::U256 p = zkhip::modular::load_words(modulus, n_words);
    f2::Elem A{zkhip::modular::load_words(a, n_words), zkhip::modular::load_words(a + n_words, n_words)},
        B{zkhip::modular::load_words(b, n_words), zkhip::modular::load_words(b + n_words, n_words)}, R;
    if (!f2::fp2_op(op, p, A, B, &R)) return ZKHIP_ERR_INVALID;
    memcpy(r, R.c0.w, 4 * n_words), memcpy(r + n_words, R.c1.w, 4 * n_words);
    return ZKHIP_OK;
}
int zkhip_fp2_host(uint32_t op, const uint32_t modulus[8], const uint32_t a[16], const uint32_t b[16], uint32_t r[16]) { return zkhip_fp2_host_x(op, 8, modulus, a, b, r); }

int zkhip_fp2_tracegen_x(zkhip_ctx* ctx, uint32_t n_words, const uint32_t* modulus, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace,
                         uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !modulus || !d_trace || !d_bitwise_trace || !d_tuple_counts || log_height > 22 || (n && !d_records)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height, T = (size_t)size_x * size_y;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "fp2_tracegen: more records than rows");
    if (size_x < 256 || size_y < 2048 || T > ((size_t)1 << 27)) return set_error(ctx, ZKHIP_ERR_INVALID, "fp2_tracegen: the tuple table must cover (x < 256, y < 2048)");
    if (!modulus_ok(modulus, n_words)) return set_error(ctx, ZKHIP_ERR_INVALID, "fp2_tracegen: the modulus must be odd, of 8 or 12 words, and fill its top words");
    Fp2Words cw{};
    memcpy(cw.p, modulus, 4 * n_words);
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "fp2_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256), bb = (unsigned)(((size_t)1 << 16) + 255) / 256;
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_fp2_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 0);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_fp2_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);
    if (n_words == 8)
        hipLaunchKernelGGL(k_fp2_trace<8>, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, ctx->stream, d_records, n, N, cw, d_trace, d_bitwise_trace, d_tuple_counts, size_y,
                           (uint32_t*)flag);
    else
        hipLaunchKernelGGL(k_fp2_trace<12>, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, ctx->stream, d_records, n, N, cw, d_trace, d_bitwise_trace, d_tuple_counts, size_y,
                           (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_fp2_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 1);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_fp2_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return tracegen_finish(ctx, flag, "fp2 tracegen (a quotient beyond L + 1 bytes, a division record whose quotient is not reduced, or an unknown operation)");
}
int zkhip_fp2_tracegen(zkhip_ctx* ctx, const uint32_t modulus[8], const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace,
                       uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    return zkhip_fp2_tracegen_x(ctx, 8, modulus, d_records, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y);
}

int zkhip_vm_fp2_tracegen_x(zkhip_ctx* ctx, uint32_t n_words, const uint32_t* modulus, const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height,
                            uint32_t* d_trace, uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || (n && !d_ts) || (n_words != 8 && n_words != 12)) return ZKHIP_ERR_INVALID;
    ZK_TRY(zkhip_fp2_tracegen_x(ctx, n_words, modulus, d_records, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y));
    const size_t N = (size_t)1 << log_height;
    KernelScope ks(ctx, "vm_fp2_timestamps");
    hipLaunchKernelGGL(k_fp2_ts, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_ts, n, N, d_trace + (size_t)f2::Cols(4 * n_words).TS * N);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}
int zkhip_vm_fp2_tracegen(zkhip_ctx* ctx, const uint32_t modulus[8], const uint32_t* d_records, const uint32_t* d_ts, size_t n, unsigned log_height,
                          uint32_t* d_trace, uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    return zkhip_vm_fp2_tracegen_x(ctx, 8, modulus, d_records, d_ts, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y);
}

}  // extern "C"
