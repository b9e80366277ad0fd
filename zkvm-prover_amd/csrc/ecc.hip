// ecc.hip -- the elliptic-curve chip on the device (include/zkhip_ecc.hpp: chord addition / tangent doubling on byte limbs, one point
// operation per row, 772 columns).  Record = op | x1[8] y1[8] x2[8] y2[8] | slope[8] (little-endian 32-bit words; the executor owns the
// one modular inversion of a call and hands the slope over).  One lane per row: the three identities' left sides as 576-bit signed
// integers (schoolbook products), binary long division by the modulus for the signed quotients and the canonical x3, y3, then the
// carries of the 3 x 64 limb equations; the row's lookups (163 byte pairs, 189 carry tuples, 2 comparisons) are counted into the bitwise
// and range-tuple tables in the same pass (wave-merged atomics, csrc/hist.hpp).  Replaces the trace generation of OpenVM's EcAddNe /
// EcDouble chips (openvm-ecc-circuit, un-vendored).
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_ecc.hpp"
#include "babybear.hpp"
#include "bigint_signed.hpp"
#include "hist.hpp"
#include "zkhip_internal.hpp"

namespace zk {
namespace {
namespace ec = zkhip::ecc;

struct EcWords {
    uint32_t p[12], a[12];
};
__global__ void k_ec_repr(uint32_t* c, size_t n, int to_m) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = to_m ? to_monty(c[i] % P) : from_monty(c[i]);
}

// NW = words of the modulus: 8 (32 limbs) or 12 (48 limbs: BLS12-381 G1, crates/circuits/batch-circuit/openvm.toml)
template <int NW>
__global__ __launch_bounds__(64) void k_ec_trace(const uint32_t* __restrict__ recs, size_t n, size_t N, EcWords cw, uint32_t* __restrict__ trace,
                                                 uint32_t* __restrict__ bitwise_range, uint32_t* __restrict__ tuple, uint32_t tuple_y, uint32_t* __restrict__ bad) {
    constexpr ec::Cols C(4 * NW);
    constexpr int L = 4 * NW, SW = Signed<NW>::SW;
    using S = Signed<NW>;
    const size_t row = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= N) return;
    if (row >= n) {
        for (size_t c = 0; c < C.WIDTH; c++) trace[c * N + row] = 0u;
        return;
    }
    const uint32_t* rec = recs + C.RECORD_WORDS * row;
    const uint32_t op = rec[0];
    if (op >= ec::N_OPS) atomicAdd(bad, 1u);
    const bool dbl = op == ec::OP_DOUBLE;
    uint32_t x1[NW], y1[NW], x2[NW], y2[NW], l[NW], x3[NW], y3[NW], q[3][NW + 1], v[SW];
    bool neg[3];
    for (int i = 0; i < NW; i++) x1[i] = rec[1 + i], y1[i] = rec[1 + NW + i], x2[i] = rec[1 + 2 * NW + i], y2[i] = rec[1 + 3 * NW + i], l[i] = rec[1 + 4 * NW + i];
    // identity 1: the slope
    for (int i = 0; i < SW; i++) v[i] = 0;
    if (dbl) {
        S::acc_product(v, l, y1, +1, 2), S::acc_product(v, x1, x1, -1, 3), S::acc_word(v, cw.a, -1);
    } else {
        S::acc_product(v, l, x2, +1, 1), S::acc_product(v, l, x1, -1, 1), S::acc_word(v, y2, -1), S::acc_word(v, y1, +1);
    }
    uint32_t res[NW];
    bool fits = S::signed_divmod(v, cw.p, q[0], res, &neg[0]);
    for (int i = 0; i < NW; i++) fits = fits && res[i] == 0;   // the slope the record carries must solve the first identity
    // identity 2: x3
    for (int i = 0; i < SW; i++) v[i] = 0;
    S::acc_product(v, l, l, +1, 1), S::acc_word(v, x1, -1), S::acc_word(v, dbl ? x1 : x2, -1);
    fits = S::signed_divmod(v, cw.p, q[1], x3, &neg[1]) && fits;
    // identity 3: y3
    for (int i = 0; i < SW; i++) v[i] = 0;
    S::acc_product(v, l, x1, +1, 1), S::acc_product(v, l, x3, -1, 1), S::acc_word(v, y1, -1);
    fits = S::signed_divmod(v, cw.p, q[2], y3, &neg[2]) && fits;
    if (!fits) atomicAdd(bad, 1u);

    auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
    auto put = [&](size_t col, uint32_t val) { trace[col * N + row] = to_monty(val); };
    const uint32_t* vars[7] = {x1, y1, x2, y2, l, x3, y3};
    for (int o = 0; o < 7; o++)
        for (int i = 0; i < L; i++) {
            put(L * o + i, byte_of(vars[o], i));
            if (!(i & 1)) hist_add(bitwise_range, byte_of(vars[o], i) * 256 + byte_of(vars[o], i + 1));
        }
    for (int e = 0; e < 3; e++) {
        for (int i = 0; i < (int)C.Q_LIMBS; i++) {
            put(C.Q + e * C.Q_LIMBS + i, byte_of(q[e], i));
            if (!(i & 1)) hist_add(bitwise_range, byte_of(q[e], i) * 256 + (i + 1 < (int)C.Q_LIMBS ? byte_of(q[e], i + 1) : 0u));
        }
        put(C.QS + e, neg[e] ? 1u : 0u);
    }
    // carries: position k of an identity's limb sum, plus the carry in, is 256 times the carry out
    for (int e = 0; e < 3; e++) {
        int64_t c = 0;
        const int64_t q_sign = neg[e] ? -1 : 1;
        for (int k = 0; k < (int)C.N_POS; k++) {
            int64_t s = c;
            for (int i = 0; i < (int)C.Q_LIMBS; i++) {
                const int j = k - i;
                if (j < 0 || j >= L) continue;
                s -= q_sign * (int64_t)byte_of(q[e], i) * byte_of(cw.p, j);
                if (i >= L) continue;
                const int64_t li = byte_of(l, i);
                if (e == 0) s += dbl ? 2 * li * byte_of(y1, j) - 3 * (int64_t)byte_of(x1, i) * byte_of(x1, j) : li * ((int64_t)byte_of(x2, j) - byte_of(x1, j));
                else if (e == 1) s += li * byte_of(l, j);
                else s += li * ((int64_t)byte_of(x1, j) - byte_of(x3, j));
            }
            if (k < L) {
                if (e == 0) s -= dbl ? (int64_t)byte_of(cw.a, k) : (int64_t)byte_of(y2, k) - byte_of(y1, k);
                else if (e == 1) s -= (int64_t)byte_of(x1, k) + byte_of(dbl ? x1 : x2, k) + byte_of(x3, k);
                else s -= (int64_t)byte_of(y1, k) + byte_of(y3, k);
            }
            if ((s & 255) != 0 && fits) atomicAdd(bad, 1u);   // (cannot happen: the identities hold)
            c = s >> 8;
            if (k < (int)C.N_CARRY) {
                const int64_t shifted = c + ec::CARRY_OFFSET;
                const uint32_t val = shifted < 0 || shifted >= (int64_t)256 * tuple_y ? 0u : (uint32_t)shifted;
                if ((int64_t)val != shifted) atomicAdd(bad, 1u);
                put(C.CX + e * C.N_CARRY + k, val & 255u), put(C.CY + e * C.N_CARRY + k, val >> 8);
                hist_add(tuple, (val & 255u) * tuple_y + (val >> 8));
            } else if (c != 0 && fits) {
                atomicAdd(bad, 1u);
            }
        }
    }
    // x3 < P, y3 < P: the most significant differing limb
    const uint32_t* outs[2] = {x3, y3};
    for (int o = 0; o < 2; o++) {
        int mark = -1;
        for (int i = L - 1; i >= 0; i--)
            if (byte_of(outs[o], i) != byte_of(cw.p, i)) {
                mark = i;
                break;
            }
        for (int i = 0; i < L; i++) put(C.MARK + L * o + i, i == mark ? 1u : 0u);
        const uint32_t diff = mark >= 0 ? byte_of(cw.p, mark) - byte_of(outs[o], mark) : 0u;
        put(C.DIFF + o, diff);
        hist_add(bitwise_range, ((diff - 1u) & 255u) * 256);
    }
    put(C.REAL, 1u), put(C.IS_DOUBLE, dbl ? 1u : 0u);
}

// the VM chip's timestamp column: row i carries the timestamp of call i
__global__ __launch_bounds__(256) void k_ec_ts(const uint32_t* __restrict__ ts, size_t n, size_t N, uint32_t* __restrict__ col) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row < N) col[row] = row < n ? to_monty(ts[row] % P) : 0u;
}

std::mutex g_mu;
struct AirKey {
    ec::Modulus p, a;
    uint32_t bitwise_bus, tuple_bus;
    bool operator<(const AirKey& o) const {
        if (p != o.p) return p < o.p;
        if (a != o.a) return a < o.a;
        if (bitwise_bus != o.bitwise_bus) return bitwise_bus < o.bitwise_bus;
        return tuple_bus < o.tuple_bus;
    }
};
std::map<AirKey, std::vector<uint32_t>> g_programs;   // kept for the life of the process

bool curve_of(uint32_t nw, const uint32_t* modulus, const uint32_t* a, ec::Curve* c) {
    if (nw != 8 && nw != 12) return false;
    c->p = zkhip::modular::load_words(modulus, nw), c->a = zkhip::modular::load_words(a, nw);
    // an odd modulus that fills its top two words (room for the (L + 1)-byte quotients), and a reduced coefficient
    return (c->p.w[0] & 1u) && c->p.w[nw - 1] != 0 && c->p.w[nw - 2] != 0 && ec::less(c->a, c->p);
}

}  // namespace
}  // namespace zk

using namespace zk;

extern "C" {

int zkhip_ec_air_x(const uint8_t* modulus, const uint8_t* a, uint32_t n_limbs, uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air* out) {
    if (!modulus || !a || !out || (n_limbs != 32 && n_limbs != 48)) return ZKHIP_ERR_INVALID;
    AirKey key;
    key.p.limbs = key.a.limbs = n_limbs;
    memcpy(key.p.data(), modulus, n_limbs), memcpy(key.a.data(), a, n_limbs);
    key.bitwise_bus = bitwise_bus, key.tuple_bus = tuple_bus;
    if (!(key.p[0] & 1u) || !key.p[n_limbs - 1]) return ZKHIP_ERR_INVALID;
    const ec::Cols C(n_limbs);
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_programs.find(key);
        if (it == g_programs.end()) {
            zkhip::air::AirBuilder b(C.WIDTH, 0);
            ec::ec_air(b, key.p, key.a, bitwise_bus, tuple_bus);
            it = g_programs.emplace(key, b.program()).first;
        }
        out->program = it->second.data(), out->program_len = it->second.size(), out->log_height = 0, out->width = C.WIDTH, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}
int zkhip_ec_air(const uint8_t modulus[32], const uint8_t a[32], uint32_t bitwise_bus, uint32_t tuple_bus, zkhip_air* out) {
    return zkhip_ec_air_x(modulus, a, 32, bitwise_bus, tuple_bus, out);
}

int zkhip_ec_host_x(uint32_t op, uint32_t n_words, const uint32_t* modulus, const uint32_t* a, const uint32_t* x1, const uint32_t* y1, const uint32_t* x2, const uint32_t* y2,
                    uint32_t* slope, uint32_t* x3, uint32_t* y3) {
    if (!modulus || !a || !x1 || !y1 || !x2 || !y2 || !slope || !x3 || !y3) return ZKHIP_ERR_INVALID;
    ec::Curve c;
    if (!curve_of(n_words, modulus, a, &c)) return ZKHIP_ERR_INVALID;
    using zkhip::modular::load_words;
    ec::U256 L, X3, Y3;
    if (!ec::ec_op(op, c, load_words(x1, n_words), load_words(y1, n_words), load_words(x2, n_words), load_words(y2, n_words), &L, &X3, &Y3)) return ZKHIP_ERR_INVALID;
    memcpy(slope, L.w, 4 * n_words), memcpy(x3, X3.w, 4 * n_words), memcpy(y3, Y3.w, 4 * n_words);
    return ZKHIP_OK;
}
int zkhip_ec_host(uint32_t op, const uint32_t modulus[8], const uint32_t a[8], const uint32_t x1[8], const uint32_t y1[8], const uint32_t x2[8],
                  const uint32_t y2[8], uint32_t slope[8], uint32_t x3[8], uint32_t y3[8]) {
    return zkhip_ec_host_x(op, 8, modulus, a, x1, y1, x2, y2, slope, x3, y3);
}

int zkhip_ec_tracegen_x(zkhip_ctx* ctx, uint32_t n_words, const uint32_t* modulus, const uint32_t* a, const uint32_t* d_records, size_t n, unsigned log_height,
                        uint32_t* d_trace, uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !modulus || !a || !d_trace || !d_bitwise_trace || !d_tuple_counts || log_height > 22 || (n && !d_records)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height, T = (size_t)size_x * size_y;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "ec_tracegen: more records than rows");
    if (size_x < 256 || size_y < 2048 || T > ((size_t)1 << 27)) return set_error(ctx, ZKHIP_ERR_INVALID, "ec_tracegen: the tuple table must cover (x < 256, y < 2048)");
    ec::Curve c;
    if (!curve_of(n_words, modulus, a, &c)) return set_error(ctx, ZKHIP_ERR_INVALID, "ec_tracegen: the modulus must be odd, of 8 or 12 words, and fill its top words, the coefficient reduced");
    EcWords cw{};
    memcpy(cw.p, modulus, 4 * n_words), memcpy(cw.a, a, 4 * n_words);
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "ec_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256), bb = (unsigned)(((size_t)1 << 16) + 255) / 256;
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_ec_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 0);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_ec_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);   // the range column of the 8-bit table
    if (n_words == 8)
        hipLaunchKernelGGL(k_ec_trace<8>, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, ctx->stream, d_records, n, N, cw, d_trace, d_bitwise_trace, d_tuple_counts, size_y,
                           (uint32_t*)flag);
    else
        hipLaunchKernelGGL(k_ec_trace<12>, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, ctx->stream, d_records, n, N, cw, d_trace, d_bitwise_trace, d_tuple_counts, size_y,
                           (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_ec_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 1);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_ec_repr, dim3(bb), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return tracegen_finish(ctx, flag, "ec tracegen (a slope that does not solve the chord / tangent identity, a quotient beyond L + 1 bytes, or an unknown operation)");
}
int zkhip_ec_tracegen(zkhip_ctx* ctx, const uint32_t modulus[8], const uint32_t a[8], const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace,
                      uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    return zkhip_ec_tracegen_x(ctx, 8, modulus, a, d_records, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y);
}

int zkhip_vm_ec_tracegen_x(zkhip_ctx* ctx, uint32_t n_words, const uint32_t* modulus, const uint32_t* a, const uint32_t* d_records, const uint32_t* d_ts, size_t n,
                           unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || (n && !d_ts) || (n_words != 8 && n_words != 12)) return ZKHIP_ERR_INVALID;
    ZK_TRY(zkhip_ec_tracegen_x(ctx, n_words, modulus, a, d_records, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y));
    const size_t N = (size_t)1 << log_height;
    KernelScope ks(ctx, "vm_ec_timestamps");
    hipLaunchKernelGGL(k_ec_ts, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_ts, n, N, d_trace + (size_t)ec::Cols(4 * n_words).TS * N);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}
int zkhip_vm_ec_tracegen(zkhip_ctx* ctx, const uint32_t modulus[8], const uint32_t a[8], const uint32_t* d_records, const uint32_t* d_ts, size_t n,
                         unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    return zkhip_vm_ec_tracegen_x(ctx, 8, modulus, a, d_records, d_ts, n, log_height, d_trace, d_bitwise_trace, d_tuple_counts, size_x, size_y);
}

}  // extern "C"
