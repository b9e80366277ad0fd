// vm_chips.hip -- the one-statement VM circuit on the device and across the ABI (include/zkhip_vm_circuit.hpp):
//   * zkhip_vm_air / zkhip_vm_program_table / zkhip_vm_decode: the segment's AIR set (programs built once, by the C++ definitions --
//     there is no second definition of these chips anywhere), its decode and program table;
//   * trace generators of the new chips: frame (gathers the program row of every executed instruction and splits operands into
//     bytes), the load/store adapter columns, the Poseidon2 chip with multiplicities,
//     and a transposer for the small chips whose rows the segmenting executor writes itself (ecall, leaf, merkle, connector);
//   * range-table multiplicities of scaled columns (requests like 4 * addr_hi).
// Replaces, for this backend, the trace generation of OpenVM's adapter / connector / persistent-memory chips (un-vendored;
// AGENTS.md:183-187 says the reference's GPU backend fills chip traces on the device).
#include <algorithm>
#include <map>
#include <mutex>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_vm_circuit.hpp"
#include "babybear.hpp"
#include "hist.hpp"
#include "zkhip_internal.hpp"

namespace zk {
int poseidon2_air_tracegen(zkhip_ctx* ctx, const uint32_t* d_inputs, size_t n_perms, unsigned log_height, uint32_t* d_trace);
namespace {

namespace vmc = zkhip::vmc;

std::once_flag g_airs_once;
vmc::AirShape g_airs[vmc::N_STATIC_AIRS];
std::mutex g_mod_mu;
std::map<std::pair<zkhip::modular::Modulus, unsigned>, vmc::AirShape> g_mod_airs;   // (modulus, 2 index + adapter) -> program
std::map<std::pair<zkhip::modular::Modulus, unsigned>, vmc::AirShape> g_fp2_airs;   // likewise for the fp2 chips
std::map<std::pair<std::pair<zkhip::modular::Modulus, zkhip::modular::Modulus>, unsigned>, vmc::AirShape> g_ec_airs;   // ((modulus, a), 2 index + adapter)

__global__ __launch_bounds__(256) void k_vm_frame(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
                                                  const uint32_t* __restrict__ zs, const uint32_t* __restrict__ rdp, const uint32_t* __restrict__ pcinc,
                                                  const uint32_t* __restrict__ pts1, const uint32_t* __restrict__ pts2, const uint32_t* __restrict__ pts3, size_t n, const uint32_t* __restrict__ program, size_t n_program, size_t N,
                                                  uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t k = 0;
    const bool valid = r < n && (k = idx[r]) < n_program;
    if (r < n && !valid) atomicAdd(bad, 1u);
    uint32_t col[vmc::FRAME_WIDTH];
#pragma unroll
    for (unsigned q = 0; q < vmc::FRAME_WIDTH; q++) col[q] = 0;
    if (valid) {
        col[0] = program[k];                                             // pc
        col[1] = to_monty(1u + vmc::TS_STEP * (uint32_t)r);              // timestamp of the instruction's first slot
#pragma unroll
        for (unsigned q = 1; q < vmc::PROGRAM_FIELDS; q++) col[1 + q] = program[(size_t)q * n_program + k];
        const uint32_t x = xs[r], y = ys[r], z = zs[r], p = rdp[r];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            col[18 + i] = to_monty((x >> (8 * i)) & 255u), col[22 + i] = to_monty((y >> (8 * i)) & 255u), col[26 + i] = to_monty((z >> (8 * i)) & 255u);
        }
        col[30] = to_monty(p & 0xffffu), col[31] = to_monty(p >> 16);
        const uint32_t inc = pcinc[r];
        if (inc >= P) atomicAdd(bad, 1u);
        col[32] = to_monty(inc % P), col[33] = MONTY_ONE;
        // register adapter: previous timestamps and gaps of the rs1 / rs2 / rd accesses (zero where the instruction skips the access)
        const uint32_t ts = 1u + vmc::TS_STEP * (uint32_t)r;
        const uint32_t used[3] = {col[10], col[11], col[13]}, at[3] = {ts, ts + 2, ts + 12}, prev[3] = {pts1[r], pts2[r], pts3[r]};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (!used[a]) continue;
            const uint32_t gap = at[a] - prev[a] - 1;
            if (prev[a] >= at[a] || (gap >> (16 + vmc::GAP_HI_BITS))) atomicAdd(bad, 1u);
            col[34 + 3 * a] = to_monty(prev[a] % P), col[35 + 3 * a] = to_monty(gap & 0xffffu), col[36 + 3 * a] = to_monty((gap >> 16) & ((1u << vmc::GAP_HI_BITS) - 1));
        }
    }
#pragma unroll
    for (unsigned q = 0; q < vmc::FRAME_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
}

// load/store adapter columns 33..47: ts | base[4] | imm_lo imm_hi | addr_lo addr_hi | carry0 carry1 | word_lo | prev_ts gap_lo gap_hi
__global__ __launch_bounds__(256) void k_vm_ls_adapter(const uint32_t* __restrict__ tss, const uint32_t* __restrict__ bases, const uint32_t* __restrict__ imms,
                                                       const uint32_t* __restrict__ ptss, size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t col[15];
#pragma unroll
    for (int q = 0; q < 15; q++) col[q] = 0;
    if (r < n) {
        const uint32_t base = bases[r], imm = imms[r];
        const uint32_t lo = (base & 0xffffu) + (imm & 0xffffu), c0 = lo >> 16;
        const uint32_t hi = (base >> 16) + (imm >> 16) + c0, c1 = hi >> 16;
        col[0] = to_monty(tss[r]);
#pragma unroll
        for (int i = 0; i < 4; i++) col[1 + i] = to_monty((base >> (8 * i)) & 255u);
        col[5] = to_monty(imm & 0xffffu), col[6] = to_monty(imm >> 16);
        col[7] = to_monty(lo & 0xffffu), col[8] = to_monty(hi & 0xffffu), col[9] = to_monty(c0), col[10] = to_monty(c1);
        col[11] = to_monty((lo & 0xffffu) >> 2);
        const uint32_t at = tss[r] + 4, prev = ptss[r], gap = at - prev - 1;
        if (prev >= at || (gap >> (16 + vmc::GAP_HI_BITS))) atomicAdd(bad, 1u);
        col[12] = to_monty(prev % P), col[13] = to_monty(gap & 0xffffu), col[14] = to_monty((gap >> 16) & ((1u << vmc::GAP_HI_BITS) - 1));
    }
#pragma unroll
    for (int q = 0; q < 15; q++) trace[(size_t)(33 + q) * N + r] = col[q];
}

__global__ __launch_bounds__(256) void k_fill_prefix(uint32_t* __restrict__ col, size_t n_ones, size_t N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) col[i] = i < n_ones ? MONTY_ONE : 0u;
}

// rows (row-major, canonical) -> column-major Montgomery trace; rows >= n get the padding row
__global__ __launch_bounds__(256) void k_rows_to_trace(const uint32_t* __restrict__ rows, size_t n, uint32_t width, size_t N, const uint32_t* __restrict__ pad,
                                                       uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * width) return;
    const size_t q = i / N, r = i % N;
    uint32_t v = r < n ? rows[r * width + q] : (pad ? pad[q] : 0u);
    if (v >= P) {
        atomicAdd(bad, 1u);
        v %= P;
    }
    trace[i] = to_monty(v);
}

__global__ void k_counts_repr(uint32_t* c, size_t n, int to_m) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = to_m ? to_monty(c[i] % P) : from_monty(c[i]);
}
__global__ __launch_bounds__(256) void k_range_counts_scaled(const uint32_t* __restrict__ values, size_t n, uint32_t scale, uint32_t T,
                                                             uint32_t* __restrict__ hist, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk[HOT_SLOTS], hc[HOT_SLOTS];
    hot_init(hk, hc);
    uint32_t n_bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint64_t v = (uint64_t)from_monty(values[i]) * scale;
        if (v >= T) {
            n_bad++;
            continue;
        }
        hot_add(hk, hc, hist, (uint32_t)v);
    }
    if (n_bad) atomicAdd(bad, n_bad);
    hot_flush(hk, hc, hist);
}

// ---- native field / extension / castf chips (include/zkhip_vm_circuit.hpp native_arith_air, native_ext_air, castf_vm_air): ONE row per
// call, core and memory adapter together; the records are what the segmenting executor saw (operands, the result words before, the word
// pointer, the timestamp, the words' previous accesses) -- the result, the inverse and the canonical-word columns are computed here.
__device__ __forceinline__ uint32_t nat_inv(uint32_t a_monty) { return mpow(a_monty, P - 2); }
// columns lo hi hi_gap top top_inv of a canonical word v < p
__device__ __forceinline__ void canonical_cols(uint32_t v, uint32_t* col) {
    const uint32_t lo = v & 0xffffu, hi = v >> 16, gap = zkhip::native::P_HI - hi;
    col[0] = to_monty(lo), col[1] = to_monty(hi), col[2] = to_monty(gap), col[3] = gap == 0 ? MONTY_ONE : 0u, col[4] = gap == 0 ? 0u : nat_inv(to_monty(gap));
}
// columns prev_ts gap_lo gap_hi of a word access at timestamp `at`
__device__ __forceinline__ void access_cols(uint32_t prev, uint32_t at, uint32_t* col, uint32_t* bad) {
    const uint32_t gap = at - prev - 1;
    if (prev >= at || (gap >> (16 + vmc::GAP_HI_BITS))) atomicAdd(bad, 1u);
    col[0] = to_monty(prev % P), col[1] = to_monty(gap & 0xffffu), col[2] = to_monty((gap >> 16) & ((1u << vmc::GAP_HI_BITS) - 1));
}
__global__ __launch_bounds__(256) void k_vm_native_arith(const uint32_t* __restrict__ recs, size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t col[vmc::NATIVE_ARITH_WIDTH];
#pragma unroll
    for (unsigned q = 0; q < vmc::NATIVE_ARITH_WIDTH; q++) col[q] = 0;
    if (r < n) {
        const uint32_t* rec = recs + r * 9;   // op | b | c | previous result word | word pointer | ts | previous timestamps of the three words
        const uint32_t op = rec[0], bw = rec[1], cw = rec[2], prev = rec[3], base = rec[4], ts = rec[5];
        const uint32_t bm = to_monty(bw % P), cm = to_monty(cw % P);
        uint32_t am = 0, inv = 0;
        if (op == 0) am = madd(bm, cm);
        else if (op == 1) am = msub(bm, cm);
        else if (op == 2) am = mmul(bm, cm);
        else if (op == 3 && cm != 0) inv = nat_inv(cm), am = mmul(bm, inv);
        else atomicAdd(bad, 1u);
        col[0] = to_monty(ts % P), col[1] = to_monty(base % P);
        col[2] = to_monty(bw & 0xffffu), col[3] = to_monty(bw >> 16), col[4] = to_monty(cw & 0xffffu), col[5] = to_monty(cw >> 16);
        canonical_cols(from_monty(am), col + 6);
        col[11] = to_monty(prev & 0xffffu), col[12] = to_monty(prev >> 16);
        if (op < 4) col[13 + op] = MONTY_ONE;
        col[17] = inv;
        for (int k = 0; k < 3; k++) access_cols(rec[6 + k], ts + 5, col + 18 + 3 * k, bad);
    }
#pragma unroll
    for (unsigned q = 0; q < vmc::NATIVE_ARITH_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
}
__device__ __forceinline__ void nat_ext_mul(const uint32_t* x, const uint32_t* y, uint32_t* z) {   // Montgomery, X^4 = 11
    const uint32_t w = to_monty(zkhip::native::W);
    uint32_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) t[i + j] = madd(t[i + j], mmul(x[i], y[j]));
    for (int i = 0; i < 3; i++) z[i] = madd(t[i], mmul(w, t[i + 4]));
    z[3] = t[3];
}
__global__ __launch_bounds__(256) void k_vm_native_ext(const uint32_t* __restrict__ recs, size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    auto put = [&](unsigned q, uint32_t v) { trace[(size_t)q * N + r] = v; };
    if (r >= n) {
        for (unsigned q = 0; q < vmc::NATIVE_EXT_WIDTH; q++) put(q, 0u);
        return;
    }
    const uint32_t* rec = recs + r * 27;   // op | x[4] | y[4] | previous result words [4] | word pointer | ts | previous timestamps of the twelve words
    const uint32_t op = rec[0], base = rec[13], ts = rec[14];
    uint32_t x[4], y[4], z[4], inv[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) x[i] = to_monty(rec[1 + i] % P), y[i] = to_monty(rec[5 + i] % P);
    if (op == 0) for (int i = 0; i < 4; i++) z[i] = madd(x[i], y[i]);
    else if (op == 1) for (int i = 0; i < 4; i++) z[i] = msub(x[i], y[i]);
    else if (op == 2) nat_ext_mul(x, y, z);
    else if (op == 3) {
        // y^-1 through the norm to F[X^2] and on to F (include/zkhip_native.hpp ext_inv): y = A + X B, D = A^2 - X^2 B^2, y^-1 = (A - X B) / D
        const uint32_t w = to_monty(zkhip::native::W);
        auto kmul = [&](const uint32_t* a, const uint32_t* b, uint32_t* o) {
            const uint32_t o0 = madd(mmul(a[0], b[0]), mmul(w, mmul(a[1], b[1]))), o1 = madd(mmul(a[0], b[1]), mmul(a[1], b[0]));
            o[0] = o0, o[1] = o1;
        };
        const uint32_t A[2] = {y[0], y[2]}, B[2] = {y[1], y[3]};
        uint32_t A2[2], B2[2], D[2], Di[2], RA[2], RB[2];
        kmul(A, A, A2), kmul(B, B, B2);
        D[0] = msub(A2[0], mmul(w, B2[1])), D[1] = msub(A2[1], B2[0]);
        const uint32_t nrm = msub(mmul(D[0], D[0]), mmul(w, mmul(D[1], D[1])));
        if (nrm == 0) atomicAdd(bad, 1u);
        const uint32_t ni = nat_inv(nrm);
        Di[0] = mmul(D[0], ni), Di[1] = mmul(msub(0u, D[1]), ni);
        kmul(A, Di, RA), kmul(B, Di, RB);
        inv[0] = RA[0], inv[2] = RA[1], inv[1] = msub(0u, RB[0]), inv[3] = msub(0u, RB[1]);
        nat_ext_mul(x, inv, z);
    } else {
        atomicAdd(bad, 1u);
        for (int i = 0; i < 4; i++) z[i] = 0;
    }
    put(0, to_monty(ts % P)), put(1, to_monty(base % P));
    for (int i = 0; i < 4; i++) {
        put(2 + 2 * i, to_monty(rec[1 + i] & 0xffffu)), put(3 + 2 * i, to_monty(rec[1 + i] >> 16));
        put(10 + 2 * i, to_monty(rec[5 + i] & 0xffffu)), put(11 + 2 * i, to_monty(rec[5 + i] >> 16));
        uint32_t c5[5];
        canonical_cols(from_monty(z[i]), c5);
        for (int k = 0; k < 5; k++) put(18 + 5 * i + k, c5[k]);
        put(38 + 2 * i, to_monty(rec[9 + i] & 0xffffu)), put(39 + 2 * i, to_monty(rec[9 + i] >> 16));
        put(46 + i, op == (uint32_t)i ? MONTY_ONE : 0u);
        put(50 + i, inv[i]);
    }
    for (int k = 0; k < 12; k++) {
        uint32_t c3[3];
        access_cols(rec[15 + k], ts + 5, c3, bad);
        for (int q = 0; q < 3; q++) put(54 + 3 * k + q, c3[q]);
    }
}
__global__ __launch_bounds__(256) void k_vm_castf(const uint32_t* __restrict__ recs, size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t col[vmc::CASTF_WIDTH];
#pragma unroll
    for (unsigned q = 0; q < vmc::CASTF_WIDTH; q++) col[q] = 0;
    if (r < n) {
        const uint32_t* rec = recs + r * 6;   // x | previous output word | word pointer | ts | previous timestamps of the two words
        const uint32_t x = rec[0], prev = rec[1], base = rec[2], ts = rec[3];
        if (x >= zkhip::native::CASTF_BOUND) atomicAdd(bad, 1u);
        col[0] = to_monty(ts % P), col[1] = to_monty(base % P);
        for (int i = 0; i < 4; i++) col[2 + i] = to_monty((x >> (8 * i)) & 255u);
        col[6] = to_monty(((x >> 24) & 63u) * 4u);
        col[7] = to_monty(prev & 0xffffu), col[8] = to_monty(prev >> 16);
        access_cols(rec[4], ts + 5, col + 9, bad), access_cols(rec[5], ts + 5, col + 12, bad);
        col[15] = MONTY_ONE;
    }
#pragma unroll
    for (unsigned q = 0; q < vmc::CASTF_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
}

int check_flag(zkhip_ctx* ctx, void* flag, const char* what) { return tracegen_finish(ctx, flag, what); }

}  // namespace
}  // namespace zk

using namespace zk;

extern "C" {

size_t zkhip_vm_n_airs(void) { return vmc::N_STATIC_AIRS; }

int zkhip_vm_air(unsigned id, zkhip_air* out, size_t* prep_width) {
    if (id >= vmc::N_STATIC_AIRS || !out) return ZKHIP_ERR_INVALID;
    try {
        std::call_once(g_airs_once, [] {
            for (unsigned i = 0; i < vmc::N_STATIC_AIRS; i++) g_airs[i] = vmc::build_air(i);
        });
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    const vmc::AirShape& s = g_airs[id];
    out->program = s.program.data(), out->program_len = s.program.size(), out->log_height = 0, out->width = s.width, out->n_pvs = s.n_pvs;
    out->prep_trace = nullptr, out->prep_commit = nullptr;
    if (prep_width) *prep_width = s.prep_width;
    return ZKHIP_OK;
}

int zkhip_vm_modmul_air(const uint8_t modulus[32], unsigned index, int adapter, zkhip_air* out) { return zkhip_vm_modmul_air_x(modulus, 32, index, adapter, out); }
int zkhip_vm_modmul_air_x(const uint8_t* modulus, uint32_t n_limbs, unsigned index, int adapter, zkhip_air* out) {
    if (!modulus || !out || index >= vmc::MAX_MODULI || (n_limbs != 32 && n_limbs != 48)) return ZKHIP_ERR_INVALID;
    zkhip::modular::Modulus m;
    m.limbs = n_limbs;
    memcpy(m.data(), modulus, n_limbs);
    try {
        std::lock_guard<std::mutex> lk(g_mod_mu);
        const auto key = std::make_pair(m, 2 * index + (adapter ? 1u : 0u));
        auto it = g_mod_airs.find(key);
        if (it == g_mod_airs.end()) it = g_mod_airs.emplace(key, vmc::build_modmul_air(m, index, adapter != 0)).first;
        const vmc::AirShape& s = it->second;
        out->program = s.program.data(), out->program_len = s.program.size(), out->log_height = 0, out->width = s.width, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}

int zkhip_vm_ec_air(const uint8_t modulus[32], const uint8_t a[32], unsigned index, int adapter, zkhip_air* out) {
    return zkhip_vm_ec_air_x(modulus, a, 32, index, adapter, out);
}
int zkhip_vm_ec_air_x(const uint8_t* modulus, const uint8_t* a, uint32_t n_limbs, unsigned index, int adapter, zkhip_air* out) {
    if (!modulus || !a || !out || index >= vmc::MAX_CURVES || (n_limbs != 32 && n_limbs != 48) || !(modulus[0] & 1u) || !modulus[n_limbs - 1]) return ZKHIP_ERR_INVALID;
    zkhip::modular::Modulus m, ca;
    m.limbs = ca.limbs = n_limbs;
    memcpy(m.data(), modulus, n_limbs), memcpy(ca.data(), a, n_limbs);
    try {
        std::lock_guard<std::mutex> lk(g_mod_mu);
        const auto key = std::make_pair(std::make_pair(m, ca), 2 * index + (adapter ? 1u : 0u));
        auto it = g_ec_airs.find(key);
        if (it == g_ec_airs.end()) it = g_ec_airs.emplace(key, vmc::build_ec_air(m, ca, index, adapter != 0)).first;
        const vmc::AirShape& s = it->second;
        out->program = s.program.data(), out->program_len = s.program.size(), out->log_height = 0, out->width = s.width, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}

int zkhip_vm_fp2_air(const uint8_t modulus[32], unsigned index, int adapter, zkhip_air* out) { return zkhip_vm_fp2_air_x(modulus, 32, index, adapter, out); }
int zkhip_vm_fp2_air_x(const uint8_t* modulus, uint32_t n_limbs, unsigned index, int adapter, zkhip_air* out) {
    if (!modulus || !out || index >= vmc::MAX_FP2 || (n_limbs != 32 && n_limbs != 48) || !(modulus[0] & 1u) || !modulus[n_limbs - 1]) return ZKHIP_ERR_INVALID;
    zkhip::modular::Modulus m;
    m.limbs = n_limbs;
    memcpy(m.data(), modulus, n_limbs);
    try {
        std::lock_guard<std::mutex> lk(g_mod_mu);
        const auto key = std::make_pair(m, 2 * index + (adapter ? 1u : 0u));
        auto it = g_fp2_airs.find(key);
        if (it == g_fp2_airs.end()) it = g_fp2_airs.emplace(key, vmc::build_fp2_air(m, index, adapter != 0)).first;
        const vmc::AirShape& s = it->second;
        out->program = s.program.data(), out->program_len = s.program.size(), out->log_height = 0, out->width = s.width, out->n_pvs = 0;
        out->prep_trace = nullptr, out->prep_commit = nullptr;
    } catch (const std::exception&) {
        return ZKHIP_ERR_INVALID;
    }
    return ZKHIP_OK;
}

int zkhip_vm_decode(uint32_t word, uint32_t pc, uint32_t out[17], int* legal) {
    if (!out) return ZKHIP_ERR_INVALID;
    const vmc::Decoded d = vmc::decode(word, pc);
    const auto f = d.fields();
    for (size_t i = 0; i < vmc::PROGRAM_FIELDS; i++) out[i] = f[i];
    if (legal) *legal = d.legal ? 1 : 0;
    return ZKHIP_OK;
}

int zkhip_vm_program_table(const uint32_t* words, size_t n_words, uint32_t pc_base, unsigned log_program, uint32_t* out) {
    if (!out || (n_words && !words) || log_program > 27 || n_words > ((size_t)1 << log_program)) return ZKHIP_ERR_INVALID;
    const std::vector<uint32_t> t = vmc::program_table(std::vector<uint32_t>(words, words + n_words), pc_base, log_program);
    memcpy(out, t.data(), t.size() * 4);
    return ZKHIP_OK;
}

int zkhip_vm_frame_tracegen(zkhip_ctx* ctx, const uint32_t* d_pc_index, const uint32_t* d_x, const uint32_t* d_y, const uint32_t* d_z,
                            const uint32_t* d_rd_prev, const uint32_t* d_pc_inc, const uint32_t* d_prev_ts_rs1, const uint32_t* d_prev_ts_rs2,
                            const uint32_t* d_prev_ts_rd, size_t n, const uint32_t* d_program, size_t n_program, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_program || log_height > 25 ||
        (n && (!d_pc_index || !d_x || !d_y || !d_z || !d_rd_prev || !d_pc_inc || !d_prev_ts_rs1 || !d_prev_ts_rs2 || !d_prev_ts_rd)))
        return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "vm_frame_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "vm_frame_tracegen");
    hipLaunchKernelGGL(k_vm_frame, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_pc_index, d_x, d_y, d_z, d_rd_prev, d_pc_inc,
                       d_prev_ts_rs1, d_prev_ts_rs2, d_prev_ts_rd, n, d_program, n_program, N, d_trace, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return check_flag(ctx, flag, "vm_frame_tracegen (instruction index beyond the program, a pc step that is not a field element, or a timestamp gap out of range)");
}

int zkhip_vm_loadstore_tracegen(zkhip_ctx* ctx, const uint32_t* d_case, const uint32_t* d_read, const uint32_t* d_prev, const uint32_t* d_ts,
                                const uint32_t* d_base, const uint32_t* d_imm, const uint32_t* d_prev_ts, size_t n, unsigned log_height, uint32_t* d_trace,
                                uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_case || !d_read || !d_prev || !d_ts || !d_base || !d_imm || !d_prev_ts)))
        return ZKHIP_ERR_INVALID;
    // the core's 33 columns (same stride: they are the first 33 columns of this chip), then the adapter's
    ZK_TRY(zkhip_rv32_loadstore_tracegen(ctx, d_case, d_read, d_prev, n, log_height, d_trace, d_bitwise_trace));
    const size_t N = (size_t)1 << log_height;
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "vm_loadstore_adapter_tracegen");
    hipLaunchKernelGGL(k_vm_ls_adapter, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_ts, d_base, d_imm, d_prev_ts, n, N, d_trace, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return check_flag(ctx, flag, "vm_loadstore_tracegen (a timestamp gap out of range)");
}

int zkhip_vm_poseidon2_tracegen(zkhip_ctx* ctx, const uint32_t* d_inputs, size_t n, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27) return ZKHIP_ERR_INVALID;
    ZK_TRY(poseidon2_air_tracegen(ctx, d_inputs, n, log_height, d_trace));
    const size_t N = (size_t)1 << log_height;
    KernelScope ks(ctx, "vm_poseidon2_multiplicities");
    hipLaunchKernelGGL(k_fill_prefix, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_trace + (size_t)ZKHIP_POSEIDON2_AIR_WIDTH * N, n, N);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

// the rows of the native field / extension / castf chips from the executor's call records (include/zkhip_vm_exec.hpp SegmentRecords)
static int native_rows(zkhip_ctx* ctx, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace, int which, const char* name) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27 || (n && !d_records)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, std::string(name) + ": more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, name);
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    if (which == 0) hipLaunchKernelGGL(k_vm_native_arith, grid, block, 0, ctx->stream, d_records, n, N, d_trace, (uint32_t*)flag);
    else if (which == 1) hipLaunchKernelGGL(k_vm_native_ext, grid, block, 0, ctx->stream, d_records, n, N, d_trace, (uint32_t*)flag);
    else hipLaunchKernelGGL(k_vm_castf, grid, block, 0, ctx->stream, d_records, n, N, d_trace, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return check_flag(ctx, flag, (std::string(name) + " (an unknown operation, a division by zero, a castf operand of 2^30 or more, or a timestamp gap out of range)").c_str());
}
int zkhip_vm_native_arith_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace) {
    return native_rows(ctx, d_records, n, log_height, d_trace, 0, "vm_native_arith_tracegen");
}
int zkhip_vm_native_ext_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace) {
    return native_rows(ctx, d_records, n, log_height, d_trace, 1, "vm_native_ext_tracegen");
}
int zkhip_vm_castf_tracegen(zkhip_ctx* ctx, const uint32_t* d_records, size_t n, unsigned log_height, uint32_t* d_trace) {
    return native_rows(ctx, d_records, n, log_height, d_trace, 2, "vm_castf_tracegen");
}

int zkhip_rows_tracegen(zkhip_ctx* ctx, const uint32_t* d_rows, size_t n, size_t width, unsigned log_height, uint32_t* d_trace, const uint32_t* pad_row) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || width == 0 || width > 4096 || log_height > 27 || (n && !d_rows)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rows_tracegen: more rows than the trace height");
    void *flag = nullptr, *scratch = nullptr;
    ZK_TRY(get_scratch(ctx, 2, 16 + 4 * 4096, &scratch));   // [error flag | the padding row]
    uint32_t* d_pad = nullptr;
    KernelScope ks(ctx, "rows_tracegen");
    if (ctx->defer_tracegen_checks && ctx->d_deferred_bad) {
        flag = ctx->d_deferred_bad;   // (counted in place: tracegen_flag, csrc/api.hip)
    } else {
        flag = scratch;
        ZK_HIP_CHECK(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
    }
    if (pad_row) {
        d_pad = (uint32_t*)scratch + 4;
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_pad, pad_row, width * 4, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // pad_row may be a temporary of the caller
    }
    hipLaunchKernelGGL(k_rows_to_trace, dim3((unsigned)((N * width + 255) / 256)), dim3(256), 0, ctx->stream, d_rows, n, (uint32_t)width, N, d_pad, d_trace,
                       (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return check_flag(ctx, flag, "rows_tracegen (a word is not a canonical field element)");
}

int zkhip_range_counts_scaled_tracegen(zkhip_ctx* ctx, const uint32_t* d_values, size_t n, uint32_t scale, unsigned log_table, uint32_t* d_counts,
                                       int accumulate) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_counts || log_table > 27 || scale == 0 || (n && !d_values)) return ZKHIP_ERR_INVALID;
    const size_t T = (size_t)1 << log_table;
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "range_counts_scaled_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!accumulate) ZK_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, T * 4, ctx->stream));
    else if (!ctx->tables_canonical) hipLaunchKernelGGL(k_counts_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 0);
    if (n) {
        const unsigned blocks = (unsigned)std::min<size_t>((n + 256 * 8 - 1) / (256 * 8), HOT_MAX_BLOCKS);
        hipLaunchKernelGGL(k_range_counts_scaled, dim3(blocks), dim3(256), 0, ctx->stream, d_values, n, scale, (uint32_t)T, d_counts, (uint32_t*)flag);
    }
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_counts_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return check_flag(ctx, flag, "range_counts_scaled_tracegen (scaled value outside the table)");
}

}  // extern "C"
