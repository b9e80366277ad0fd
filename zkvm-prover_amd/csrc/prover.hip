// prover.hip -- the STARK prover behind zkhip_keygen / zkhip_prove (what
// `sdk.prove(..)` -> `StarkEngine::prove` does for one segment, crates/prover/src/prover/mod.rs:355-357).
//
// Pipeline (north_star order): trace commit (coset LDE + Merkle-Poseidon2) -> alpha -> per-row
// constraint evaluation / quotient -> quotient-chunk LDE + commit -> zeta -> openings ->
// FRI batching (reduced openings) -> FRI fold loop with per-round commit + PoW -> query PoW ->
// query openings.  Protocol and proof layout: DESIGN.md; byte-for-byte twin of oracle/stark.c.
//
// MI355X design points:
//  * Everything a proof needs is resident: LDEs, trees, FRI layers live in one workspace sized
//    at keygen (a 2^22 x 300 chunk trace needs 13.6 GB of the 288 GB; zkhip_pk_workspace_bytes).
//  * No host round trip inside a proof: the transcript is a device object (transcript.hip), the
//    challenges are read by the next kernel straight from HBM, PoW nonces are searched on the
//    device, and the proof is assembled in a device buffer whose layout is static (FRI proofs
//    have fixed shape).  One D2H copy at the end.
//  * Hot kernels are one-row-per-lane over column-major matrices (coalesced 256 B per wave per
//    column); reductions over rows (openings) are two-stage and deterministic.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>

#include "air_compile.hpp"
#include "poseidon2_coop.hpp"
#include "quotient_jit.hpp"
#include "transcript.hpp"
#include "zkhip_internal.hpp"

namespace zk {

static constexpr unsigned QBS = 256;  // quotient kernel block size

// ---------------------------------------------------------------------------------------------
// small device helpers
__device__ __forceinline__ Ext ld_ext(const uint32_t* p) { return Ext{{p[0], p[1], p[2], p[3]}}; }
__device__ __forceinline__ void st_ext(uint32_t* p, const Ext& e) {
    p[0] = e.c[0], p[1] = e.c[1], p[2] = e.c[2], p[3] = e.c[3];
}
// w^i for i < 2^log_m from the half-size table
__device__ __forceinline__ uint32_t root_pow(const uint32_t* tw, unsigned tw_shift, unsigned log_m, uint32_t i) {
    uint32_t half = 1u << (log_m - 1);
    return i < half ? tw[(size_t)i << tw_shift] : mneg(tw[(size_t)(i - half) << tw_shift]);
}

// ---------------------------------------------------------------------------------------------
// K5: per-row constraint interpreter.  One LDE row per lane, slots in LDS as [slot][lane].
struct QuotArgs {
    const uint32_t* code;
    uint32_t n_instr;
    const uint32_t* consts;
    const uint32_t* pvs;    // Montgomery
    const uint32_t* apow;   // n_cons extension elements: alpha^(n_cons-1-k)
    const uint32_t* lde;    // column-major, stride = M
    const uint32_t* perm;   // permutation LDE (column-major, stride = M) or null
    const uint32_t* lchal;  // N_CHAL interaction challenge coordinates or null
    const uint32_t* expo;   // 4 coordinates of the exposed cumulative sum or null
    const uint32_t* prep;   // preprocessed LDE (column-major, stride = M) or null
    uint32_t* q;            // 4 columns of M (quotient values, bit-reversed LDE order)
    const uint32_t* inv_zh; // 2^b values, index = natural index mod 2^b
    const uint32_t* zh;     // 2^b values
    uint32_t gen;           // coset shift (Montgomery)
    uint32_t w_n_inv;       // w_N^-1
    unsigned lh, b;
    uint32_t n_rows;        // rows to evaluate: the first N * (quotient chunks) of the M LDE rows (M for the stage-level entry)
};
// Z_H(x) / (x - 1) and Z_H(x) / (x - w_N^-1) for the first n_rows rows of a bit-reversed LDE of 2^h points (coset shift `gen`): the selector
// tables of the shared-rows constraint kernel (csrc/quotient_jit.hpp), generated with the key -- the values its plain form computes per wave
__global__ __launch_bounds__(256) void k_gen_selectors(uint32_t* __restrict__ first, uint32_t* __restrict__ last, uint32_t n_rows, uint32_t h, uint32_t b, uint32_t gen,
                                                       uint32_t w_m, uint32_t w_n_inv, const uint32_t* __restrict__ zh_t) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t i = __brev(r) >> (32 - h);
    const uint32_t x = mmul(gen, mpow(w_m, i));
    const uint32_t zh = zh_t[i & ((1u << b) - 1u)];
    first[r] = mmul(zh, minv(msub(x, MONTY_ONE)));
    last[r] = mmul(zh, minv(msub(x, w_n_inv)));
}

// every chip that runs the interpreter, in ONE launch: descriptor array + block prefix table (the twiddle table belongs
// to the context, not to the key, so it is a launch parameter)
struct QuotMulti {
    const QuotArgs* args;
    const uint32_t* first;  // n + 1
    uint32_t n;
    const uint32_t* tw_fwd;
    unsigned tw_log;
};

__global__ __launch_bounds__(QBS) void k_quotient(QuotMulti m) {
    extern __shared__ uint32_t slots[];
    uint32_t chip = 0;
    {
        uint32_t lo = 0, hi = m.n - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (m.first[mid] <= blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        chip = lo;
    }
    const QuotArgs& a = m.args[chip];
    const unsigned tid = threadIdx.x;
    const unsigned h = a.lh + a.b;
    const size_t M = (size_t)1 << h;
    const uint32_t r = (blockIdx.x - m.first[chip]) * QBS + tid;
    if (r >= a.n_rows) return;
    const uint32_t i = bitrev32(r, h);
    const uint32_t rn = bitrev32((i + (1u << a.b)) & (uint32_t)(M - 1), h);
    const uint32_t x = mmul(a.gen, root_pow(m.tw_fwd, m.tw_log - h, h, i));
    const uint32_t zh = a.zh[i & ((1u << a.b) - 1u)];
    const uint32_t sel_first = mmul(zh, minv(msub(x, MONTY_ONE)));
    const uint32_t sel_trans = msub(x, a.w_n_inv);
    const uint32_t sel_last = mmul(zh, minv(sel_trans));
    Ext acc = ext_zero();
    auto fetch = [&](uint32_t w) -> uint32_t {
        const uint32_t kind = w >> 28, idx = w & 0x0fffffffu;
        switch (kind) {
            case K_SLOT:
                return slots[idx * QBS + tid];
            case K_VAR: {
                const uint32_t col = idx & 0x07ffffffu;
                return a.lde[(size_t)col * M + ((idx >> 27) ? rn : r)];
            }
            case K_PUB:
                return a.pvs[idx];
            case K_CONST:
                return a.consts[idx];
            case K_PERM: {
                const uint32_t col = idx & 0x07ffffffu;
                return a.perm[(size_t)col * M + ((idx >> 27) ? rn : r)];
            }
            case K_CHAL:
                return a.lchal[idx];
            case K_EXPO:
                return a.expo[idx];
            case K_PREP: {
                const uint32_t col = idx & 0x07ffffffu;
                return a.prep[(size_t)col * M + ((idx >> 27) ? rn : r)];
            }
            default:
                return idx == 0 ? sel_first : (idx == 1 ? sel_last : sel_trans);
        }
    };
    for (uint32_t pc = 0; pc < a.n_instr; pc++) {
        const uint32_t w0 = a.code[3 * pc], wa = a.code[3 * pc + 1], wb = a.code[3 * pc + 2];
        const uint32_t op = w0 & 0xffu, dst = w0 >> 8;
        const uint32_t va = fetch(wa);
        if (op == Q_ASSERT) {
            const uint32_t* ap = a.apow + 4 * (size_t)dst;
            acc.c[0] = madd(acc.c[0], mmul(ap[0], va));
            acc.c[1] = madd(acc.c[1], mmul(ap[1], va));
            acc.c[2] = madd(acc.c[2], mmul(ap[2], va));
            acc.c[3] = madd(acc.c[3], mmul(ap[3], va));
        } else if (op == Q_NEG) {
            slots[dst * QBS + tid] = mneg(va);
        } else {
            const uint32_t vb = fetch(wb);
            slots[dst * QBS + tid] = op == Q_ADD ? madd(va, vb) : (op == Q_SUB ? msub(va, vb) : mmul(va, vb));
        }
    }
    const uint32_t izh = a.inv_zh[i & ((1u << a.b) - 1u)];
#pragma unroll
    for (int k = 0; k < 4; k++) a.q[(size_t)k * M + r] = mmul(acc.c[k], izh);
}

// A short, wide chip (few LDE rows, a long constraint program) runs the interpreter as several CONSTRAINT SLICES side by side: slice s
// evaluates a contiguous range of the constraints into a partial quotient of its own (the quotient is linear in the constraints'
// random combination); this kernel adds the partial columns up.  q, parts: 4 columns of stride M; rows below n_rows.
__global__ __launch_bounds__(256) void k_quot_sum_slices(uint32_t* __restrict__ q, const uint32_t* __restrict__ parts, uint32_t n_slices, size_t M, uint32_t n_rows) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 4 * (size_t)n_rows) return;
    const size_t idx = (t / n_rows) * M + (t % n_rows);
    uint32_t acc = parts[idx];
    for (uint32_t s = 1; s < n_slices; s++) acc = madd(acc, parts[(size_t)s * 4 * M + idx]);
    q[idx] = acc;
}

// alpha^(n_cons-1-k) for every chip at once: block = chip, one lane each (n_cons is a few hundred at most)
struct PowDesc {
    uint32_t* out;
    uint32_t n, pad;
};
__global__ void k_ext_powers_multi(const uint32_t* base, const PowDesc* __restrict__ descs) {
    if (threadIdx.x != 0) return;
    const PowDesc d = descs[blockIdx.x];
    Ext bse = ld_ext(base), cur = ext_one();
    for (uint32_t k = 0; k < d.n; k++) {
        st_ext(d.out + 4 * (size_t)(d.n - 1 - k), cur);
        cur = ext_mul(cur, bse);
    }
}
// out[k] = base^(reverse ? n-1-k : k) * 1, n extension elements (single lane: n is a few hundred)
__global__ void k_ext_powers(const uint32_t* base, uint32_t n, int reverse, uint32_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Ext bse = ld_ext(base), cur = ext_one();
    for (uint32_t k = 0; k < n; k++) {
        st_ext(out + 4 * (size_t)(reverse ? n - 1 - k : k), cur);
        cur = ext_mul(cur, bse);
    }
}

// ---------------------------------------------------------------------------------------------
// LogUp phase (bus interactions; K6).  lchal = gamma, beta^1 .. beta^LOGUP_MAX_FIELDS as base coordinates.
__global__ void k_logup_chal(const uint32_t* gb, uint32_t* lchal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Ext gamma = ld_ext(gb), beta = ld_ext(gb + 4);
    st_ext(lchal, gamma);
    Ext cur = beta;
    for (unsigned i = 1; i <= LOGUP_MAX_FIELDS; i++) {
        st_ext(lchal + 4 * i, cur);
        cur = ext_mul(cur, beta);
    }
}

// Interaction operands are expressions of the current row: each interaction has a small slot program (the
// lowering of air_compile.hpp with the operands as roots: fields first, count last).  Descriptor table on the
// device, LU_STRIDE words per interaction:
//   [0] first instruction  [1] n_instr  [2] bus+1 (Montgomery)  [3] sign  [4] n_fields  [5] first constant
//   [6] permutation column group (consecutive interactions of one group share a phi column: their terms add)
constexpr unsigned LU_STRIDE = 7;
constexpr unsigned LU_BS = 256;
struct LogupArgs {
    const uint32_t* trace;  // column-major, stride N, Montgomery
    const uint32_t* prep;   // preprocessed trace, same layout (or null)
    const uint32_t* pvs;
    const uint32_t* tab;     // descriptors
    const uint32_t* code;    // 3 words per instruction, all interactions back to back
    const uint32_t* consts;  // Montgomery
    const uint32_t* lchal;
    uint32_t* den;   // n_int x N extension elements, [j][r]
    uint32_t* num;   // n_int x N signed multiplicities
    uint32_t* perm;  // 4 (n_groups + 1) columns x N
    uint32_t* sums;  // N extension elements: row sums, then (after the scan) the running sum
    uint32_t* expo;  // 4 words
    size_t N;
    uint32_t n_int, n_groups;
};
// All chips with bus interactions of a proof go through each LogUp kernel in ONE launch: the descriptors of the chips sit
// in a device array, the grid is the concatenation of the chips' block ranges and a block finds its chip by a binary
// search over the (scalar) prefix table.  42 chips used to cost 42 launches per kernel, each far too small to fill the GPU.
struct LogupMulti {
    const LogupArgs* args;       // n chips
    const uint32_t* rows_first;  // n + 1: first 256-row block of each chip in the flattened row grid
    const uint32_t* den_first;   // n + 1: first block in the flattened denominator grid (row blocks x n_int)
    uint32_t n;
};
__device__ __forceinline__ uint32_t chip_of_block(const uint32_t* first, uint32_t n, uint32_t b) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (first[mid] <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
// denominator gamma + bus + 1 + sum_i beta^(i+1) f_i and numerator +-count of one row of one interaction
__global__ __launch_bounds__(LU_BS) void k_logup_denoms(LogupMulti m) {
    extern __shared__ uint32_t slots[];
    const uint32_t chip = chip_of_block(m.den_first, m.n, blockIdx.x);
    const LogupArgs& a = m.args[chip];
    const uint32_t lb = blockIdx.x - m.den_first[chip];
    const uint32_t rb = (uint32_t)((a.N + LU_BS - 1) / LU_BS);
    const unsigned tid = threadIdx.x;
    const size_t r = (size_t)(lb % rb) * LU_BS + tid;
    if (r >= a.N) return;
    const uint32_t j = lb / rb;
    const uint32_t* t = a.tab + (size_t)j * LU_STRIDE;
    const uint32_t* code = a.code + 3 * (size_t)t[0];
    const uint32_t n_instr = t[1], n_fields = t[4];
    const uint32_t* consts = a.consts + t[5];
    Ext den = ld_ext(a.lchal);
    den.c[0] = madd(den.c[0], t[2]);
    uint32_t num = 0;
    const uint32_t* trace = a.trace;
    const uint32_t* prep = a.prep;
    const uint32_t* pvs = a.pvs;
    const size_t N = a.N;
    auto fetch = [&](uint32_t w) -> uint32_t {
        const uint32_t kind = w >> 28, idx = w & 0x07ffffffu;
        switch (kind) {
            case K_SLOT: return slots[idx * LU_BS + tid];
            case K_VAR: return trace[(size_t)idx * N + r];
            case K_PREP: return prep[(size_t)idx * N + r];
            case K_PUB: return pvs[idx];
            default: return consts[idx];
        }
    };
    for (uint32_t pc = 0; pc < n_instr; pc++) {
        const uint32_t w0 = code[3 * pc], wa = code[3 * pc + 1], wb = code[3 * pc + 2];
        const uint32_t op = w0 & 0xffu, dst = w0 >> 8;
        const uint32_t va = fetch(wa);
        if (op == Q_ASSERT) {
            if (dst < n_fields) den = ext_add(den, ext_mul_base(ld_ext(a.lchal + 4 * (dst + 1)), va));
            else num = t[3] ? mneg(va) : va;
        } else if (op == Q_NEG) {
            slots[dst * LU_BS + tid] = mneg(va);
        } else {
            const uint32_t vb = fetch(wb);
            slots[dst * LU_BS + tid] = op == Q_ADD ? madd(va, vb) : (op == Q_SUB ? msub(va, vb) : mmul(va, vb));
        }
    }
    reinterpret_cast<uint4*>(a.den)[(size_t)j * N + r] = make_uint4(den.c[0], den.c[1], den.c[2], den.c[3]);
    a.num[(size_t)j * N + r] = num;
}
// after the batch inversion den holds phi_j[r]: write the phi columns and the row sums
__global__ __launch_bounds__(256) void k_logup_rows(LogupMulti m) {
    const uint32_t chip = chip_of_block(m.rows_first, m.n, blockIdx.x);
    const LogupArgs& a = m.args[chip];
    const size_t r = (size_t)(blockIdx.x - m.rows_first[chip]) * 256 + threadIdx.x;
    if (r >= a.N) return;
    Ext sum = ext_zero(), grp = ext_zero();
    for (uint32_t j = 0; j < a.n_int; j++) {
        const uint4 v = reinterpret_cast<const uint4*>(a.den)[(size_t)j * a.N + r];
        grp = ext_add(grp, Ext{{v.x, v.y, v.z, v.w}});
        const uint32_t g = a.tab[(size_t)j * LU_STRIDE + 6];
        if (j + 1 == a.n_int || a.tab[(size_t)(j + 1) * LU_STRIDE + 6] != g) {  // last interaction of its group
#pragma unroll
            for (int q = 0; q < 4; q++) a.perm[(size_t)(4 * g + q) * a.N + r] = grp.c[q];
            sum = ext_add(sum, grp);
            grp = ext_zero();
        }
    }
    reinterpret_cast<uint4*>(a.sums)[r] = make_uint4(sum.c[0], sum.c[1], sum.c[2], sum.c[3]);
}
// running-sum columns and the exposed total
__global__ __launch_bounds__(256) void k_logup_sums(LogupMulti m) {
    const uint32_t chip = chip_of_block(m.rows_first, m.n, blockIdx.x);
    const LogupArgs& a = m.args[chip];
    const size_t r = (size_t)(blockIdx.x - m.rows_first[chip]) * 256 + threadIdx.x;
    if (r >= a.N) return;
    const uint4 v = reinterpret_cast<const uint4*>(a.sums)[r];
    const uint32_t c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        a.perm[(size_t)(4 * a.n_groups + q) * a.N + r] = c[q];
        if (r == a.N - 1) a.expo[q] = c[q];
    }
}

// ---------------------------------------------------------------------------------------------
// openings: p(z) = (z^N - s^N)/(N s^N) * sum_i p(s w^i) * x_i/(z - x_i), x_i = s w^i (natural order)
// One opening group = the columns of all matrices that share barycentric weights (a height, a coset shift, 1 or 2 points).
// All groups of a proof are processed by ONE launch per kernel over this descriptor table (static per key).
struct OpenGroupDev {
    uint32_t lh, n_pts;
    uint32_t shift, w_n, s_pow_n, scale_base;  // Montgomery: coset shift s, generator of H, s^N, 1/(N s^N)
    uint32_t col_first, n_cols;                // slice of the column pointer / destination tables
    uint32_t n_tiles;                          // 512-row tiles
    uint32_t n_chunks;                         // column chunks: a wave reduces ONE chunk of one tile (a group of few rows and thousands of columns -- the
                                               // limb chips at 2^10 rows: 2 tiles x 2969 columns -- was two waves walking 2969 columns: 6.4 ms of a 54 ms proof)
    uint32_t tile_first;                       // first (tile, chunk) unit inside the flattened grid of ITS point-count class
    uint32_t wblk_first;                       // first 256-row block inside the flattened weights grid
    uint32_t fin_first;                        // first (column, point) block inside the flattened finish grid
    uint64_t weights_off;                      // ext offset of weights[pt][i] inside d_weights
    uint64_t partial_off;                      // word offset of partial[tile][col][pt][4] inside d_partial
};
struct OpenMulti {
    const OpenGroupDev* g;
    uint32_t n;
    const uint32_t* idx2;  // indices of the groups with two points (trace-like matrices), tile_first ascending
    uint32_t n2;
    const uint32_t* idx1;  // one point (quotient chunks)
    uint32_t n1;
    const uint32_t* zeta;
    uint32_t* pts;         // per group: pts[2] then scale[2] (ext each)
    uint32_t* weights;
    uint32_t* partial;
    const uint32_t* const* col_ptrs;
    const uint32_t* dst;
    uint32_t* opened;
    const uint32_t* tw_fwd;
    unsigned tw_log;
};
// points z, z*w and the scales (z_pt^N - s^N)/(N s^N) of every group: block = group, one lane
__global__ void k_open_points(OpenMulti m) {
    if (threadIdx.x != 0) return;
    const OpenGroupDev a = m.g[blockIdx.x];
    uint32_t* pts = m.pts + 16 * (size_t)blockIdx.x;
    Ext z = ld_ext(m.zeta);
    for (unsigned p = 0; p < a.n_pts; p++) {
        Ext zp = p == 0 ? z : ext_mul_base(z, a.w_n);
        st_ext(pts + 4 * p, zp);
        Ext t = zp;
        for (unsigned k = 0; k < a.lh; k++) t = ext_mul(t, t);
        t.c[0] = msub(t.c[0], a.s_pow_n);
        st_ext(pts + 8 + 4 * p, ext_mul_base(t, a.scale_base));
    }
}

// pts[0] = z, pts[1] = z * w (the two opening points of a height, for the FRI batching denominators)
__global__ void k_two_points(const uint32_t* zeta, uint32_t w_n, uint32_t* pts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Ext z = ld_ext(zeta);
    st_ext(pts, z);
    st_ext(pts + 4, ext_mul_base(z, w_n));
}

// weights[pt][i] = x_i / (z_pt - x_i)
__global__ __launch_bounds__(256) void k_bary_weights(OpenMulti m) {
    uint32_t gi = 0;
    {
        uint32_t lo = 0, hi = m.n - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (m.g[mid].wblk_first <= blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        gi = lo;
    }
    const OpenGroupDev& a = m.g[gi];
    const unsigned lh = a.lh;
    const size_t N = (size_t)1 << lh;
    const size_t i = (size_t)(blockIdx.x - a.wblk_first) * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t* pts = m.pts + 16 * (size_t)gi;
    uint32_t* weights = m.weights + 4 * a.weights_off;
    uint32_t x = mmul(a.shift, lh == 0 ? MONTY_ONE : root_pow(m.tw_fwd, m.tw_log - lh, lh, (uint32_t)i));
    for (unsigned p = 0; p < a.n_pts; p++) {
        Ext d = ld_ext(pts + 4 * p);
        d.c[0] = msub(d.c[0], x);
        Ext w = ext_mul_base(ext_inv(d), x);
        reinterpret_cast<uint4*>(weights)[(size_t)p * N + i] = make_uint4(w.c[0], w.c[1], w.c[2], w.c[3]);
    }
}

// partial[tile][col][pt] = sum over the tile's rows of col[r] * w[pt][r].  One tile = one WAVE =
// 64*RK rows (lane = row, RK rows per lane, their weights held in registers); per column the lane
// sums are combined with DPP/shuffle adds and lane 0 stores the wave's partial.  No LDS, no
// barriers, so the loads of the next column are in flight while the current one is multiplied.
// The flattened tile grid covers every group with NPTS points; a wave finds its group by its tile index.
template <int NPTS, int RK>
__global__ __launch_bounds__(256) void k_col_reduce(OpenMulti m) {
    const uint32_t* idx = NPTS == 2 ? m.idx2 : m.idx1;
    const uint32_t n_idx = NPTS == 2 ? m.n2 : m.n1;
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t gtile = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t gi;
    {
        uint32_t lo = 0, hi = n_idx - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (m.g[idx[mid]].tile_first <= gtile) lo = mid;
            else hi = mid - 1;
        }
        gi = idx[lo];
    }
    const OpenGroupDev& a = m.g[gi];
    const size_t unit = gtile - a.tile_first;
    if (unit >= (size_t)a.n_tiles * a.n_chunks) return;  // beyond the last group's units (grid rounded up to 4 waves per block)
    const size_t tile = unit % a.n_tiles;
    const size_t N = (size_t)1 << a.lh;
    const uint32_t n_cols = a.n_cols;
    const uint32_t per_chunk = (n_cols + a.n_chunks - 1) / a.n_chunks;
    const uint32_t c_begin = (uint32_t)(unit / a.n_tiles) * per_chunk, c_end = c_begin + per_chunk < n_cols ? c_begin + per_chunk : n_cols;
    if (c_begin >= c_end) return;
    const uint32_t* const* __restrict__ col_ptrs = m.col_ptrs + a.col_first;
    const uint32_t* __restrict__ weights = m.weights + 4 * a.weights_off;
    uint32_t* __restrict__ partial = m.partial + a.partial_off;
    const size_t row0 = tile * 64 * RK + lane;
    // Weights and cells are centred to (-p/2, p/2] and multiplied as signed words: four products fit one 64-bit
    // accumulator under the bound of the signed Montgomery step (|t| < 1.21 p^2 keeps the result inside 32 bits), so a
    // lane's RK = 8 rows cost 8 v_mad_i64_i32 + one extra product that carries the first group's residue into the second
    // (d1 * (2^32 mod p) * 2^-32 == d1) + two 2-instruction reductions per coordinate, instead of 8 x (product + modular add).
    static_assert(RK == 8, "two groups of four rows");
    int32_t w[RK][NPTS][4];
#pragma unroll
    for (int k = 0; k < RK; k++) {
        size_t r = row0 + (size_t)k * 64;
#pragma unroll
        for (int p = 0; p < NPTS; p++) {
            uint4 v = r < N ? reinterpret_cast<const uint4*>(weights)[(size_t)p * N + r] : make_uint4(0, 0, 0, 0);
            w[k][p][0] = center_signed(v.x), w[k][p][1] = center_signed(v.y), w[k][p][2] = center_signed(v.z), w[k][p][3] = center_signed(v.w);
        }
    }
    uint32_t x[RK], xn[RK];
#pragma unroll
    for (int k = 0; k < RK; k++) {
        size_t r = row0 + (size_t)k * 64;
        x[k] = r < N ? col_ptrs[c_begin][r] : 0u;
    }
    for (uint32_t c = c_begin; c < c_end; c++) {
        if (c + 1 < c_end) {
            const uint32_t* col = col_ptrs[c + 1];  // columns of every matrix of this height: wave-uniform pointer table
#pragma unroll
            for (int k = 0; k < RK; k++) {
                size_t r = row0 + (size_t)k * 64;
                xn[k] = r < N ? col[r] : 0u;
            }
        }
        int32_t xs[RK];
#pragma unroll
        for (int k = 0; k < RK; k++) xs[k] = center_signed(x[k]);
        uint32_t acc[8];  // [p][q] for two points; the second half is unused (zero) with one point
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (p >= NPTS) {
                    acc[p * 4 + q] = 0;
                    continue;
                }
                int64_t t = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) t += (int64_t)w[k][p][q] * xs[k];
                const int32_t d1 = smred64(t);                       // |d1| < 0.97 p
                t = (int64_t)d1 * (int32_t)MONTY_ONE;                // 0.13 p^2 at most
#pragma unroll
                for (int k = 4; k < 8; k++) t += (int64_t)w[k][p][q] * xs[k];
                acc[p * 4 + q] = canon_signed_wide(smred64(t));      // |.| < 1.03 p
            }
        // Sum over the 64 lanes, all eight values at once: each exchange step halves the number of values a lane
        // carries (v_permlane32_swap / v_permlane16_swap move the halves that change hands, then one select + row
        // rotation), three more steps finish the one value left: 10 modular additions instead of 48.
        uint32_t u4[4], u2[2], u1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const auto r = __builtin_amdgcn_permlane32_swap(acc[j], acc[j + 4], false, false);
            u4[j] = madd(r[0], r[1]);  // lanes 0-31: value j, lanes 32-63: value j + 4
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const auto r = __builtin_amdgcn_permlane16_swap(u4[j], u4[j + 2], false, false);
            u2[j] = madd(r[0], r[1]);  // 16-lane rows 0 / 1: values j, j + 2 (rows 2 / 3: + 4)
        }
        {
            const bool up = (lane & 8u) != 0;
            const uint32_t keep = up ? u2[1] : u2[0], send = up ? u2[0] : u2[1];
            u1 = madd(keep, dpp<ZK_ROR(8)>(send));  // lanes 0-7 of a row: value + 0, lanes 8-15: value + 1
        }
        u1 = madd(u1, dpp<0x141>(u1));        // row_half_mirror: lane i <-> 7 - i inside each group of 8
        u1 = madd(u1, dpp<0xB1>(u1));         // quad_perm [1,0,3,2]
        u1 = madd(u1, dpp<ZK_QROT2>(u1));     // quad_perm [2,3,0,1]
        if ((lane & 7u) == 0) {
            // value index = 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1)
            const unsigned vi = ((lane >> 5) << 2) | (((lane >> 4) & 1u) << 1) | ((lane >> 3) & 1u);
            if (vi < NPTS * 4) partial[(tile * n_cols + c) * (NPTS * 4) + vi] = u1;
        }
#pragma unroll
        for (int k = 0; k < RK; k++) x[k] = xn[k];
    }
}

// opened[dst[c][pt]] = scale[pt] * sum_tiles partial[tile][c][pt]; block = one (column, point) of one group
__global__ __launch_bounds__(64) void k_open_finish(OpenMulti m) {
    uint32_t gi = 0;
    {
        uint32_t lo = 0, hi = m.n - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (m.g[mid].fin_first <= blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        gi = lo;
    }
    const OpenGroupDev& a = m.g[gi];
    const uint32_t lb = blockIdx.x - a.fin_first, n_pts = a.n_pts, n_cols = a.n_cols, n_tiles = a.n_tiles;
    const uint32_t c = lb / n_pts, p = lb % n_pts, lane = threadIdx.x;
    const uint32_t* partial = m.partial + a.partial_off;
    const uint32_t* scale = m.pts + 16 * (size_t)gi + 8;
    const uint32_t* dst = m.dst + 2 * (size_t)a.col_first;
    uint32_t acc[4] = {0, 0, 0, 0};
    for (uint32_t t = lane; t < n_tiles; t += 64) {
        const uint32_t* s = partial + ((size_t)t * n_cols + c) * (n_pts * 4) + p * 4;
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = madd(acc[q], s[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t v = acc[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = madd(v, __shfl_xor(v, off, 64));
        acc[q] = v;
    }
    if (lane == 0) {
        Ext r = ext_mul(Ext{{acc[0], acc[1], acc[2], acc[3]}}, ld_ext(scale + 4 * p));
        st_ext(m.opened + 4 * (size_t)dst[2 * c + p], r);  // dst: ext index of (column, point) inside the opened-value array
    }
}

// ---------------------------------------------------------------------------------------------
// FRI batching: per matrix and point, ro[r] += off_pt * (ry_pt - rrow[r]) / (z_pt - x_r)
// One committed matrix as the FRI-batching kernels see it (static per proving key, commitment order within a height)
struct RoItem {
    const uint32_t* mat;   // column-major LDE matrix, stride 2^h
    uint32_t opened_off;   // ext offset of its opened values: [pt][width]
    uint32_t width, n_pts;
    uint32_t pad;
    uint64_t num_reduced;  // alpha power offset of its first point
};
// out[k] = base^k for k < n, one lane per power (square-and-multiply: any order of exact field products gives the
// same element as the serial chain)
__global__ __launch_bounds__(256) void k_ext_powers_par(const uint32_t* base, uint32_t n, uint32_t* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) st_ext(out + 4 * (size_t)k, ext_pow(ld_ext(base), k));
}
// one wave per matrix (grid = all matrices of the proof): ry_pt = sum_k alpha^k * opened[pt][k] (lane-strided partial
// sums, shuffle reduction), off_pt = alpha^(num_reduced + pt * width); slot[item] = {ry0, ry1, off0, off1}
__global__ __launch_bounds__(64) void k_reduce_prep(const RoItem* items, const uint32_t* alpha, const uint32_t* apow,
                                                    const uint32_t* opened, uint32_t* slots) {
    const RoItem it = items[blockIdx.x];
    const unsigned lane = threadIdx.x;
    const uint32_t* op = opened + 4 * (size_t)it.opened_off;
    Ext ry[2] = {ext_zero(), ext_zero()};
    for (uint32_t k = lane; k < it.width; k += 64) {
        const Ext ak = ld_ext(apow + 4 * (size_t)k);
        for (uint32_t p = 0; p < it.n_pts; p++) ry[p] = ext_add(ry[p], ext_mul(ak, ld_ext(op + 4 * ((size_t)p * it.width + k))));
    }
    for (uint32_t p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t v = ry[p].c[q];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v = madd(v, __shfl_xor(v, o, 64));
            ry[p].c[q] = v;
        }
    if (lane != 0) return;
    const Ext aw = ld_ext(apow + 4 * (size_t)it.width);  // alpha^width
    Ext off = ext_pow(ld_ext(alpha), it.num_reduced);
    uint32_t* sl = slots + 16 * (size_t)blockIdx.x;
    for (uint32_t p = 0; p < 2; p++) {
        st_ext(sl + 4 * p, ry[p]);
        st_ext(sl + 8 + 4 * p, off);
        off = ext_mul(off, aw);
    }
}

struct ReduceArgs {
    const RoItem* items;   // the matrices of this height, commitment order
    const uint32_t* slots; // {ry0, ry1, off0, off1} per item (same indexing as items)
    uint32_t n_items;
    const uint32_t* apow;
    const uint32_t* inv;   // [row][2] ext: 1 / (z_p - x_row)
    uint32_t* ro;          // 2^h ext
    unsigned h;
};
// ro[r] = sum over the matrices m of this height and their points p of  off_{m,p} * (ry_{m,p} - rrow_m[r]) / (z_p - x_r),
// rrow_m[r] = sum_k alpha^k M_m[r][k].  One launch per height: every matrix row is read once, ro is written once.
__global__ __launch_bounds__(256) void k_reduced_openings(ReduceArgs a) {
    const size_t M = (size_t)1 << a.h;
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    Ext inv[2];
    {
        const uint4 i0 = reinterpret_cast<const uint4*>(a.inv)[2 * r], i1 = reinterpret_cast<const uint4*>(a.inv)[2 * r + 1];
        inv[0] = Ext{{i0.x, i0.y, i0.z, i0.w}};
        inv[1] = Ext{{i1.x, i1.y, i1.z, i1.w}};
    }
    Ext out = ext_zero();
    for (uint32_t m = 0; m < a.n_items; m++) {
        const RoItem it = a.items[m];
        // sum_k apow[k] * mat[k][r]: products of residues are summed in 64 bits, four at a time
        // (4 p^2 < 2^64), the group sums are banked in split 64-bit accumulators (LazyAcc) and reduced once per
        // matrix row -- 4 v_mad_u64_u32 + 4 adds per column instead of 4 full modular multiply-adds; 8 column loads are
        // in flight per lane
        LazyAcc lz[4];
        uint32_t acc[4] = {0, 0, 0, 0};
        const uint32_t* col = it.mat + r;
        uint32_t k = 0;
        for (; k + 8 <= it.width; k += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = col[(size_t)(k + u) * M];
#pragma unroll
            for (int g = 0; g < 8; g += 4) {
                uint64_t t[4] = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t* ap = a.apow + 4 * (size_t)(k + g + u);
#pragma unroll
                    for (int q = 0; q < 4; q++) t[q] += (uint64_t)ap[q] * v[g + u];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) lz[q].add_group(t[q]);
            }
        }
        if (it.width >= 8) {
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = lz[q].reduce();
        }
        for (; k < it.width; k++) {
            const uint32_t v = col[(size_t)k * M];
            const uint32_t* ap = a.apow + 4 * (size_t)k;
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = madd(acc[q], mmul(ap[q], v));
        }
        const Ext rrow{{acc[0], acc[1], acc[2], acc[3]}};
        const uint32_t* sl = a.slots + 16 * (size_t)m;
        for (uint32_t p = 0; p < it.n_pts; p++) {
            const Ext u = ext_mul(ext_sub(ld_ext(sl + 4 * p), rrow), inv[p]);
            out = ext_add(out, ext_mul(u, ld_ext(sl + 8 + 4 * p)));
        }
    }
    reinterpret_cast<uint4*>(a.ro)[r] = make_uint4(out.c[0], out.c[1], out.c[2], out.c[3]);
}

// inv[r][p] = 1 / (z_p - x_r) for the two opening points of a height, x_r = g * w^bitrev(r): one extension inversion
// per row (Montgomery's trick on the pair) instead of one per row per matrix per point
__global__ __launch_bounds__(256) void k_ro_denoms(const uint32_t* pts, const uint32_t* tw_fwd, uint32_t gen, unsigned h,
                                                   unsigned tw_shift, uint32_t* inv) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ((size_t)1 << h)) return;
    const uint32_t x = mmul(gen, root_pow(tw_fwd, tw_shift, h, bitrev32((uint32_t)r, h)));
    Ext d0 = ld_ext(pts), d1 = ld_ext(pts + 4);
    d0.c[0] = msub(d0.c[0], x);
    d1.c[0] = msub(d1.c[0], x);
    const Ext ip = ext_inv(ext_mul(d0, d1));
    const Ext i0 = ext_mul(ip, d1), i1 = ext_mul(ip, d0);
    reinterpret_cast<uint4*>(inv)[2 * r] = make_uint4(i0.c[0], i0.c[1], i0.c[2], i0.c[3]);
    reinterpret_cast<uint4*>(inv)[2 * r + 1] = make_uint4(i1.c[0], i1.c[1], i1.c[2], i1.c[3]);
}

// ---------------------------------------------------------------------------------------------
// FRI layer leaves: digest i = sponge(layer[2i] || layer[2i+1]) (8 words = one permutation)
__global__ __launch_bounds__(256) void k_hash_pairs(const uint4* __restrict__ layer, size_t n_leaves,
                                                    uint32_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    uint4 a = layer[2 * i], b = layer[2 * i + 1];
    uint32_t s[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, 0, 0, 0, 0, 0, 0, 0, 0};
    poseidon2_permute_rolled(s);
    uint4* o = reinterpret_cast<uint4*>(out + i * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// diagnosis (zkhip_config.self_check): every tree's digests as the previous proof left them
static std::mutex g_shadow_mu;
static std::map<const zkhip_tree*, std::vector<uint32_t>> g_shadow;
// diagnosis (zkhip_config.self_check): the leaf digests of a FRI layer against the pairs they hash
__global__ __launch_bounds__(256) void k_check_pairs(const uint4* __restrict__ layer, size_t n_leaves, const uint32_t* __restrict__ digests, uint32_t* report) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    uint4 a = layer[2 * i], b = layer[2 * i + 1];
    uint32_t s[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, 0, 0, 0, 0, 0, 0, 0, 0};
    poseidon2_permute_rolled(s);
    bool same = true;
    for (int q = 0; q < 8; q++) same = same && s[q] == digests[i * 8 + q];
    if (!same) atomicAdd(&report[0], 1u), atomicMin(&report[1], (uint32_t)(i & 0xffffffu));
}

struct FriLayerDesc {
    const uint32_t* layer;    // 2^log_len ext
    const uint32_t* digests;  // tree over 2^(log_len-1) leaves
    uint32_t log_len;
    uint32_t out_off;         // word offset inside one query's record
};
// grid (n_queries, n_layers): sibling value + authentication path, canonical
__global__ __launch_bounds__(64) void k_fri_query(const FriLayerDesc* layers, const uint32_t* indices,
                                                  uint32_t* proof_queries, size_t query_pitch) {
    const FriLayerDesc L = layers[blockIdx.y];
    const uint32_t il = indices[blockIdx.x] >> blockIdx.y;
    uint32_t* o = proof_queries + (size_t)blockIdx.x * query_pitch + L.out_off;
    const unsigned t = threadIdx.x;
    if (t < 4) o[t] = from_monty(L.layer[4 * (size_t)(il ^ 1u) + t]);
    const unsigned tree_h = L.log_len - 1;
    const uint32_t leaf = il >> 1;
    size_t layer_off = 0;
    for (unsigned l = 0; l < tree_h; l++) {
        size_t sib = (leaf >> l) ^ 1u;
        if (t < 8) o[4 + 8 * l + t] = from_monty(L.digests[(layer_off + sib) * 8 + t]);
        layer_off += (size_t)1 << (tree_h - l);
    }
}

// Coefficients of the final polynomial (log_final_poly_len > 0): the last FRI layer holds n_last = 2^(b+lfp) evaluations,
// bit-reversed, over the subgroup of that size; thread j computes c_j = 1/n_last * sum_i F[bitrev(i)] * w^(-ij) for
// j < 2^lfp (a naive inverse DFT: n_last <= 4096 values, once per proof).  One block.
__global__ void k_fri_final_poly(const uint32_t* last, unsigned log_last, unsigned n_fin, uint32_t w_inv, uint32_t inv_n, uint32_t* out) {
    const unsigned j = threadIdx.x;
    if (j >= n_fin) return;
    const uint32_t wj = mpow(w_inv, j);
    uint32_t cur = MONTY_ONE;
    Ext acc = ext_zero();
    for (uint32_t i = 0; i < (1u << log_last); i++) {
        acc = ext_add(acc, ext_mul_base(ld_ext(last + 4 * (size_t)bitrev32(i, log_last)), cur));
        cur = mmul(cur, wj);
    }
    st_ext(out + 4 * (size_t)j, ext_mul_base(acc, inv_n));
}

__global__ void k_copy_canon(const uint32_t* src, uint32_t* dst, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = from_monty(src[i]);
}

}  // namespace zk

using namespace zk;

// ---------------------------------------------------------------------------------------------
struct AirPlan {
    unsigned lh = 0, h = 0;
    size_t width = 0, n_pvs = 0, N = 0, M = 0;
    unsigned nq = 1;   // quotient chunks (AirProgram::qd()); the quotient is evaluated on the first MQ = N * nq rows of the LDE
    size_t MQ = 0;
    size_t quot_first = 0;  // rank of chunk 0 among the quotient matrices of the proof
    AirProgram prog;
    std::vector<uint32_t> program_words;
    uint32_t digest[8];  // sponge of the bytecode, canonical
    // device
    uint32_t* d_code = nullptr;
    uint32_t n_instr = 0;
    uint32_t* d_consts = nullptr;
    unsigned n_slots = 1;
    uint32_t* d_zh = nullptr;      // 2^b
    uint32_t* d_inv_zh = nullptr;  // 2^b
    uint32_t* d_pvs = nullptr;     // n_pvs (Montgomery), refreshed per proof
    uint32_t* d_lde = nullptr;     // width columns x M
    // constraint slices of the interpreter form (short, wide chips): code / constants / instruction count per slice, partial quotients
    std::vector<uint32_t*> d_slice_code, d_slice_consts;
    std::vector<uint32_t> slice_n_instr;
    uint32_t* d_q_part = nullptr;  // n_slices x 4 columns x M
    uint32_t* d_q = nullptr;       // 4 columns x M (quotient values, bit-reversed LDE order)
    uint32_t* d_qnat = nullptr;    // nq chunks x 4 columns x N (natural order)
    uint32_t* d_qlde = nullptr;    // nq chunks x 4 columns x M
    uint32_t* d_apow_q = nullptr;  // n_cons ext
    size_t opened_main_off = 0, opened_perm_off = 0, opened_quot_off = 0;  // in ext units inside d_opened
    // LogUp phase (AIRs with bus interactions)
    uint32_t n_int = 0;
    size_t perm_w = 0, lu_index = 0;   // width of the permutation matrix; rank among the AIRs with interactions
    uint32_t* d_lu_tab = nullptr;      // interaction descriptors (LU_STRIDE words each)
    uint32_t* d_lu_code = nullptr;     // operand programs of all interactions
    uint32_t* d_lu_consts = nullptr;
    unsigned lu_slots = 1;             // LDS slots the widest operand program needs
    uint32_t* d_lu_den = nullptr;      // n_int x N ext
    uint32_t* d_lu_num = nullptr;      // n_int x N
    uint32_t* d_lu_sums = nullptr;     // N ext
    uint32_t* d_perm = nullptr;        // perm_w columns x N (natural order)
    uint32_t* d_perm_lde = nullptr;    // perm_w columns x M
    // cached main partition (OpenVM-v1 cached main): the first cw columns of the main trace are committed in a tree of
    // their own at prove time; the other columns join the common main commitment
    size_t cw = 0, opened_cached_off = 0;
    zkhip_tree* t_cached = nullptr;
    // preprocessed trace: uploaded, extended and committed at keygen, resident for the life of the key
    size_t prep_w = 0, opened_prep_off = 0;
    uint32_t* d_prep = nullptr;        // prep_w columns x N (natural order, Montgomery)
    uint32_t* d_prep_lde = nullptr;    // prep_w columns x M
    zkhip_tree* t_prep = nullptr;
    uint32_t prep_commit[8] = {};      // canonical
    hipModule_t jit_mod = nullptr;   // keygen-compiled constraint kernel (null -> interpreter)
    hipFunction_t jit_fn = nullptr;
    unsigned jit_rows_per_block = 256;   // 0: the LDS-tiled form (a fixed grid walks 64-row tiles); 64 + 256 NW: the shared-rows form, NW waves per 64 rows (csrc/quotient_jit.hpp)
    size_t jit_tab_words = 0;            // the parameter table's words; the shared-rows form keeps its two selector tables (MQ words each) behind them
    uint32_t* d_jit_tab = nullptr;   // per-instance leaf parameters of the shape classes
};

struct zkhip_pk {
    zkhip_params params;
    unsigned b = 1, nch = 2, hmax = 0, n_layers = 0, lfp = 0;  // nch = 2^b cosets of the LDE (AIR a uses the first airs[a].nq of them for its quotient)
    size_t n_quot = 0;                                            // quotient matrices of a proof = sum of nq
    std::vector<AirPlan> airs;
    std::vector<uint32_t> preamble;  // canonical words observed before anything else (pvs patched in)
    std::vector<size_t> preamble_pv_off, preamble_prep_off;
    // workspace
    void* d_ws = nullptr;
    size_t ws_bytes = 0;
    DevTranscript* d_tr = nullptr;
    uint32_t* d_preamble = nullptr;
    uint32_t* d_chal = nullptr;     // alpha[4] zeta[4] alpha_f[4] then betas[4*n_layers]
    uint32_t* d_opened = nullptr;   // n_open ext (Montgomery)
    size_t n_open = 0;
    uint32_t* d_pts = nullptr;      // scratch: points / scales / ry / off (ext each)
    uint32_t* d_weights = nullptr;  // 2 x Nmax ext
    uint32_t* d_partial = nullptr;
    uint32_t* d_apow_f = nullptr;   // max width ext
    std::vector<uint32_t*> d_ro;    // per log height (nullptr if none)
    uint32_t* d_ro_inv = nullptr;   // 2 x 2^hmax ext: inverse denominators of the height being reduced
    // openings: the columns of all matrices that share barycentric weights (same height, same coset shift) are reduced
    // by one launch; main-trace columns are patched into the pointer table per proof
    struct OpenGroup {
        unsigned lh, n_pts;
        int chunk;  // -1: trace-like matrices (shift 1, two points), j: quotient chunk j of every AIR of this height
        uint32_t first, n_cols;
    };
    std::vector<OpenGroup> open_groups;
    OpenGroupDev* d_og = nullptr;  // the groups as device descriptors (all groups of a proof: one launch per kernel)
    uint32_t *d_og2_idx = nullptr, *d_og1_idx = nullptr;
    uint32_t og_n = 0, og2_n = 0, og1_n = 0, og2_tiles = 0, og1_tiles = 0, og_wblocks = 0, og_fin_blocks = 0;
    std::vector<const uint32_t*> open_ptrs_host;          // pointer table (host copy)
    struct MainCol {
        uint32_t index, air, col;
    };
    std::vector<MainCol> open_main_cols;                   // table entries that point into the caller's traces
    // coset LDEs: chips of one height (>= 2^12 rows) are extended together; kind 0 = main traces (sources patched per
    // proof), 1 = permutation traces, 2 + j = quotient chunk j.  The source pointers follow the opening pointers in the
    // same table; the destination pointers are static.
    struct LdeGroup {
        unsigned lh;
        int kind;
        uint32_t first, n_cols;  // slice of the pointer tables (sources: after open_total_cols)
    };
    std::vector<LdeGroup> lde_groups;
    size_t lde_total_cols = 0;
    uint32_t** d_lde_dst = nullptr;
    const uint32_t** d_open_ptrs = nullptr;
    const uint32_t** h_open_ptrs_pinned = nullptr;
    // per-proof host->device staging (pinned): [preamble | 4 header words | Montgomery public values of every AIR]; the
    // event marks the last copy that reads it, so a second prove_async waits only if those copies are still pending
    uint32_t* h_stage_pinned = nullptr;
    size_t stage_words = 0;
    // LogUp phase of all chips as one batch: descriptor array (main-trace pointers patched per proof through the staging
    // buffer), block prefix tables, the shared denominator / numerator regions, the segmented-scan table
    LogupArgs* d_lu_args = nullptr;
    std::vector<LogupArgs> lu_args_host;
    uint32_t *d_lu_rows_first = nullptr, *d_lu_den_first = nullptr;
    uint32_t lu_rows_blocks = 0, lu_den_blocks = 0, lu_scan_blocks = 0, lu_max_slots = 1;
    bool lu_scan_multi = false;
    uint32_t *d_lu_den_all = nullptr, *d_lu_num_all = nullptr;
    size_t lu_den_elems = 0;
    ScanSeg* d_lu_scan = nullptr;
    // quotient phase as batches: interpreter descriptors of the chips without a compiled kernel, the alpha-power
    // descriptors of every chip, the bit-reversal copies of every quotient chunk
    std::vector<QuotArgs> quot_args_host;  // one per AIR (the compiled kernels take their fields as parameters)
    QuotArgs* d_quot_args = nullptr;
    uint32_t* d_quot_first = nullptr;
    uint32_t quot_n = 0, quot_blocks = 0, quot_max_slots = 1;
    PowDesc* d_pow_desc = nullptr;
    BitrevSeg *d_br_tiled = nullptr, *d_br_small = nullptr;
    uint32_t br_n_tiled = 0, br_n_small = 0, br_blocks_tiled = 0, br_blocks_small = 0;
    hipEvent_t stage_ev = nullptr;
    uint32_t* d_open_dst = nullptr;
    size_t open_total_cols = 0;
    RoItem* d_ro_items = nullptr;   // every committed matrix, grouped by height (ascending), commitment order inside
    uint32_t* d_ro_slots = nullptr; // 16 words per item
    struct RoGroup {
        unsigned lh, h;
        uint32_t first, n;
    };
    std::vector<RoGroup> ro_groups;
    uint32_t n_ro_items = 0;
    std::vector<uint32_t*> d_flayer;  // n_layers+1
    size_t n_prep = 0;              // AIRs with a preprocessed trace
    size_t n_cached = 0;            // AIRs with a cached main partition
    size_t off_roots_cached = 0, q_cached_words = 0;
    size_t max_w = 0;               // widest committed matrix
    size_t n_lu = 0;                // AIRs with bus interactions
    unsigned h_perm = 0;            // log height of the permutation commitment
    uint32_t* d_lchal = nullptr;    // N_CHAL challenge coordinates, then the raw gamma, beta (8 words)
    uint32_t* d_exposed = nullptr;  // 4 x n_lu (Montgomery)
    zkhip_tree* t_main = nullptr;
    zkhip_tree* t_perm = nullptr;
    zkhip_tree* t_quot = nullptr;
    std::vector<zkhip_tree*> t_fri;
    FriLayerDesc* d_fri_desc = nullptr;
    uint32_t* d_indices = nullptr;
    uint32_t* d_proof = nullptr;
    // proof layout (words)
    size_t proof_words = 0, off_roots = 0, off_opened = 0, off_fri = 0, off_final = 0, off_qpow = 0, off_queries = 0;
    size_t off_root_perm = 0, off_exposed = 0, off_root_quot = 0;
    size_t query_pitch = 0, q_main_words = 0, q_prep_words = 0, q_perm_words = 0, q_quot_words = 0;
};

namespace {

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

uint32_t host_pow(uint32_t a, uint64_t e) { return mpow(a, e); }

int upload(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return ZKHIP_OK;
    // on the context's stream, then wait: ordered before every later kernel of this context (a blocking hipMemcpy runs on
    // the legacy default stream, which a non-blocking stream does not wait for), and `src` may be a temporary
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

}  // namespace

extern "C" {

// Compiles the constraint kernels of `airs` (hipRTC needs no GPU) into `cache_dir`, where key generation looks for them
// (zkhip_config.jit_cache_dir): __graft_entry__.build() does this for the chips of the guest flow, so that the first key generation of a
// process on a fresh machine loads code objects instead of compiling for tens of seconds.  The source depends on the program and the
// blow-up only.  *n_ok = kernels present afterwards; AIRs whose program the code generator refuses are skipped (they run interpreted).
int zkhip_jit_prewarm(const zkhip_air* airs, size_t n_airs, unsigned log_blowup, const char* cache_dir, size_t* n_ok) {
    if (!airs || !cache_dir || !cache_dir[0] || log_blowup < 1 || log_blowup > 4) return ZKHIP_ERR_INVALID;
    size_t ok = 0;
    for (size_t a = 0; a < n_airs; a++) {
        if (!airs[a].program || airs[a].width == 0) return ZKHIP_ERR_INVALID;
        AirProgram p;
        std::string err;
        if (parse_air(airs[a].program, airs[a].program_len, airs[a].width, &p, &err) != 0) return ZKHIP_ERR_INVALID;
        // what key generation would compile (the rule of zkhip_keygen): evaluations of >= 2^jit_min_log_work row-instructions at the
        // AIR's height, and AIRs the interpreter form refuses
        CompiledAir comp;
        const bool interp_ok = compile_air(p, &comp, &err) == 0;
        const double work = (double)((size_t)1 << (airs[a].log_height + log_blowup)) * (double)(comp.code.size() / 3);
        if (interp_ok && work < (double)(1ull << std::min(process_config().jit_min_log_work, 62u)) && process_config().jit != 2) continue;
        std::vector<uint32_t> tab;
        std::vector<char> code;
        std::string msg;
        if (quot_jit_code(p, airs[a].log_height, log_blowup, &tab, &code, &msg, nullptr, cache_dir)) ok++;
    }
    if (n_ok) *n_ok = ok;
    return ZKHIP_OK;
}

int zkhip_keygen(zkhip_ctx* ctx, const zkhip_params* params, const zkhip_air* airs, size_t n_airs, zkhip_pk** out) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !params || !airs || !out || n_airs == 0) return ZKHIP_ERR_INVALID;
    if (params->log_final_poly_len > ZKHIP_MAX_LOG_FINAL_POLY) return set_error(ctx, ZKHIP_ERR_INVALID, "log_final_poly_len out of range");
    for (size_t a = 0; a < n_airs; a++)
        if (airs[a].log_height < params->log_final_poly_len)  // its reduced openings would never join the fold loop
            return set_error(ctx, ZKHIP_ERR_INVALID, "AIR " + std::to_string(a) + ": trace shorter than the final polynomial (2^log_final_poly_len rows)");
    if (params->log_blowup < 1 || params->log_blowup > 4) return set_error(ctx, ZKHIP_ERR_INVALID, "log_blowup out of range");
    if (params->commit_pow_bits > 30 || params->query_pow_bits > 30 || params->num_queries == 0)
        return set_error(ctx, ZKHIP_ERR_INVALID, "bad FRI parameters");
    zkhip_pk* pk = new zkhip_pk();
    pk->params = *params;
    pk->b = params->log_blowup;
    pk->nch = 1u << pk->b;
    pk->airs.resize(n_airs);
    std::string err;
    size_t n_open = 0, max_n = 0, max_w = 4, tw_main = 0, tw_perm = 0;
    for (size_t a = 0; a < n_airs; a++) {
        AirPlan& A = pk->airs[a];
        if (airs[a].log_height + pk->b > 27 || airs[a].width == 0) {
            delete pk;
            return set_error(ctx, ZKHIP_ERR_INVALID, "AIR height/width out of range");
        }
        A.program_words.assign(airs[a].program, airs[a].program + airs[a].program_len);
        for (uint32_t w : A.program_words)
            if (w >= P) {
                delete pk;
                return set_error(ctx, ZKHIP_ERR_INVALID, "AIR program word not a field element");
            }
        if (parse_air(A.program_words.data(), A.program_words.size(), airs[a].width, &A.prog, &err) != 0 ||
            A.prog.n_pvs != airs[a].n_pvs) {
            delete pk;
            return set_error(ctx, ZKHIP_ERR_INVALID, "AIR " + std::to_string(a) + ": " + (err.empty() ? "n_pvs mismatch" : err));
        }
        if (A.prog.log_qd() > pk->b) {
            delete pk;
            return set_error(ctx, ZKHIP_ERR_CONSTRAINT, "constraint degree exceeds 2^log_blowup + 1");
        }
        A.lh = airs[a].log_height;
        A.h = A.lh + pk->b;
        A.width = airs[a].width;
        A.n_pvs = airs[a].n_pvs;
        A.N = (size_t)1 << A.lh;
        A.M = (size_t)1 << A.h;
        A.nq = A.prog.qd();
        A.MQ = A.N * A.nq;
        A.quot_first = pk->n_quot;
        pk->n_quot += A.nq;
        pk->hmax = std::max(pk->hmax, A.h);
        max_n = std::max(max_n, A.N);
        max_w = std::max(max_w, A.width);
        A.cw = A.prog.cached_width;
        if (A.cw) pk->n_cached++;
        tw_main += A.width - A.cw;
        A.prep_w = A.prog.prep_width;
        if (A.prep_w) {
            if (!airs[a].prep_trace) {
                delete pk;
                return set_error(ctx, ZKHIP_ERR_INVALID, "AIR " + std::to_string(a) + " declares a preprocessed trace but none was given");
            }
            for (size_t i = 0; i < A.prep_w * A.N; i++)
                if (airs[a].prep_trace[i] >= P) {
                    delete pk;
                    return set_error(ctx, ZKHIP_ERR_INVALID, "preprocessed trace value not canonical");
                }
            pk->n_prep++;
            max_w = std::max(max_w, A.prep_w);
        }
        A.n_int = (uint32_t)A.prog.ints.size();
        A.perm_w = A.prog.perm_width();
        if (A.n_int) {
            A.lu_index = pk->n_lu++;
            pk->h_perm = std::max(pk->h_perm, A.h);
            max_w = std::max(max_w, A.perm_w);
            tw_perm += A.perm_w;
        }
        // program digest (binds the proof to the constraint set)
        std::vector<uint32_t> pm(A.program_words.size());
        for (size_t i = 0; i < pm.size(); i++) pm[i] = to_monty(A.program_words[i]);
        uint32_t dg[8];
        p2_hash_slice(pm.data(), pm.size(), dg);
        for (int i = 0; i < 8; i++) A.digest[i] = from_monty(dg[i]);
    }
    pk->lfp = params->log_final_poly_len;
    pk->n_layers = pk->hmax - pk->b - pk->lfp;
    // opened-value layout: main (all airs), permutation (airs with interactions), quotient (all airs)
    for (size_t a = 0; a < n_airs; a++) {
        pk->airs[a].opened_main_off = n_open;  // the COMMON part of the main trace
        n_open += 2 * (pk->airs[a].width - pk->airs[a].cw);
    }
    for (size_t a = 0; a < n_airs; a++) {
        pk->airs[a].opened_cached_off = n_open;
        n_open += 2 * pk->airs[a].cw;
    }
    for (size_t a = 0; a < n_airs; a++) {
        pk->airs[a].opened_prep_off = n_open;
        n_open += 2 * pk->airs[a].prep_w;
    }
    for (size_t a = 0; a < n_airs; a++) {
        pk->airs[a].opened_perm_off = n_open;
        n_open += 2 * pk->airs[a].perm_w;
    }
    for (size_t a = 0; a < n_airs; a++) {
        pk->airs[a].opened_quot_off = n_open;
        n_open += 4 * (size_t)pk->airs[a].nq;
    }
    pk->n_open = n_open;
    // preamble
    {
        auto& pre = pk->preamble;
        pre = {PROTO_TAG, (uint32_t)n_airs, params->log_blowup, params->log_final_poly_len, params->num_queries,
               params->commit_pow_bits, params->query_pow_bits};
        for (size_t a = 0; a < n_airs; a++) {
            const AirPlan& A = pk->airs[a];
            pre.push_back(A.lh);
            pre.push_back((uint32_t)A.width);
            pre.push_back((uint32_t)A.n_pvs);
            for (int i = 0; i < 8; i++) pre.push_back(A.digest[i]);
            pk->preamble_prep_off.push_back(pre.size());
            if (A.prep_w)
                for (int i = 0; i < 8; i++) pre.push_back(0);  // commitment, known once the table is committed below
            pk->preamble_pv_off.push_back(pre.size());
            for (size_t i = 0; i < A.n_pvs; i++) pre.push_back(0);
        }
    }
    // proof layout
    {
        size_t w = 4;
        pk->off_roots = w;
        w += 8;
        pk->off_roots_cached = w;
        w += 8 * pk->n_cached;
        if (pk->n_lu) {
            pk->off_root_perm = w;
            pk->off_exposed = w + 8;
            w += 8 + 4 * pk->n_lu;
        }
        pk->off_root_quot = w;
        w += 8;
        pk->off_opened = w;
        w += 4 * n_open;
        pk->off_fri = w;
        w += 9 * (size_t)pk->n_layers;
        pk->off_final = w;
        w += (size_t)4 << pk->lfp;
        pk->off_qpow = w;
        w += 1;
        pk->off_queries = w;
        pk->q_main_words = tw_main + 8 * (size_t)pk->hmax;
        for (size_t a = 0; a < n_airs; a++) {
            if (pk->airs[a].cw) pk->q_cached_words += pk->airs[a].cw + 8 * (size_t)pk->airs[a].h;
            if (pk->airs[a].prep_w) pk->q_prep_words += pk->airs[a].prep_w + 8 * (size_t)pk->airs[a].h;
        }
        pk->q_perm_words = pk->n_lu ? tw_perm + 8 * (size_t)pk->h_perm : 0;
        pk->q_quot_words = 4 * pk->n_quot + 8 * (size_t)pk->hmax;
        size_t pitch = pk->q_main_words + pk->q_cached_words + pk->q_prep_words + pk->q_perm_words + pk->q_quot_words;
        for (unsigned l = 0; l < pk->n_layers; l++) pitch += 4 + 8 * (size_t)(pk->hmax - l - 1);
        pk->query_pitch = pitch;
        w += pitch * params->num_queries;
        pk->proof_words = w;
    }
    {
        std::vector<AirProgram> progs(n_airs);
        std::vector<unsigned> lhs(n_airs);
        for (size_t a = 0; a < n_airs; a++) progs[a] = pk->airs[a].prog, lhs[a] = pk->airs[a].lh;
        if (!logup_bus_counts_bounded(progs.data(), lhs.data(), n_airs)) {
            delete pk;
            return set_error(ctx, ZKHIP_ERR_INVALID, "a bus carries >= p interaction rows: the LogUp argument is unsound at these trace heights");
        }
    }
    // ---- workspace plan ----
    Bump bp;
    const unsigned nch = pk->nch;
    struct AirOff {
        size_t code, consts, zh, inv_zh, pvs, lde, q, qnat, qlde, apow_q, q_part;
        std::vector<size_t> slice_code, slice_consts;
        size_t lu_tab, lu_code, lu_consts, lu_den, lu_totals, lu_sums, perm, perm_lde, prep, prep_lde, dig_prep, dig_cached;
    };
    std::vector<AirOff> ao(n_airs);
    std::vector<CompiledAir> comp(n_airs);
    std::vector<std::vector<CompiledAir>> slices(n_airs);
    struct LuHost {
        std::vector<uint32_t> tab, code, consts;
    };
    std::vector<LuHost> lu(n_airs);
    size_t lu_den_elems = 0, lu_tot_elems = 0;
    for (size_t a = 0; a < n_airs; a++) {
        AirPlan& A = pk->airs[a];
        // The interpreter lowering can refuse a valid AIR (more live intermediates than its LDS slots); the compiled
        // kernel keeps intermediates in registers and has no such limit, so the refusal is fatal only if the JIT is
        // unavailable too.
        const bool interp_ok = compile_air(A.prog, &comp[a], &err) == 0;
        const std::string interp_err = err;
        if (!interp_ok) comp[a] = CompiledAir();
        A.n_instr = (uint32_t)(comp[a].code.size() / 3);
        A.n_slots = comp[a].n_slots;
        // keygen-time compiled constraint kernel when the evaluation is large enough to repay the
        // ~2-4 s hipRTC compile (2^26 row-instructions ~ a few ms of interpreter time per proof);
        // ZKHIP_FORCE_JIT=1 / ZKHIP_NO_JIT=1 override
        // ZKHIP_JIT_MIN_LOG_WORK=k moves the threshold to 2^k row-instructions (a fixed app that is proven many times
        // can afford to compile every chip: ~1.5 s each)
        const unsigned jit_log = ctx->cfg.jit_min_log_work;
        const bool big = (double)((size_t)1 << (airs[a].log_height + pk->b)) * (double)A.n_instr >= (double)(1ull << std::min(jit_log, 62u));
        std::string jit_msg = "not attempted";
        if ((big || !interp_ok || ctx->cfg.jit == 2) && ctx->cfg.jit != 0) {
            std::string msg;
            std::vector<uint32_t> tab;
            const auto jit_t0 = std::chrono::steady_clock::now();
            struct JitTimer {   // ZKHIP_KEYGEN_TIMING=1: what each chip's compiled constraint kernel cost (hipRTC or the on-disk cache)
                std::chrono::steady_clock::time_point t0;
                size_t air, n_nodes;
                bool forced;
                ~JitTimer() {
                    if (getenv("ZKHIP_KEYGEN_TIMING"))
                        fprintf(stderr, "[zkhip keygen] AIR %zu: constraint kernel %.2f s (%zu nodes%s)\n", air,
                                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), n_nodes, forced ? ", interpreter form refused" : "");
                }
            } jit_timer{jit_t0, a, (size_t)A.prog.n_nodes, !interp_ok};
            if (quot_jit_build(A.prog, airs[a].log_height, pk->b, &A.jit_mod, &A.jit_fn, &tab, &msg, &A.jit_rows_per_block, ctx->cfg.jit_cache_dir) &&
                hipMalloc(&A.d_jit_tab, (tab.size() + ((A.jit_rows_per_block & 255u) == 64 ? 2 * A.MQ : 0)) * 4) == hipSuccess &&
                hipMemcpyAsync(A.d_jit_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                hipStreamSynchronize(ctx->stream) == hipSuccess) {
                A.jit_tab_words = tab.size();
            } else {
                if (A.jit_mod) (void)hipModuleUnload(A.jit_mod);
                A.jit_mod = nullptr;
                A.jit_fn = nullptr;
                ctx->last_error = "constraint JIT unavailable, using the interpreter kernel: " + msg;
                jit_msg = msg;
            }
        }
        if (!interp_ok && !A.jit_fn) {
            zkhip_pk_destroy(ctx, pk);  // also unloads the modules of the AIRs already compiled
            return set_error(ctx, ZKHIP_ERR_INVALID, "AIR " + std::to_string(a) + ": " + interp_err + " (interpreter form; no compiled form available either: " + jit_msg + ")");
        }
        // Constraint slices: with fewer than ~2 workgroups per CU the interpreter's time is one lane's walk through the whole program
        // (a dependent load per instruction); cutting the constraints into S contiguous ranges gives S times the lanes, each with
        // 1 / S of the walk.  Cut points: the ASSERTs of the whole program nearest to the multiples of n_instr / S.
        // ZKHIP_NO_QUOT_SLICES=1 switches it off (measurements).
        if (!A.jit_fn && interp_ok && A.n_instr >= 2048 && (A.MQ + QBS - 1) / QBS <= 256 && ctx->cfg.quot_slices) {
            const uint32_t want = std::min<uint32_t>(16u, std::min<uint32_t>(A.n_instr / 1024u, (uint32_t)(2048u / ((A.MQ + QBS - 1) / QBS))));
            std::vector<uint32_t> cut{0};   // constraint indices where slices start
            if (want >= 2) {
                const std::vector<uint32_t>& code = comp[a].code;
                uint32_t next = 1, seen_max = 0;
                for (uint32_t pc = 0; pc < A.n_instr && next < want; pc++) {
                    if ((code[3 * pc] & 0xffu) != Q_ASSERT) continue;
                    seen_max = std::max(seen_max, (code[3 * pc] >> 8) + 1u);
                    if (pc >= (uint64_t)A.n_instr * next / want && seen_max > cut.back() && seen_max < A.prog.n_cons) cut.push_back(seen_max), next++;
                }
            }
            if (cut.size() >= 2) {
                cut.push_back(A.prog.n_cons);
                bool ok = true;
                for (size_t sl = 0; sl + 1 < cut.size() && ok; sl++) {
                    std::vector<uint32_t> roots(A.prog.cons + cut[sl], A.prog.cons + cut[sl + 1]);
                    CompiledAir part;
                    std::string e2;
                    ok = compile_air(A.prog, &part, &e2, &roots) == 0;
                    for (size_t pc = 0; ok && pc < part.code.size() / 3; pc++)   // an ASSERT carries its index among the roots: back to the constraint's number
                        if ((part.code[3 * pc] & 0xffu) == Q_ASSERT) part.code[3 * pc] = Q_ASSERT | (((part.code[3 * pc] >> 8) + cut[sl]) << 8);
                    if (ok) slices[a].push_back(std::move(part));
                }
                if (!ok) slices[a].clear();
            }
            for (const CompiledAir& part : slices[a]) {
                A.n_slots = std::max(A.n_slots, part.n_slots);
                A.slice_n_instr.push_back((uint32_t)(part.code.size() / 3));
                ao[a].slice_code.push_back(bp.take(part.code.size() * 4 + 16));
                ao[a].slice_consts.push_back(bp.take(part.consts.size() * 4));
            }
            if (!slices[a].empty()) ao[a].q_part = bp.take(slices[a].size() * A.M * 4 * 4);
        }
        ao[a].code = bp.take(comp[a].code.size() * 4 + 16);
        ao[a].consts = bp.take(comp[a].consts.size() * 4);
        ao[a].zh = bp.take(nch * 4);
        ao[a].inv_zh = bp.take(nch * 4);
        ao[a].pvs = bp.take(A.n_pvs * 4 + 16);
        ao[a].lde = bp.take(A.M * A.width * 4);
        ao[a].q = bp.take(A.M * 4 * 4);
        ao[a].qnat = bp.take(A.MQ * 4 * 4);
        ao[a].qlde = bp.take((size_t)A.nq * A.M * 4 * 4);
        ao[a].apow_q = bp.take((size_t)(A.prog.n_cons + 1) * 16);
        if (A.cw) ao[a].dig_cached = bp.take(merkle_digest_count(A.h) * 32);
        if (A.prep_w) {
            ao[a].prep = bp.take(A.prep_w * A.N * 4);
            ao[a].prep_lde = bp.take(A.prep_w * A.M * 4);
            ao[a].dig_prep = bp.take(merkle_digest_count(A.h) * 32);
        }
        if (A.n_int) {
            // one operand program per interaction: roots = fields, then count
            for (uint32_t j = 0; j < A.n_int; j++) {
                const Interaction& it = A.prog.ints[j];
                std::vector<uint32_t> roots(it.fields, it.fields + it.n_fields);
                roots.push_back(it.count);
                CompiledAir cj;
                if (compile_air(A.prog, &cj, &err, &roots) != 0) {
                    zkhip_pk_destroy(ctx, pk);
                    return set_error(ctx, ZKHIP_ERR_INVALID, "AIR " + std::to_string(a) + " interaction " + std::to_string(j) + ": " + err);
                }
                const uint32_t desc[LU_STRIDE] = {(uint32_t)(lu[a].code.size() / 3), (uint32_t)(cj.code.size() / 3), to_monty(it.bus + 1),
                                                  it.sign, it.n_fields, (uint32_t)lu[a].consts.size(), it.group};
                lu[a].tab.insert(lu[a].tab.end(), desc, desc + LU_STRIDE);
                lu[a].code.insert(lu[a].code.end(), cj.code.begin(), cj.code.end());
                lu[a].consts.insert(lu[a].consts.end(), cj.consts.begin(), cj.consts.end());
                A.lu_slots = std::max(A.lu_slots, cj.n_slots);
            }
            ao[a].lu_code = bp.take(lu[a].code.size() * 4 + 16);
            ao[a].lu_consts = bp.take(lu[a].consts.size() * 4 + 16);
            ao[a].lu_tab = bp.take((size_t)A.n_int * LU_STRIDE * 4);
            ao[a].lu_den = lu_den_elems;  // element offset into the shared regions (one batch inversion for all chips)
            lu_den_elems += (size_t)A.n_int * A.N;
            ao[a].lu_sums = bp.take(A.N * 16);
            ao[a].lu_totals = lu_tot_elems;
            lu_tot_elems += scan_blocks_of(A.N);
            ao[a].perm = bp.take(A.perm_w * A.N * 4);
            ao[a].perm_lde = bp.take(A.perm_w * A.M * 4);
        }
    }
    size_t n_quot_entries = n_airs + 1;
    for (size_t a = 0; a < n_airs; a++) n_quot_entries += slices[a].size();
    const size_t o_quot_args = bp.take(n_quot_entries * sizeof(QuotArgs)), o_quot_first = bp.take((n_quot_entries + 1) * 4);
    const size_t o_pow_desc = bp.take((n_airs + 1) * sizeof(PowDesc));
    const size_t o_br_tiled = bp.take((pk->n_quot + 1) * sizeof(BitrevSeg)), o_br_small = bp.take((pk->n_quot + 1) * sizeof(BitrevSeg));
    const size_t o_lu_den_all = bp.take(lu_den_elems * 16 + 16), o_lu_num_all = bp.take(lu_den_elems * 4 + 16);
    const size_t o_lu_totals = bp.take(lu_tot_elems * 16 + 16), o_lu_args = bp.take((pk->n_lu + 1) * sizeof(LogupArgs));
    const size_t o_lu_rows_first = bp.take((pk->n_lu + 2) * 4), o_lu_den_first = bp.take((pk->n_lu + 2) * 4);
    const size_t o_lu_scan = bp.take((pk->n_lu + 1) * sizeof(ScanSeg));
    size_t o_lchal = bp.take((N_CHAL + 8) * 4);
    size_t o_exposed = bp.take((4 * pk->n_lu + 4) * 4);
    size_t o_dig_perm = pk->n_lu ? bp.take(merkle_digest_count(pk->h_perm) * 32) : 0;
    size_t o_tr = bp.take(sizeof(DevTranscript));
    size_t o_pre = bp.take(pk->preamble.size() * 4);
    size_t o_chal = bp.take((3 + pk->n_layers + ((size_t)1 << pk->lfp)) * 16);  // + the final polynomial's coefficients
    size_t o_opened = bp.take(n_open * 16);
    size_t o_pts = 0, o_weights = 0;  // sized once the opening groups are known (below)
    // opening groups (sizes only here; the pointer tables are filled once the workspace is laid out)
    size_t partial_words = 16;
    {
        std::vector<unsigned> lhs;
        for (size_t a = 0; a < n_airs; a++) lhs.push_back(pk->airs[a].lh);
        std::sort(lhs.begin(), lhs.end());
        lhs.erase(std::unique(lhs.begin(), lhs.end()), lhs.end());
        uint32_t first = 0;
        for (unsigned lh : lhs)
            for (int chunk = -1; chunk < (int)nch; chunk++) {
                uint32_t nc = 0;
                for (size_t a = 0; a < n_airs; a++) {
                    const AirPlan& A = pk->airs[a];
                    if (A.lh != lh || chunk >= (int)A.nq) continue;
                    nc += chunk < 0 ? (uint32_t)(A.width + A.prep_w + A.perm_w) : 4u;
                }
                if (!nc) continue;  // no AIR of this height has that many chunks
                const unsigned n_pts = chunk < 0 ? 2 : 1;
                pk->open_groups.push_back({lh, n_pts, chunk, first, nc});
                first += nc;
                const size_t n_tiles = (((size_t)1 << lh) + 511) / 512;
                partial_words = std::max(partial_words, n_tiles * nc * n_pts * 4);
            }
        pk->open_total_cols = first;
        uint32_t lfirst = 0;
        for (unsigned lh : lhs) {
            if (lh < 12) continue;  // smaller transforms are single-tile kernels, extended per matrix
            for (int kind = 0; kind < 2 + (int)nch; kind++) {
                uint32_t nc = 0;
                for (size_t a = 0; a < n_airs; a++) {
                    const AirPlan& A = pk->airs[a];
                    if (A.lh != lh || kind - 2 >= (int)A.nq) continue;
                    nc += kind == 0 ? (uint32_t)A.width : kind == 1 ? (uint32_t)A.perm_w : 4u;
                }
                if (nc) pk->lde_groups.push_back({lh, kind, lfirst, nc});
                lfirst += nc;
            }
        }
        pk->lde_total_cols = lfirst;
    }
    // every group keeps its own weights / partial sums / points: all groups run in one launch per kernel
    size_t og_weights_ext = 0, og_partial_words = 16;
    for (const auto& g : pk->open_groups) {
        og_weights_ext += (size_t)g.n_pts << g.lh;
        og_partial_words += ((((size_t)1 << g.lh) + 511) / 512) * g.n_cols * g.n_pts * 4;
    }
    (void)partial_words;
    o_pts = bp.take((pk->open_groups.size() + 1) * 16 * 4);
    o_weights = bp.take(og_weights_ext * 16 + 16);
    size_t o_partial = bp.take(og_partial_words * 4);
    const size_t o_og = bp.take((pk->open_groups.size() + 1) * sizeof(OpenGroupDev));
    const size_t o_og2 = bp.take((pk->open_groups.size() + 1) * 4), o_og1 = bp.take((pk->open_groups.size() + 1) * 4);
    size_t o_open_ptrs = bp.take((pk->open_total_cols + pk->lde_total_cols) * sizeof(uint32_t*) + 16);
    size_t o_lde_dst = bp.take(pk->lde_total_cols * sizeof(uint32_t*) + 16);
    size_t o_open_dst = bp.take(pk->open_total_cols * 8 + 16);
    size_t o_apow_f = bp.take((max_w + 1) * 16);
    pk->max_w = max_w;
    size_t o_ro_inv = bp.take(((size_t)32) << pk->hmax);
    pk->n_ro_items = (uint32_t)(n_airs + pk->n_cached + pk->n_prep + pk->n_lu + pk->n_quot);
    size_t o_ro_items = bp.take(pk->n_ro_items * sizeof(RoItem));
    size_t o_ro_slots = bp.take((size_t)pk->n_ro_items * 64);
    std::vector<size_t> o_ro(pk->hmax + 1, (size_t)-1);
    for (size_t a = 0; a < n_airs; a++)
        if (o_ro[pk->airs[a].h] == (size_t)-1) o_ro[pk->airs[a].h] = bp.take(((size_t)16) << pk->airs[a].h);
    std::vector<size_t> o_flayer(pk->n_layers + 1);
    for (unsigned l = 1; l <= pk->n_layers; l++) o_flayer[l] = bp.take((size_t)16 << (pk->hmax - l));
    size_t o_dig_main = bp.take(merkle_digest_count(pk->hmax) * 32);
    size_t o_dig_quot = bp.take(merkle_digest_count(pk->hmax) * 32);
    std::vector<size_t> o_dig_fri(pk->n_layers);
    for (unsigned l = 0; l < pk->n_layers; l++) o_dig_fri[l] = bp.take(merkle_digest_count(pk->hmax - l - 1) * 32);
    size_t o_desc = bp.take(sizeof(FriLayerDesc) * (pk->n_layers + 1));
    size_t o_idx = bp.take(params->num_queries * 4);
    size_t o_proof = bp.take(pk->proof_words * 4);
    pk->ws_bytes = bp.off;
    if (hipMalloc(&pk->d_ws, pk->ws_bytes) != hipSuccess) {
        pk->d_ws = nullptr;
        zkhip_pk_destroy(ctx, pk);
        return set_error(ctx, ZKHIP_ERR_NOMEM, "workspace of " + std::to_string(bp.off) + " bytes");
    }
    char* base = (char*)pk->d_ws;
    int rc = ZKHIP_OK;
    std::vector<zkhip_matrix> mm, pmm, qm;
    for (size_t a = 0; a < n_airs && rc == ZKHIP_OK; a++) {
        AirPlan& A = pk->airs[a];
        A.d_code = (uint32_t*)(base + ao[a].code);
        A.d_consts = (uint32_t*)(base + ao[a].consts);
        A.d_zh = (uint32_t*)(base + ao[a].zh);
        A.d_inv_zh = (uint32_t*)(base + ao[a].inv_zh);
        A.d_pvs = (uint32_t*)(base + ao[a].pvs);
        A.d_lde = (uint32_t*)(base + ao[a].lde);
        A.d_q = (uint32_t*)(base + ao[a].q);
        A.d_qnat = (uint32_t*)(base + ao[a].qnat);
        A.d_qlde = (uint32_t*)(base + ao[a].qlde);
        A.d_apow_q = (uint32_t*)(base + ao[a].apow_q);
        rc = upload(ctx, A.d_code, comp[a].code.data(), comp[a].code.size() * 4);
        if (rc == ZKHIP_OK) rc = upload(ctx, A.d_consts, comp[a].consts.data(), comp[a].consts.size() * 4);
        for (size_t sl = 0; sl < slices[a].size() && rc == ZKHIP_OK; sl++) {
            A.d_slice_code.push_back((uint32_t*)(base + ao[a].slice_code[sl])), A.d_slice_consts.push_back((uint32_t*)(base + ao[a].slice_consts[sl]));
            rc = upload(ctx, A.d_slice_code[sl], slices[a][sl].code.data(), slices[a][sl].code.size() * 4);
            if (rc == ZKHIP_OK) rc = upload(ctx, A.d_slice_consts[sl], slices[a][sl].consts.data(), slices[a][sl].consts.size() * 4);
        }
        if (!slices[a].empty()) A.d_q_part = (uint32_t*)(base + ao[a].q_part);
        // vanishing polynomial of H on the LDE coset: x^N = g^N * w_{2^b}^(i mod 2^b)
        std::vector<uint32_t> zh(nch), izh(nch);
        uint32_t gN = host_pow(to_monty(FIELD_GEN_CANON), A.N), wb = two_adic_generator(pk->b);
        for (unsigned j = 0; j < nch; j++) {
            zh[j] = msub(mmul(gN, host_pow(wb, j)), MONTY_ONE);
            izh[j] = minv(zh[j]);
        }
        if (rc == ZKHIP_OK) rc = upload(ctx, A.d_zh, zh.data(), nch * 4);
        if (rc == ZKHIP_OK) rc = upload(ctx, A.d_inv_zh, izh.data(), nch * 4);
        if (rc == ZKHIP_OK && A.jit_fn && (A.jit_rows_per_block & 255u) == 64) {   // the shared-rows constraint kernel reads its selectors from a table
            KernelScope ks(ctx, "gen_selectors");
            hipLaunchKernelGGL(k_gen_selectors, dim3((unsigned)((A.MQ + 255) / 256)), dim3(256), 0, ctx->stream, A.d_jit_tab + A.jit_tab_words, A.d_jit_tab + A.jit_tab_words + A.MQ,
                               (uint32_t)A.MQ, A.h, pk->b, to_monty(FIELD_GEN_CANON), two_adic_generator(A.h), minv(two_adic_generator(A.lh)), A.d_zh);
            if (hipGetLastError() != hipSuccess) rc = set_error(ctx, ZKHIP_ERR_HIP, "k_gen_selectors");
        }
        if (A.prep_w) {
            A.d_prep = (uint32_t*)(base + ao[a].prep);
            A.d_prep_lde = (uint32_t*)(base + ao[a].prep_lde);
            if (rc == ZKHIP_OK) rc = upload(ctx, A.d_prep, airs[a].prep_trace, A.prep_w * A.N * 4);
            if (rc == ZKHIP_OK) rc = convert_repr(ctx, A.d_prep, A.prep_w * A.N, true);
            if (rc == ZKHIP_OK) rc = lde_batch(ctx, A.d_prep, A.N, A.d_prep_lde, A.M, A.lh, pk->b, A.prep_w, to_monty(FIELD_GEN_CANON));
            zkhip_matrix pmx{A.d_prep_lde, A.M, A.h, A.prep_w};
            if (rc == ZKHIP_OK) rc = merkle_plan(ctx, &pmx, 1, (uint32_t*)(base + ao[a].dig_prep), &A.t_prep);
            if (rc == ZKHIP_OK) rc = merkle_build(ctx, A.t_prep, false);
            if (rc == ZKHIP_OK) {
                uint32_t rm[8];
                if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
                    hipMemcpyAsync(rm, zkhip_tree_root_device(A.t_prep), 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipStreamSynchronize(ctx->stream) != hipSuccess)
                    rc = set_error(ctx, ZKHIP_ERR_HIP, "reading the preprocessed commitment failed");
                for (int k = 0; k < 8; k++) A.prep_commit[k] = from_monty(rm[k]);
            }
        }
        if (A.n_int) {
            A.d_lu_tab = (uint32_t*)(base + ao[a].lu_tab);
            A.d_lu_den = (uint32_t*)(base + o_lu_den_all) + 4 * ao[a].lu_den;
            A.d_lu_num = (uint32_t*)(base + o_lu_num_all) + ao[a].lu_den;
            A.d_lu_sums = (uint32_t*)(base + ao[a].lu_sums);
            A.d_perm = (uint32_t*)(base + ao[a].perm);
            A.d_perm_lde = (uint32_t*)(base + ao[a].perm_lde);
            A.d_lu_code = (uint32_t*)(base + ao[a].lu_code);
            A.d_lu_consts = (uint32_t*)(base + ao[a].lu_consts);
            if (rc == ZKHIP_OK) rc = upload(ctx, A.d_lu_tab, lu[a].tab.data(), lu[a].tab.size() * 4);
            if (rc == ZKHIP_OK) rc = upload(ctx, A.d_lu_code, lu[a].code.data(), lu[a].code.size() * 4);
            if (rc == ZKHIP_OK) rc = upload(ctx, A.d_lu_consts, lu[a].consts.data(), lu[a].consts.size() * 4);
            pmm.push_back(zkhip_matrix{A.d_perm_lde, A.M, A.h, A.perm_w});
        }
        mm.push_back(zkhip_matrix{A.d_lde + A.cw * A.M, A.M, A.h, A.width - A.cw});  // common part
        for (unsigned j = 0; j < A.nq; j++) qm.push_back(zkhip_matrix{A.d_qlde + (size_t)j * 4 * A.M, A.M, A.h, 4});
    }
    pk->d_tr = (DevTranscript*)(base + o_tr);
    pk->d_preamble = (uint32_t*)(base + o_pre);
    pk->d_chal = (uint32_t*)(base + o_chal);
    pk->d_lchal = (uint32_t*)(base + o_lchal);
    pk->d_exposed = (uint32_t*)(base + o_exposed);
    // LogUp batch tables (all chips with interactions, in AIR order = lu_index order)
    pk->d_lu_den_all = (uint32_t*)(base + o_lu_den_all);
    pk->d_lu_num_all = (uint32_t*)(base + o_lu_num_all);
    pk->lu_den_elems = lu_den_elems;
    pk->d_lu_args = (LogupArgs*)(base + o_lu_args);
    pk->d_lu_rows_first = (uint32_t*)(base + o_lu_rows_first);
    pk->d_lu_den_first = (uint32_t*)(base + o_lu_den_first);
    pk->d_lu_scan = (ScanSeg*)(base + o_lu_scan);
    pk->d_opened = (uint32_t*)(base + o_opened);
    pk->d_pts = (uint32_t*)(base + o_pts);
    pk->d_weights = (uint32_t*)(base + o_weights);
    pk->d_partial = (uint32_t*)(base + o_partial);
    {
        pk->d_og = (OpenGroupDev*)(base + o_og);
        pk->d_og2_idx = (uint32_t*)(base + o_og2);
        pk->d_og1_idx = (uint32_t*)(base + o_og1);
        std::vector<OpenGroupDev> ogs;
        std::vector<uint32_t> i2, i1;
        uint64_t w_off = 0, p_off = 0;
        const uint32_t gen_m = to_monty(FIELD_GEN_CANON);
        for (const auto& g : pk->open_groups) {
            if (g.n_cols == 0) continue;
            const size_t N = (size_t)1 << g.lh;
            OpenGroupDev d;
            d.lh = g.lh, d.n_pts = g.n_pts;
            d.shift = g.chunk < 0 ? MONTY_ONE : mmul(gen_m, host_pow(two_adic_generator(g.lh + pk->b), bitrev32((uint32_t)g.chunk, pk->b)));
            d.w_n = two_adic_generator(g.lh);
            d.s_pow_n = host_pow(d.shift, N);
            d.scale_base = minv(mmul(to_monty((uint32_t)(N % P)), d.s_pow_n));
            d.col_first = g.first, d.n_cols = g.n_cols;
            d.n_tiles = (uint32_t)((N + 511) / 512);
            // chunks of at least 16 columns, as many as give the group ~1024 waves
            d.n_chunks = (uint32_t)std::max<size_t>(1, std::min<size_t>((g.n_cols + 15) / 16, (1024 + d.n_tiles - 1) / d.n_tiles));
            uint32_t& tiles = g.n_pts == 2 ? pk->og2_tiles : pk->og1_tiles;
            d.tile_first = tiles;
            tiles += d.n_tiles * d.n_chunks;
            d.wblk_first = pk->og_wblocks;
            pk->og_wblocks += (uint32_t)((N + 255) / 256);
            d.fin_first = pk->og_fin_blocks;
            pk->og_fin_blocks += g.n_cols * g.n_pts;
            d.weights_off = w_off, w_off += (uint64_t)g.n_pts * N;
            d.partial_off = p_off, p_off += (uint64_t)d.n_tiles * g.n_cols * g.n_pts * 4;
            (g.n_pts == 2 ? i2 : i1).push_back((uint32_t)ogs.size());
            ogs.push_back(d);
        }
        pk->og_n = (uint32_t)ogs.size(), pk->og2_n = (uint32_t)i2.size(), pk->og1_n = (uint32_t)i1.size();
        if (rc == ZKHIP_OK && !ogs.empty()) rc = upload(ctx, pk->d_og, ogs.data(), ogs.size() * sizeof(OpenGroupDev));
        if (rc == ZKHIP_OK && !i2.empty()) rc = upload(ctx, pk->d_og2_idx, i2.data(), i2.size() * 4);
        if (rc == ZKHIP_OK && !i1.empty()) rc = upload(ctx, pk->d_og1_idx, i1.data(), i1.size() * 4);
    }
    pk->d_apow_f = (uint32_t*)(base + o_apow_f);
    pk->d_ro_inv = (uint32_t*)(base + o_ro_inv);
    pk->d_open_ptrs = (const uint32_t**)(base + o_open_ptrs);
    pk->d_open_dst = (uint32_t*)(base + o_open_dst);
    {
        std::vector<uint32_t> dst(2 * pk->open_total_cols + 2, 0);
        pk->open_ptrs_host.assign(pk->open_total_cols + pk->lde_total_cols + 1, nullptr);
        for (const auto& g : pk->open_groups) {
            uint32_t c = g.first;
            for (size_t a = 0; a < n_airs; a++) {
                AirPlan& A = pk->airs[a];
                if (A.lh != g.lh || g.chunk >= (int)A.nq) continue;
                auto put = [&](const uint32_t* base_ptr, size_t width, size_t opened_off, bool is_main, size_t col0 = 0) {
                    for (size_t k = 0; k < width; k++, c++) {
                        pk->open_ptrs_host[c] = base_ptr ? base_ptr + k * A.N : nullptr;
                        if (is_main) pk->open_main_cols.push_back({c, (uint32_t)a, (uint32_t)(col0 + k)});
                        dst[2 * c] = (uint32_t)(opened_off + k);
                        dst[2 * c + 1] = (uint32_t)(opened_off + width + k);
                    }
                };
                if (g.chunk < 0) {
                    if (A.cw) put(nullptr, A.cw, A.opened_cached_off, true);  // trace columns stay in order: cached, then common
                    put(nullptr, A.width - A.cw, A.opened_main_off, true, A.cw);
                    if (A.prep_w) put(A.d_prep, A.prep_w, A.opened_prep_off, false);
                    if (A.n_int) put(A.d_perm, A.perm_w, A.opened_perm_off, false);
                } else {
                    put(A.d_qnat + (size_t)g.chunk * 4 * A.N, 4, A.opened_quot_off + 4 * (size_t)g.chunk, false);
                }
            }
        }
        if (rc == ZKHIP_OK) rc = upload(ctx, pk->d_open_dst, dst.data(), 2 * pk->open_total_cols * 4);
        // LDE groups: sources behind the opening pointers, destinations in their own static table
        pk->d_lde_dst = (uint32_t**)(base + o_lde_dst);
        std::vector<uint32_t*> ldst(pk->lde_total_cols + 1, nullptr);
        for (const auto& g : pk->lde_groups) {
            uint32_t c = g.first;
            for (size_t a = 0; a < n_airs; a++) {
                AirPlan& A = pk->airs[a];
                if (A.lh != g.lh || g.kind - 2 >= (int)A.nq) continue;
                const size_t w = g.kind == 0 ? A.width : g.kind == 1 ? A.perm_w : 4;
                for (size_t k = 0; k < w; k++, c++) {
                    const uint32_t idx = (uint32_t)(pk->open_total_cols + c);
                    if (g.kind == 0) {
                        pk->open_main_cols.push_back({idx, (uint32_t)a, (uint32_t)k});
                        ldst[c] = A.d_lde + k * A.M;
                    } else if (g.kind == 1) {
                        pk->open_ptrs_host[idx] = A.d_perm + k * A.N;
                        ldst[c] = A.d_perm_lde + k * A.M;
                    } else {
                        const size_t j = (size_t)g.kind - 2;
                        pk->open_ptrs_host[idx] = A.d_qnat + (j * 4 + k) * A.N;
                        ldst[c] = A.d_qlde + (j * 4 + k) * A.M;
                    }
                }
            }
        }
        if (rc == ZKHIP_OK) rc = upload(ctx, pk->d_lde_dst, ldst.data(), pk->lde_total_cols * sizeof(uint32_t*));
        if (rc == ZKHIP_OK && hipHostMalloc((void**)&pk->h_open_ptrs_pinned, (pk->open_total_cols + pk->lde_total_cols + 1) * sizeof(uint32_t*), 0) != hipSuccess)
            rc = set_error(ctx, ZKHIP_ERR_NOMEM, "pinned pointer table");
    }
    pk->d_ro_items = (RoItem*)(base + o_ro_items);
    pk->d_ro_slots = (uint32_t*)(base + o_ro_slots);
    {
        // the matrices in commitment order (main, preprocessed, permutation, quotient chunks) fix every alpha offset;
        // they are then grouped by height so that one launch reduces a whole height
        struct HostItem {
            RoItem it;
            unsigned lh, h;
        };
        std::vector<HostItem> items;
        std::vector<uint64_t> num_reduced(pk->hmax + 1, 0);
        // rounds: 0 common main, 4 cached main partitions (right after the main batch), 1 preprocessed, 2 permutation, 3 quotient
        for (int round : {0, 4, 1, 2, 3})
            for (size_t a = 0; a < n_airs; a++) {
                AirPlan& A = pk->airs[a];
                if (round == 1 && !A.prep_w) continue;
                if (round == 2 && !A.n_int) continue;
                if (round == 4 && !A.cw) continue;
                const unsigned n_mats = round == 3 ? A.nq : 1;
                for (unsigned j = 0; j < n_mats; j++) {
                    HostItem hi;
                    hi.it.width = round == 0   ? (uint32_t)(A.width - A.cw)
                                  : round == 4 ? (uint32_t)A.cw
                                  : round == 1 ? (uint32_t)A.prep_w
                                  : round == 2 ? (uint32_t)A.perm_w
                                               : 4u;
                    hi.it.n_pts = round == 3 ? 1 : 2;
                    hi.it.mat = round == 0   ? A.d_lde + A.cw * A.M
                                : round == 4 ? A.d_lde
                                : round == 1 ? A.d_prep_lde
                                : round == 2 ? A.d_perm_lde
                                             : A.d_qlde + (size_t)j * 4 * A.M;
                    hi.it.opened_off = (uint32_t)(round == 0   ? A.opened_main_off
                                                  : round == 4 ? A.opened_cached_off
                                                  : round == 1 ? A.opened_prep_off
                                                  : round == 2 ? A.opened_perm_off
                                                               : A.opened_quot_off + 4 * (size_t)j);
                    hi.it.pad = 0;
                    hi.it.num_reduced = num_reduced[A.h];
                    hi.lh = A.lh, hi.h = A.h;
                    num_reduced[A.h] += (uint64_t)hi.it.width * hi.it.n_pts;
                    items.push_back(hi);
                }
            }
        std::stable_sort(items.begin(), items.end(), [](const HostItem& x, const HostItem& y) { return x.h < y.h; });
        std::vector<RoItem> flat;
        for (size_t i = 0; i < items.size(); i++) {
            if (pk->ro_groups.empty() || pk->ro_groups.back().h != items[i].h)
                pk->ro_groups.push_back({items[i].lh, items[i].h, (uint32_t)i, 0u});
            pk->ro_groups.back().n++;
            flat.push_back(items[i].it);
        }
        if (rc == ZKHIP_OK) rc = upload(ctx, pk->d_ro_items, flat.data(), flat.size() * sizeof(RoItem));
    }
    pk->d_ro.assign(pk->hmax + 1, nullptr);
    for (unsigned h = 0; h <= pk->hmax; h++)
        if (o_ro[h] != (size_t)-1) pk->d_ro[h] = (uint32_t*)(base + o_ro[h]);
    pk->d_flayer.assign(pk->n_layers + 1, nullptr);
    pk->d_flayer[0] = pk->d_ro[pk->hmax];
    for (unsigned l = 1; l <= pk->n_layers; l++) pk->d_flayer[l] = (uint32_t*)(base + o_flayer[l]);
    pk->d_indices = (uint32_t*)(base + o_idx);
    pk->d_proof = (uint32_t*)(base + o_proof);
    pk->d_fri_desc = (FriLayerDesc*)(base + o_desc);
    if (rc == ZKHIP_OK) rc = merkle_plan(ctx, mm.data(), mm.size(), (uint32_t*)(base + o_dig_main), &pk->t_main);
    for (size_t a = 0; a < n_airs && rc == ZKHIP_OK; a++) {
        AirPlan& A = pk->airs[a];
        if (!A.cw) continue;
        zkhip_matrix cmx{A.d_lde, A.M, A.h, A.cw};
        rc = merkle_plan(ctx, &cmx, 1, (uint32_t*)(base + ao[a].dig_cached), &A.t_cached);
    }
    if (rc == ZKHIP_OK && pk->n_lu) rc = merkle_plan(ctx, pmm.data(), pmm.size(), (uint32_t*)(base + o_dig_perm), &pk->t_perm);
    if (rc == ZKHIP_OK) rc = merkle_plan(ctx, qm.data(), qm.size(), (uint32_t*)(base + o_dig_quot), &pk->t_quot);
    pk->t_fri.assign(pk->n_layers, nullptr);
    std::vector<FriLayerDesc> desc(pk->n_layers);
    for (size_t a = 0; a < n_airs; a++)
        if (pk->airs[a].prep_w)
            for (int i = 0; i < 8; i++) pk->preamble[pk->preamble_prep_off[a] + i] = pk->airs[a].prep_commit[i];
    size_t qoff = pk->q_main_words + pk->q_cached_words + pk->q_prep_words + pk->q_perm_words + pk->q_quot_words;
    for (unsigned l = 0; l < pk->n_layers && rc == ZKHIP_OK; l++) {
        rc = merkle_plan_leaves(ctx, pk->hmax - l - 1, (uint32_t*)(base + o_dig_fri[l]), &pk->t_fri[l]);
        desc[l].layer = pk->d_flayer[l];
        desc[l].digests = (uint32_t*)(base + o_dig_fri[l]);
        desc[l].log_len = pk->hmax - l;
        desc[l].out_off = (uint32_t)qoff;
        qoff += 4 + 8 * (size_t)(pk->hmax - l - 1);
    }
    if (rc == ZKHIP_OK) rc = upload(ctx, pk->d_fri_desc, desc.data(), desc.size() * sizeof(FriLayerDesc));
    if (rc == ZKHIP_OK) rc = ensure_twiddles(ctx, pk->hmax);
    if (rc == ZKHIP_OK) {
        pk->d_quot_args = (QuotArgs*)(base + o_quot_args);
        pk->d_quot_first = (uint32_t*)(base + o_quot_first);
        pk->d_pow_desc = (PowDesc*)(base + o_pow_desc);
        pk->d_br_tiled = (BitrevSeg*)(base + o_br_tiled);
        pk->d_br_small = (BitrevSeg*)(base + o_br_small);
        std::vector<QuotArgs> interp;
        std::vector<uint32_t> first;
        std::vector<PowDesc> pows;
        std::vector<BitrevSeg> tiled, small;
        uint32_t acc = 0;
        const uint32_t gen_m = to_monty(FIELD_GEN_CANON);
        for (size_t a = 0; a < n_airs; a++) {
            AirPlan& A = pk->airs[a];
            QuotArgs qa;
            qa.code = A.d_code;
            qa.n_instr = A.n_instr;
            qa.consts = A.d_consts;
            qa.pvs = A.d_pvs;
            qa.apow = A.d_apow_q;
            qa.lde = A.d_lde;
            qa.perm = A.d_perm_lde;
            qa.lchal = pk->d_lchal;
            qa.expo = pk->d_exposed + 4 * A.lu_index;
            qa.prep = A.d_prep_lde;
            qa.q = A.d_q;
            qa.inv_zh = A.d_inv_zh;
            qa.zh = A.d_zh;
            qa.gen = gen_m;
            qa.w_n_inv = minv(two_adic_generator(A.lh));
            qa.lh = A.lh;
            qa.b = pk->b;
            qa.n_rows = (uint32_t)A.MQ;
            pk->quot_args_host.push_back(qa);
            pows.push_back(PowDesc{A.d_apow_q, A.prog.n_cons, 0});
            if (!A.jit_fn) {
                const size_t n_entries = A.d_slice_code.empty() ? 1 : A.d_slice_code.size();
                for (size_t sl = 0; sl < n_entries; sl++) {
                    QuotArgs e = qa;
                    if (!A.d_slice_code.empty()) e.code = A.d_slice_code[sl], e.consts = A.d_slice_consts[sl], e.n_instr = A.slice_n_instr[sl], e.q = A.d_q_part + sl * 4 * A.M;
                    interp.push_back(e);
                    first.push_back(acc);
                    acc += (uint32_t)((A.MQ + QBS - 1) / QBS);
                }
                pk->quot_max_slots = std::max(pk->quot_max_slots, (uint32_t)A.n_slots);
            }
            for (unsigned j = 0; j < A.nq; j++) {
                // column k of chunk j starts at q + k*M + j*N and goes to qnat + (4j + k)*N
                BitrevSeg sg{A.d_q + (size_t)j * A.N, A.d_qnat + (size_t)j * 4 * A.N, A.M, A.N, A.lh, 4, 0, 0};
                auto& v = A.lh >= 10 ? tiled : small;
                uint32_t& blocks = A.lh >= 10 ? pk->br_blocks_tiled : pk->br_blocks_small;
                sg.first_block = blocks;
                blocks += ntt_bitrev_copy_blocks(A.lh, 4);
                v.push_back(sg);
            }
        }
        first.push_back(acc);
        pk->quot_n = (uint32_t)interp.size(), pk->quot_blocks = acc;
        pk->br_n_tiled = (uint32_t)tiled.size(), pk->br_n_small = (uint32_t)small.size();
        if (!interp.empty()) rc = upload(ctx, pk->d_quot_args, interp.data(), interp.size() * sizeof(QuotArgs));
        if (rc == ZKHIP_OK) rc = upload(ctx, pk->d_quot_first, first.data(), first.size() * 4);
        if (rc == ZKHIP_OK) rc = upload(ctx, pk->d_pow_desc, pows.data(), pows.size() * sizeof(PowDesc));
        if (rc == ZKHIP_OK && !tiled.empty()) rc = upload(ctx, pk->d_br_tiled, tiled.data(), tiled.size() * sizeof(BitrevSeg));
        if (rc == ZKHIP_OK && !small.empty()) rc = upload(ctx, pk->d_br_small, small.data(), small.size() * sizeof(BitrevSeg));
    }
    if (rc == ZKHIP_OK && pk->n_lu) {
        std::vector<uint32_t> rows_first, den_first;
        std::vector<ScanSeg> segs;
        uint32_t rb_acc = 0, den_acc = 0, scan_acc = 0;
        for (size_t a = 0; a < n_airs; a++) {
            AirPlan& A = pk->airs[a];
            if (!A.n_int) continue;
            LogupArgs la;
            la.trace = nullptr;  // the caller's buffer: patched per proof
            la.prep = A.d_prep;
            la.pvs = A.d_pvs;
            la.tab = A.d_lu_tab;
            la.code = A.d_lu_code;
            la.consts = A.d_lu_consts;
            la.lchal = pk->d_lchal;
            la.den = A.d_lu_den;
            la.num = A.d_lu_num;
            la.perm = A.d_perm;
            la.sums = A.d_lu_sums;
            la.expo = pk->d_exposed + 4 * A.lu_index;
            la.N = A.N;
            la.n_int = A.n_int;
            la.n_groups = (uint32_t)A.prog.n_groups();
            pk->lu_args_host.push_back(la);
            const uint32_t rb = (uint32_t)((A.N + 255) / 256), sb = scan_blocks_of(A.N);
            rows_first.push_back(rb_acc), den_first.push_back(den_acc);
            segs.push_back(ScanSeg{A.d_lu_sums, (uint32_t*)(base + o_lu_totals) + 4 * ao[a].lu_totals, (uint64_t)A.N, scan_acc, sb});
            rb_acc += rb, den_acc += rb * A.n_int, scan_acc += sb;
            if (sb > 1) pk->lu_scan_multi = true;
            pk->lu_max_slots = std::max(pk->lu_max_slots, (uint32_t)A.lu_slots);
        }
        rows_first.push_back(rb_acc), den_first.push_back(den_acc);
        pk->lu_rows_blocks = rb_acc, pk->lu_den_blocks = den_acc, pk->lu_scan_blocks = scan_acc;
        rc = upload(ctx, pk->d_lu_rows_first, rows_first.data(), rows_first.size() * 4);
        if (rc == ZKHIP_OK) rc = upload(ctx, pk->d_lu_den_first, den_first.data(), den_first.size() * 4);
        if (rc == ZKHIP_OK) rc = upload(ctx, pk->d_lu_scan, segs.data(), segs.size() * sizeof(ScanSeg));
    }
    if (rc == ZKHIP_OK) {
        size_t n_pv = 0;
        for (const auto& A : pk->airs) n_pv += A.n_pvs;
        // [LogupArgs x n_lu (8-byte aligned: first)] [preamble] [4 header words] [public values]
        pk->stage_words = pk->n_lu * (sizeof(LogupArgs) / 4) + pk->preamble.size() + 4 + n_pv;
        if (hipHostMalloc((void**)&pk->h_stage_pinned, pk->stage_words * 4, 0) != hipSuccess ||
            hipEventCreateWithFlags(&pk->stage_ev, hipEventDisableTiming) != hipSuccess)
            rc = set_error(ctx, ZKHIP_ERR_NOMEM, "pinned staging buffer");
    }
    if (rc != ZKHIP_OK) {
        zkhip_pk_destroy(ctx, pk);
        return rc;
    }
    *out = pk;
    return ZKHIP_OK;
}

void zkhip_pk_destroy(zkhip_ctx* ctx, zkhip_pk* pk) {
    ZK_BIND_DEVICE(ctx);
    if (!pk) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (pk->t_main) zkhip_tree_destroy(ctx, pk->t_main);
    if (pk->h_open_ptrs_pinned) (void)hipHostFree(pk->h_open_ptrs_pinned);
    if (pk->h_stage_pinned) (void)hipHostFree(pk->h_stage_pinned);
    if (pk->stage_ev) (void)hipEventDestroy(pk->stage_ev);
    if (pk->t_perm) zkhip_tree_destroy(ctx, pk->t_perm);
    for (auto& A : pk->airs) {
        if (A.t_prep) zkhip_tree_destroy(ctx, A.t_prep);
        if (A.t_cached) zkhip_tree_destroy(ctx, A.t_cached);
    }
    if (pk->t_quot) zkhip_tree_destroy(ctx, pk->t_quot);
    for (auto* t : pk->t_fri)
        if (t) zkhip_tree_destroy(ctx, t);
    for (auto& A : pk->airs) {
        if (A.jit_mod) (void)hipModuleUnload(A.jit_mod);
        if (A.d_jit_tab) (void)hipFree(A.d_jit_tab);
    }
    if (pk->d_ws) (void)hipFree(pk->d_ws);
    delete pk;
}

size_t zkhip_proof_size(const zkhip_pk* pk) { return pk ? pk->proof_words * 4 : 0; }
size_t zkhip_pk_workspace_bytes(const zkhip_pk* pk) { return pk ? pk->ws_bytes : 0; }

int zkhip_pk_prep_commitment(zkhip_ctx* ctx, const zkhip_pk* pk, size_t air_index, uint32_t out[8]) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !pk || !out || air_index >= pk->airs.size()) return ZKHIP_ERR_INVALID;
    if (!pk->airs[air_index].prep_w) return set_error(ctx, ZKHIP_ERR_INVALID, "AIR has no preprocessed trace");
    memcpy(out, pk->airs[air_index].prep_commit, 32);
    return ZKHIP_OK;
}

int zkhip_prove_async(zkhip_ctx* ctx, const zkhip_pk* pkc, const uint32_t* const* d_traces,
                      const uint32_t* const* pvs) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !pkc || !d_traces) return ZKHIP_ERR_INVALID;
    zkhip_pk* pk = const_cast<zkhip_pk*>(pkc);
    const unsigned b = pk->b, nch = pk->nch, hmax = pk->hmax;
    const size_t n_airs = pk->airs.size();
    hipStream_t st = ctx->stream;
    ZK_TRY(ensure_twiddles(ctx, hmax));
    const uint32_t gen = to_monty(FIELD_GEN_CANON);
    uint32_t* d_alpha = pk->d_chal;
    uint32_t* d_zeta = pk->d_chal + 4;
    uint32_t* d_alpha_f = pk->d_chal + 8;
    uint32_t* d_betas = pk->d_chal + 12;

    // ---- 0. preamble + public values: one pinned staging buffer, no host synchronisation ----
    {
        for (size_t a = 0; a < n_airs; a++) {
            const AirPlan& A = pk->airs[a];
            if (A.n_pvs && (!pvs || !pvs[a])) return set_error(ctx, ZKHIP_ERR_INVALID, "missing public values");
            for (size_t i = 0; i < A.n_pvs; i++)
                if (pvs[a][i] >= P) return set_error(ctx, ZKHIP_ERR_INVALID, "public value not canonical");
        }
        ZK_HIP_CHECK(ctx, hipEventSynchronize(pk->stage_ev));  // returns at once unless the previous proof's copies are pending
        const size_t n_pre = pk->preamble.size();
        LogupArgs* lu_stage = reinterpret_cast<LogupArgs*>(pk->h_stage_pinned);
        uint32_t* pre = pk->h_stage_pinned + pk->n_lu * (sizeof(LogupArgs) / 4);
        if (pk->n_lu) {
            size_t k = 0;
            for (size_t a = 0; a < n_airs; a++)
                if (pk->airs[a].n_int) {
                    if (!d_traces[a]) return set_error(ctx, ZKHIP_ERR_INVALID, "missing trace");
                    lu_stage[k] = pk->lu_args_host[k];
                    lu_stage[k].trace = d_traces[a];
                    k++;
                }
            ZK_HIP_CHECK(ctx, hipMemcpyAsync(pk->d_lu_args, lu_stage, pk->n_lu * sizeof(LogupArgs), hipMemcpyHostToDevice, st));
        }
        uint32_t* hdr = pre + n_pre;
        uint32_t* pm = hdr + 4;
        memcpy(pre, pk->preamble.data(), n_pre * 4);
        hdr[0] = PROOF_MAGIC + (pk->n_lu ? 1u : 0u) + (pk->n_prep ? 2u : 0u) + (pk->n_cached ? 4u : 0u), hdr[1] = (uint32_t)n_airs, hdr[2] = hmax, hdr[3] = pk->n_layers;
        for (size_t a = 0; a < n_airs; a++) {
            const AirPlan& A = pk->airs[a];
            for (size_t i = 0; i < A.n_pvs; i++) {
                pre[pk->preamble_pv_off[a] + i] = pvs[a][i];
                pm[i] = to_monty(pvs[a][i]);
            }
            if (A.n_pvs) ZK_HIP_CHECK(ctx, hipMemcpyAsync(A.d_pvs, pm, A.n_pvs * 4, hipMemcpyHostToDevice, st));
            pm += A.n_pvs;
        }
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(pk->d_preamble, pre, n_pre * 4, hipMemcpyHostToDevice, st));
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(pk->d_proof, hdr, 16, hipMemcpyHostToDevice, st));
        ZK_HIP_CHECK(ctx, hipEventRecord(pk->stage_ev, st));
        ZK_TRY(transcript_init(ctx, pk->d_tr));
        ZK_TRY(transcript_observe(ctx, pk->d_tr, pk->d_preamble, (uint32_t)n_pre, true));
    }

    for (size_t a = 0; a < n_airs; a++)
        if (!d_traces[a]) return set_error(ctx, ZKHIP_ERR_INVALID, "missing trace");
    // pointer tables: main-trace columns live in the caller's buffers; their entries (for the batched LDEs and for the
    // batched openings) are refreshed when the caller passes other buffers than last time
    {
        bool changed = false;
        for (const auto& mc : pk->open_main_cols) {
            const uint32_t* ptr = d_traces[mc.air] + (size_t)mc.col * pk->airs[mc.air].N;
            if (pk->open_ptrs_host[mc.index] != ptr) {
                pk->open_ptrs_host[mc.index] = ptr;
                changed = true;
            }
        }
        if (changed) {
            // the pinned staging buffer may still feed the previous proof's copy
            const size_t n_ptrs = pk->open_total_cols + pk->lde_total_cols;
            ZK_HIP_CHECK(ctx, hipStreamSynchronize(st));
            memcpy(pk->h_open_ptrs_pinned, pk->open_ptrs_host.data(), n_ptrs * sizeof(uint32_t*));
            ZK_HIP_CHECK(ctx, hipMemcpyAsync(pk->d_open_ptrs, pk->h_open_ptrs_pinned, n_ptrs * sizeof(uint32_t*),
                                             hipMemcpyHostToDevice, st));
        }
    }
    auto lde_groups_of = [&](int kind_lo, int kind_hi) -> int {
        for (const auto& g : pk->lde_groups) {
            if (g.kind < kind_lo || g.kind > kind_hi) continue;
            uint32_t shift = gen;
            if (g.kind >= 2) {
                const uint32_t sj = mmul(gen, host_pow(two_adic_generator(g.lh + b), bitrev32((uint32_t)(g.kind - 2), b)));
                shift = mmul(gen, minv(sj));
            }
            ZK_TRY(lde_batch_cols(ctx, pk->d_open_ptrs + pk->open_total_cols + g.first, pk->d_lde_dst + g.first, g.n_cols, g.lh, b, shift));
        }
        return ZKHIP_OK;
    };

    // ---- 1. trace commit: coset LDE + Merkle ----
    // Pipelined form (zkhip_set_commit_pipeline): the tallest chips' columns are extended block by block on the side stream
    // while the main stream's row sponge absorbs the blocks already done (state parked in HBM between blocks).
    const zkhip_pk::LdeGroup* top = nullptr;
    if (ctx->commit_parts >= 2 && pk->n_cached == 0)  // with cached partitions the leaf columns are not the LDE group's
        for (const auto& g : pk->lde_groups)
            if (g.kind == 0 && g.lh + b == hmax && g.lh >= 16 && g.n_cols >= 16 * ctx->commit_parts) top = &g;
    bool leaves_ready = false;
    if (top) {
        if (!ctx->side_stream) {
            if (ctx->side_cus) {
                // CU partition: bit i of a CU mask is CU i, numbered round-robin over the XCDs and then over their shader
                // engines, so the low `side_cus` bits are an even slice of every XCD (its L2 and its share of the HBM channels)
                const unsigned words = ((unsigned)ctx->cu_count + 31) / 32;
                std::vector<uint32_t> m_side(words, 0), m_hash(words, 0);
                for (unsigned i = 0; i < (unsigned)ctx->cu_count; i++) (i < ctx->side_cus ? m_side : m_hash)[i / 32] |= 1u << (i % 32);
                ZK_HIP_CHECK(ctx, hipExtStreamCreateWithCUMask(&ctx->side_stream, words, m_side.data()));
                ZK_HIP_CHECK(ctx, hipExtStreamCreateWithCUMask(&ctx->hash_stream, words, m_hash.data()));
            } else {
                ZK_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            }
            for (auto& e : ctx->pipe_ev)
                if (!e) ZK_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const unsigned parts = ctx->commit_parts;
        const uint32_t per = ((top->n_cols + parts - 1) / parts + 7u) & ~7u;  // block boundaries on sponge-rate multiples
        void* d_state = nullptr;
        ZK_TRY(get_scratch(ctx, 6, ((size_t)16 << hmax) * 4, &d_state));
        // the shorter chips first, on the main stream (the NTT scratch buffers of the context serve one stream at a time)
        for (const auto& g : pk->lde_groups)
            if (g.kind == 0 && &g != top)
                ZK_TRY(lde_batch_cols(ctx, pk->d_open_ptrs + pk->open_total_cols + g.first, pk->d_lde_dst + g.first, g.n_cols, g.lh, b, gen));
        for (size_t a = 0; a < n_airs; a++) {
            AirPlan& A = pk->airs[a];
            if (A.lh < 12) ZK_TRY(lde_batch(ctx, d_traces[a], A.N, A.d_lde, A.M, A.lh, b, A.width, gen));
        }
        // the side stream starts after everything already queued on the main stream (pointer tables, previous proof)
        hipStream_t hs = ctx->hash_stream ? ctx->hash_stream : st;  // where the sponge of the pipeline runs
        ZK_HIP_CHECK(ctx, hipEventRecord(ctx->pipe_ev[8], st));
        ZK_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->pipe_ev[8], 0));
        if (hs != st) ZK_HIP_CHECK(ctx, hipStreamWaitEvent(hs, ctx->pipe_ev[8], 0));
        unsigned k = 0;
        for (uint32_t c0 = 0; c0 < top->n_cols; c0 += per, k++) {
            const uint32_t c1 = std::min(top->n_cols, c0 + per);
            ctx->stream = ctx->side_stream;
            int rc = lde_batch_cols(ctx, pk->d_open_ptrs + pk->open_total_cols + top->first + c0, pk->d_lde_dst + top->first + c0,
                                    c1 - c0, top->lh, b, gen);
            hipError_t he = hipEventRecord(ctx->pipe_ev[k], ctx->side_stream);
            ctx->stream = st;
            if (rc != ZKHIP_OK) return rc;
            ZK_HIP_CHECK(ctx, he);
            ZK_HIP_CHECK(ctx, hipStreamWaitEvent(hs, ctx->pipe_ev[k], 0));
            ctx->stream = hs;
            rc = merkle_leaves_part(ctx, pk->t_main, c0, c1, c0 == 0, c1 == top->n_cols, (uint32_t*)d_state);
            ctx->stream = st;
            if (rc != ZKHIP_OK) return rc;
        }
        if (hs != st) {
            ZK_HIP_CHECK(ctx, hipEventRecord(ctx->pipe_ev[9], hs));
            ZK_HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->pipe_ev[9], 0));
        }
        leaves_ready = true;
    } else {
        ZK_TRY(lde_groups_of(0, 0));  // chips of >= 2^12 rows, one batch per height
        for (size_t a = 0; a < n_airs; a++) {
            AirPlan& A = pk->airs[a];
            if (A.lh < 12) ZK_TRY(lde_batch(ctx, d_traces[a], A.N, A.d_lde, A.M, A.lh, b, A.width, gen));
        }
    }
    ZK_TRY(merkle_build(ctx, pk->t_main, leaves_ready));
    const uint32_t* d_root_main = zkhip_tree_root_device(pk->t_main);
    // main-trace commitments in the reference's order: cached partitions (AIR order), then the common main
    for (size_t a = 0; a < n_airs; a++) {
        AirPlan& A = pk->airs[a];
        if (!A.cw) continue;
        ZK_TRY(merkle_build(ctx, A.t_cached, false));
        ZK_TRY(transcript_observe(ctx, pk->d_tr, zkhip_tree_root_device(A.t_cached), 8, false));
    }
    ZK_TRY(transcript_observe(ctx, pk->d_tr, d_root_main, 8, false));

    // ---- 1b. LogUp phase: permutation trace of every AIR with bus interactions, LDE + commit ----
    const uint32_t* d_root_perm = nullptr;
    if (pk->n_lu) {
        uint32_t* d_gb = pk->d_lchal + N_CHAL;
        ZK_TRY(transcript_sample(ctx, pk->d_tr, d_gb, nullptr, 4));
        ZK_TRY(transcript_sample(ctx, pk->d_tr, d_gb + 4, nullptr, 4));
        {
            KernelScope ks(ctx, "logup_chal");
            hipLaunchKernelGGL(k_logup_chal, dim3(1), dim3(64), 0, st, d_gb, pk->d_lchal);
        }
        {
            const LogupMulti lm{pk->d_lu_args, pk->d_lu_rows_first, pk->d_lu_den_first, (uint32_t)pk->n_lu};
            {
                KernelScope ks(ctx, "logup_denoms");
                hipLaunchKernelGGL(k_logup_denoms, dim3(pk->lu_den_blocks), dim3(LU_BS), (size_t)pk->lu_max_slots * LU_BS * 4, st, lm);
            }
            // one batch inversion over the denominators of every chip (numerators fused)
            ZK_TRY(launch_batch_inverse(ctx, pk->d_lu_den_all, pk->d_lu_den_all, pk->lu_den_elems, pk->d_lu_num_all));
            {
                KernelScope ks(ctx, "logup_rows");
                hipLaunchKernelGGL(k_logup_rows, dim3(pk->lu_rows_blocks), dim3(256), 0, st, lm);
            }
            ZK_TRY(ext_inclusive_scan_multi(ctx, pk->d_lu_scan, (uint32_t)pk->n_lu, pk->lu_scan_blocks, pk->lu_scan_multi));
            {
                KernelScope ks(ctx, "logup_sums");
                hipLaunchKernelGGL(k_logup_sums, dim3(pk->lu_rows_blocks), dim3(256), 0, st, lm);
            }
            ZK_HIP_CHECK(ctx, hipGetLastError());
        }
        for (size_t a = 0; a < n_airs; a++) {
            AirPlan& A = pk->airs[a];
            if (A.n_int && A.lh < 12) ZK_TRY(lde_batch(ctx, A.d_perm, A.N, A.d_perm_lde, A.M, A.lh, b, A.perm_w, gen));
        }
        ZK_TRY(lde_groups_of(1, 1));
        ZK_TRY(merkle_build(ctx, pk->t_perm, false));
        d_root_perm = zkhip_tree_root_device(pk->t_perm);
        ZK_TRY(transcript_observe(ctx, pk->d_tr, d_root_perm, 8, false));
        ZK_TRY(transcript_observe(ctx, pk->d_tr, pk->d_exposed, (uint32_t)(4 * pk->n_lu), false));
    }
    ZK_TRY(transcript_sample(ctx, pk->d_tr, d_alpha, nullptr, 4));

    // ---- 2. quotient ----
    {
        KernelScope ks(ctx, "ext_powers");  // alpha powers of every chip: one launch, one lane per chip
        hipLaunchKernelGGL(k_ext_powers_multi, dim3((unsigned)n_airs), dim3(64), 0, st, d_alpha, pk->d_pow_desc);
    }
    // the compiled kernels (one per chip, each writes its own quotient columns) round-robin over the context's further streams: a proof of fifty
    // chips has a dozen of them that occupy a few workgroups for half a millisecond each (zkhip_config.quot_streams); forked HERE, so that they also run
    // beside the interpreter's launch
    // (only the SHORT ones: at most two workgroups per CU.  A kernel that fills the chip gains nothing beside another one -- the Fibonacci guest's
    // six large chips lost 3 - 10 % when they were spread too)
    // (jit_rows_per_block: 256 = the plain form, 64 = the shared-rows form -- sixteen waves per 64 rows --, 0 = the LDS-tiled form, a fixed grid)
    auto jit_blocks = [&](const AirPlan& A) {
        return (A.jit_rows_per_block & 255u) == 64 ? (unsigned)(A.MQ / 64) : A.jit_rows_per_block ? (unsigned)((A.MQ + 255) / 256) : (unsigned)std::min<size_t>(A.MQ / 64, 4 * 256);
    };
    unsigned n_jit = 0, fan = 0;
    // (an error between fork and join returns from this function: the side streams are drained first -- buffers the next proof reuses may
    // still be written by a kernel queued on them: ADVICE round 5)
    struct DrainSideStreams {
        zkhip_ctx* ctx;
        const unsigned& fan;
        bool joined = false;
        ~DrainSideStreams() {
            if (joined) return;
            for (unsigned q = 0; q < fan; q++)
                if (ctx->quot_streams[q]) (void)hipStreamSynchronize(ctx->quot_streams[q]);
        }
    } drain_side_streams{ctx, fan};
    for (size_t a = 0; a < n_airs; a++) n_jit += pk->airs[a].jit_fn && jit_blocks(pk->airs[a]) <= 2u * (unsigned)ctx->cu_count ? 1u : 0u;
    if (n_jit >= 3 && !ctx->profiling && ctx->cfg.quot_streams) {
        fan = std::min(4u, ctx->cfg.quot_streams);
        if (!ctx->quot_fork) ZK_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->quot_fork, hipEventDisableTiming));
        for (unsigned q = 0; q < fan; q++) {
            if (!ctx->quot_streams[q]) ZK_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->quot_streams[q], hipStreamNonBlocking));
            if (!ctx->quot_join[q]) ZK_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->quot_join[q], hipEventDisableTiming));
        }
        ZK_HIP_CHECK(ctx, hipEventRecord(ctx->quot_fork, st));
        for (unsigned q = 0; q < fan; q++) ZK_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->quot_streams[q], ctx->quot_fork, 0));
    }
    // chips without a compiled kernel: the interpreter runs over all of them in one launch
    if (pk->quot_n) {
        KernelScope ks(ctx, "quotient_eval");
        const QuotMulti qm{pk->d_quot_args, pk->d_quot_first, pk->quot_n, ctx->d_tw_fwd, ctx->tw_log};
        if ((size_t)pk->quot_max_slots * QBS * 4 > 65536)   // (more than the default dynamic LDS: a chip with > 64 live intermediates)
            ZK_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_quotient, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(Q_MAX_SLOTS * QBS * 4)));
        hipLaunchKernelGGL(k_quotient, dim3(pk->quot_blocks), dim3(QBS), (size_t)pk->quot_max_slots * QBS * 4, st, qm);
        for (size_t a = 0; a < n_airs; a++) {
            AirPlan& A = pk->airs[a];
            if (A.d_slice_code.empty()) continue;
            hipLaunchKernelGGL(k_quot_sum_slices, dim3((unsigned)((4 * A.MQ + 255) / 256)), dim3(256), 0, st, A.d_q, A.d_q_part, (uint32_t)A.d_slice_code.size(), A.M,
                               (uint32_t)A.MQ);
        }
        ZK_HIP_CHECK(ctx, hipGetLastError());
    }
    unsigned i_jit = 0;
    for (size_t a = 0; a < n_airs; a++) {
        AirPlan& A = pk->airs[a];
        if (!A.jit_fn) continue;
        // (the large kernels stay on the proof's own stream, one after the other)
        const bool small_one = fan && jit_blocks(A) <= 2u * (unsigned)ctx->cu_count;
        hipStream_t qs = small_one ? ctx->quot_streams[i_jit++ % fan] : st;
        QuotArgs qa = pk->quot_args_host[a];
        const uint32_t* tw_fwd = ctx->d_tw_fwd;
        unsigned tw_shift = ctx->tw_log - A.h;
        KernelScope ks(ctx, "quotient_eval_jit");
        uint32_t h_bits = A.h, nq_rows = (uint32_t)A.MQ;
        void* params[] = {(void*)&qa.lde, (void*)&qa.q,      (void*)&qa.pvs,      (void*)&qa.apow,
                          (void*)&tw_fwd, (void*)&qa.zh,  (void*)&qa.inv_zh,   (void*)&A.d_jit_tab,
                          (void*)&qa.gen, (void*)&qa.w_n_inv, (void*)&tw_shift, (void*)&qa.perm,
                          (void*)&qa.lchal, (void*)&qa.expo, (void*)&qa.prep, (void*)&h_bits, (void*)&nq_rows};
        // (tiled form: one workgroup per CU holds its 150 KB tile; a few rounds of workgroups per CU even the tail out)
        const unsigned blocks = jit_blocks(A);
        ZK_HIP_CHECK(ctx, hipModuleLaunchKernel(A.jit_fn, blocks, 1, 1, (A.jit_rows_per_block & 255u) == 64 ? 64 * (A.jit_rows_per_block >> 8) : A.jit_rows_per_block ? 256 : 64 * QUOT_TILE_WAVES, 1, 1, 0, qs, params, nullptr));
    }
    for (unsigned q = 0; q < fan; q++) {
        ZK_HIP_CHECK(ctx, hipEventRecord(ctx->quot_join[q], ctx->quot_streams[q]));
        ZK_HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->quot_join[q], 0));
    }
    drain_side_streams.joined = true;
    // chunk j = rows [jN,(j+1)N) of q, bit-reversed within the chunk: bring to natural order (all chips, all chunks: one
    // launch per kernel form), then extend from s_j*H to g*K (shift g/s_j) -- p3's quotient-chunk commitment
    ZK_TRY(ntt_bitrev_copy_multi(ctx, pk->d_br_tiled, pk->br_n_tiled, pk->br_blocks_tiled, pk->d_br_small, pk->br_n_small,
                                 pk->br_blocks_small));
    for (size_t a = 0; a < n_airs; a++) {
        AirPlan& A = pk->airs[a];
        if (A.lh >= 12) continue;
        for (unsigned j = 0; j < A.nq; j++) {
            uint32_t sj = mmul(gen, host_pow(two_adic_generator(A.h), bitrev32(j, b)));
            ZK_TRY(lde_batch(ctx, A.d_qnat + (size_t)j * 4 * A.N, A.N, A.d_qlde + (size_t)j * 4 * A.M, A.M, A.lh, b, 4, mmul(gen, minv(sj))));
        }
    }
    ZK_TRY(lde_groups_of(2, 2 + (int)nch - 1));  // chunk j of every chip of one height in one batch
    ZK_TRY(merkle_build(ctx, pk->t_quot, false));
    const uint32_t* d_root_quot = zkhip_tree_root_device(pk->t_quot);
    ZK_TRY(transcript_observe(ctx, pk->d_tr, d_root_quot, 8, false));
    ZK_TRY(transcript_sample(ctx, pk->d_tr, d_zeta, nullptr, 4));

    // ---- 3. openings at zeta (and zeta*w for the trace-like matrices) ----
    // Barycentric weights depend only on (height, coset shift, points), so the columns of ALL matrices that share them
    // -- main, preprocessed and permutation matrices of every AIR of one height; chunk j of every AIR of one height --
    // are reduced by one launch over a pointer table.  Main-trace columns live in the caller's buffers: their table
    // entries are refreshed when the caller passes other buffers than last time.
    // every (height, chunk) group in one launch per kernel: points + scales, barycentric weights, column reduction (one
    // launch per point count), final sums
    if (pk->og_n) {
        const OpenMulti om{pk->d_og, pk->og_n, pk->d_og2_idx, pk->og2_n, pk->d_og1_idx, pk->og1_n, d_zeta, pk->d_pts, pk->d_weights,
                           pk->d_partial, pk->d_open_ptrs, pk->d_open_dst, pk->d_opened, ctx->d_tw_fwd, ctx->tw_log};
        {
            KernelScope ks(ctx, "open_points");
            hipLaunchKernelGGL(k_open_points, dim3(pk->og_n), dim3(64), 0, st, om);
        }
        {
            KernelScope ks(ctx, "bary_weights");
            hipLaunchKernelGGL(k_bary_weights, dim3(pk->og_wblocks), dim3(256), 0, st, om);
        }
        {
            KernelScope ks(ctx, "open_col_reduce");
            if (pk->og2_n) hipLaunchKernelGGL((k_col_reduce<2, 8>), dim3((pk->og2_tiles + 3) / 4), dim3(256), 0, st, om);
            if (pk->og1_n) hipLaunchKernelGGL((k_col_reduce<1, 8>), dim3((pk->og1_tiles + 3) / 4), dim3(256), 0, st, om);
        }
        {
            KernelScope ks(ctx, "open_finish");
            hipLaunchKernelGGL(k_open_finish, dim3(pk->og_fin_blocks), dim3(64), 0, st, om);
        }
        ZK_HIP_CHECK(ctx, hipGetLastError());
    }
    ZK_TRY(transcript_observe(ctx, pk->d_tr, pk->d_opened, (uint32_t)(4 * pk->n_open), false));
    ZK_TRY(transcript_sample(ctx, pk->d_tr, d_alpha_f, nullptr, 4));

    // ---- 4. FRI batching: reduced openings per LDE height ----
    {
        {
            KernelScope ks(ctx, "ext_powers");
            const uint32_t np = (uint32_t)pk->max_w + 1;
            hipLaunchKernelGGL(k_ext_powers_par, dim3((np + 255) / 256), dim3(256), 0, st, d_alpha_f, np, pk->d_apow_f);
        }
        {
            KernelScope ks(ctx, "reduce_prep");
            hipLaunchKernelGGL(k_reduce_prep, dim3(pk->n_ro_items), dim3(64), 0, st, pk->d_ro_items, d_alpha_f, pk->d_apow_f,
                               pk->d_opened, pk->d_ro_slots);
        }
        for (const auto& g : pk->ro_groups) {
            {
                KernelScope ks(ctx, "reduce_prep");
                hipLaunchKernelGGL(k_two_points, dim3(1), dim3(64), 0, st, d_zeta, two_adic_generator(g.lh), pk->d_pts);  // zeta, zeta * w_N
                hipLaunchKernelGGL(k_ro_denoms, dim3((unsigned)((((size_t)1 << g.h) + 255) / 256)), dim3(256), 0, st, pk->d_pts,
                                   ctx->d_tw_fwd, gen, g.h, ctx->tw_log - g.h, pk->d_ro_inv);
            }
            ReduceArgs ra;
            ra.items = pk->d_ro_items + g.first;
            ra.slots = pk->d_ro_slots + 16 * (size_t)g.first;
            ra.n_items = g.n;
            ra.apow = pk->d_apow_f;
            ra.inv = pk->d_ro_inv;
            ra.ro = pk->d_ro[g.h];
            ra.h = g.h;
            {
                KernelScope ks(ctx, "reduced_openings");
                hipLaunchKernelGGL(k_reduced_openings, dim3((unsigned)((((size_t)1 << g.h) + 255) / 256)), dim3(256), 0, st, ra);
            }
            ZK_HIP_CHECK(ctx, hipGetLastError());
        }
    }

    // ---- 5. FRI commit phase ----
    uint32_t* pf = pk->d_proof;
    for (unsigned l = 0; l < pk->n_layers; l++) {
        const unsigned log_len = hmax - l;
        const size_t half = (size_t)1 << (log_len - 1);
        {
            KernelScope ks(ctx, "fri_hash_pairs");
            hipLaunchKernelGGL(k_hash_pairs, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st,
                               (const uint4*)pk->d_flayer[l], half, pk->t_fri[l]->d_digests);
        }
        ZK_TRY(merkle_build(ctx, pk->t_fri[l], true));
        const uint32_t* d_root = zkhip_tree_root_device(pk->t_fri[l]);
        // observe(root); proof <- root; grind; proof <- witness; sample(beta): one launch (four before)
        ZK_TRY(transcript_fri_round(ctx, pk->d_tr, d_root, pk->params.commit_pow_bits, pf + pk->off_fri + 9 * (size_t)l, d_betas + 4 * l));
        ZK_TRY(fri_fold(ctx, pk->d_flayer[l], pk->d_flayer[l + 1], log_len - 1, d_betas + 4 * l,
                        pk->d_ro[log_len - 1], pk->d_ro[log_len - 1] != nullptr));
    }
    // final polynomial: a constant (first value of the last layer) for log_final_poly_len = 0, else the 2^lfp coefficients
    // interpolated from the last layer's 2^(b+lfp) evaluations
    const uint32_t* d_final = pk->d_flayer[pk->n_layers];
    const uint32_t n_fin_words = 4u << pk->lfp;
    if (pk->lfp) {
        uint32_t* d_coef = d_betas + 4 * (size_t)pk->n_layers;
        const unsigned log_last = pk->b + pk->lfp;
        const uint32_t w_inv = minv(two_adic_generator(log_last));
        const uint32_t inv_n = minv(to_monty(1u << log_last));
        KernelScope ks(ctx, "fri_final_poly");
        hipLaunchKernelGGL(k_fri_final_poly, dim3(1), dim3(256), 0, st, d_final, log_last, 1u << pk->lfp, w_inv, inv_n, d_coef);
        d_final = d_coef;
    }
    ZK_TRY(transcript_observe(ctx, pk->d_tr, d_final, n_fin_words, false));
    {
        KernelScope ks(ctx, "copy_canon");
        hipLaunchKernelGGL(k_copy_canon, dim3((n_fin_words + 63) / 64), dim3(64), 0, st, d_final, pf + pk->off_final, n_fin_words);
        hipLaunchKernelGGL(k_copy_canon, dim3(1), dim3(64), 0, st, d_root_main, pf + pk->off_roots, 8u);
        {
            size_t k = 0;
            for (size_t a = 0; a < n_airs; a++)
                if (pk->airs[a].cw)
                    hipLaunchKernelGGL(k_copy_canon, dim3(1), dim3(64), 0, st, zkhip_tree_root_device(pk->airs[a].t_cached),
                                       pf + pk->off_roots_cached + 8 * k++, 8u);
        }
        hipLaunchKernelGGL(k_copy_canon, dim3(1), dim3(64), 0, st, d_root_quot, pf + pk->off_root_quot, 8u);
        if (pk->n_lu) {
            hipLaunchKernelGGL(k_copy_canon, dim3(1), dim3(64), 0, st, d_root_perm, pf + pk->off_root_perm, 8u);
            const uint32_t ne = (uint32_t)(4 * pk->n_lu);
            hipLaunchKernelGGL(k_copy_canon, dim3((ne + 63) / 64), dim3(64), 0, st, pk->d_exposed, pf + pk->off_exposed, ne);
        }
        uint32_t n = (uint32_t)(4 * pk->n_open);
        hipLaunchKernelGGL(k_copy_canon, dim3((n + 255) / 256), dim3(256), 0, st, pk->d_opened, pf + pk->off_opened, n);
    }
    ZK_TRY(transcript_grind(ctx, pk->d_tr, pk->params.query_pow_bits, pf + pk->off_qpow));

    // ---- 6. queries ----
    ZK_TRY(transcript_sample_bits(ctx, pk->d_tr, pk->d_indices, pk->params.num_queries, hmax));
    uint32_t* pq = pf + pk->off_queries;
    ZK_TRY(merkle_open_device(ctx, pk->t_main, pk->d_indices, 0, pk->params.num_queries, pq, pk->query_pitch));
    {
        size_t off = pk->q_main_words;
        for (size_t a = 0; a < n_airs; a++) {
            const AirPlan& A = pk->airs[a];
            if (!A.cw) continue;
            ZK_TRY(merkle_open_device(ctx, A.t_cached, pk->d_indices, hmax - A.h, pk->params.num_queries, pq + off, pk->query_pitch));
            off += A.cw + 8 * (size_t)A.h;
        }
        for (size_t a = 0; a < n_airs; a++) {
            const AirPlan& A = pk->airs[a];
            if (!A.prep_w) continue;
            ZK_TRY(merkle_open_device(ctx, A.t_prep, pk->d_indices, hmax - A.h, pk->params.num_queries, pq + off, pk->query_pitch));
            off += A.prep_w + 8 * (size_t)A.h;
        }
    }
    if (pk->n_lu)
        ZK_TRY(merkle_open_device(ctx, pk->t_perm, pk->d_indices, hmax - pk->h_perm, pk->params.num_queries,
                                  pq + pk->q_main_words + pk->q_cached_words + pk->q_prep_words, pk->query_pitch));
    ZK_TRY(merkle_open_device(ctx, pk->t_quot, pk->d_indices, 0, pk->params.num_queries,
                              pq + pk->q_main_words + pk->q_cached_words + pk->q_prep_words + pk->q_perm_words, pk->query_pitch));
    if (pk->n_layers) {
        KernelScope ks(ctx, "fri_query");
        hipLaunchKernelGGL(k_fri_query, dim3(pk->params.num_queries, pk->n_layers), dim3(64), 0, st, pk->d_fri_desc,
                           pk->d_indices, pq, pk->query_pitch);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    if (ctx->cfg.self_check) {
        // diagnosis: every plain layer of every tree of this proof against its children, the FRI leaves against the layers they hash
        std::vector<std::pair<std::string, const zkhip_tree*>> trees{{"main", pk->t_main}, {"quotient", pk->t_quot}};
        if (pk->n_lu) trees.push_back({"permutation", pk->t_perm});
        for (size_t a = 0; a < n_airs; a++) {
            if (pk->airs[a].t_cached) trees.push_back({"cached " + std::to_string(a), pk->airs[a].t_cached});
            if (pk->airs[a].t_prep) trees.push_back({"preprocessed " + std::to_string(a), pk->airs[a].t_prep});
        }
        for (unsigned l = 0; l < pk->n_layers; l++) trees.push_back({"fri " + std::to_string(l), pk->t_fri[l]});
        const size_t n_rep = trees.size() + pk->n_layers;
        std::vector<uint32_t> rep(2 * n_rep);
        for (size_t i = 0; i < n_rep; i++) rep[2 * i] = 0, rep[2 * i + 1] = 0xffffffffu;
        uint32_t* d_rep = nullptr;
        ZK_HIP_CHECK(ctx, hipMalloc(&d_rep, rep.size() * 4));
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_rep, rep.data(), rep.size() * 4, hipMemcpyHostToDevice, st));
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        for (size_t i = 0; i < trees.size(); i++) ZK_TRY(merkle_check_tree(ctx, trees[i].second, d_rep + 2 * i));
        for (unsigned l = 0; l < pk->n_layers; l++) {
            const size_t half = (size_t)1 << (hmax - l - 1);
            hipLaunchKernelGGL(k_check_pairs, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st, (const uint4*)pk->d_flayer[l], half,
                               (const uint32_t*)pk->t_fri[l]->d_digests, d_rep + 2 * (trees.size() + l));
        }
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(rep.data(), d_rep, rep.size() * 4, hipMemcpyDeviceToHost, st));
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        std::string bad;
        for (size_t i = 0; i < trees.size(); i++)
            if (rep[2 * i]) {
                // every differing node of the tree (first 8): what memory holds (read by copies, not kernels), what its stored children hash
                // to, what the PREVIOUS proof of this key left in the node's place, and where else either value lies -- a lost store leaves
                // the previous proof's node, a misdirected store puts the right node somewhere else, a foreign write leaves neither
                const zkhip_tree* t = trees[i].second;
                const unsigned lh = t->log_height;
                std::vector<uint32_t> all(merkle_digest_count(lh) * 8);
                ZK_HIP_CHECK(ctx, hipMemcpy(all.data(), t->d_digests, all.size() * 4, hipMemcpyDeviceToHost));
                std::vector<uint32_t> shadow;
                size_t shadow_first = 0;   // (first node of the shadow copy)
                {
                    std::lock_guard<std::mutex> lk(g_shadow_mu);
                    auto it = g_shadow.find(t);
                    if (it != g_shadow.end()) shadow = it->second, shadow_first = merkle_digest_count(lh) - shadow.size() / 8;
                }
                auto words = [](const uint32_t* w) {
                    char b[80];
                    std::snprintf(b, sizeof b, "%08x %08x %08x %08x %08x %08x %08x %08x", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
                    return std::string(b);
                };
                auto where = [&](size_t node) {   // node number -> "layer l index i"
                    unsigned l = 0;
                    while (l < lh && node >= t->layer_off[l + 1]) l++;
                    return "layer " + std::to_string(l) + " index " + std::to_string(node - t->layer_off[l]);
                };
                auto find_in = [&](const std::vector<uint32_t>& hay, size_t first_node, const uint32_t* w, size_t skip_node) {
                    std::string r;
                    for (size_t n = 0; n * 8 + 8 <= hay.size(); n++)
                        if (n + first_node != skip_node && memcmp(&hay[n * 8], w, 32) == 0) r += (r.empty() ? "" : ", ") + where(n + first_node);
                    return r.empty() ? std::string("nowhere") : r;
                };
                unsigned shown = 0;
                for (unsigned l = 1; l <= lh && shown < 8; l++) {
                    const unsigned level = lh - l;
                    if (level < t->level_cnt.size() && t->level_cnt[level]) continue;
                    for (size_t idx = 0; idx < ((size_t)1 << level) && shown < 8; idx++) {
                        uint32_t kids[16];
                        memcpy(kids, &all[(t->layer_off[l - 1] + 2 * idx) * 8], 64);
                        poseidon2_permute_host(kids);
                        const size_t node = t->layer_off[l] + idx;
                        const uint32_t* stored = &all[node * 8];
                        if (memcmp(kids, stored, 32) == 0) continue;
                        shown++;
                        unsigned diff_mask = 0;
                        for (int q = 0; q < 8; q++) diff_mask |= (stored[q] != kids[q]) << q;
                        char head[200];
                        std::snprintf(head, sizeof head, "{node layer %u index %zu (byte offset 0x%zx in the store, device address %p) words differing: 0x%02x; ", l, idx, node * 32,
                                      (const void*)(t->d_digests + node * 8), diff_mask);
                        bad += head;
                        bad += "stored [" + words(stored) + "]; children hash to [" + words(kids) + "]; ";
                        if (!shadow.empty() && node >= shadow_first) {
                            const uint32_t* prevw = &shadow[(node - shadow_first) * 8];
                            bad += "previous proof had [" + words(prevw) + "] there: " + (memcmp(prevw, stored, 32) == 0 ? "THE SAME (a store that did not land)" : "different") + "; ";
                            bad += "the stored value lies in the previous proof's store at: " + find_in(shadow, shadow_first, stored, (size_t)-1) + "; ";
                        } else {
                            bad += "no copy of the previous proof's node; ";
                        }
                        bad += "the stored value lies elsewhere in this store at: " + find_in(all, 0, stored, node) + "; the right value lies elsewhere at: " + find_in(all, 0, kids, node) + "} ";
                    }
                }
                uint32_t again[2] = {0, 0xffffffffu};
                ZK_HIP_CHECK(ctx, hipMemcpy(d_rep, again, 8, hipMemcpyHostToDevice));
                ZK_TRY(merkle_check_tree(ctx, t, d_rep));
                ZK_HIP_CHECK(ctx, hipStreamSynchronize(st));
                ZK_HIP_CHECK(ctx, hipMemcpy(again, d_rep, 8, hipMemcpyDeviceToHost));
                bad += "[the host finds " + std::to_string(shown) + " differing nodes in the copy; a second device check finds " + std::to_string(again[0]) + "] ";
            }
        (void)hipFree(d_rep);
        for (size_t i = 0; i < trees.size(); i++) {   // what this proof leaves, for the next one's diagnosis
            const zkhip_tree* t = trees[i].second;
            // (the tail of the store: the layers of <= 2^13 nodes -- all the fused kernels' layers; a large tree's whole store would take longer than the proof)
            const size_t total = merkle_digest_count(t->log_height), keep = std::min(total, ((size_t)2 << 13) - 1);
            std::vector<uint32_t> copy(keep * 8);
            ZK_HIP_CHECK(ctx, hipMemcpy(copy.data(), t->d_digests + (total - keep) * 8, copy.size() * 4, hipMemcpyDeviceToHost));
            std::lock_guard<std::mutex> lk(g_shadow_mu);
            g_shadow[t].swap(copy);
        }
        for (size_t i = 0; i < n_rep; i++)
            if (rep[2 * i]) {
                const std::string name = i < trees.size() ? "tree " + trees[i].first + " (height 2^" + std::to_string(trees[i].second->log_height) + ")"
                                                          : "leaves of fri " + std::to_string(i - trees.size());
                bad += name + ": " + std::to_string(rep[2 * i]) + " nodes differ from the hash of their children, first layer " + std::to_string(rep[2 * i + 1] >> 24) +
                       " index " + std::to_string(rep[2 * i + 1] & 0xffffffu) + "; ";
            }
        if (!bad.empty()) {
            std::fprintf(stderr, "[zkhip self-check] %s\n", bad.c_str());
            return set_error(ctx, ZKHIP_ERR_HIP, "self-check: " + bad);
        }
    }
    return ZKHIP_OK;
}

// K5 on its own: the quotient values of ONE AIR over its LDE domain, through the interpreter kernel.  Stage-level entry like
// zkhip_lde_batch / zkhip_merkle_commit (SURVEY.md 8(b) `zkhip_constraint_eval`); zkhip_prove runs the same kernel over all chips.
int zkhip_constraint_eval(zkhip_ctx* ctx, const uint32_t* program, size_t program_len, unsigned log_height, unsigned log_blowup,
                          size_t width, const uint32_t* d_lde, const uint32_t* pvs, size_t n_pvs, const uint32_t alpha[4], uint32_t* d_q) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !program || !d_lde || !alpha || !d_q || width == 0 || log_blowup < 1 || log_blowup > 4 || log_height + log_blowup > 27)
        return ZKHIP_ERR_INVALID;
    AirProgram prog;
    std::string err;
    if (parse_air(program, program_len, width, &prog, &err) != 0) return set_error(ctx, ZKHIP_ERR_INVALID, "constraint_eval: " + err);
    if (prog.n_pvs != n_pvs || (n_pvs && !pvs)) return set_error(ctx, ZKHIP_ERR_INVALID, "constraint_eval: public values do not match the program");
    if (!prog.ints.empty() || prog.prep_width) return set_error(ctx, ZKHIP_ERR_INVALID, "constraint_eval: AIRs with bus interactions or preprocessed traces are proven through zkhip_prove");
    if (prog.max_degree > (1u << log_blowup) + 1) return set_error(ctx, ZKHIP_ERR_CONSTRAINT, "constraint degree exceeds 2^log_blowup + 1");
    CompiledAir comp;
    if (compile_air(prog, &comp, &err) != 0) return set_error(ctx, ZKHIP_ERR_INVALID, "constraint_eval: " + err);
    const unsigned b = log_blowup, nch = 1u << b, h = log_height + b;
    const size_t N = (size_t)1 << log_height, M = (size_t)1 << h;
    ZK_TRY(ensure_twiddles(ctx, h));
    // small tables: code | consts | pvs | alpha (1 ext) | apow (n_cons ext) | zh | inv_zh | descriptor | prefix
    std::vector<uint32_t> pm(n_pvs + 1), zh(nch), izh(nch), am(4);
    for (size_t i = 0; i < n_pvs; i++) {
        if (pvs[i] >= P) return set_error(ctx, ZKHIP_ERR_INVALID, "public value not canonical");
        pm[i] = to_monty(pvs[i]);
    }
    for (int k = 0; k < 4; k++) {
        if (alpha[k] >= P) return set_error(ctx, ZKHIP_ERR_INVALID, "alpha not canonical");
        am[k] = to_monty(alpha[k]);
    }
    const uint32_t gN = host_pow(to_monty(FIELD_GEN_CANON), N), wb = two_adic_generator(b);
    for (unsigned j = 0; j < nch; j++) zh[j] = msub(mmul(gN, host_pow(wb, j)), MONTY_ONE), izh[j] = minv(zh[j]);
    Bump bp;
    const size_t o_code = bp.take(comp.code.size() * 4 + 16), o_consts = bp.take(comp.consts.size() * 4 + 16), o_pvs = bp.take(n_pvs * 4 + 16);
    const size_t o_alpha = bp.take(16), o_apow = bp.take((size_t)(prog.n_cons + 1) * 16), o_zh = bp.take(nch * 4), o_izh = bp.take(nch * 4);
    const size_t o_args = bp.take(sizeof(QuotArgs)), o_first = bp.take(16), o_pow = bp.take(sizeof(PowDesc));
    void* buf = nullptr;
    ZK_TRY(get_scratch(ctx, 7, bp.off, &buf));
    char* base = (char*)buf;
    ZK_TRY(upload(ctx, base + o_code, comp.code.data(), comp.code.size() * 4));
    ZK_TRY(upload(ctx, base + o_consts, comp.consts.data(), comp.consts.size() * 4));
    ZK_TRY(upload(ctx, base + o_pvs, pm.data(), n_pvs * 4));
    ZK_TRY(upload(ctx, base + o_alpha, am.data(), 16));
    ZK_TRY(upload(ctx, base + o_zh, zh.data(), nch * 4));
    ZK_TRY(upload(ctx, base + o_izh, izh.data(), nch * 4));
    QuotArgs qa;
    memset(&qa, 0, sizeof qa);
    qa.code = (uint32_t*)(base + o_code), qa.n_instr = (uint32_t)(comp.code.size() / 3), qa.consts = (uint32_t*)(base + o_consts);
    qa.pvs = (uint32_t*)(base + o_pvs), qa.apow = (uint32_t*)(base + o_apow), qa.lde = d_lde, qa.q = d_q;
    qa.inv_zh = (uint32_t*)(base + o_izh), qa.zh = (uint32_t*)(base + o_zh), qa.gen = to_monty(FIELD_GEN_CANON);
    qa.w_n_inv = minv(two_adic_generator(log_height)), qa.lh = log_height, qa.b = b, qa.n_rows = (uint32_t)M;
    const uint32_t first[2] = {0, (uint32_t)((M + QBS - 1) / QBS)};
    const PowDesc pd{(uint32_t*)(base + o_apow), prog.n_cons, 0};
    ZK_TRY(upload(ctx, base + o_args, &qa, sizeof qa));
    ZK_TRY(upload(ctx, base + o_first, first, 8));
    ZK_TRY(upload(ctx, base + o_pow, &pd, sizeof pd));
    {
        KernelScope ks(ctx, "ext_powers");
        hipLaunchKernelGGL(k_ext_powers_multi, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)(base + o_alpha), (const PowDesc*)(base + o_pow));
    }
    {
        KernelScope ks(ctx, "quotient_eval");
        const QuotMulti qm{(const QuotArgs*)(base + o_args), (const uint32_t*)(base + o_first), 1u, ctx->d_tw_fwd, ctx->tw_log};
        if ((size_t)comp.n_slots * QBS * 4 > 65536)
            ZK_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_quotient, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(Q_MAX_SLOTS * QBS * 4)));
        hipLaunchKernelGGL(k_quotient, dim3(first[1]), dim3(QBS), (size_t)comp.n_slots * QBS * 4, ctx->stream, qm);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the scratch tables may be reused by the next call
    return ZKHIP_OK;
}

int zkhip_proof_fetch(zkhip_ctx* ctx, const zkhip_pk* pk, uint8_t* out, size_t cap, size_t* out_len) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !pk || !out) return ZKHIP_ERR_INVALID;
    size_t bytes = pk->proof_words * 4;
    if (out_len) *out_len = bytes;
    if (cap < bytes) return set_error(ctx, ZKHIP_ERR_SMALL_BUFFER, "proof buffer too small");
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(out, pk->d_proof, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DevTranscript h;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(&h, pk->d_tr, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h.error) return set_error(ctx, ZKHIP_ERR_POW_FAILED, "proof-of-work search failed");
    if (ctx->cfg.self_check) {   // diagnosis: the same bytes a second time
        std::vector<uint8_t> again(bytes);
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(again.data(), pk->d_proof, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (memcmp(again.data(), out, bytes) != 0) {
            size_t first = 0, n = 0;
            for (size_t i = 0; i < bytes; i++)
                if (again[i] != out[i]) {
                    if (!n) first = i;
                    n++;
                }
            std::fprintf(stderr, "[zkhip self-check] two copies of one proof differ in %zu bytes from byte %zu of %zu\n", n, first, bytes);
            return set_error(ctx, ZKHIP_ERR_HIP, "self-check: two device-to-host copies of one proof differ");
        }
    }
    return ZKHIP_OK;
}

int zkhip_prove(zkhip_ctx* ctx, const zkhip_pk* pk, const uint32_t* const* d_traces, const uint32_t* const* pvs,
                uint8_t* out, size_t cap, size_t* out_len) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !pk || !out) return ZKHIP_ERR_INVALID;
    if (cap < pk->proof_words * 4) {
        if (out_len) *out_len = pk->proof_words * 4;
        return set_error(ctx, ZKHIP_ERR_SMALL_BUFFER, "proof buffer too small");
    }
    ZK_TRY(zkhip_prove_async(ctx, pk, d_traces, pvs));
    return zkhip_proof_fetch(ctx, pk, out, cap, out_len);
}

}  // extern "C"
