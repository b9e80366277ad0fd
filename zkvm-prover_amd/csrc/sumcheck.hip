// sumcheck.hip -- LogUp and sum-check building blocks (K6, K7 of SURVEY.md 2.3; a7.4 of 8(a)).
//
// The backend the reference pins (openvm-stark-backend 2.0.0, Cargo.lock:5337) proves bus
// interactions with LogUp (fractional sums 1/(alpha + ...)) and constraints / stacking with
// sum-check rounds over multilinear tables.  Their device work reduces to
// four protocol-independent streaming kernels over extension-field arrays, built here to the same
// parity bar as the rest (oracle/sumcheck.c):
//   * batch inversion (Montgomery's trick, 8 elements per lane: 3 ext products + 1/8 inversion each),
//   * running sum of num/den (three-phase scan; field addition is exact, so any association order
//     gives the same element),
//   * MLE fold  out[i] = f[2i] + r (f[2i+1] - f[2i])        -- same HBM shape as the FRI fold,
//   * sum-check round s(t) = sum_i prod_j (a_j + t (b_j - a_j)), t = 0..k, with a deterministic
//     two-stage reduction.
// All are HBM-streaming (16-byte lanes, coalesced 1 KiB per wave instruction).
#include <algorithm>

#include "lds_barrier.hpp"
#include "zkhip_internal.hpp"

namespace zk {

__device__ __forceinline__ Ext ld4(const uint32_t* p, size_t i) {
    uint4 v = reinterpret_cast<const uint4*>(p)[i];
    return Ext{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void st4(uint32_t* p, size_t i, const Ext& e) {
    reinterpret_cast<uint4*>(p)[i] = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]);
}

// ---- batch inversion ---------------------------------------------------------------------------
constexpr int BI_K = 8;
// in == out is allowed: a lane reads all of its elements before it writes any of them
__global__ __launch_bounds__(256) void k_ext_batch_inverse(const uint32_t* in, uint32_t* out, size_t n,
                                                           const uint32_t* __restrict__ num) {
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // A zero element (a LogUp denominator vanishes with probability ~2^-120 per row) must not poison the other seven of
    // its group: it takes part in the running product as 1 and its own "inverse" is 0 -- what Fermat's a^(p^4-2) gives,
    // i.e. what the oracle computes element by element.  (The row's LogUp constraint then fails unless its count is 0,
    // exactly as with the reference's per-element inversion.)
    Ext x[BI_K], pre[BI_K];
    unsigned zero_mask = 0;
#pragma unroll
    for (int k = 0; k < BI_K; k++) {
        size_t i = gid + (size_t)k * nthreads;
        x[k] = i < n ? ld4(in, i) : ext_one();
        if ((x[k].c[0] | x[k].c[1] | x[k].c[2] | x[k].c[3]) == 0) {
            zero_mask |= 1u << k;
            x[k] = ext_one();
        }
        pre[k] = k == 0 ? x[0] : ext_mul(pre[k - 1], x[k]);
    }
    Ext inv = ext_inv(pre[BI_K - 1]);
#pragma unroll
    for (int k = BI_K - 1; k >= 0; k--) {
        Ext o = k == 0 ? inv : ext_mul(inv, pre[k - 1]);
        inv = ext_mul(inv, x[k]);
        size_t i = gid + (size_t)k * nthreads;
        if (i < n) {
            if (zero_mask & (1u << k)) o = ext_zero();
            if (num) o = ext_mul_base(o, num[i]);  // LogUp term num/den
            st4(out, i, o);
        }
    }
}

int launch_batch_inverse(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n, const uint32_t* d_num) {
    if (n == 0) return ZKHIP_OK;
    KernelScope ks(ctx, "ext_batch_inverse");
    size_t threads = (n + BI_K - 1) / BI_K;
    unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(k_ext_batch_inverse, dim3(blocks), dim3(256), 0, ctx->stream, d_in, d_out, n, d_num);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

// ---- inclusive scan over extension elements ----------------------------------------------------
constexpr int SC_K = 8;               // consecutive elements per lane
constexpr int SC_BLOCK = 256 * SC_K;  // elements per workgroup

__device__ __forceinline__ Ext wave_inclusive_scan(Ext v, unsigned lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Ext o{{(uint32_t)__shfl_up((int)v.c[0], off, 64), (uint32_t)__shfl_up((int)v.c[1], off, 64),
               (uint32_t)__shfl_up((int)v.c[2], off, 64), (uint32_t)__shfl_up((int)v.c[3], off, 64)}};
        if (lane >= (unsigned)off) v = ext_add(v, o);
    }
    return v;
}

// phase 1: scan inside each workgroup, emit the workgroup total
__global__ __launch_bounds__(256) void k_scan_local(uint32_t* __restrict__ data, size_t n, uint32_t* __restrict__ totals) {
    __shared__ uint32_t wave_tot[4][4];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * SC_BLOCK + (size_t)tid * SC_K;
    Ext v[SC_K];
    Ext run = ext_zero();
#pragma unroll
    for (int k = 0; k < SC_K; k++) {
        v[k] = base + k < n ? ld4(data, base + k) : ext_zero();
        run = ext_add(run, v[k]);
        v[k] = run;
    }
    Ext incl = wave_inclusive_scan(run, lane);
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 4; q++) wave_tot[wave][q] = incl.c[q];
    }
    zk_syncthreads();
    Ext off = ext_sub(incl, run);  // exclusive prefix of this lane inside its wave
    for (unsigned w = 0; w < wave; w++)
        off = ext_add(off, Ext{{wave_tot[w][0], wave_tot[w][1], wave_tot[w][2], wave_tot[w][3]}});
#pragma unroll
    for (int k = 0; k < SC_K; k++)
        if (base + k < n) st4(data, base + k, ext_add(v[k], off));
    if (tid == 255) {
        Ext t = ext_add(off, run);
        st4(totals, blockIdx.x, t);
    }
}
// phase 2: one workgroup scans the workgroup totals in place (exclusive -> stored as inclusive)
__global__ __launch_bounds__(1024) void k_scan_totals(uint32_t* __restrict__ totals, size_t n_blocks) {
    __shared__ uint32_t wave_tot[16][4];
    __shared__ uint32_t carry[4];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < 4) carry[tid] = 0;
    zk_syncthreads();
    for (size_t start = 0; start < n_blocks; start += 1024) {
        size_t i = start + tid;
        Ext v = i < n_blocks ? ld4(totals, i) : ext_zero();
        Ext incl = wave_inclusive_scan(v, lane);
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 4; q++) wave_tot[wave][q] = incl.c[q];
        }
        zk_syncthreads();
        Ext off{{carry[0], carry[1], carry[2], carry[3]}};
        for (unsigned w = 0; w < wave; w++)
            off = ext_add(off, Ext{{wave_tot[w][0], wave_tot[w][1], wave_tot[w][2], wave_tot[w][3]}});
        Ext r = ext_add(incl, off);
        if (i < n_blocks) st4(totals, i, r);
        zk_syncthreads();
        if (tid == 1023) {
#pragma unroll
            for (int q = 0; q < 4; q++) carry[q] = r.c[q];
        }
        zk_syncthreads();
    }
}
// phase 3: add the preceding workgroups' total
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ data, size_t n, const uint32_t* __restrict__ totals) {
    if (blockIdx.x == 0) return;
    const Ext off = ld4(totals, blockIdx.x - 1);
    const size_t base = (size_t)blockIdx.x * SC_BLOCK;
    for (unsigned e = threadIdx.x; e < SC_BLOCK; e += 256)
        if (base + e < n) st4(data, base + e, ext_add(ld4(data, base + e), off));
}

int ext_inclusive_scan(zkhip_ctx* ctx, uint32_t* d_out, size_t n) {
    if (n == 0) return ZKHIP_OK;
    size_t n_blocks = (n + SC_BLOCK - 1) / SC_BLOCK;
    void* totals;
    ZK_TRY(get_scratch(ctx, 5, n_blocks * 16, &totals));
    KernelScope ks(ctx, "logup_scan");
    hipLaunchKernelGGL(k_scan_local, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, d_out, n, (uint32_t*)totals);
    if (n_blocks > 1) {
        hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, ctx->stream, (uint32_t*)totals, n_blocks);
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, d_out, n, (const uint32_t*)totals);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

// ---- the same scan over several independent arrays in three launches (the LogUp running sums of every chip of a proof:
//      42 chips used to cost 42 x 3 launches, each a few microseconds of work) ----
// block b of the flattened grid belongs to the segment whose [first_block, first_block + n_blocks) holds it
__device__ __forceinline__ uint32_t seg_of_block(const ScanSeg* segs, uint32_t n_seg, uint32_t b) {
    uint32_t lo = 0, hi = n_seg - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (segs[mid].first_block <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(256) void k_scan_local_multi(const ScanSeg* __restrict__ segs, uint32_t n_seg) {
    __shared__ uint32_t wave_tot[4][4];
    const ScanSeg sg = segs[seg_of_block(segs, n_seg, blockIdx.x)];
    const uint32_t lb = blockIdx.x - sg.first_block;
    uint32_t* data = sg.data;
    const size_t n = sg.n;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const size_t base = (size_t)lb * SC_BLOCK + (size_t)tid * SC_K;
    Ext v[SC_K];
    Ext run = ext_zero();
#pragma unroll
    for (int k = 0; k < SC_K; k++) {
        v[k] = base + k < n ? ld4(data, base + k) : ext_zero();
        run = ext_add(run, v[k]);
        v[k] = run;
    }
    Ext incl = wave_inclusive_scan(run, lane);
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 4; q++) wave_tot[wave][q] = incl.c[q];
    }
    zk_syncthreads();
    Ext off = ext_sub(incl, run);
    for (unsigned w = 0; w < wave; w++)
        off = ext_add(off, Ext{{wave_tot[w][0], wave_tot[w][1], wave_tot[w][2], wave_tot[w][3]}});
#pragma unroll
    for (int k = 0; k < SC_K; k++)
        if (base + k < n) st4(data, base + k, ext_add(v[k], off));
    if (tid == 255) st4(sg.totals, lb, ext_add(off, run));
}
// one workgroup per segment scans that segment's workgroup totals
__global__ __launch_bounds__(1024) void k_scan_totals_multi(const ScanSeg* __restrict__ segs) {
    __shared__ uint32_t wave_tot[16][4];
    __shared__ uint32_t carry[4];
    const ScanSeg sg = segs[blockIdx.x];
    const size_t n_blocks = sg.n_blocks;
    if (n_blocks <= 1) return;
    uint32_t* totals = sg.totals;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < 4) carry[tid] = 0;
    zk_syncthreads();
    for (size_t start = 0; start < n_blocks; start += 1024) {
        size_t i = start + tid;
        Ext v = i < n_blocks ? ld4(totals, i) : ext_zero();
        Ext incl = wave_inclusive_scan(v, lane);
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 4; q++) wave_tot[wave][q] = incl.c[q];
        }
        zk_syncthreads();
        Ext off{{carry[0], carry[1], carry[2], carry[3]}};
        for (unsigned w = 0; w < wave; w++)
            off = ext_add(off, Ext{{wave_tot[w][0], wave_tot[w][1], wave_tot[w][2], wave_tot[w][3]}});
        Ext r = ext_add(incl, off);
        if (i < n_blocks) st4(totals, i, r);
        zk_syncthreads();
        if (tid == 1023) {
#pragma unroll
            for (int q = 0; q < 4; q++) carry[q] = r.c[q];
        }
        zk_syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_add_multi(const ScanSeg* __restrict__ segs, uint32_t n_seg) {
    const ScanSeg sg = segs[seg_of_block(segs, n_seg, blockIdx.x)];
    const uint32_t lb = blockIdx.x - sg.first_block;
    if (lb == 0) return;
    const Ext off = ld4(sg.totals, lb - 1);
    const size_t base = (size_t)lb * SC_BLOCK;
    for (unsigned e = threadIdx.x; e < SC_BLOCK; e += 256)
        if (base + e < sg.n) st4(sg.data, base + e, ext_add(ld4(sg.data, base + e), off));
}
uint32_t scan_blocks_of(size_t n) { return (uint32_t)((n + SC_BLOCK - 1) / SC_BLOCK); }
int ext_inclusive_scan_multi(zkhip_ctx* ctx, const ScanSeg* d_segs, uint32_t n_seg, uint32_t total_blocks, bool any_multi_block) {
    if (n_seg == 0 || total_blocks == 0) return ZKHIP_OK;
    KernelScope ks(ctx, "logup_scan");
    hipLaunchKernelGGL(k_scan_local_multi, dim3(total_blocks), dim3(256), 0, ctx->stream, d_segs, n_seg);
    if (any_multi_block) {
        hipLaunchKernelGGL(k_scan_totals_multi, dim3(n_seg), dim3(1024), 0, ctx->stream, d_segs);
        hipLaunchKernelGGL(k_scan_add_multi, dim3(total_blocks), dim3(256), 0, ctx->stream, d_segs, n_seg);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int logup_running_sum(zkhip_ctx* ctx, const uint32_t* d_den, const uint32_t* d_num, size_t n, uint32_t* d_out) {
    if (n == 0) return ZKHIP_OK;
    ZK_TRY(launch_batch_inverse(ctx, d_den, d_out, n, d_num));
    return ext_inclusive_scan(ctx, d_out, n);
}

// ---- MLE fold ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mle_fold(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n,
                                                  const uint32_t* __restrict__ r_p) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ext r{{r_p[0], r_p[1], r_p[2], r_p[3]}};
    uint4 a = in[2 * i], b = in[2 * i + 1];
    Ext d{{msub(b.x, a.x), msub(b.y, a.y), msub(b.z, a.z), msub(b.w, a.w)}};
    Ext t = ext_mul(r, d);
    out[i] = make_uint4(madd(a.x, t.c[0]), madd(a.y, t.c[1]), madd(a.z, t.c[2]), madd(a.w, t.c[3]));
}

// ---- sum-check round ---------------------------------------------------------------------------
constexpr int SCR_MAXK = 4;
struct SumcheckArgs {
    const uint32_t* tab[SCR_MAXK];
    uint32_t k;
    size_t n_half;
};
// partial[block][t] for t = 0..k
__global__ __launch_bounds__(256) void k_sumcheck_partial(SumcheckArgs a, uint32_t* __restrict__ partial) {
    __shared__ uint32_t red[4][(SCR_MAXK + 1) * 4];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    Ext acc[SCR_MAXK + 1];
#pragma unroll
    for (int t = 0; t <= SCR_MAXK; t++) acc[t] = ext_zero();
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < a.n_half; i += (size_t)gridDim.x * 256) {
        Ext v[SCR_MAXK], d[SCR_MAXK];
#pragma unroll
        for (int j = 0; j < SCR_MAXK; j++)
            if ((uint32_t)j < a.k) {
                v[j] = ld4(a.tab[j], 2 * i);
                d[j] = ext_sub(ld4(a.tab[j], 2 * i + 1), v[j]);
            }
#pragma unroll
        for (int t = 0; t <= SCR_MAXK; t++)
            if ((uint32_t)t <= a.k) {
                Ext p = v[0];
#pragma unroll
                for (int j = 1; j < SCR_MAXK; j++)
                    if ((uint32_t)j < a.k) p = ext_mul(p, v[j]);
                acc[t] = ext_add(acc[t], p);
#pragma unroll
                for (int j = 0; j < SCR_MAXK; j++)
                    if ((uint32_t)j < a.k) v[j] = ext_add(v[j], d[j]);  // next evaluation point
            }
    }
#pragma unroll
    for (int t = 0; t <= SCR_MAXK; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t x = acc[t].c[q];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x = madd(x, __shfl_xor(x, off, 64));
            if (lane == 0) red[wave][t * 4 + q] = x;
        }
    zk_syncthreads();
    if (tid < (a.k + 1) * 4) {
        uint32_t x = madd(madd(red[0][tid], red[1][tid]), madd(red[2][tid], red[3][tid]));
        partial[(size_t)blockIdx.x * ((SCR_MAXK + 1) * 4) + tid] = x;
    }
}
__global__ __launch_bounds__(64) void k_sumcheck_final(const uint32_t* __restrict__ partial, unsigned n_blocks, unsigned k,
                                                       uint32_t* __restrict__ out_canon) {
    const unsigned slot = blockIdx.x;  // t*4+q
    uint32_t x = 0;
    for (unsigned b = threadIdx.x; b < n_blocks; b += 64) x = madd(x, partial[(size_t)b * ((SCR_MAXK + 1) * 4) + slot]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = madd(x, __shfl_xor(x, off, 64));
    if (threadIdx.x == 0) out_canon[slot] = from_monty(x);
    (void)k;
}

}  // namespace zk

using namespace zk;

extern "C" {

int zkhip_ext_batch_inverse(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_in || !d_out) return ZKHIP_ERR_INVALID;
    return launch_batch_inverse(ctx, d_in, d_out, n, nullptr);
}

int zkhip_logup_running_sum(zkhip_ctx* ctx, const uint32_t* d_den, const uint32_t* d_num, size_t n, uint32_t* d_out,
                            uint32_t* total_out) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_den || !d_num || !d_out) return ZKHIP_ERR_INVALID;
    ZK_TRY(logup_running_sum(ctx, d_den, d_num, n, d_out));
    if (total_out) {
        uint32_t t[4] = {0, 0, 0, 0};
        if (n) ZK_TRY(zkhip_d2h(ctx, t, d_out + 4 * (n - 1), 16));
        for (int i = 0; i < 4; i++) total_out[i] = from_monty(t[i]);
    }
    return ZKHIP_OK;
}

int zkhip_mle_fold(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n, const uint32_t r[4]) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_in || !d_out || !r) return ZKHIP_ERR_INVALID;
    if (n == 0) return ZKHIP_OK;
    uint32_t rm[4];
    for (int i = 0; i < 4; i++) {
        if (r[i] >= P) return set_error(ctx, ZKHIP_ERR_INVALID, "r not canonical");
        rm[i] = to_monty(r[i]);
    }
    void* d_r;
    ZK_TRY(get_scratch(ctx, 2, 16, &d_r));
    ZK_TRY(zkhip_h2d(ctx, d_r, rm, 16));
    KernelScope ks(ctx, "mle_fold");
    hipLaunchKernelGGL(k_mle_fold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4*)d_in,
                       (uint4*)d_out, n, (const uint32_t*)d_r);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int zkhip_sumcheck_round(zkhip_ctx* ctx, const uint32_t* const* d_tables, size_t k, size_t n_half, uint32_t* out) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_tables || !out || k < 1 || k > SCR_MAXK) return ZKHIP_ERR_INVALID;
    SumcheckArgs a;
    for (size_t j = 0; j < SCR_MAXK; j++) a.tab[j] = j < k ? d_tables[j] : nullptr;
    for (size_t j = 0; j < k; j++)
        if (!d_tables[j]) return ZKHIP_ERR_INVALID;
    a.k = (uint32_t)k;
    a.n_half = n_half;
    unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>(1, (n_half + 255) / 256), (size_t)ctx->cu_count * 8);
    void *d_partial, *d_out;
    ZK_TRY(get_scratch(ctx, 5, (size_t)blocks * (SCR_MAXK + 1) * 16, &d_partial));
    ZK_TRY(get_scratch(ctx, 3, (SCR_MAXK + 1) * 16, &d_out));
    {
        KernelScope ks(ctx, "sumcheck_round");
        hipLaunchKernelGGL(k_sumcheck_partial, dim3(blocks), dim3(256), 0, ctx->stream, a, (uint32_t*)d_partial);
        hipLaunchKernelGGL(k_sumcheck_final, dim3((unsigned)((k + 1) * 4)), dim3(64), 0, ctx->stream,
                           (const uint32_t*)d_partial, blocks, (unsigned)k, (uint32_t*)d_out);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return zkhip_d2h(ctx, out, d_out, (k + 1) * 16);
}

}  // extern "C"
