// fri.hip -- arity-2 FRI fold over the quartic extension (K8).
//
// Restates p3-fri's TwoAdicFriFolding::fold_row for arity 2 (the reference's parameters:
// crates/circuits/chunk-circuit/openvm.toml:1-6): with the layer in bit-reversed domain order,
// (e0, e1) = (in[2i], in[2i+1]) are the values at x and -x, x = w_{h+1}^{bitrev_h(i)}, and
//     out[i] = e0 + (beta - x) * (e1 - e0) / (-2x).
// When a reduced-opening vector of the folded length exists it is mixed in as
// out[i] += beta^2 * add[i] (p3-fri commit phase).
//
// HBM-bound streaming kernel: 32 B in, 16 B out per lane, one ext*ext product; 1/x comes from
// the resident inverse-twiddle table (index = bitrev(i), no inversion on the device).
#include "zkhip_internal.hpp"

namespace zk {

__global__ __launch_bounds__(256) void k_fri_fold(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                  unsigned log_n_out, const uint32_t* __restrict__ beta_p,
                                                  const uint4* __restrict__ add, const uint32_t* __restrict__ tw_fwd,
                                                  const uint32_t* __restrict__ tw_inv, unsigned tw_shift) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << log_n_out)) return;
    Ext beta{{beta_p[0], beta_p[1], beta_p[2], beta_p[3]}};
    uint4 a = in[2 * i], b = in[2 * i + 1];
    size_t e = (size_t)bitrev32((uint32_t)i, log_n_out) << tw_shift;
    uint32_t x = tw_fwd[e], xinv = tw_inv[e];
    // c = 1/(-2x) = -(1/2) * (1/x)
    const uint32_t half = (MONTY_ONE & 1u) ? ((MONTY_ONE >> 1) + ((P + 1u) >> 1)) : (MONTY_ONE >> 1);
    uint32_t c = mneg(mmul(xinv, half));
    Ext d{{mmul(msub(b.x, a.x), c), mmul(msub(b.y, a.y), c), mmul(msub(b.z, a.z), c), mmul(msub(b.w, a.w), c)}};
    Ext bx = beta;
    bx.c[0] = msub(bx.c[0], x);
    Ext t = ext_mul(bx, d);
    Ext r{{madd(a.x, t.c[0]), madd(a.y, t.c[1]), madd(a.z, t.c[2]), madd(a.w, t.c[3])}};
    if (add) {
        Ext b2 = ext_mul(beta, beta);
        uint4 v = add[i];
        Ext av{{v.x, v.y, v.z, v.w}};
        r = ext_add(r, ext_mul(b2, av));
    }
    out[i] = make_uint4(r.c[0], r.c[1], r.c[2], r.c[3]);
}

int fri_fold(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, unsigned log_n_out, const uint32_t* d_beta,
             const uint32_t* d_add, bool has_add) {
    ZK_TRY(ensure_twiddles(ctx, log_n_out + 1));
    KernelScope ks(ctx, "fri_fold");
    size_t n = (size_t)1 << log_n_out;
    hipLaunchKernelGGL(k_fri_fold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint4*)d_in, (uint4*)d_out, log_n_out, d_beta, has_add ? (const uint4*)d_add : nullptr,
                       ctx->d_tw_fwd, ctx->d_tw_inv, ctx->tw_log - (log_n_out + 1));
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace zk
