// transcript.hip -- device-resident Fiat-Shamir transcript (T4) and proof-of-work grinding (K9).
//
// Semantics restate p3-challenger 0.4.3 DuplexChallenger<BabyBear, Poseidon2, 16, 8>
// (Cargo.lock:5576): observe buffers inputs and duplexes at RATE, sample pops
// from the end of the 8-word output buffer, sample_bits masks the canonical value, grind
// returns the smallest witness (SURVEY.md 7 "hard parts" (d)).  PoW widths: openvm.toml:5-6.
//
// MI355X design: the sponge state lives in HBM and every observe / sample is a one-wave kernel
// on the prover's stream, reading commitments straight from the Merkle kernels' output and
// writing challenges where the next kernel reads them.  The prover therefore never
// synchronises with the host between stages (the reference's CUDA engine round-trips each
// commitment through the host transcript).
#include <algorithm>
#include <cstdlib>

#include "lds_barrier.hpp"
#include "poseidon2_coop.hpp"
#include "transcript.hpp"
#include "zkhip_internal.hpp"

namespace zk {

// All transcript kernels run one wave; lanes 0..15 hold the sponge state (poseidon2_coop.hpp).
struct TrRegs {
    uint32_t s;  // state word of this lane (lane < 16)
    uint32_t n_in, n_out;
};
__device__ __forceinline__ TrRegs tr_load(const DevTranscript* t, unsigned lane) {
    TrRegs r;
    r.s = t->state[lane & 15u];
    r.n_in = t->n_in;
    r.n_out = t->n_out;
    return r;
}
__device__ __forceinline__ void tr_store(DevTranscript* t, const TrRegs& r, unsigned lane) {
    if (lane < 16) t->state[lane] = r.s;
    if (lane == 0) {
        t->n_in = r.n_in;
        t->n_out = r.n_out;
    }
}
__device__ __forceinline__ void tr_duplex(TrRegs& r, unsigned lane) {
    r.s = coop_permute(r.s, lane & 15u);
    r.n_in = 0;
    r.n_out = 8;
}
__device__ __forceinline__ void tr_observe1(TrRegs& r, unsigned lane, uint32_t v) {
    r.n_out = 0;
    if ((lane & 15u) == r.n_in) r.s = v;
    r.n_in++;
    if (r.n_in == 8) tr_duplex(r, lane);
}
__device__ __forceinline__ uint32_t tr_sample1(TrRegs& r, unsigned lane) {
    if (r.n_in != 0 || r.n_out == 0) tr_duplex(r, lane);
    r.n_out--;
    return __shfl(r.s, (int)r.n_out, 64);
}

__global__ void k_tr_init(DevTranscript* t) {
    if (threadIdx.x < 16) t->state[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        t->n_in = t->n_out = 0;
        t->pow_found = 0xffffffffu;   // ARMED: between two grinds pow_found = 0xffffffff and pad (the blocks-done counter) = 0
        t->pow_applied = 0;
        t->error = 0;
        t->pad = 0;
    }
}

// src_canonical != 0: src holds canonical words (converted on the fly).  The sponge is serial by construction (one
// permutation per 8 words), so the latency of a block is what counts: once the input buffer is aligned, lanes 0..7 fetch a
// whole block with ONE load and the next block is already in flight while this one is permuted (word-by-word uniform loads cost
// more than the permutation itself; a chunk proof observes ~12 k opened words here).
__global__ __launch_bounds__(64) void k_tr_observe(DevTranscript* t, const uint32_t* src, uint32_t n, int src_canonical) {
    const unsigned lane = threadIdx.x;
    TrRegs r = tr_load(t, lane);
    uint32_t i = 0;
    for (; i < n && r.n_in != 0; i++) {
        uint32_t v = src[i];
        tr_observe1(r, lane, src_canonical ? to_monty(v) : v);
    }
    if (i + 8 <= n) {
        const unsigned l8 = lane & 7u;
        const CoopConsts cc = coop_load_consts(lane & 15u);
        uint32_t nxt = src[i + l8];
        while (i + 8 <= n) {
            uint32_t cur = src_canonical ? to_monty(nxt) : nxt;
            i += 8;
            if (i + 8 <= n) nxt = src[i + l8];
            if ((lane & 15u) < 8) r.s = cur;
            r.s = coop_permute_regs(r.s, lane & 15u, cc);
        }
        r.n_in = 0, r.n_out = 8;
    }
    for (; i < n; i++) {
        uint32_t v = src[i];
        tr_observe1(r, lane, src_canonical ? to_monty(v) : v);
    }
    tr_store(t, r, lane);
}

// dst_monty: optional Montgomery output; dst_canon: optional canonical output (e.g. the proof)
__global__ __launch_bounds__(64) void k_tr_sample(DevTranscript* t, uint32_t* dst_monty, uint32_t* dst_canon, uint32_t n) {
    const unsigned lane = threadIdx.x;
    TrRegs r = tr_load(t, lane);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t v = tr_sample1(r, lane);
        if (lane == 0) {
            if (dst_monty) dst_monty[i] = v;
            if (dst_canon) dst_canon[i] = from_monty(v);
        }
    }
    tr_store(t, r, lane);
}

__global__ __launch_bounds__(64) void k_tr_sample_bits(DevTranscript* t, uint32_t* dst, uint32_t n, unsigned bits) {
    const unsigned lane = threadIdx.x;
    TrRegs r = tr_load(t, lane);
    uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t v = from_monty(tr_sample1(r, lane)) & mask;
        if (lane == 0) dst[i] = v;
    }
    tr_store(t, r, lane);
}

// Applies the found witness to the live transcript and stores it (canonical); leaves the transcript armed for the next grind.
__device__ __forceinline__ void grind_apply(DevTranscript* t, unsigned bits, uint32_t* witness_out, unsigned lane) {
    const uint32_t w = __hip_atomic_load(&t->pow_found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w == 0xffffffffu) {
        if (lane == 0) t->error |= 1u, t->pad = 0;
        return;
    }
    TrRegs r = tr_load(t, lane);
    tr_observe1(r, lane, to_monty(w));
    const uint32_t mask = bits ? ((1u << bits) - 1u) : 0u;
    const uint32_t v = from_monty(tr_sample1(r, lane)) & mask;
    tr_store(t, r, lane);
    if (lane == 0) {
        if (v != 0) t->error |= 2u;
        if (witness_out) *witness_out = w;
        t->pow_applied = 1;
        t->pow_found = 0xffffffffu, t->pad = 0;   // armed again
    }
}

// Proof-of-work search AND the application of the witness in ONE launch (round 4: a grind was three launches -- arm, search, finish --
// and a proof has ~20 of them).  Every lane tests candidates base + gid, base advancing by the grid size.  observe(w);
// sample_bits(bits) costs exactly one permutation whichever branch of the duplex logic is taken.  A block leaves once a witness below
// its next candidate window is known; since every block scans its candidates in increasing order, the atomicMin result is the global
// minimum (= the first witness a sequential scan finds).  The LAST block to leave (a counter in the transcript) applies the witness
// with its first wave and re-arms the transcript.
__global__ __launch_bounds__(256) void k_grind(DevTranscript* t, unsigned bits, uint32_t limit, uint32_t* witness_out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t s0[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s0[i] = t->state[i];
    const uint32_t n_in = t->n_in;
    for (uint32_t base = 0; base < limit; base += stride) {
        // volatile read: other blocks publish with atomicMin (device scope)
        const uint32_t found = __hip_atomic_load(&t->pow_found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (found < base) break;
        const uint32_t w = base + gid;
        if (w >= limit || w < base) continue;
        uint32_t s[16];
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = s0[i];
        const uint32_t wm = to_monty(w);
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((uint32_t)i == n_in) s[i] = wm;
        poseidon2_permute(s);
        if ((from_monty(s[7]) & mask) == 0) atomicMin(&t->pow_found, w);
    }
    // every lane of the block has read the state it needed and published what it found: count the block out
    __shared__ uint32_t last;
    __threadfence();
    zk_syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&t->pad, 1u) == gridDim.x - 1 ? 1u : 0u;
    zk_syncthreads();
    if (!last || threadIdx.x >= 64) return;
    __threadfence();
    grind_apply(t, bits, witness_out, threadIdx.x);
}

// One ROUND of the FRI commit phase on the transcript in ONE launch: observe the layer's root (8 words), grind `bits` bits, sample the
// folding challenge (4 words) -- four launches before (observe, copy of the root into the proof, grind, sample), twenty rounds per proof.
// Every lane absorbs the root into its OWN copy of the state before it tests its candidates (one more permutation at most, beside the
// one per candidate); the last block to leave redoes the absorption on the live transcript with its first wave, applies the witness,
// samples, and writes [root (canonical) | witness] into the proof and the challenge where the fold reads it.  bits = 0: no search (p3's
// grind(0): witness 0).
__global__ __launch_bounds__(256) void k_fri_round_transcript(DevTranscript* t, const uint32_t* __restrict__ root, unsigned bits, uint32_t limit,
                                                               uint32_t* __restrict__ proof_out, uint32_t* __restrict__ beta_out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t mask = bits ? (1u << bits) - 1u : 0u;
    if (bits) {
        uint32_t s0[16];
#pragma unroll
        for (int i = 0; i < 16; i++) s0[i] = t->state[i];
        uint32_t n_in = t->n_in;
        for (int k = 0; k < 8; k++) {   // observe(root[k]) on the private copy
            const uint32_t v = root[k];
#pragma unroll
            for (int i = 0; i < 8; i++)
                if ((uint32_t)i == n_in) s0[i] = v;
            if (++n_in == 8) poseidon2_permute(s0), n_in = 0;
        }
        for (uint32_t base = 0; base < limit; base += stride) {
            const uint32_t found = __hip_atomic_load(&t->pow_found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (found < base) break;
            const uint32_t w = base + gid;
            if (w >= limit || w < base) continue;
            uint32_t s[16];
#pragma unroll
            for (int i = 0; i < 16; i++) s[i] = s0[i];
            const uint32_t wm = to_monty(w);
#pragma unroll
            for (int i = 0; i < 8; i++)
                if ((uint32_t)i == n_in) s[i] = wm;
            poseidon2_permute(s);
            if ((from_monty(s[7]) & mask) == 0) atomicMin(&t->pow_found, w);
        }
    }
    __shared__ uint32_t last;
    __threadfence();
    zk_syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&t->pad, 1u) == gridDim.x - 1 ? 1u : 0u;
    zk_syncthreads();
    if (!last || threadIdx.x >= 64) return;
    __threadfence();
    const unsigned lane = threadIdx.x;
    const uint32_t w = bits ? __hip_atomic_load(&t->pow_found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (w == 0xffffffffu) {
        if (lane == 0) t->error |= 1u, t->pad = 0;
        return;
    }
    TrRegs r = tr_load(t, lane);
    for (int k = 0; k < 8; k++) {
        const uint32_t v = root[k];
        tr_observe1(r, lane, v);
        if (lane == 0) proof_out[k] = from_monty(v);
    }
    tr_observe1(r, lane, to_monty(w));
    const uint32_t chk = from_monty(tr_sample1(r, lane)) & mask;
    for (int k = 0; k < 4; k++) {
        const uint32_t b = tr_sample1(r, lane);
        if (lane == 0) beta_out[k] = b;
    }
    tr_store(t, r, lane);
    if (lane == 0) {
        if (chk != 0) t->error |= 2u;
        proof_out[8] = w;
        t->pow_applied = 1;
        t->pow_found = 0xffffffffu, t->pad = 0;   // armed again
    }
}

// grind(0): p3 semantics -- observe witness 0 and sample (no search)
__global__ __launch_bounds__(64) void k_grind_finish(DevTranscript* t, unsigned bits, uint32_t* witness_out) { grind_apply(t, bits, witness_out, threadIdx.x); }

__global__ void k_grind_arm(DevTranscript* t, uint32_t init) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        t->pow_found = init;
        t->pow_applied = 0;
    }
}

int transcript_init(zkhip_ctx* ctx, DevTranscript* d_t) {
    KernelScope ks(ctx, "transcript");
    hipLaunchKernelGGL(k_tr_init, dim3(1), dim3(64), 0, ctx->stream, d_t);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}
// Long absorptions on the host (csrc/poseidon2_avx512.cpp): the sponge is a chain of dependent permutations, ~1.5 us each on the device
// whatever the occupancy, ~0.25 us in a 512-bit register of the host -- and the words absorbed here (the opened values) are part of the
// proof, so they cross PCIe anyway.  One stream synchronisation; the state goes back with a 96-byte copy.  zkhip_config.host_sponge = 0 (or a
// CPU without AVX-512) keeps everything on the device; host_sponge_min_words moves the threshold (default 8192 words: below
// that the device's ~0.2 us per word costs less than the synchronisation, after which the host has to catch up with its launches).
void poseidon2_permute_avx512(uint32_t s[16]);
static bool host_sponge_available() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    return ok;
}
static int observe_on_host(zkhip_ctx* ctx, DevTranscript* d_t, const uint32_t* d_src, uint32_t n, bool canonical) {
    const size_t bytes = sizeof(DevTranscript) + (size_t)n * 4;
    if (ctx->h_sponge_bytes < bytes) {
        if (ctx->h_sponge) (void)hipHostFree(ctx->h_sponge);
        ctx->h_sponge = nullptr, ctx->h_sponge_bytes = 0;
        if (hipHostMalloc(&ctx->h_sponge, bytes * 2, hipHostMallocDefault) != hipSuccess) return set_error(ctx, ZKHIP_ERR_NOMEM, "pinned staging of the host sponge");
        ctx->h_sponge_bytes = bytes * 2;
    }
    DevTranscript* t = (DevTranscript*)ctx->h_sponge;
    uint32_t* w = (uint32_t*)(t + 1);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(t, d_t, sizeof(DevTranscript), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(w, d_src, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // the duplex challenger's observe, word by word: inputs overwrite the front of the state, eight of them trigger a permutation
    // (tr_observe1 above)
    uint32_t n_in = t->n_in;
    for (uint32_t i = 0; i < n; i++) {
        t->state[n_in++] = canonical ? to_monty(w[i]) : w[i];
        if (n_in == 8) poseidon2_permute_avx512(t->state), n_in = 0;
    }
    t->n_in = n_in, t->n_out = n_in == 0 ? 8u : 0u;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_t, t, sizeof(DevTranscript), hipMemcpyHostToDevice, ctx->stream));
    // (the staging buffer is reused by the next call: the copy must have left it)
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int transcript_observe(zkhip_ctx* ctx, DevTranscript* d_t, const uint32_t* d_src, uint32_t n, bool canonical) {
    if (n == 0) return ZKHIP_OK;
    KernelScope ks(ctx, "transcript");
    if (ctx->cfg.host_sponge && n >= ctx->cfg.host_sponge_min_words && host_sponge_available()) return observe_on_host(ctx, d_t, d_src, n, canonical);
    hipLaunchKernelGGL(k_tr_observe, dim3(1), dim3(64), 0, ctx->stream, d_t, d_src, n, canonical ? 1 : 0);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}
int transcript_sample(zkhip_ctx* ctx, DevTranscript* d_t, uint32_t* d_monty, uint32_t* d_canon, uint32_t n) {
    if (n == 0) return ZKHIP_OK;
    KernelScope ks(ctx, "transcript");
    hipLaunchKernelGGL(k_tr_sample, dim3(1), dim3(64), 0, ctx->stream, d_t, d_monty, d_canon, n);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}
int transcript_sample_bits(zkhip_ctx* ctx, DevTranscript* d_t, uint32_t* d_dst, uint32_t n, unsigned bits) {
    if (n == 0) return ZKHIP_OK;
    KernelScope ks(ctx, "transcript");
    hipLaunchKernelGGL(k_tr_sample_bits, dim3(1), dim3(64), 0, ctx->stream, d_t, d_dst, n, bits);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

// One launch per grind (the search's last block applies the witness); never needs the host.
int transcript_grind(zkhip_ctx* ctx, DevTranscript* d_t, unsigned bits, uint32_t* d_witness_out) {
    if (bits > 30) return set_error(ctx, ZKHIP_ERR_INVALID, "pow bits > 30");
    KernelScope ks(ctx, "pow_grind");
    if (bits == 0) {
        // p3 semantics: grind(0) observes witness 0 and samples
        hipLaunchKernelGGL(k_grind_arm, dim3(1), dim3(64), 0, ctx->stream, d_t, 0u);
        hipLaunchKernelGGL(k_grind_finish, dim3(1), dim3(64), 0, ctx->stream, d_t, bits, d_witness_out);
    } else {
        // a sweep covers 2^bits candidates (the expected position of the witness): measured on a node proof's 21 grinds 1.08 ms against
        // 1.22 ms with 2^(bits+1) and 1.64 ms with 2^(bits+2) per sweep -- a permutation per lane is latency, more lanes only queue
        const unsigned grind_shift = std::min(ctx->cfg.grind_sweep_shift, 8u);
        unsigned blocks = std::min(4096u, std::max(64u, ((1u << grind_shift) << bits) / 256u));
        hipLaunchKernelGGL(k_grind, dim3(blocks), dim3(256), 0, ctx->stream, d_t, bits, (uint32_t)P, d_witness_out);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

// observe(root) + grind(bits) + sample(4) of one FRI commit round in one launch; d_proof_out: 9 words [root | witness] (canonical)
int transcript_fri_round(zkhip_ctx* ctx, DevTranscript* d_t, const uint32_t* d_root, unsigned bits, uint32_t* d_proof_out, uint32_t* d_beta_out) {
    if (bits > 30) return set_error(ctx, ZKHIP_ERR_INVALID, "pow bits > 30");
    KernelScope ks(ctx, "fri_round_transcript");
    const unsigned grind_shift = std::min(ctx->cfg.grind_sweep_shift, 8u);
    const unsigned blocks = bits ? std::min(4096u, std::max(64u, ((1u << grind_shift) << bits) / 256u)) : 1u;
    hipLaunchKernelGGL(k_fri_round_transcript, dim3(blocks), dim3(256), 0, ctx->stream, d_t, d_root, bits, (uint32_t)P, d_proof_out, d_beta_out);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace zk
