// merkle.hip -- Poseidon2 Merkle commitment (K2 row sponge, K3 compression layers, K4 openings).
//
// Replaces the engine's "Merkle-Poseidon2 commit" of RS-encoded matrices (SURVEY.md 8(a) a7.2,
// a7.6): leaf i = PaddingFreeSponge<16,8,8> over the concatenated row i of every tallest matrix,
// parent = TruncatedPermutation(left || right), and a matrix of height s is injected into the
// layer of size s as node = compress(node, sponge(row)) -- the published p3 MerkleTreeMmcs rule.
// Digest = 8 words (crates/types/src/proof.rs:209).
//
// gfx950 mapping: one matrix row per lane.  Matrices are column-major, so lane i reading row i
// of column c is a 256-byte coalesced wave access with no transposition; the 16-word sponge
// state lives in VGPRs, round constants are scalar loads (rolled rounds, poseidon2_coop.hpp), the column-pointer table is wave-uniform
// (scalar loads).  The kernel is integer-VALU bound (~600 Montgomery products per permutation),
// not HBM bound -- see DESIGN.md "Rooflines".
#include <algorithm>

#include "lds_barrier.hpp"
#include "poseidon2_coop.hpp"
#include "zkhip_internal.hpp"

namespace zk {

// one copy of the permutation per call site: full and ragged 8-column blocks share it.  (Requesting the
// cells of block j+1 before permuting block j changes nothing: 47.17 vs 47.23 ms -- the other resident
// waves already hide the loads.)
__device__ __forceinline__ void absorb_rows(uint32_t (&s)[16], const uint32_t* const* __restrict__ cols,
                                            uint32_t n_cols, uint32_t row) {
#pragma unroll 1
    for (uint32_t j = 0; j < n_cols; j += 8) {
        if (j + 8 <= n_cols) {
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = cols[j + k][row];
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (j + k < n_cols) s[k] = cols[j + k][row];
        }
        poseidon2_permute_rolled(s);
    }
}

// (launched with 256 lanes per workgroup, or with 768 -- zkhip_config.hash_block: two workgroups of 12 waves fill 6 of a SIMD's 8 wave slots
// and a third does not fit, so two slots, 176 VGPRs and the whole LDS of every CU stay free for a memory-bound kernel of another stream)
__global__ __launch_bounds__(768) void k_hash_rows(const uint32_t* const* __restrict__ cols, uint32_t n_cols,
                                                   size_t n_rows, uint32_t* __restrict__ out) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;  // heights are <= 2^27: 32-bit offsets
    if (row >= n_rows) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0;
    absorb_rows(s, cols, n_cols, row);
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)row * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// The row sponges of SEVERAL levels of a mixed-height tree in one launch (round 4): the leaf level and every level with injected
// matrices that is too large for the cooperative layer form.  A proof of many chips of many heights hashes most of its rows at the
// levels below the tallest one; hashed inside the layer kernel (one lane per node: compress, sponge, compress) those levels ran at
// one or two waves per SIMD -- here their rows run beside the leaf level's at full occupancy, and the layer kernel is left with two
// permutations per node (k_compress_layer_inj2).  The row digest of node i of a level is parked in the node's own slot of the digest
// store until the layer kernel replaces it.
// (second session of round 5) EVERY level with injected matrices is hashed here, whatever its size (zkhip_config.rows_in_bulk): a segment proof of the
// chunk-circuit configuration carries wide chips of few rows (Keccak-f: 2634 columns x 2^14 LDE rows = 330 permutations per row; the limb chips: ~3000
// columns at 2^11 - 2^12 rows), and the cooperative tree kernels ran those sponges on every 16-lane row of every workgroup of the group's FIRST layer --
// rows that a deeper layer no longer uses included: 12.7 + 8.9 ms of a 74 ms proof in two launches (docs/round5_b.md).  A level's blocks take
// one of two forms: one lane per row, or -- a level of few rows and many columns, whose one-lane chain of hundreds of permutations would outlast the
// rest of the launch -- one 16-lane row per matrix row (the cooperative permutation: ~2.7x the VALU work, an eighth of the chain).  The levels with the
// longest chains come first in the grid and run at wave priority 3, so that the chains end under the bulk of the launch instead of behind it.
constexpr unsigned ZK_MAX_ROW_LEVELS = 28;
struct RowHashLevels {
    uint32_t n_levels;
    uint32_t first_block[ZK_MAX_ROW_LEVELS + 1];   // blocks [first_block[k], first_block[k + 1]) hash entry k's rows
    uint32_t col_off[ZK_MAX_ROW_LEVELS], n_cols[ZK_MAX_ROW_LEVELS];
    uint8_t log_rows[ZK_MAX_ROW_LEVELS];
    uint8_t mode[ZK_MAX_ROW_LEVELS];               // bit 0: cooperative form (16 lanes per row), bit 1: wave priority 3
    uint64_t out_off[ZK_MAX_ROW_LEVELS];           // word offset of the level's layer in the digest store
};
__global__ __launch_bounds__(256) void k_hash_rows_multi(const uint32_t* const* __restrict__ cols, RowHashLevels L, uint32_t* __restrict__ digests) {
    uint32_t k = 0;
    while (k + 1 < L.n_levels && blockIdx.x >= L.first_block[k + 1]) k++;   // (wave-uniform)
    const uint32_t mode = L.mode[k];
    if (mode & 2u) __builtin_amdgcn_s_setprio(3);
    const size_t n_rows = (size_t)1 << L.log_rows[k];
    const uint32_t blk = blockIdx.x - L.first_block[k];
    if (mode & 1u) {
        // sponge over the row: lane q < 8 of the 16-lane row absorbs column c + q, lanes 8..15 are the capacity; the next block's cells are
        // requested before this block's permutation
        const unsigned lane = threadIdx.x & 15u;
        const size_t row = (size_t)blk * 16 + (threadIdx.x >> 4);
        const size_t rr = row < n_rows ? row : 0;   // all 16 lanes of every row stay active through the DPP permutation
        const uint32_t* const* ic = cols + L.col_off[k];
        const uint32_t n = L.n_cols[k];
        uint32_t h = 0, nxt = (lane < 8 && lane < n) ? ic[lane][rr] : 0;
        for (uint32_t c = 0; c < n; c += 8) {
            if (lane < 8 && c + lane < n) h = nxt;
            if (lane < 8 && c + 8 + lane < n) nxt = ic[c + 8 + lane][rr];
            h = coop_permute(h, lane);
        }
        if (row < n_rows && lane < 8) digests[L.out_off[k] + row * 8 + lane] = h;
        return;
    }
    const uint32_t row = blk * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0;
    absorb_rows(s, cols + L.col_off[k], L.n_cols[k], row);
    uint4* o = reinterpret_cast<uint4*>(digests + L.out_off[k] + (size_t)row * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// next[i] = compress(compress(prev[2 i], prev[2 i + 1]), row digest parked in next[i]) -- the layer step of a level whose rows
// k_hash_rows_multi has hashed
__global__ __launch_bounds__(256) void k_compress_layer_inj2(const uint32_t* __restrict__ prev, uint32_t* __restrict__ next, size_t n_next) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_next) return;
    uint32_t s[16];
    const uint4* p = reinterpret_cast<const uint4*>(prev + i * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = p[q];
        s[4 * q] = v.x, s[4 * q + 1] = v.y, s[4 * q + 2] = v.z, s[4 * q + 3] = v.w;
    }
    uint4* o = reinterpret_cast<uint4*>(next + i * 8);
    const uint4 r0 = o[0], r1 = o[1];
#pragma unroll 1
    for (int step = 0; step < 2; step++) {
        if (step == 1) s[8] = r0.x, s[9] = r0.y, s[10] = r0.z, s[11] = r0.w, s[12] = r1.x, s[13] = r1.y, s[14] = r1.z, s[15] = r1.w;
        poseidon2_permute_rolled(s);
    }
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// One step of the same sponge over a block of columns, the 16-word state parked in HBM between steps ([16][rows]: a
// lane's loads and stores are coalesced with its neighbours').  Lets the trace commit run as a pipeline: while the
// sponge absorbs block k the LDE of block k+1 is computed on another stream.
__global__ __launch_bounds__(256) void k_hash_rows_part(const uint32_t* const* __restrict__ cols, uint32_t n_cols, size_t n_rows,
                                                        uint32_t* __restrict__ state, int first, int last,
                                                        uint32_t* __restrict__ out) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    uint32_t s[16];
    if (first) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = 0;
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = state[(size_t)i * n_rows + row];
    }
    absorb_rows(s, cols, n_cols, row);
    if (last) {
        uint4* o = reinterpret_cast<uint4*>(out + (size_t)row * 8);
        o[0] = make_uint4(s[0], s[1], s[2], s[3]);
        o[1] = make_uint4(s[4], s[5], s[6], s[7]);
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) state[(size_t)i * n_rows + row] = s[i];
    }
}

// next[i] = compress(prev[2i], prev[2i+1]); with injected matrices: compress(that, sponge(row i))
__global__ __launch_bounds__(256) void k_compress_layer(const uint32_t* __restrict__ prev,
                                                        uint32_t* __restrict__ next, size_t n_next) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_next) return;
    uint32_t s[16];
    const uint4* p = reinterpret_cast<const uint4*>(prev + i * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = p[q];
        s[4 * q] = v.x, s[4 * q + 1] = v.y, s[4 * q + 2] = v.z, s[4 * q + 3] = v.w;
    }
    poseidon2_permute(s);  // the unrolled form is 5 % faster here (short kernel, one call site)
    uint4* o = reinterpret_cast<uint4*>(next + i * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// The layer step with shorter matrices injected: node = compress(compress(left, right), sponge(row i)).  ONE call site
// of the rolled permutation serves the compression, every 8-column block of the row sponge and the final compression
// (three inlined copies are ~80 KB of code, more than the instruction cache; with many chips of many heights most rows
// of a proof are hashed here, not in k_hash_rows).
__global__ __launch_bounds__(256) void k_compress_layer_inj(const uint32_t* __restrict__ prev, uint32_t* __restrict__ next,
                                                            size_t n_next, const uint32_t* const* __restrict__ inj_cols,
                                                            uint32_t n_inj_cols) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_next) return;
    uint32_t cur[16], node[8];
    const uint4* p = reinterpret_cast<const uint4*>(prev + i * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = p[q];
        cur[4 * q] = v.x, cur[4 * q + 1] = v.y, cur[4 * q + 2] = v.z, cur[4 * q + 3] = v.w;
    }
    const uint32_t n_blocks = (n_inj_cols + 7) / 8;
#pragma unroll 1
    for (uint32_t step = 0; step < n_blocks + 2; step++) {
        if (step >= 1 && step <= n_blocks) {
            if (step == 1) {  // the pair is compressed: keep its digest, start the row sponge
#pragma unroll
                for (int k = 0; k < 8; k++) node[k] = cur[k];
#pragma unroll
                for (int k = 0; k < 16; k++) cur[k] = 0;
            }
            const uint32_t j = 8 * (step - 1);
            if (j + 8 <= n_inj_cols) {
#pragma unroll
                for (int k = 0; k < 8; k++) cur[k] = inj_cols[j + k][i];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (j + k < n_inj_cols) cur[k] = inj_cols[j + k][i];
            }
        } else if (step == n_blocks + 1) {  // compress(node digest, row digest)
#pragma unroll
            for (int k = 0; k < 8; k++) cur[8 + k] = cur[k];
#pragma unroll
            for (int k = 0; k < 8; k++) cur[k] = node[k];
        }
        poseidon2_permute_rolled(cur);
    }
    uint4* o = reinterpret_cast<uint4*>(next + i * 8);
    o[0] = make_uint4(cur[0], cur[1], cur[2], cur[3]);
    o[1] = make_uint4(cur[4], cur[5], cur[6], cur[7]);
}

__global__ void k_permute_batch(uint32_t* states, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[16];
    uint4* p = reinterpret_cast<uint4*>(states + i * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = p[q];
        s[4 * q] = v.x, s[4 * q + 1] = v.y, s[4 * q + 2] = v.z, s[4 * q + 3] = v.w;
    }
    poseidon2_permute_rolled(s);
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = make_uint4(s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]);
}

// The same layer step with one node per 16-lane row (cooperative permutation, poseidon2_coop.hpp): ~8x shorter
// dependent chain per node at 16x the lanes.  Used for layers of <= 2^15 nodes, which cannot fill the chip with one
// lane per node anyway -- there a layer costs one permutation LATENCY, and trees have a dozen such layers (more when
// matrices of many heights are injected).
__global__ __launch_bounds__(256) void k_compress_layer_coop(const uint32_t* __restrict__ prev, uint32_t* __restrict__ next,
                                                             size_t n_next, const uint32_t* const* __restrict__ inj_cols,
                                                             uint32_t n_inj_cols, int parked) {
    const unsigned lane = threadIdx.x & 15u;
    const size_t i = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const size_t ii = i < n_next ? i : 0;  // all 16 lanes of every row stay active through the DPP permutation
    const CoopConsts cc = coop_load_consts(lane);
    uint32_t x = coop_permute_regs(prev[ii * 16 + lane], lane, cc);
    if (n_inj_cols && parked) {
        // the row's digest is parked in the node's slot (k_hash_rows_multi): lane 8 + k reads word k
        const uint32_t hh = next[ii * 8 + (lane & 7u)];
        x = coop_permute_regs(lane < 8 ? x : hh, lane, cc);
    } else if (n_inj_cols) {
        uint32_t h = 0;  // sponge over the injected row: lane k < 8 absorbs column j + k, lanes 8..15 are the capacity
        for (uint32_t j = 0; j < n_inj_cols; j += 8) {
            if (lane < 8 && j + lane < n_inj_cols) h = inj_cols[j + lane][ii];
            h = coop_permute_regs(h, lane, cc);
        }
        const uint32_t hh = dpp<ZK_ROR(8)>(h);  // lane 8 + k reads the digest word k
        x = coop_permute_regs(lane < 8 ? x : hh, lane, cc);
    }
    if (i < n_next && lane < 8) next[i * 8 + lane] = x;
}

// Up to FIVE plain layers in one launch (round 4: the flow's profile had 8 100 cooperative layer launches of 11.6 us for 65 proofs -- a
// few microseconds of permutations behind a launch each).  A workgroup produces 16 nodes of the first layer (one 16-lane row per
// node), keeps them in LDS, and goes on with the 8 / 4 / 2 / 1 parents that depend on nothing else: a permutation latency per layer, no
// launch in between.  Every layer is written to the tree's digest store as before.
struct CoopMulti {
    uint64_t out_off[5];   // word offsets of the produced layers in the digest store
    uint32_t m;            // layers to produce (1..5)
};
__global__ __launch_bounds__(256) void k_compress_coop_multi(const uint32_t* prev, uint32_t* digests, size_t n_next, CoopMulti cm) {   // (prev points into digests: no __restrict__)
    __shared__ uint32_t buf[2][16 * 8];
    const unsigned lane = threadIdx.x & 15u, row = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + row;
    const size_t ii = i < n_next ? i : 0;  // all 16 lanes of every row stay active through the DPP permutation
    const CoopConsts cc = coop_load_consts(lane);
    uint32_t x = coop_permute_regs(prev[ii * 16 + lane], lane, cc);
    // every layer's word stays in a register of its own and goes to the digest store at the END of the kernel, behind the last
    // permutation: the store's data register is never written again (the stale node of the guest flow)
    uint32_t keep[5];
    keep[0] = x;
    if (lane < 8) buf[0][row * 8 + lane] = x;
    unsigned cur = 0, rows = 16;
#pragma unroll
    for (uint32_t j = 1; j < 5; j++) {
        if (j >= cm.m) break;
        zk_syncthreads();
        rows >>= 1;
        const unsigned r = row < rows ? row : 0;
        x = coop_permute_regs(buf[cur][16 * r + lane], lane, cc);   // the two children's digests are neighbours in the buffer
        keep[j] = x;
        if (row < rows && lane < 8) buf[cur ^ 1][row * 8 + lane] = x;
        cur ^= 1;
    }
    if (lane >= 8) return;
#pragma unroll
    for (uint32_t j = 0; j < 5; j++) {
        if (j >= cm.m) break;
        const size_t o = ((size_t)blockIdx.x * 16 >> j) + row;
        if (row < (16u >> j) && o < (n_next >> j)) digests[cm.out_off[j] + o * 8 + lane] = keep[j];
    }
}

// Up to FIVE layers in one launch, plain or with injected rows (round 5: a segment proof's main, permutation and quotient trees carry chips
// of a dozen heights, so nearly every small layer has injected rows and was a launch of k_compress_layer_coop of its own -- 127 such
// launches per proof of the guest flow, a permutation latency behind a launch each).  A workgroup of 16 rows produces 16 nodes of the
// first layer (node = compress(left, right), then compress(node, sponge(row of the matrices injected at this height)) where the layer has
// any), hands them on through the LDS and goes on with the 8 / 4 / 2 / 1 parents.  With n_next <= 16 the one workgroup ends at the root.
// The loop is rolled: zk_syncthreads() states the LDS wait in front of every barrier (docs/stale_node.md).
struct CoopFused {
    uint64_t out_off[5];   // word offsets of the produced layers in the digest store
    uint32_t inj_off[5];   // first injected column of the layer in the column-pointer table
    uint32_t inj_cnt[5];   // injected columns of the layer (0: a plain layer)
    uint32_t parked[5];    // the rows of the layer's matrices are hashed already: the digest waits in the node's slot (k_hash_rows_multi)
    uint32_t m;            // layers to produce (1..5)
};
__global__ __launch_bounds__(256) void k_compress_coop_fused(const uint32_t* prev, uint32_t* digests, size_t n_next, const uint32_t* const* __restrict__ cols, CoopFused cf) {
    __shared__ uint32_t buf[2][16 * 8];
    const unsigned lane = threadIdx.x & 15u, row = threadIdx.x >> 4;
    const CoopConsts cc = coop_load_consts(lane);
    size_t node = (size_t)blockIdx.x * 16 + row;
    unsigned rows = 16;
    uint32_t in = prev[(node < n_next ? node : 0) * 16 + lane];   // all 16 lanes of every row stay active through the DPP permutation
#pragma unroll 1
    for (uint32_t j = 0; j < cf.m; j++) {
        const size_t n_j = n_next >> j;
        const bool active = row < rows && node < n_j;
        const size_t idx = active ? node : 0;
        uint32_t x = coop_permute_regs(in, lane, cc);
        const uint32_t n_inj = cf.inj_cnt[j];
        if (n_inj && cf.parked[j]) {
            const uint32_t hh = digests[cf.out_off[j] + idx * 8 + (lane & 7u)];   // lane 8 + k reads word k of the row's digest
            x = coop_permute_regs(lane < 8 ? x : hh, lane, cc);
        } else if (n_inj) {
            const uint32_t* const* ic = cols + cf.inj_off[j];
            uint32_t h = 0;   // sponge over the injected row: lane k < 8 absorbs column c + k, lanes 8..15 are the capacity
            for (uint32_t c = 0; c < n_inj; c += 8) {
                if (lane < 8 && c + lane < n_inj) h = ic[c + lane][idx];
                h = coop_permute_regs(h, lane, cc);
            }
            const uint32_t hh = dpp<ZK_ROR(8)>(h);   // lane 8 + k reads the digest word k
            x = coop_permute_regs(lane < 8 ? x : hh, lane, cc);
        }
        if (active && lane < 8) digests[cf.out_off[j] + node * 8 + lane] = x;
        if (j + 1 == cf.m) break;
        if (row < rows && lane < 8) buf[j & 1][row * 8 + lane] = x;
        zk_syncthreads();
        rows >>= 1;
        node = ((size_t)blockIdx.x * 16 >> (j + 1)) + row;
        in = buf[j & 1][16 * (row < rows ? row : 0) + lane];   // the two children's digests are neighbours in the buffer
    }
}

// Top of a tree in ONE launch: from a layer of <= 512 nodes down to the root.  Each 16-lane row
// computes one compression cooperatively (poseidon2_coop.hpp), the working layer lives in LDS,
// every produced layer is also written to the tree's digest store for later openings.
__global__ __launch_bounds__(1024) void k_compress_top(uint32_t* __restrict__ digests, unsigned lh, unsigned l0) {
    __shared__ uint32_t buf[2][512 * 8];
    const unsigned tid = threadIdx.x, lane = tid & 15u, grp = tid >> 4;  // 64 groups
    auto layer_off = [&](unsigned l) -> size_t { return ((size_t)2 << lh) - ((size_t)2 << (lh - l)); };
    unsigned n = 1u << (lh - l0);
    const CoopConsts cc = coop_load_consts(lane);
    for (unsigned e = tid; e < n * 8; e += 1024) buf[0][e] = digests[layer_off(l0) * 8 + e];
    zk_syncthreads();
    unsigned cur = 0;
    // layers of more than 64 nodes (top_max_log 7, 8): several nodes per row, stored as they are made; from 64 nodes down a row makes
    // one node per layer, keeps it in a register of its own and stores it at the END of the kernel (as k_compress_coop_multi does)
    uint32_t keep[7];
    unsigned l = l0 + 1;
    for (; l <= lh && (1u << (lh - l)) > 64u; l++) {
        const unsigned n_next = 1u << (lh - l);
        uint32_t* out = digests + layer_off(l) * 8;
        for (unsigned i = grp; i < n_next; i += 64) {
            uint32_t x = buf[cur][16 * i + lane];
            x = coop_permute_regs(x, lane, cc);
            if (lane < 8) {
                buf[cur ^ 1][8 * i + lane] = x;
                out[8 * i + lane] = x;
            }
        }
        zk_syncthreads();
        cur ^= 1;
    }
    const unsigned l1 = l;   // first layer of <= 64 nodes
#pragma unroll
    for (unsigned k = 0; k < 7; k++) {
        if (l1 + k > lh) break;
        const unsigned n_next = 1u << (lh - l1 - k);
        // all 16 lanes of a row stay active through the DPP permutation
        const unsigned ii = grp < n_next ? grp : 0;
        const uint32_t x = coop_permute_regs(buf[cur][16 * ii + lane], lane, cc);
        keep[k] = x;
        if (grp < n_next && lane < 8) buf[cur ^ 1][8 * grp + lane] = x;
        zk_syncthreads();
        cur ^= 1;
    }
    if (lane >= 8) return;
#pragma unroll
    for (unsigned k = 0; k < 7; k++) {
        if (l1 + k > lh) break;
        if (grp < (1u << (lh - l1 - k))) digests[layer_off(l1 + k) * 8 + 8 * grp + lane] = keep[k];
    }
}

// The round-4 bodies of the two fused kernels ("store a layer's word, then compute on"), kept behind zkhip_config.tree_store_early for the
// A/B of the stale node (docs/stale_node.md): TEST ONLY -- the form that stored a wrong node in ~3 % of guest-flow runs.  They are compiled
// only under -DZKHIP_TEST_KERNELS, i.e. into libzkhip_test.so (csrc/Makefile), which the stress tests load; libzkhip.so does not hold them and
// refuses zkhip_config.tree_store_early.
#ifdef ZKHIP_TEST_KERNELS
__global__ __launch_bounds__(256) void k_compress_coop_multi_early(const uint32_t* __restrict__ prev, uint32_t* __restrict__ digests, size_t n_next, CoopMulti cm) {
    __shared__ uint32_t buf[2][16 * 8];
    const unsigned lane = threadIdx.x & 15u, row = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + row;
    const size_t ii = i < n_next ? i : 0;
    const CoopConsts cc = coop_load_consts(lane);
    uint32_t x = coop_permute_regs(prev[ii * 16 + lane], lane, cc);
    if (lane < 8) {
        if (i < n_next) digests[cm.out_off[0] + i * 8 + lane] = x;
        buf[0][row * 8 + lane] = x;
    }
    unsigned cur = 0, rows = 16;
    for (uint32_t j = 1; j < cm.m; j++) {
        __syncthreads();
        rows >>= 1;
        const unsigned r = row < rows ? row : 0;
        x = coop_permute_regs(buf[cur][16 * r + lane], lane, cc);
        const size_t o = ((size_t)blockIdx.x * 16 >> j) + row;
        if (row < rows && lane < 8) {
            if (o < (n_next >> j)) digests[cm.out_off[j] + o * 8 + lane] = x;
            buf[cur ^ 1][row * 8 + lane] = x;
        }
        cur ^= 1;
    }
}
__global__ __launch_bounds__(1024) void k_compress_top_early(uint32_t* __restrict__ digests, unsigned lh, unsigned l0) {
    __shared__ uint32_t buf[2][512 * 8];
    const unsigned tid = threadIdx.x, lane = tid & 15u, grp = tid >> 4;  // 64 groups
    auto layer_off = [&](unsigned l) -> size_t { return ((size_t)2 << lh) - ((size_t)2 << (lh - l)); };
    unsigned n = 1u << (lh - l0);
    const CoopConsts cc = coop_load_consts(lane);
    for (unsigned e = tid; e < n * 8; e += 1024) buf[0][e] = digests[layer_off(l0) * 8 + e];
    __syncthreads();
    unsigned cur = 0;
    for (unsigned l = l0 + 1; l <= lh; l++) {
        const unsigned n_next = 1u << (lh - l);
        uint32_t* out = digests + layer_off(l) * 8;
        for (unsigned i = grp; i < ((n_next + 63u) & ~63u); i += 64) {
            const unsigned ii = i < n_next ? i : 0;
            uint32_t x = buf[cur][16 * ii + lane];
            x = coop_permute_regs(x, lane, cc);
            if (i < n_next && lane < 8) {
                buf[cur ^ 1][8 * i + lane] = x;
                out[8 * i + lane] = x;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}
#endif  // ZKHIP_TEST_KERNELS

// Diagnosis (zkhip_config.self_check): every node of a plain layer against the compression of its children, one lane per node, through the
// plain (non-cooperative) permutation.  report[0] = mismatching nodes, report[1] = the smallest (layer << 24 | index) among them.
__global__ __launch_bounds__(256) void k_check_layer(const uint32_t* __restrict__ prev, const uint32_t* __restrict__ next, size_t n_next, uint32_t layer,
                                                     uint32_t* report) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_next) return;
    uint32_t s[16];
    for (int q = 0; q < 16; q++) s[q] = prev[i * 16 + q];
    poseidon2_permute_rolled(s);
    bool same = true;
    for (int q = 0; q < 8; q++) same = same && s[q] == next[i * 8 + q];
    if (!same) atomicAdd(&report[0], 1u), atomicMin(&report[1], (layer << 24) | (uint32_t)(i & 0xffffffu));
}
int merkle_check_tree(zkhip_ctx* ctx, const zkhip_tree* t, uint32_t* d_report) {
    const unsigned lh = t->log_height;
    for (unsigned l = 1; l <= lh; l++) {
        const unsigned level = lh - l;
        if (level < t->level_cnt.size() && t->level_cnt[level]) continue;   // (a layer with injected rows: not checked here)
        const size_t cnt = (size_t)1 << level;
        hipLaunchKernelGGL(k_check_layer, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, t->d_digests + t->layer_off[l - 1] * 8,
                           t->d_digests + t->layer_off[l] * 8, cnt, l, d_report);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int permute_batch(zkhip_ctx* ctx, uint32_t* d_states, size_t n) {
    if (n == 0) return ZKHIP_OK;
    KernelScope ks(ctx, "poseidon2_permute_batch");
    hipLaunchKernelGGL(k_permute_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_states, n);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

size_t merkle_digest_count(unsigned log_height) { return ((size_t)2 << log_height) - 1; }

// Layout of the per-tree pointer store (device):
//   [ all columns in caller order : total_width pointers ]
//   [ per level L = log_height .. 0 : pointers of the columns of matrices with that height ]
//   then u32 shifts[total_width] (log_height - mat.log_height) for the open kernel
// Plans a tree over `mats`: builds and uploads the column-pointer tables and reserves the digest
// store.  `d_digests` may be caller-provided workspace (then the tree does not own it).
int merkle_plan(zkhip_ctx* ctx, const zkhip_matrix* mats, size_t n_mats, uint32_t* d_digests, zkhip_tree** out) {
    if (n_mats == 0) return set_error(ctx, ZKHIP_ERR_INVALID, "merkle: no matrices");
    zkhip_tree* t = new zkhip_tree();
    t->mats.assign(mats, mats + n_mats);
    unsigned lh = 0;
    size_t total_w = 0;
    for (size_t m = 0; m < n_mats; m++) {
        lh = std::max(lh, mats[m].log_height);
        total_w += mats[m].width;
        if (mats[m].stride < ((size_t)1 << mats[m].log_height)) {
            delete t;
            return set_error(ctx, ZKHIP_ERR_INVALID, "merkle: stride < height");
        }
    }
    t->log_height = lh;
    t->total_width = total_w;
    std::vector<const uint32_t*> ptrs;
    std::vector<uint32_t> shifts;
    for (size_t m = 0; m < n_mats; m++)
        for (size_t c = 0; c < mats[m].width; c++) {
            ptrs.push_back(mats[m].data + c * mats[m].stride);
            shifts.push_back(lh - mats[m].log_height);
        }
    t->level_off.assign(lh + 1, 0);
    t->level_cnt.assign(lh + 1, 0);
    for (unsigned level = 0; level <= lh; level++) {
        t->level_off[level] = ptrs.size();
        size_t cnt = 0;
        for (size_t m = 0; m < n_mats; m++)
            if (mats[m].log_height == level)
                for (size_t c = 0; c < mats[m].width; c++, cnt++) ptrs.push_back(mats[m].data + c * mats[m].stride);
        t->level_cnt[level] = cnt;
    }
    size_t ptr_bytes = ptrs.size() * sizeof(void*), shift_bytes = shifts.size() * sizeof(uint32_t);
    if (hipMalloc(&t->d_colptrs, ptr_bytes + shift_bytes + 16) != hipSuccess) {
        delete t;
        return set_error(ctx, ZKHIP_ERR_NOMEM, "merkle: pointer table alloc");
    }
    t->shifts_off = ptr_bytes;
    hipError_t e1 = hipMemcpyAsync(t->d_colptrs, ptrs.data(), ptr_bytes, hipMemcpyHostToDevice, ctx->stream);
    hipError_t e2 = shift_bytes ? hipMemcpyAsync((char*)t->d_colptrs + ptr_bytes, shifts.data(), shift_bytes,
                                                 hipMemcpyHostToDevice, ctx->stream)
                                : hipSuccess;
    if (e1 == hipSuccess && e2 == hipSuccess) e1 = hipStreamSynchronize(ctx->stream);  // host vectors are temporaries
    if (e1 != hipSuccess || e2 != hipSuccess) {
        (void)hipFree(t->d_colptrs);
        delete t;
        return set_error(ctx, ZKHIP_ERR_HIP, "merkle: pointer table upload");
    }
    if (d_digests) {
        t->d_digests = d_digests;
        t->owns_digests = false;
    } else if (hipMalloc(&t->d_digests, merkle_digest_count(lh) * 8 * sizeof(uint32_t)) != hipSuccess) {
        (void)hipFree(t->d_colptrs);
        delete t;
        return set_error(ctx, ZKHIP_ERR_NOMEM, "merkle: digest alloc");
    }
    t->layer_off.resize(lh + 1);
    size_t off = 0;
    for (unsigned l = 0; l <= lh; l++) {
        t->layer_off[l] = off;
        off += (size_t)1 << (lh - l);
    }
    *out = t;
    return ZKHIP_OK;
}

// Plans a tree whose leaf digests are produced by the caller (FRI layers): no matrices.
int merkle_plan_leaves(zkhip_ctx* ctx, unsigned log_height, uint32_t* d_digests, zkhip_tree** out) {
    zkhip_tree* t = new zkhip_tree();
    t->log_height = log_height;
    t->level_off.assign(log_height + 1, 0);
    t->level_cnt.assign(log_height + 1, 0);
    if (d_digests) {
        t->d_digests = d_digests;
        t->owns_digests = false;
    } else if (hipMalloc(&t->d_digests, merkle_digest_count(log_height) * 8 * sizeof(uint32_t)) != hipSuccess) {
        delete t;
        return set_error(ctx, ZKHIP_ERR_NOMEM, "merkle: digest alloc");
    }
    t->layer_off.resize(log_height + 1);
    size_t off = 0;
    for (unsigned l = 0; l <= log_height; l++) {
        t->layer_off[l] = off;
        off += (size_t)1 << (log_height - l);
    }
    *out = t;
    return ZKHIP_OK;
}

int merkle_leaves_part(zkhip_ctx* ctx, zkhip_tree* t, size_t col_begin, size_t col_end, bool first, bool last, uint32_t* d_state) {
    const unsigned lh = t->log_height;
    const size_t n = (size_t)1 << lh;
    if (col_begin % 8 || col_end > t->level_cnt[lh] || col_begin >= col_end || (!last && (col_end % 8))) return ZKHIP_ERR_INVALID;
    const uint32_t* const* d_ptrs = (const uint32_t* const*)t->d_colptrs;
    KernelScope ks(ctx, "poseidon2_hash_rows");
    hipLaunchKernelGGL(k_hash_rows_part, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_ptrs + t->level_off[lh] + col_begin, (uint32_t)(col_end - col_begin), n, d_state, first ? 1 : 0,
                       last ? 1 : 0, t->d_digests);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

// Launches the hashing kernels of a planned tree on the ctx stream (asynchronous).
int merkle_build(zkhip_ctx* ctx, zkhip_tree* t, bool leaves_ready) {
    const unsigned lh = t->log_height;
    const uint32_t* const* d_ptrs = (const uint32_t* const*)t->d_colptrs;
    size_t n = (size_t)1 << lh;
    // the levels whose rows are hashed in bulk: the leaf level and every level with injected matrices (zkhip_config.rows_in_bulk; 0: only the
    // levels above the cooperative layer form's size, at most eight, as rounds 4 - 5)
    const unsigned coop_inj_max_log = ctx->cfg.coop_inj_max_log;
    const bool all_levels = ctx->cfg.rows_in_bulk != 0;
    std::vector<char> bulk(lh + 1, 0);
    struct Entry {
        unsigned level;
        size_t cols, perms;   // permutations per row
        bool coop;
    };
    std::vector<Entry> entries;
    auto want_level = [&](unsigned level) {
        if (entries.size() == (all_levels ? ZK_MAX_ROW_LEVELS : 8u)) return;
        entries.push_back({level, t->level_cnt[level], (t->level_cnt[level] + 7) / 8, false});
        bulk[level] = 1;
    };
    if (!leaves_ready) want_level(lh);
    for (unsigned level = lh; level-- > 0;)
        if (t->level_cnt[level] && (all_levels || ((size_t)1 << level) > ((size_t)1 << coop_inj_max_log))) want_level(level);
    RowHashLevels L{};
    if (!entries.empty()) {
        // A level's chain: `perms` permutations one after the other on a lane (~9 us each when the wave is served first), a third of that in
        // the cooperative form.  The launch as a whole: all permutations at the chip's rate (~7500 per us).  A level of at most 2^rows_coop_max_log
        // rows whose one-lane chain would outlast the launch takes the cooperative form; levels are ordered by the length of their chains, and
        // every chain of more than 64 permutations runs at wave priority 3.
        double total = 0;
        for (const Entry& e : entries) total += (double)e.perms * (double)((size_t)1 << e.level);
        const double launch_us = total / 7500.0;
        if (all_levels)
            for (Entry& e : entries)
                e.coop = e.level <= ctx->cfg.rows_coop_max_log && e.perms >= 16 && (double)e.perms * 9.0 > std::max(launch_us, 300.0);
        auto chain_us = [](const Entry& e) { return (double)e.perms * (e.coop ? 3.0 : 9.0); };
        if (all_levels) std::stable_sort(entries.begin(), entries.end(), [&](const Entry& a, const Entry& b) { return chain_us(a) > chain_us(b); });
        for (const Entry& e : entries) {
            const uint32_t k = L.n_levels++;
            const size_t rows = (size_t)1 << e.level, per_block = e.coop ? 16 : 256;
            L.first_block[k + 1] = L.first_block[k] + (uint32_t)((rows + per_block - 1) / per_block);
            L.col_off[k] = (uint32_t)t->level_off[e.level], L.n_cols[k] = (uint32_t)e.cols;
            L.log_rows[k] = (uint8_t)e.level, L.out_off[k] = (uint64_t)t->layer_off[lh - e.level] * 8;
            L.mode[k] = (uint8_t)((e.coop ? 1u : 0u) | (all_levels && e.perms > 64 ? 2u : 0u));
        }
    }
    if (L.n_levels == 1 && !leaves_ready) {
        KernelScope ks(ctx, "poseidon2_hash_rows");
        const unsigned hb = n >= ((size_t)1 << 20) ? std::max(64u, std::min(768u, ctx->cfg.hash_block)) : 256u;   // (small trees: more workgroups matter more)
        hipLaunchKernelGGL(k_hash_rows, dim3((unsigned)((n + hb - 1) / hb)), dim3(hb), 0, ctx->stream,
                           d_ptrs + t->level_off[lh], (uint32_t)t->level_cnt[lh], n, t->d_digests);
    } else if (L.n_levels) {
        KernelScope ks(ctx, "poseidon2_hash_rows");
        hipLaunchKernelGGL(k_hash_rows_multi, dim3(L.first_block[L.n_levels]), dim3(256), 0, ctx->stream, d_ptrs, L, t->d_digests);
    }
    // levels at which no shorter matrix is injected, counted from the root
    unsigned clean_top = 0;
    while (clean_top < lh && t->level_cnt[clean_top] == 0) clean_top++;  // levels 0..clean_top-1 are plain
    for (unsigned l = 1; l <= lh; l++) {
        unsigned level = lh - l;
        size_t cnt = (size_t)1 << level;
        // (the top of the tree in one workgroup: from 2^top_max_log nodes down; above that a layer is a launch of its own -- a layer of 256
        // nodes is four sequential passes of the one workgroup, a permutation latency each)
        if (cnt <= ((size_t)1 << std::min(ctx->cfg.top_max_log, 8u)) && level < clean_top) {
            // this and all remaining levels in one launch
            KernelScope ks(ctx, "poseidon2_compress_top");
#ifdef ZKHIP_TEST_KERNELS
            if (ctx->cfg.tree_store_early) hipLaunchKernelGGL(k_compress_top_early, dim3(1), dim3(1024), 0, ctx->stream, t->d_digests, lh, l - 1);
            else
#endif
                hipLaunchKernelGGL(k_compress_top, dim3(1), dim3(1024), 0, ctx->stream, t->d_digests, lh, l - 1);
            break;
        }
        KernelScope ks(ctx, "poseidon2_compress_layer");
        // (ZKHIP_COOP_MAX_LOG / ZKHIP_COOP_INJ_MAX_LOG: the largest layer, plain / with injected rows, that takes the cooperative form)
        const unsigned coop_max = ctx->cfg.coop_max_log, coop_inj_max = ctx->cfg.coop_inj_max_log;
        if (ctx->cfg.coop_fused && cnt <= ((size_t)1 << std::min(coop_max, coop_inj_max))) {
            // this layer and up to four above it in one launch, injected rows included; a group stops where the one-workgroup top kernel
            // takes over (a clean top), and the groups are cut so that the last one ends at the root
            const unsigned top_log = std::min(ctx->cfg.top_max_log, 8u);
            unsigned avail = 0;   // layers from this one up that the fused form may take
            while (avail <= level) {
                const unsigned lv = level - avail;
                if (avail > 0 && ((size_t)1 << lv) <= ((size_t)1 << top_log) && lv < clean_top) break;   // the top kernel's from here
                avail++;
            }
            unsigned m = avail % 5 ? avail % 5 : 5;
            bool any_inj = false;
            for (unsigned q = 0; q < m; q++) any_inj = any_inj || t->level_cnt[level - q] != 0;
            if (any_inj || m >= 2 || level == 0) {
                CoopFused cf{};
                cf.m = m;
                for (unsigned q = 0; q < m; q++) {
                    cf.out_off[q] = (uint64_t)t->layer_off[l + q] * 8;
                    cf.inj_off[q] = (uint32_t)t->level_off[level - q], cf.inj_cnt[q] = (uint32_t)t->level_cnt[level - q];
                    cf.parked[q] = bulk[level - q] ? 1u : 0u;
                }
                hipLaunchKernelGGL(k_compress_coop_fused, dim3((unsigned)((cnt + 15) / 16)), dim3(256), 0, ctx->stream, t->d_digests + t->layer_off[l - 1] * 8, t->d_digests, cnt,
                                   d_ptrs, cf);
                l += m - 1;
                continue;
            }
        }
        if (t->level_cnt[level] == 0 && cnt <= ((size_t)1 << coop_max) && cnt >= 16) {
            // this plain layer and the plain layers above it that the top kernel does not take, up to five, in one launch
            const unsigned top_log = std::min(ctx->cfg.top_max_log, 8u);
            CoopMulti cm{};
            unsigned m = 0;
            while (m < 5 && level >= m) {
                const unsigned lv = level - m;
                if (t->level_cnt[lv] != 0) break;
                if (m > 0 && ((size_t)1 << lv) <= ((size_t)1 << top_log) && lv < clean_top) break;   // the top kernel's from here
                cm.out_off[m] = (uint64_t)t->layer_off[l + m] * 8;
                m++;
            }
            if (m >= 2) {
                cm.m = m;
#ifdef ZKHIP_TEST_KERNELS
                if (ctx->cfg.tree_store_early)
                    hipLaunchKernelGGL(k_compress_coop_multi_early, dim3((unsigned)(cnt / 16)), dim3(256), 0, ctx->stream, t->d_digests + t->layer_off[l - 1] * 8, t->d_digests, cnt, cm);
                else
#endif
                    hipLaunchKernelGGL(k_compress_coop_multi, dim3((unsigned)(cnt / 16)), dim3(256), 0, ctx->stream, t->d_digests + t->layer_off[l - 1] * 8, t->d_digests, cnt, cm);
                l += m - 1;
                continue;
            }
        }
        if (cnt <= ((size_t)1 << (t->level_cnt[level] ? coop_inj_max : coop_max)))
            hipLaunchKernelGGL(k_compress_layer_coop, dim3((unsigned)((cnt + 15) / 16)), dim3(256), 0, ctx->stream,
                               t->d_digests + t->layer_off[l - 1] * 8, t->d_digests + t->layer_off[l] * 8, cnt,
                               d_ptrs ? d_ptrs + t->level_off[level] : nullptr, (uint32_t)t->level_cnt[level], bulk[level] ? 1 : 0);
        else if (t->level_cnt[level] && bulk[level])
            hipLaunchKernelGGL(k_compress_layer_inj2, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                               t->d_digests + t->layer_off[l - 1] * 8, t->d_digests + t->layer_off[l] * 8, cnt);
        else if (t->level_cnt[level])
            hipLaunchKernelGGL(k_compress_layer_inj, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                               t->d_digests + t->layer_off[l - 1] * 8, t->d_digests + t->layer_off[l] * 8, cnt,
                               d_ptrs + t->level_off[level], (uint32_t)t->level_cnt[level]);
        else
            hipLaunchKernelGGL(k_compress_layer, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                               t->d_digests + t->layer_off[l - 1] * 8, t->d_digests + t->layer_off[l] * 8, cnt);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int merkle_commit(zkhip_ctx* ctx, const zkhip_matrix* mats, size_t n_mats, zkhip_tree** out) {
    zkhip_tree* t = nullptr;
    ZK_TRY(merkle_plan(ctx, mats, n_mats, nullptr, &t));
    int rc = merkle_build(ctx, t, false);
    if (rc != ZKHIP_OK) {
        (void)hipFree(t->d_colptrs);
        if (t->owns_digests) (void)hipFree(t->d_digests);
        delete t;
        return rc;
    }
    *out = t;
    return ZKHIP_OK;
}

// One block per opened index: rows of every matrix (canonical) then sibling digests bottom-up.
__global__ void k_merkle_open(const uint32_t* const* __restrict__ cols, const uint32_t* __restrict__ shifts,
                              uint32_t total_w, const uint32_t* __restrict__ digests, unsigned log_height,
                              const uint32_t* __restrict__ indices, unsigned index_shift,
                              uint32_t* __restrict__ out, size_t pitch) {
    const size_t q = blockIdx.x;
    const size_t index = indices[q] >> index_shift;
    uint32_t* o = out + q * pitch;
    for (uint32_t c = threadIdx.x; c < total_w; c += blockDim.x) o[c] = from_monty(cols[c][index >> shifts[c]]);
    size_t layer_off = 0;
    for (unsigned l = 0; l < log_height; l++) {
        size_t sib = (index >> l) ^ 1;
        if (threadIdx.x < 8) o[total_w + 8 * l + threadIdx.x] = from_monty(digests[(layer_off + sib) * 8 + threadIdx.x]);
        layer_off += (size_t)1 << (log_height - l);
    }
}

int merkle_open_device(zkhip_ctx* ctx, const zkhip_tree* t, const uint32_t* d_indices, unsigned index_shift,
                       size_t n, uint32_t* d_out, size_t out_pitch_words) {
    if (n == 0) return ZKHIP_OK;
    KernelScope ks(ctx, "merkle_open");
    const uint32_t* const* d_ptrs = (const uint32_t* const*)t->d_colptrs;
    const uint32_t* d_shifts = (const uint32_t*)((const char*)t->d_colptrs + t->shifts_off);
    hipLaunchKernelGGL(k_merkle_open, dim3((unsigned)n), dim3(128), 0, ctx->stream, d_ptrs, d_shifts,
                       (uint32_t)t->total_width, t->d_digests, t->log_height, d_indices, index_shift, d_out,
                       out_pitch_words);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace zk
