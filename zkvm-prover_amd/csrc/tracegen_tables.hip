// tracegen_tables.hip -- device-side trace generation for three more periphery chips of the
// reference's chunk circuit (crates/circuits/chunk-circuit/openvm.toml:8-59: rv32i/rv32m/bigint/... all lean on them):
//
//  * range-tuple checker  (OpenVM RangeTupleCheckerChip<2>, sizes [256, 8192] under `[app_vm_config.rv32m]
//    range_tuple_checker_sizes`, openvm.toml): the whole trace is ONE multiplicity column over the table of all tuples
//    (x, y), row x * size_y + y -- counted with atomics from the requesting columns where they lie;
//  * bitwise-operation lookup (OpenVM BitwiseOperationLookupChip<8>): table of all (x, y), x, y < 2^bits, row
//    (x << bits) + y; TWO multiplicity columns: range requests (x, y) and XOR requests (x, y, x ^ y);
//  * volatile memory boundary (OpenVM VolatileBoundaryChip): one row per touched address, SORTED by (address space,
//    pointer), carrying initial / final data and the final timestamp, plus the sortedness witness (the gap to the
//    next key split into 16-bit limbs for the range checker).  The sort is rocPRIM's device radix sort (through the
//    hipCUB front end shipped with ROCm), the fill one coalesced pass.
//
// The un-vendored OpenVM crates hold the reference generators (`generate_proving_ctx` of each chip's GPU twin,
// AGENTS.md:183-187); the layouts below restate the published chip structure and are matched cell for cell by
// oracle/tracegen.c (tests/test_gpu_tracegen_tables.py), the AIRs are in zkvm-prover_amd/air.py.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>

#include "lds_barrier.hpp"
#include "../../include/zkhip.h"
#include "hist.hpp"
#include "babybear.hpp"
#include "zkhip_internal.hpp"

namespace zk {
namespace {

__global__ void k_tab_repr(uint32_t* c, size_t n, int to_m) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = to_m ? to_monty(c[i] % P) : from_monty(c[i]);
}

// hist[x * size_y + y] += 1 for every request; x / y Montgomery words of the requesting trace columns
__global__ __launch_bounds__(256) void k_tuple_counts(const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys, size_t n,
                                                      uint32_t size_x, uint32_t size_y, uint32_t* __restrict__ hist,
                                                      uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk[HOT_SLOTS], hc[HOT_SLOTS];   // hot entries counted in LDS, merged once per workgroup (csrc/hist.hpp)
    hot_init(hk, hc);
    uint32_t n_bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t x = from_monty(xs[i]), y = from_monty(ys[i]);
        if (x >= size_x || y >= size_y) {
            n_bad++;
            continue;
        }
        hot_add(hk, hc, hist, (uint32_t)((size_t)x * size_y + y));
    }
    if (n_bad) atomicAdd(bad, n_bad);
    hot_flush(hk, hc, hist);
}

// column 0 (range requests) / column 1 (xor requests) of the table of all (x, y): row (x << bits) + y
__global__ __launch_bounds__(256) void k_bitwise_counts(const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
                                                        const uint32_t* __restrict__ ops, size_t n, unsigned bits,
                                                        uint32_t* __restrict__ hist, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk[HOT_SLOTS], hc[HOT_SLOTS];   // hot entries counted in LDS, merged once per workgroup (csrc/hist.hpp)
    hot_init(hk, hc);
    uint32_t n_bad = 0;
    const uint32_t lim = 1u << bits;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t x = from_monty(xs[i]), y = from_monty(ys[i]), op = from_monty(ops[i]);
        if (x >= lim || y >= lim || op > 1) {
            n_bad++;
            continue;
        }
        hot_add(hk, hc, hist, (uint32_t)(((size_t)op << (2 * bits)) + ((size_t)x << bits) + y));
    }
    if (n_bad) atomicAdd(bad, n_bad);
    hot_flush(hk, hc, hist);
}

__global__ void k_boundary_keys(const uint32_t* __restrict__ as, const uint32_t* __restrict__ ptr, size_t n, uint64_t* keys,
                                uint32_t* idx, uint32_t* bad, unsigned as_bits, unsigned ptr_bits) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if ((as_bits < 32 && (as[i] >> as_bits)) || (ptr_bits < 32 && (ptr[i] >> ptr_bits))) atomicAdd(bad, 1u);
    keys[i] = ((uint64_t)as[i] << 32) | ptr[i];
    idx[i] = (uint32_t)i;
}
// row r < n: record idx[r] of the sorted order; gap to the next key minus one in 16-bit limbs (0 on the last valid row);
// rows >= n are zero (is_valid = 0).  Columns: as, ptr, initial, final, final_ts, is_valid, gap_lo, gap_hi.
__global__ __launch_bounds__(256) void k_boundary_fill(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx,
                                                       const uint32_t* __restrict__ init, const uint32_t* __restrict__ fin,
                                                       const uint32_t* __restrict__ ts, size_t n, size_t N, unsigned ptr_bits,
                                                       uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t c[ZKHIP_MEMORY_BOUNDARY_WIDTH] = {};
    if (r < n) {
        const uint64_t k = keys[r];
        const uint32_t j = idx[r];
        c[0] = to_monty((uint32_t)(k >> 32)), c[1] = to_monty((uint32_t)k);
        c[2] = init[j], c[3] = fin[j];
        c[4] = to_monty(ts[j] % P);
        c[5] = MONTY_ONE;
        if (r + 1 < n) {
            const uint64_t kn = keys[r + 1];
            if (kn == k) atomicAdd(bad, 1u);  // an address listed twice
            // the AIR's key: as * 2^ptr_bits + ptr
            const uint64_t a = ((k >> 32) << ptr_bits) + (uint32_t)k, b = ((kn >> 32) << ptr_bits) + (uint32_t)kn;
            const uint64_t gap = b - a - 1;
            c[6] = to_monty((uint32_t)(gap & 0xffffu));
            c[7] = to_monty((uint32_t)((gap >> 16) % P));
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_MEMORY_BOUNDARY_WIDTH; q++) trace[(size_t)q * N + r] = c[q];
}

int finish_counts(zkhip_ctx* ctx, void* flag, const char* what) { return tracegen_finish(ctx, flag, std::string(what) + " (requests outside the table)"); }

}  // namespace
}  // namespace zk

using namespace zk;

extern "C" int zkhip_range_tuple_counts_tracegen(zkhip_ctx* ctx, const uint32_t* d_x, const uint32_t* d_y, size_t n, uint32_t size_x,
                                                 uint32_t size_y, uint32_t* d_counts, int accumulate) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_counts || (n && (!d_x || !d_y)) || size_x == 0 || size_y == 0) return ZKHIP_ERR_INVALID;
    const size_t T = (size_t)size_x * size_y;
    if (T > ((size_t)1 << 27) || (T & (T - 1))) return set_error(ctx, ZKHIP_ERR_INVALID, "range tuple table: size_x * size_y must be a power of two <= 2^27");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "range_tuple_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!accumulate) ZK_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, T * 4, ctx->stream));
    else if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 0);
    if (n) {
        const unsigned blocks = (unsigned)std::min<size_t>((n + 256 * 8 - 1) / (256 * 8), HOT_MAX_BLOCKS);
        hipLaunchKernelGGL(k_tuple_counts, dim3(blocks), dim3(256), 0, ctx->stream, d_x, d_y, n, size_x, size_y, d_counts, (uint32_t*)flag);
    }
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "range_tuple_counts_tracegen");
}

extern "C" int zkhip_bitwise_lookup_tracegen(zkhip_ctx* ctx, const uint32_t* d_x, const uint32_t* d_y, const uint32_t* d_op, size_t n,
                                             unsigned num_bits, uint32_t* d_trace, int accumulate) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || (n && (!d_x || !d_y || !d_op)) || num_bits == 0 || num_bits > 12) return ZKHIP_ERR_INVALID;
    const size_t T = (size_t)2 << (2 * num_bits);  // two columns
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "bitwise_lookup_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!accumulate) ZK_HIP_CHECK(ctx, hipMemsetAsync(d_trace, 0, T * 4, ctx->stream));
    else if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_trace, T, 0);
    if (n) {
        const unsigned blocks = (unsigned)std::min<size_t>((n + 256 * 8 - 1) / (256 * 8), HOT_MAX_BLOCKS);
        hipLaunchKernelGGL(k_bitwise_counts, dim3(blocks), dim3(256), 0, ctx->stream, d_x, d_y, d_op, n, num_bits, d_trace, (uint32_t*)flag);
    }
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_trace, T, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "bitwise_lookup_tracegen");
}

extern "C" int zkhip_memory_boundary_tracegen(zkhip_ctx* ctx, const uint32_t* d_addr_space, const uint32_t* d_pointer,
                                              const uint32_t* d_initial, const uint32_t* d_final, const uint32_t* d_timestamp, size_t n,
                                              unsigned as_bits, unsigned pointer_bits, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27 || as_bits == 0 || pointer_bits == 0 || as_bits + pointer_bits > 44 || pointer_bits > 32 || as_bits > 32)
        return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N || (n && (!d_addr_space || !d_pointer || !d_initial || !d_final || !d_timestamp)))
        return set_error(ctx, ZKHIP_ERR_INVALID, "memory_boundary_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "memory_boundary_tracegen");
    uint64_t* keys_sorted = nullptr;
    uint32_t* idx_sorted = nullptr;
    if (n) {
        // scratch: keys in / out (u64), indices in / out (u32), radix-sort workspace
        size_t tmp_bytes = 0;
        ZK_HIP_CHECK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                                             (uint32_t*)nullptr, (int)n, 0, 64, ctx->stream));   // size query
        const size_t kb = (n * 8 + 255) & ~(size_t)255, ib = (n * 4 + 255) & ~(size_t)255;
        void* buf = nullptr;
        ZK_TRY(get_scratch(ctx, 3, 2 * kb + 2 * ib + tmp_bytes + 256, &buf));
        uint64_t* keys = (uint64_t*)buf;
        keys_sorted = (uint64_t*)((char*)buf + kb);
        uint32_t* idx = (uint32_t*)((char*)buf + 2 * kb);
        idx_sorted = (uint32_t*)((char*)buf + 2 * kb + ib);
        void* tmp = (char*)buf + 2 * kb + 2 * ib;
        hipLaunchKernelGGL(k_boundary_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_addr_space, d_pointer, n, keys,
                           idx, (uint32_t*)flag, as_bits, pointer_bits);
        ZK_HIP_CHECK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_sorted, idx, idx_sorted, (int)n, 0, 64, ctx->stream));
    }
    hipLaunchKernelGGL(k_boundary_fill, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, keys_sorted, idx_sorted, d_initial,
                       d_final, d_timestamp, n, N, pointer_bits, d_trace, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    uint32_t h_bad = 0;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(&h_bad, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_bad) return set_error(ctx, ZKHIP_ERR_INVALID, "memory_boundary_tracegen: " + std::to_string(h_bad) + " records out of range or duplicated");
    return ZKHIP_OK;
}

// ---- an instruction chip: RV32 base ALU core (OpenVM rv32im BaseAluCoreAir: ADD / SUB / XOR / OR / AND on 4 x 8-bit limbs) --------
// From execution records (opcode, rs1 value b, rs2 / immediate value c) the kernel fills one trace row per record AND counts, with
// atomics, the bitwise-lookup requests the row's interactions make -- the way the reference's GPU chips fill their traces and bump
// the periphery tables in one pass (AGENTS.md:183-187).  Columns (ZKHIP_RV32_ALU_WIDTH = 18, stride 2^log_height, Montgomery):
//   a[4] | b[4] | c[4] | is_add is_sub is_xor is_or is_and | is_valid        (little-endian limbs; a = result)
// Per limb the chip sends (x, y, x ^ y, 1) to the bitwise lookup bus: (b_i, c_i) for the bitwise opcodes, (a_i, a_i) for ADD / SUB
// (a ^ a = 0: a range check of the result limb).  The AIR is air.py rv32_alu_core_air(); rows >= n are zero.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_alu(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ bs,
                                                  const uint32_t* __restrict__ cs, size_t n, size_t N, uint32_t* __restrict__ trace,
                                                  uint32_t* __restrict__ xor_counts, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_x[HOT_SLOTS], hc_x[HOT_SLOTS];
    hot_init(hk_x, hc_x);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_ALU_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r], b = bs[r], c = cs[r];
        if (op > 4) {
            atomicAdd(bad, 1u);
        } else {
            const uint32_t a = op == 0 ? b + c : op == 1 ? b - c : op == 2 ? (b ^ c) : op == 3 ? (b | c) : (b & c);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t ai = (a >> (8 * i)) & 255u, bi = (b >> (8 * i)) & 255u, ci = (c >> (8 * i)) & 255u;
                col[i] = to_monty(ai), col[4 + i] = to_monty(bi), col[8 + i] = to_monty(ci);
                const uint32_t x = op >= 2 ? bi : ai, y = op >= 2 ? ci : ai;
                hot_add(hk_x, hc_x, xor_counts, (x << 8) | y);
            }
            col[12 + op] = MONTY_ONE;
            col[17] = MONTY_ONE;
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_ALU_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_x, hc_x, xor_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_alu_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_b, const uint32_t* d_c, size_t n,
                                       unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_opcode || !d_b || !d_c))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_alu_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "rv32_alu_tracegen");
    // the XOR multiplicity column of the 8-bit bitwise lookup table (column 1 of its 2 x 2^16 trace): Montgomery -> counts -> Montgomery
    uint32_t* xor_col = d_bitwise_trace + (1u << 16);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, xor_col, (size_t)1 << 16, 0);
    hipLaunchKernelGGL(k_rv32_alu, dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), dim3(256), 0, ctx->stream, d_opcode, d_b, d_c, n, N, d_trace, xor_col,
                       (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, xor_col, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "rv32_alu_tracegen (opcode > 4)");
}

// ---- RV32 multiplication core (OpenVM rv32im MultiplicationCoreAir: MUL on 4 x 8-bit limbs) ----------------------------------------
// Record = the two 32-bit operands.  Row: a[4] (low 32 bits of b * c) | b[4] | c[4] | is_valid  (ZKHIP_RV32_MUL_WIDTH = 13).  The
// carries carry_i = (sum_{k<=i} b_k c_{i-k} + carry_{i-1} - a_i) / 256 are not columns: the AIR recomputes them as expressions and
// sends (a_i, carry_i) to the range-TUPLE checker -- sizes [256, 8192] in the reference's config exist for exactly this pair
// (crates/circuits/chunk-circuit/openvm.toml `range_tuple_checker_sizes`).  The kernel counts those four requests per record into
// the tuple table's trace while it fills the row.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_mul(const uint32_t* __restrict__ bs, const uint32_t* __restrict__ cs, size_t n, size_t N,
                                                  uint32_t* __restrict__ trace, uint32_t* __restrict__ tuple_counts, uint32_t size_y) {
    __shared__ uint32_t hk_t[HOT_SLOTS], hc_t[HOT_SLOTS];
    hot_init(hk_t, hc_t);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_MUL_WIDTH] = {};
    if (r < n) {
        const uint32_t b = bs[r], c = cs[r];
        uint32_t bl[4], cl[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bl[i] = (b >> (8 * i)) & 255u, cl[i] = (c >> (8 * i)) & 255u;
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("" : "+v"(bl[i]), "+v"(cl[i]));   // no v_dot4_u32_u8 rewrite: see k_rv32_mulh
        // schoolbook columns of the low word: sums stay below 4 * 255^2 + 1024 < 2^19
        const uint32_t s[4] = {bl[0] * cl[0], bl[0] * cl[1] + bl[1] * cl[0], bl[0] * cl[2] + bl[1] * cl[1] + bl[2] * cl[0],
                               bl[0] * cl[3] + bl[1] * cl[2] + bl[2] * cl[1] + bl[3] * cl[0]};
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t acc = s[i] + carry;
            const uint32_t ai = acc & 255u;
            carry = acc >> 8;  // < 1024
            col[i] = to_monty(ai), col[4 + i] = to_monty(bl[i]), col[8 + i] = to_monty(cl[i]);
            hot_add(hk_t, hc_t, tuple_counts, ai * size_y + carry);
        }
        col[12] = MONTY_ONE;
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_MUL_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_t, hc_t, tuple_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_mul_tracegen(zkhip_ctx* ctx, const uint32_t* d_b, const uint32_t* d_c, size_t n, unsigned log_height,
                                       uint32_t* d_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_tuple_counts || log_height > 27 || (n && (!d_b || !d_c))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height, T = (size_t)size_x * size_y;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_mul_tracegen: more records than rows");
    if (size_x < 256 || size_y < 1024 || T > ((size_t)1 << 27)) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_mul_tracegen: the tuple table must cover (limb < 256, carry < 1024)");
    KernelScope ks(ctx, "rv32_mul_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 0);
    hipLaunchKernelGGL(k_rv32_mul, dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), dim3(256), 0, ctx->stream, d_b, d_c, n, N, d_trace, d_tuple_counts, size_y);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return ZKHIP_OK;
}

// ---- system chips: the program chip and the execution frames that look instructions up in it ----------------------------------
// OpenVM's ProgramAir keeps the program (pc, opcode, operands a..g: 9 fields per instruction) as a CACHED main partition -- its
// commitment is part of the verifying key's identity and is reused by every segment proof -- and one common column, the number
// of times each instruction was executed; it RECEIVES each instruction that often on the program bus.  The reference's stored
// proofs have exactly this shape for their first AIR (cached width 9, common width 1).  The execution side here is one row
// per executed instruction that SENDS the instruction's nine fields (in OpenVM every instruction chip's adapter does that; a
// stand-alone frame chip keeps the pair self-contained).  From the list of executed instruction indices:
//   k_program_freq  : histogram -> the frequency column (the program chip's whole common trace)
//   k_exec_frames   : gather    -> frame rows [9 program fields | is_valid]
namespace zk {
namespace {
// a loop's instructions are the hot bins of this histogram: the workgroup counts the program's first 2^13 rows in LDS and merges
// once (k_range_counts in tracegen.hip has the measurements), rows beyond that go to HBM through the wave-aggregated increment
__global__ __launch_bounds__(256) void k_program_freq(const uint32_t* __restrict__ idx, size_t n, size_t N, uint32_t* __restrict__ freq,
                                                      uint32_t* __restrict__ bad) {
    __shared__ uint32_t bins[1u << 13];
    const uint32_t L = N < (1u << 13) ? (uint32_t)N : (1u << 13);
    for (uint32_t i = threadIdx.x; i < L; i += 256) bins[i] = 0;
    zk_syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t k = idx[i];
        if (k >= N) atomicAdd(bad, 1u);
        else if (k < L) atomicAdd(&bins[k], 1u);
        else hist_add(freq, k);
    }
    zk_syncthreads();
    for (uint32_t i = threadIdx.x; i < L; i += 256)
        if (bins[i]) atomicAdd(&freq[i], bins[i]);
}
__global__ __launch_bounds__(256) void k_exec_frames(const uint32_t* __restrict__ idx, size_t n, const uint32_t* __restrict__ program,
                                                     size_t n_program, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t k = 0;
    const bool valid = r < n && (k = idx[r]) < n_program;
    if (r < n && !valid) atomicAdd(bad, 1u);
#pragma unroll
    for (int q = 0; q < ZKHIP_PROGRAM_FIELDS; q++) trace[(size_t)q * N + r] = valid ? program[(size_t)q * n_program + k] : 0u;
    trace[(size_t)ZKHIP_PROGRAM_FIELDS * N + r] = valid ? MONTY_ONE : 0u;
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_program_freq_tracegen(zkhip_ctx* ctx, const uint32_t* d_pc_index, size_t n, unsigned log_height, uint32_t* d_freq) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_freq || log_height > 27 || (n && !d_pc_index)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "program_freq_tracegen");
    ZK_HIP_CHECK(ctx, hipMemsetAsync(d_freq, 0, N * 4, ctx->stream));
    if (n)
        hipLaunchKernelGGL(k_program_freq, dim3((unsigned)std::min<size_t>((n + 256 * 32 - 1) / (256 * 32), 1024)), dim3(256), 0, ctx->stream, d_pc_index, n, N,
                           d_freq, (uint32_t*)flag);
    hipLaunchKernelGGL(k_tab_repr, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_freq, N, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "program_freq_tracegen (instruction index beyond the program)");
}

extern "C" int zkhip_exec_frame_tracegen(zkhip_ctx* ctx, const uint32_t* d_pc_index, size_t n, const uint32_t* d_program, size_t n_program,
                                         unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_program || log_height > 27 || (n && !d_pc_index)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "exec_frame_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "exec_frame_tracegen");
    hipLaunchKernelGGL(k_exec_frames, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_pc_index, n, d_program, n_program, N, d_trace,
                       (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "exec_frame_tracegen (instruction index beyond the program)");
}

// ---- RV32 less-than core (OpenVM rv32im LessThanCoreAir<4, 8>: SLT / SLTU) -------------------------------------------------------
// Record = (opcode 0 = SLT signed, 1 = SLTU unsigned; operands b, c).  Row (ZKHIP_RV32_LT_WIDTH = 18 columns):
//   b[4] | c[4] | cmp | is_slt is_sltu | b_msb_f c_msb_f | marker[4] | diff_val
// b_msb_f / c_msb_f are the most significant limbs as FIELD elements: the limb itself for SLTU, limb - 256 for a negative SLT
// operand; marker[i] = 1 at the most significant limb where b and c differ (none if b = c), diff_val = |c_i - b_i| there (1..255).
// The row sends two RANGE requests to the bitwise lookup: (b_msb_f + 128 is_slt, c_msb_f + 128 is_slt) -- which is what makes the
// signed limbs lie in [-128, 127] -- and (diff_val - 1, 0) when a marker is set; both are counted here into the range column of
// the lookup table's trace.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_lt(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ bs,
                                                 const uint32_t* __restrict__ cs, size_t n, size_t N, uint32_t* __restrict__ trace,
                                                 uint32_t* __restrict__ range_counts, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_r[HOT_SLOTS], hc_r[HOT_SLOTS];
    hot_init(hk_r, hc_r);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_LT_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r], b = bs[r], c = cs[r];
        if (op > 1) {
            atomicAdd(bad, 1u);
        } else {
            const bool is_slt = op == 0;
            int bl[4], cl[4];
#pragma unroll
            for (int i = 0; i < 4; i++) bl[i] = (int)((b >> (8 * i)) & 255u), cl[i] = (int)((c >> (8 * i)) & 255u);
            const int bm = is_slt && bl[3] >= 128 ? bl[3] - 256 : bl[3], cm = is_slt && cl[3] >= 128 ? cl[3] - 256 : cl[3];
            int mark = -1, diff = 0;
#pragma unroll
            for (int i = 3; i >= 0; i--) {
                const int x = i == 3 ? bm : bl[i], y = i == 3 ? cm : cl[i];
                if (mark < 0 && x != y) mark = i, diff = y - x;
            }
            const uint32_t cmp = mark >= 0 && diff > 0;
            const int dv = diff > 0 ? diff : -diff;
#pragma unroll
            for (int i = 0; i < 4; i++) col[i] = to_monty((uint32_t)bl[i]), col[4 + i] = to_monty((uint32_t)cl[i]);
            col[8] = cmp ? MONTY_ONE : 0u;
            col[9 + op] = MONTY_ONE;
            col[11] = to_monty(bm < 0 ? P - (uint32_t)(-bm) : (uint32_t)bm);
            col[12] = to_monty(cm < 0 ? P - (uint32_t)(-cm) : (uint32_t)cm);
            if (mark >= 0) col[13 + mark] = MONTY_ONE, col[17] = to_monty((uint32_t)dv);
            const uint32_t sh = is_slt ? 128u : 0u;
            hot_add(hk_r, hc_r, range_counts, ((uint32_t)(bm + (int)sh) << 8) | (uint32_t)(cm + (int)sh));
            if (mark >= 0) hot_add(hk_r, hc_r, range_counts, (uint32_t)(dv - 1) << 8);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_LT_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_r, hc_r, range_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_lt_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_b, const uint32_t* d_c, size_t n,
                                      unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_opcode || !d_b || !d_c))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_lt_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "rv32_lt_tracegen");
    // the RANGE multiplicity column of the 8-bit bitwise lookup table (column 0 of its 2 x 2^16 trace)
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);
    hipLaunchKernelGGL(k_rv32_lt, dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), dim3(256), 0, ctx->stream, d_opcode, d_b, d_c, n, N, d_trace, d_bitwise_trace,
                       (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "rv32_lt_tracegen (opcode > 1)");
}

// ---- memory access chip: the offline memory-checking argument's per-access rows ---------------------------------------------------
// One row per cell access of the execution's memory log (include/zkhip_vm.hpp ExecRecords::acc_*): the row RECEIVES the state the
// previous access of the cell left, (as, ptr, prev_data, prev_ts), and SENDS the state it leaves, (as, ptr, data, ts), on the memory
// bus; the boundary chip (zkhip_memory_boundary_tracegen) sends every touched cell's initial state and receives its final one, so the
// bus balances exactly when the log is a consistent history of the memory.  Time moves forward: ts - prev_ts - 1 = gap_lo + 2^16
// gap_hi with both limbs (and the 16-bit cell value) sent to the range checker.  Columns (ZKHIP_MEMORY_ACCESS_WIDTH = 10):
//   as | ptr | prev_data | prev_ts | data | ts | is_read | is_valid | gap_lo | gap_hi
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_memory_access(const uint32_t* __restrict__ as, const uint32_t* __restrict__ ptr,
                                                       const uint32_t* __restrict__ prev_data, const uint32_t* __restrict__ prev_ts,
                                                       const uint32_t* __restrict__ data, const uint32_t* __restrict__ ts,
                                                       const uint32_t* __restrict__ is_read, size_t n, size_t N, uint32_t* __restrict__ trace,
                                                       uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t col[ZKHIP_MEMORY_ACCESS_WIDTH] = {};
    if (r < n) {
        const uint32_t pd = prev_data[r], pt = prev_ts[r], d = data[r], t = ts[r], rd = is_read[r];
        if (as[r] >= P || ptr[r] >= P || pd >= 65536u || d >= 65536u || t >= P || pt >= t || rd > 1u || (rd && d != pd)) {
            atomicAdd(bad, 1u);
        } else {
            const uint32_t gap = t - pt - 1u;
            col[0] = to_monty(as[r]), col[1] = to_monty(ptr[r]), col[2] = to_monty(pd), col[3] = to_monty(pt), col[4] = to_monty(d);
            col[5] = to_monty(t), col[6] = rd ? MONTY_ONE : 0u, col[7] = MONTY_ONE, col[8] = to_monty(gap & 0xffffu), col[9] = to_monty(gap >> 16);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_MEMORY_ACCESS_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_memory_access_tracegen(zkhip_ctx* ctx, const uint32_t* d_addr_space, const uint32_t* d_pointer, const uint32_t* d_prev_data,
                                            const uint32_t* d_prev_ts, const uint32_t* d_data, const uint32_t* d_ts, const uint32_t* d_is_read,
                                            size_t n, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27 || (n && (!d_addr_space || !d_pointer || !d_prev_data || !d_prev_ts || !d_data || !d_ts || !d_is_read)))
        return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "memory_access_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "memory_access_tracegen");
    hipLaunchKernelGGL(k_memory_access, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_addr_space, d_pointer, d_prev_data, d_prev_ts,
                       d_data, d_ts, d_is_read, n, N, d_trace, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "memory_access_tracegen (value above 16 bits, time not moving forward, or a read that changes its cell)");
}

// ---- RV32 shift core (OpenVM rv32im ShiftCoreAir<4, 8>: SLL / SRL / SRA) -----------------------------------------------------------
// Record = (opcode 0 = SLL, 1 = SRL, 2 = SRA; value b; shift operand c, of which c mod 32 counts).  Row (ZKHIP_RV32_SHIFT_WIDTH = 32):
//   a[4] | b[4] | c0 | is_sll is_srl is_sra | bit_marker[8] | limb_marker[4] | carry[4] | sign | q | mult_left | mult_right
// (air.py rv32_shift_core_air()).  Lookup requests counted into the 8-bit bitwise table's trace in the same pass: range pairs
// (carry[i], mult - 1 - carry[i]) x 4, (a0, a1), (a2, a3), (q, 32 q); for SRA the XOR request (b[3], 128).
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_shift(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ bs,
                                                    const uint32_t* __restrict__ cs, size_t n, size_t N, uint32_t* __restrict__ trace,
                                                    uint32_t* __restrict__ range_counts, uint32_t* __restrict__ xor_counts,
                                                    uint32_t* __restrict__ bad) {
    __shared__ uint32_t hot_k[HOT_SLOTS], hot_c[HOT_SLOTS];   // one cache for both columns of the bitwise table (xor = range + 2^16)
    hot_init(hot_k, hot_c);
    const uint32_t xor_off = (uint32_t)(xor_counts - range_counts);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_SHIFT_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r], b = bs[r], c = cs[r];
        if (op > 2) {
            atomicAdd(bad, 1u);
        } else {
            const uint32_t s = c & 31u, bsh = s & 7u, lsh = s >> 3, c0 = c & 255u, q = c0 >> 5, mult = 1u << bsh;
            const uint32_t a = op == 0 ? b << s : op == 1 ? b >> s : (uint32_t)((int32_t)b >> s);
            const uint32_t sign = op == 2 ? b >> 31 : 0u;
            uint32_t bl[4], al[4], cy[4];
#pragma unroll
            for (int i = 0; i < 4; i++) bl[i] = (b >> (8 * i)) & 255u, al[i] = (a >> (8 * i)) & 255u;
            if (op == 0) {
                uint32_t carry = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) carry = cy[k] = (bl[k] * mult + carry) >> 8;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) cy[k] = bl[k] & (mult - 1u);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) col[i] = to_monty(al[i]), col[4 + i] = to_monty(bl[i]), col[24 + i] = to_monty(cy[i]);
            col[8] = to_monty(c0);
            col[9 + op] = MONTY_ONE;
            col[12 + bsh] = MONTY_ONE;
            col[20 + lsh] = MONTY_ONE;
            col[28] = sign ? MONTY_ONE : 0u;
            col[29] = to_monty(q);
            col[op == 0 ? 30 : 31] = to_monty(mult);
#pragma unroll
            for (int i = 0; i < 4; i++) hot_add(hot_k, hot_c, range_counts, (cy[i] << 8) | (mult - 1u - cy[i]));
            hot_add(hot_k, hot_c, range_counts, (al[0] << 8) | al[1]);
            hot_add(hot_k, hot_c, range_counts, (al[2] << 8) | al[3]);
            hot_add(hot_k, hot_c, range_counts, (q << 8) | (32u * q));
            if (op == 2) hot_add(hot_k, hot_c, range_counts, xor_off + ((bl[3] << 8) | 128u));
        }
    }
#pragma unroll
    for (int qq = 0; qq < ZKHIP_RV32_SHIFT_WIDTH; qq++) trace[(size_t)qq * N + r] = col[qq];
    }
    hot_flush(hot_k, hot_c, range_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_shift_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_b, const uint32_t* d_c, size_t n,
                                         unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_opcode || !d_b || !d_c))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_shift_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "rv32_shift_tracegen");
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(512), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)2 << 16, 0);  // both columns
    hipLaunchKernelGGL(k_rv32_shift, dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), dim3(256), 0, ctx->stream, d_opcode, d_b, d_c, n, N, d_trace,
                       d_bitwise_trace, d_bitwise_trace + (1u << 16), (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(512), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)2 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "rv32_shift_tracegen (opcode > 2)");
}

// ---- RV32 branch-equal core (OpenVM rv32im BranchEqualCoreAir<4>: BEQ / BNE) ------------------------------------------------------
// Record = (opcode 0 = BEQ, 1 = BNE; operands a, b; the branch offset imm as a field element's canonical value, i.e. p - |imm| for a
// backward branch).  Row (ZKHIP_RV32_BRANCH_EQ_WIDTH = 17): a[4] | b[4] | taken | imm | is_beq is_bne | diff_inv_marker[4] | pc_inc;
// the marker of the first differing limb pair holds (a_i - b_i)^-1 -- one field inversion per unequal row, on the device.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_branch_eq(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ as, const uint32_t* __restrict__ bs,
                                                        const uint32_t* __restrict__ imms, size_t n, size_t N, uint32_t* __restrict__ trace,
                                                        uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t col[ZKHIP_RV32_BRANCH_EQ_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r], a = as[r], b = bs[r], imm = imms[r];
        if (op > 1 || imm >= P) {
            atomicAdd(bad, 1u);
        } else {
            const bool taken = op == 0 ? a == b : a != b;
            int first = -1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t ai = (a >> (8 * i)) & 255u, bi = (b >> (8 * i)) & 255u;
                col[i] = to_monty(ai), col[4 + i] = to_monty(bi);
                if (first < 0 && ai != bi) first = i;
            }
            if (first >= 0) {
                const uint32_t ai = (a >> (8 * first)) & 255u, bi = (b >> (8 * first)) & 255u;
                col[12 + first] = minv(msub(to_monty(ai), to_monty(bi)));
            }
            col[8] = taken ? MONTY_ONE : 0u;
            col[9] = to_monty(imm);
            col[10 + op] = MONTY_ONE;
            col[16] = taken ? to_monty(imm) : to_monty(4u);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_BRANCH_EQ_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_branch_eq_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_a, const uint32_t* d_b, const uint32_t* d_imm,
                                             size_t n, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27 || (n && (!d_opcode || !d_a || !d_b || !d_imm))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_branch_eq_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "rv32_branch_eq_tracegen");
    hipLaunchKernelGGL(k_rv32_branch_eq, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_opcode, d_a, d_b, d_imm, n, N, d_trace,
                       (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "rv32_branch_eq_tracegen (opcode > 1 or offset not a field element)");
}

// ---- RV32 branch-less-than core (OpenVM rv32im BranchLessThanCoreAir<4, 8>: BLT / BLTU / BGE / BGEU) -------------------------------
// Record = (opcode 0 = BLT, 1 = BLTU, 2 = BGE, 3 = BGEU; operands a, b; the offset as a canonical field element).  Row
// (ZKHIP_RV32_BRANCH_LT_WIDTH = 23): a[4] | b[4] | cmp_lt | taken | imm | 4 opcode flags | a_msb_f b_msb_f | marker[4] | diff_val | pc_inc;
// the comparison columns are those of the less-than chip, and so are the two range requests counted into the bitwise table.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_branch_lt(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ as, const uint32_t* __restrict__ bs,
                                                        const uint32_t* __restrict__ imms, size_t n, size_t N, uint32_t* __restrict__ trace,
                                                        uint32_t* __restrict__ range_counts, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_r[HOT_SLOTS], hc_r[HOT_SLOTS];
    hot_init(hk_r, hc_r);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_BRANCH_LT_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r], a = as[r], b = bs[r], imm = imms[r];
        if (op > 3 || imm >= P) {
            atomicAdd(bad, 1u);
        } else {
            const bool is_signed = (op & 1u) == 0, is_ge = op >= 2;
            int al[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; i++) al[i] = (int)((a >> (8 * i)) & 255u), bl[i] = (int)((b >> (8 * i)) & 255u);
            const int am = is_signed && al[3] >= 128 ? al[3] - 256 : al[3], bm = is_signed && bl[3] >= 128 ? bl[3] - 256 : bl[3];
            int mark = -1, diff = 0;
#pragma unroll
            for (int i = 3; i >= 0; i--) {
                const int x = i == 3 ? am : al[i], y = i == 3 ? bm : bl[i];
                if (mark < 0 && x != y) mark = i, diff = y - x;
            }
            const bool lt = mark >= 0 && diff > 0, taken = lt != is_ge;
            const int dv = diff > 0 ? diff : -diff;
#pragma unroll
            for (int i = 0; i < 4; i++) col[i] = to_monty((uint32_t)al[i]), col[4 + i] = to_monty((uint32_t)bl[i]);
            col[8] = lt ? MONTY_ONE : 0u, col[9] = taken ? MONTY_ONE : 0u, col[10] = to_monty(imm);
            col[11 + op] = MONTY_ONE;
            col[15] = to_monty(am < 0 ? P - (uint32_t)(-am) : (uint32_t)am);
            col[16] = to_monty(bm < 0 ? P - (uint32_t)(-bm) : (uint32_t)bm);
            if (mark >= 0) col[17 + mark] = MONTY_ONE, col[21] = to_monty((uint32_t)dv);
            col[22] = taken ? to_monty(imm) : to_monty(4u);
            const int sh = is_signed ? 128 : 0;
            hot_add(hk_r, hc_r, range_counts, ((uint32_t)(am + sh) << 8) | (uint32_t)(bm + sh));
            if (mark >= 0) hot_add(hk_r, hc_r, range_counts, (uint32_t)(dv - 1) << 8);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_BRANCH_LT_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_r, hc_r, range_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_branch_lt_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_a, const uint32_t* d_b, const uint32_t* d_imm,
                                             size_t n, unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_opcode || !d_a || !d_b || !d_imm))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_branch_lt_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "rv32_branch_lt_tracegen");
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);
    hipLaunchKernelGGL(k_rv32_branch_lt, dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), dim3(256), 0, ctx->stream, d_opcode, d_a, d_b, d_imm, n, N, d_trace,
                       d_bitwise_trace, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "rv32_branch_lt_tracegen (opcode > 3 or offset not a field element)");
}

// ---- RV32 JAL / LUI, AUIPC and JALR cores (OpenVM rv32im Rv32JalLuiCoreAir, Rv32AuipcCoreAir, Rv32JalrCoreAir) ---------------------
// One thread per record / row as the chips above; the rows' range requests are counted into column 0 of the bitwise table.
namespace zk {
namespace {
__device__ __forceinline__ void bump_range(uint32_t* range_counts, uint32_t x, uint32_t y) { hist_add(range_counts, (x << 8) | y); }
__device__ __forceinline__ void bump_range_hot(uint32_t* keys, uint32_t* cnts, uint32_t* range_counts, uint32_t x, uint32_t y) {
    hot_add(keys, cnts, range_counts, (x << 8) | y);
}

__global__ __launch_bounds__(256) void k_rv32_jal_lui(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ pcs, const uint32_t* __restrict__ imms,
                                                      size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ range_counts,
                                                      uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_r[HOT_SLOTS], hc_r[HOT_SLOTS];
    hot_init(hk_r, hc_r);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_JAL_LUI_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r], pc = pcs[r], imm = imms[r];
        const bool ok = op == 0 ? (imm < P && pc < (1u << 30) - 4) : (op == 1 && (imm >> 20) == 0 && pc < P);
        if (!ok) {
            atomicAdd(bad, 1u);
        } else {
            const uint32_t rd = op == 0 ? pc + 4 : imm << 12;
            col[0] = to_monty(pc), col[1] = to_monty(imm);
#pragma unroll
            for (int i = 0; i < 4; i++) col[2 + i] = to_monty((rd >> (8 * i)) & 255u);
            col[6 + op] = MONTY_ONE;
            col[8] = op == 0 ? to_monty(imm) : to_monty(4u);
            bump_range_hot(hk_r, hc_r, range_counts, rd & 255u, (rd >> 8) & 255u);
            bump_range_hot(hk_r, hc_r, range_counts, (rd >> 16) & 255u, rd >> 24);
            if (op == 0) bump_range_hot(hk_r, hc_r, range_counts, (rd >> 24) * 4, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_JAL_LUI_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_r, hc_r, range_counts);
}

__global__ __launch_bounds__(256) void k_rv32_auipc(const uint32_t* __restrict__ pcs, const uint32_t* __restrict__ imms, size_t n, size_t N,
                                                    uint32_t* __restrict__ trace, uint32_t* __restrict__ range_counts, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_r[HOT_SLOTS], hc_r[HOT_SLOTS];
    hot_init(hk_r, hc_r);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_AUIPC_WIDTH] = {};
    if (r < n) {
        const uint32_t pc = pcs[r], imm = imms[r];
        if (pc >= (1u << 30) || (imm >> 20) != 0) {
            atomicAdd(bad, 1u);
        } else {
            const uint32_t rd = pc + (imm << 12), im16 = imm << 4;
            uint32_t pl[4], il[3], dl[4];
#pragma unroll
            for (int i = 0; i < 4; i++) pl[i] = (pc >> (8 * i)) & 255u, dl[i] = (rd >> (8 * i)) & 255u;
#pragma unroll
            for (int i = 0; i < 3; i++) il[i] = (im16 >> (8 * i)) & 255u;
            col[0] = to_monty(pc), col[1] = to_monty(imm);
#pragma unroll
            for (int i = 0; i < 4; i++) col[2 + i] = to_monty(pl[i]), col[9 + i] = to_monty(dl[i]);
#pragma unroll
            for (int i = 0; i < 3; i++) col[6 + i] = to_monty(il[i]);
            col[13] = MONTY_ONE;
            bump_range_hot(hk_r, hc_r, range_counts, pl[0], pl[1]), bump_range_hot(hk_r, hc_r, range_counts, pl[2], 4 * pl[3]), bump_range_hot(hk_r, hc_r, range_counts, il[0], il[1]);
            bump_range_hot(hk_r, hc_r, range_counts, il[2], dl[1]), bump_range_hot(hk_r, hc_r, range_counts, dl[2], dl[3]);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_AUIPC_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_r, hc_r, range_counts);
}

__global__ __launch_bounds__(256) void k_rv32_jalr(const uint32_t* __restrict__ pcs, const uint32_t* __restrict__ rs1s, const uint32_t* __restrict__ imms,
                                                   size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ range_counts,
                                                   uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_r[HOT_SLOTS], hc_r[HOT_SLOTS];
    hot_init(hk_r, hc_r);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_JALR_WIDTH] = {};
    if (r < n) {
        const uint32_t pc = pcs[r], rs1 = rs1s[r], imm = imms[r];
        const uint32_t sign = (imm >> 11) & 1u, t = rs1 + (sign ? imm | 0xfffff000u : imm), rd = pc + 4, to_pc = t & ~1u;
        if (pc >= (1u << 30) - 4 || (imm >> 12) != 0 || to_pc >= (1u << 30)) {   // program counters are 30-bit values
            atomicAdd(bad, 1u);
        } else {
            col[0] = to_monty(pc), col[1] = to_monty(imm), col[2] = to_monty(imm & 255u), col[3] = to_monty(imm >> 8), col[4] = sign ? MONTY_ONE : 0u;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                col[5 + i] = to_monty((rs1 >> (8 * i)) & 255u);
                col[9 + i] = to_monty((rd >> (8 * i)) & 255u);
                col[13 + i] = to_monty((t >> (8 * i)) & 255u);
            }
            col[17] = (t & 1u) ? MONTY_ONE : 0u, col[18] = to_monty(to_pc), col[19] = MONTY_ONE;
            bump_range_hot(hk_r, hc_r, range_counts, imm & 255u, ((imm >> 8) - 8 * sign) * 32);
            bump_range_hot(hk_r, hc_r, range_counts, (t & 255u) >> 1, (t >> 8) & 255u);
            bump_range_hot(hk_r, hc_r, range_counts, (t >> 16) & 255u, (t >> 24) * 4);
            bump_range_hot(hk_r, hc_r, range_counts, rd & 255u, (rd >> 8) & 255u);
            bump_range_hot(hk_r, hc_r, range_counts, (rd >> 16) & 255u, (rd >> 24) * 4);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_JALR_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_r, hc_r, range_counts);
}

// shared launch frame of the three: flag, table to counts, the chip's kernel, table back to Montgomery form
template <typename Launch>
int jump_chip_tracegen(zkhip_ctx* ctx, const char* name, size_t n, unsigned log_height, uint32_t* d_bitwise_trace, Launch&& launch) {
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, name);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);
    launch(dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), N, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, name);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_jal_lui_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_pc, const uint32_t* d_imm, size_t n,
                                           unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_opcode || !d_pc || !d_imm))) return ZKHIP_ERR_INVALID;
    return jump_chip_tracegen(ctx, "rv32_jal_lui_tracegen", n, log_height, d_bitwise_trace, [&](dim3 grid, size_t N, uint32_t* flag) {
        hipLaunchKernelGGL(k_rv32_jal_lui, grid, dim3(256), 0, ctx->stream, d_opcode, d_pc, d_imm, n, N, d_trace, d_bitwise_trace, flag);
    });
}

extern "C" int zkhip_rv32_auipc_tracegen(zkhip_ctx* ctx, const uint32_t* d_pc, const uint32_t* d_imm, size_t n, unsigned log_height, uint32_t* d_trace,
                                         uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_pc || !d_imm))) return ZKHIP_ERR_INVALID;
    return jump_chip_tracegen(ctx, "rv32_auipc_tracegen", n, log_height, d_bitwise_trace, [&](dim3 grid, size_t N, uint32_t* flag) {
        hipLaunchKernelGGL(k_rv32_auipc, grid, dim3(256), 0, ctx->stream, d_pc, d_imm, n, N, d_trace, d_bitwise_trace, flag);
    });
}

extern "C" int zkhip_rv32_jalr_tracegen(zkhip_ctx* ctx, const uint32_t* d_pc, const uint32_t* d_rs1, const uint32_t* d_imm, size_t n,
                                        unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_pc || !d_rs1 || !d_imm))) return ZKHIP_ERR_INVALID;
    return jump_chip_tracegen(ctx, "rv32_jalr_tracegen", n, log_height, d_bitwise_trace, [&](dim3 grid, size_t N, uint32_t* flag) {
        hipLaunchKernelGGL(k_rv32_jalr, grid, dim3(256), 0, ctx->stream, d_pc, d_rs1, d_imm, n, N, d_trace, d_bitwise_trace, flag);
    });
}

// ---- RV32 high-multiplication core (OpenVM rv32im MulHCoreAir<4, 8>: MULH / MULHSU / MULHU) ------------------------------------------
// Record = (opcode 0 = MULH, 1 = MULHSU, 2 = MULHU; operands).  Row (ZKHIP_RV32_MULH_WIDTH = 21): a[4] | b[4] | c[4] | a_mul[4] | b_sign c_sign |
// 3 opcode flags.  The operands are extended by their sign limbs and multiplied limb by limb as the AIR states it; the eight
// (limb, carry) pairs go to the range-tuple table, the two sign requests to the bitwise table's range column.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_mulh(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ bs, const uint32_t* __restrict__ cs,
                                                   size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ tuple_counts, uint32_t size_y,
                                                   uint32_t* __restrict__ range_counts, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_t[HOT_SLOTS], hc_t[HOT_SLOTS];
    hot_init(hk_t, hc_t);
    __shared__ uint32_t hk_r[HOT_SLOTS], hc_r[HOT_SLOTS];
    hot_init(hk_r, hc_r);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_MULH_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r], b = bs[r], c = cs[r];
        if (op > 2) {
            atomicAdd(bad, 1u);
        } else {
            const uint32_t b_sign = op != 2 ? b >> 31 : 0u, c_sign = op == 0 ? c >> 31 : 0u;
            uint32_t l[8], m[8];
#pragma unroll
            for (int i = 0; i < 4; i++) l[i] = (b >> (8 * i)) & 255u, m[i] = (c >> (8 * i)) & 255u;
#pragma unroll
            for (int i = 4; i < 8; i++) l[i] = b_sign * 255u, m[i] = c_sign * 255u;
            // The limbs are made opaque to the optimiser: knowing them to be bytes, LLVM (ROCm 7.2) rewrites the column sums into
            // v_perm_b32 + v_dot4_u32_u8 and gets product limbs 1 and 2 wrong on gfx950 (found by the oracle parity test).
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" : "+v"(l[i]), "+v"(m[i]));
            uint32_t carry = 0;   // sums stay below 8 * 255^2 + 2048 < 2^20
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint32_t acc = carry;
#pragma unroll
                for (int k = 0; k <= i; k++) acc += l[k] * m[i - k];
                const uint32_t limb = acc & 255u;
                carry = acc >> 8;   // < 2048
                col[i < 4 ? 12 + i : i - 4] = to_monty(limb);
                hot_add(hk_t, hc_t, tuple_counts, limb * size_y + carry);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) col[4 + i] = to_monty(l[i]), col[8 + i] = to_monty(m[i]);
            col[16] = b_sign ? MONTY_ONE : 0u, col[17] = c_sign ? MONTY_ONE : 0u;
            col[18 + op] = MONTY_ONE;
            if (op != 2) bump_range_hot(hk_r, hc_r, range_counts, 2 * (l[3] - 128 * b_sign), 0);
            if (op == 0) bump_range_hot(hk_r, hc_r, range_counts, 2 * (m[3] - 128 * c_sign), 0);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_MULH_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_t, hc_t, tuple_counts);
    hot_flush(hk_r, hc_r, range_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_mulh_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_b, const uint32_t* d_c, size_t n, unsigned log_height,
                                        uint32_t* d_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_tuple_counts || !d_bitwise_trace || log_height > 27 || (n && (!d_opcode || !d_b || !d_c))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height, T = (size_t)size_x * size_y;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_mulh_tracegen: more records than rows");
    if (size_x < 256 || size_y < 2048 || T > ((size_t)1 << 27))
        return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_mulh_tracegen: the tuple table must cover (limb < 256, carry < 2048)");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "rv32_mulh_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 0);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);
    hipLaunchKernelGGL(k_rv32_mulh, dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), dim3(256), 0, ctx->stream, d_opcode, d_b, d_c, n, N, d_trace, d_tuple_counts, size_y,
                       d_bitwise_trace, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 1);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "rv32_mulh_tracegen (opcode > 2)");
}

// ---- RV32 load/store cores (OpenVM rv32im LoadStoreCoreAir<4> + LoadSignExtendCoreAir<4, 8>) in one chip -------------------------------
// Record = (case 0..19: LW, LHU@0 LHU@2, LBU@0..3, SW, SH@0 SH@2, SB@0..3, LH@0 LH@2, LB@0..3 -- the number after @ is the byte offset
// inside the aligned word; read word: memory for loads, the register for stores; prev word: what the destination held).
// Row (ZKHIP_RV32_LOADSTORE_WIDTH = 33): read[4] | prev[4] | write[4] | case flag[20] | sign.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_loadstore(const uint32_t* __restrict__ cases, const uint32_t* __restrict__ reads,
                                                        const uint32_t* __restrict__ prevs, size_t n, size_t N, uint32_t* __restrict__ trace,
                                                        uint32_t* __restrict__ range_counts, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_r[HOT_SLOTS], hc_r[HOT_SLOTS];
    hot_init(hk_r, hc_r);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t rd = 0, pv = 0, wr = 0, cs = 0xffffffffu, sign = 0;
    if (r < n) {
        cs = cases[r], rd = reads[r], pv = prevs[r];
        if (cs > 19) {
            atomicAdd(bad, 1u);
            cs = 0xffffffffu, rd = pv = 0;
        } else {
            // bytes and byte offset of the case: words (4, offset 0) at 0 and 7, halves at 1, 2, 8, 9, 14, 15, bytes elsewhere
            const bool word = cs == 0 || cs == 7, half = cs == 1 || cs == 2 || cs == 8 || cs == 9 || cs == 14 || cs == 15;
            const unsigned nb = word ? 4 : half ? 2 : 1;
            const unsigned base = cs < 1 ? 0 : cs < 3 ? 1 : cs < 7 ? 3 : cs < 8 ? 7 : cs < 10 ? 8 : cs < 14 ? 10 : cs < 16 ? 14 : 16;
            const unsigned sh = 8u * (half ? 2 * (cs - base) : (cs - base));
            const uint32_t mask = nb == 4 ? 0xffffffffu : (1u << (8 * nb)) - 1;
            if (cs >= 7 && cs <= 13) {
                wr = (pv & ~(mask << sh)) | ((rd & mask) << sh);
            } else {
                wr = (rd >> sh) & mask;
                if (cs >= 14) {
                    sign = (wr >> (8 * nb - 1)) & 1u;
                    if (sign) wr |= ~mask;
                    bump_range_hot(hk_r, hc_r, range_counts, 2 * (((wr >> (8 * (nb - 1))) & 255u) - 128 * sign), 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        trace[(size_t)i * N + r] = to_monty((rd >> (8 * i)) & 255u);
        trace[(size_t)(4 + i) * N + r] = to_monty((pv >> (8 * i)) & 255u);
        trace[(size_t)(8 + i) * N + r] = to_monty((wr >> (8 * i)) & 255u);
    }
#pragma unroll
    for (unsigned q = 0; q < 20; q++) trace[(size_t)(12 + q) * N + r] = q == cs ? MONTY_ONE : 0u;
    trace[(size_t)32 * N + r] = sign ? MONTY_ONE : 0u;
    }
    hot_flush(hk_r, hc_r, range_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_loadstore_tracegen(zkhip_ctx* ctx, const uint32_t* d_case, const uint32_t* d_read, const uint32_t* d_prev, size_t n,
                                             unsigned log_height, uint32_t* d_trace, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_bitwise_trace || log_height > 27 || (n && (!d_case || !d_read || !d_prev))) return ZKHIP_ERR_INVALID;
    return jump_chip_tracegen(ctx, "rv32_loadstore_tracegen", n, log_height, d_bitwise_trace, [&](dim3 grid, size_t N, uint32_t* flag) {
        hipLaunchKernelGGL(k_rv32_loadstore, grid, dim3(256), 0, ctx->stream, d_case, d_read, d_prev, n, N, d_trace, d_bitwise_trace, flag);
    });
}

// ---- RV32 division core (the job of OpenVM rv32im DivRemCoreAir<4, 8>: DIV / DIVU / REM / REMU) ---------------------------------------
// Record = (opcode 0 = DIV, 1 = DIVU, 2 = REM, 3 = REMU; dividend b; divisor c).  Row (ZKHIP_RV32_DIVREM_WIDTH = 41): b[4] | c[4] | q[4] | r[4] |
// c_abs[4] | r_abs[4] | b_sign c_sign q_sign r_sign | k_c k_r | zero_divisor c_sum_inv | marker[4] | diff | 4 opcode flags (air.py
// rv32_divrem_core_air() states what they prove).  One hardware division per row, one field inversion per non-zero divisor; the
// eight (limb, carry) pairs of c q + r - b go to the range-tuple table, six range requests to the bitwise table.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_rv32_divrem(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ bs, const uint32_t* __restrict__ cs,
                                                     size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ tuple_counts, uint32_t size_y,
                                                     uint32_t* __restrict__ range_counts, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_t[HOT_SLOTS], hc_t[HOT_SLOTS];
    hot_init(hk_t, hc_t);
    __shared__ uint32_t hk_r[HOT_SLOTS], hc_r[HOT_SLOTS];
    hot_init(hk_r, hc_r);
    for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < N; row += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_RV32_DIVREM_WIDTH] = {};
    if (row < n) {
        const uint32_t op = opc[row], b = bs[row], c = cs[row];
        if (op > 3) {
            atomicAdd(bad, 1u);
        } else {
            const bool is_signed = (op & 1u) == 0, overflow = is_signed && b == 0x80000000u && c == 0xffffffffu;
            uint32_t q, r;
            if (c == 0) q = 0xffffffffu, r = b;
            else if (overflow) q = b, r = 0;
            else if (is_signed) q = (uint32_t)((int32_t)b / (int32_t)c), r = (uint32_t)((int32_t)b % (int32_t)c);
            else q = b / c, r = b % c;
            const uint32_t b_sign = is_signed ? b >> 31 : 0u, c_sign = is_signed ? c >> 31 : 0u, r_sign = is_signed ? r >> 31 : 0u;
            const uint32_t q_sign = is_signed && !overflow ? q >> 31 : 0u;
            const uint32_t ca = c_sign ? 0u - c : c, ra = r_sign ? 0u - r : r;
            const uint32_t w[6] = {b, c, q, r, ca, ra};
#pragma unroll
            for (int g = 0; g < 6; g++)
#pragma unroll
                for (int i = 0; i < 4; i++) col[4 * g + i] = to_monty((w[g] >> (8 * i)) & 255u);
            col[24] = b_sign ? MONTY_ONE : 0u, col[25] = c_sign ? MONTY_ONE : 0u, col[26] = q_sign ? MONTY_ONE : 0u, col[27] = r_sign ? MONTY_ONE : 0u;
            col[28] = c_sign && (c & 0xffffu) ? MONTY_ONE : 0u, col[29] = r_sign && (r & 0xffffu) ? MONTY_ONE : 0u;
            col[37 + op] = MONTY_ONE;
            if (c == 0) {
                col[30] = MONTY_ONE;
            } else {
                col[31] = minv(to_monty((c & 255u) + ((c >> 8) & 255u) + ((c >> 16) & 255u) + (c >> 24)));
                int mark = 0;   // ra < ca: the magnitudes differ somewhere
#pragma unroll
                for (int i = 3; i >= 0; i--)
                    if (((ca ^ ra) >> (8 * i)) & 255u) {
                        mark = i;
                        break;
                    }
                const uint32_t d = ((ca >> (8 * mark)) & 255u) - ((ra >> (8 * mark)) & 255u);
                col[32 + mark] = MONTY_ONE, col[36] = to_monty(d);
                bump_range_hot(hk_r, hc_r, range_counts, d - 1, 0);
            }
            // c q + r - b over eight sign-extended limbs; the limbs are opaque to the optimiser (see k_rv32_mulh)
            uint32_t l[8], m[8];
            int rr[8], bb[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                l[i] = i < 4 ? (c >> (8 * i)) & 255u : 255u * c_sign, m[i] = i < 4 ? (q >> (8 * i)) & 255u : 255u * q_sign;
                rr[i] = (int)(i < 4 ? (r >> (8 * i)) & 255u : 255u * r_sign), bb[i] = (int)(i < 4 ? (b >> (8 * i)) & 255u : 255u * b_sign);
                asm volatile("" : "+v"(l[i]), "+v"(m[i]));
            }
            int carry = 0;   // sums are multiples of 256 in [0, 2^19)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int acc = carry + rr[i] - bb[i];
#pragma unroll
                for (int k = 0; k <= i; k++) acc += (int)(l[k] * m[i - k]);
                carry = acc >> 8;
                hot_add(hk_t, hc_t, tuple_counts, (i < 4 ? m[i] : (uint32_t)rr[i - 4]) * size_y + (uint32_t)carry);
            }
            if (is_signed) bump_range_hot(hk_r, hc_r, range_counts, 2 * ((b >> 24) - 128 * b_sign), 2 * ((c >> 24) - 128 * c_sign));
            bump_range_hot(hk_r, hc_r, range_counts, ca & 255u, (ca >> 8) & 255u), bump_range_hot(hk_r, hc_r, range_counts, (ca >> 16) & 255u, ca >> 24);
            bump_range_hot(hk_r, hc_r, range_counts, ra & 255u, (ra >> 8) & 255u), bump_range_hot(hk_r, hc_r, range_counts, (ra >> 16) & 255u, ra >> 24);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_RV32_DIVREM_WIDTH; q++) trace[(size_t)q * N + row] = col[q];
    }
    hot_flush(hk_t, hc_t, tuple_counts);
    hot_flush(hk_r, hc_r, range_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_rv32_divrem_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_b, const uint32_t* d_c, size_t n, unsigned log_height,
                                          uint32_t* d_trace, uint32_t* d_tuple_counts, uint32_t size_x, uint32_t size_y, uint32_t* d_bitwise_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_tuple_counts || !d_bitwise_trace || log_height > 27 || (n && (!d_opcode || !d_b || !d_c))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height, T = (size_t)size_x * size_y;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_divrem_tracegen: more records than rows");
    if (size_x < 256 || size_y < 2048 || T > ((size_t)1 << 27))
        return set_error(ctx, ZKHIP_ERR_INVALID, "rv32_divrem_tracegen: the tuple table must cover (limb < 256, carry < 2048)");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "rv32_divrem_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 0);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 0);
    hipLaunchKernelGGL(k_rv32_divrem, dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), dim3(256), 0, ctx->stream, d_opcode, d_b, d_c, n, N, d_trace, d_tuple_counts,
                       size_y, d_bitwise_trace, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_tuple_counts, T, 1);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(256), dim3(256), 0, ctx->stream, d_bitwise_trace, (size_t)1 << 16, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "rv32_divrem_tracegen (opcode > 3)");
}

// ---- NATIVE chips: base-field and extension-field arithmetic (OpenVM native FieldArithmeticCoreAir / FieldExtensionCoreAir) -----------
// The recursion programs the reference's aggregation circuits run execute on these: a = b op c over BabyBear,
// z = x op y over F[X] / (X^4 - 11).  Records are field elements; one field inversion per division, on the device.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_field_arith(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ bs, const uint32_t* __restrict__ cs,
                                                     size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t col[ZKHIP_FIELD_ARITH_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r], bc = bs[r], cc = cs[r];
        if (op > 3 || bc >= P || cc >= P || (op == 3 && cc == 0)) {
            atomicAdd(bad, 1u);
        } else {
            const uint32_t b = to_monty(bc), c = to_monty(cc);
            const uint32_t inv = op == 3 ? minv(c) : 0u;
            col[0] = op == 0 ? madd(b, c) : op == 1 ? msub(b, c) : op == 2 ? mmul(b, c) : mmul(b, inv);
            col[1] = b, col[2] = c, col[3 + op] = MONTY_ONE, col[7] = inv;
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_FIELD_ARITH_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
}

__global__ __launch_bounds__(256) void k_field_ext(const uint32_t* __restrict__ opc, const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
                                                   size_t n, size_t N, uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t col[ZKHIP_FIELD_EXT_WIDTH] = {};
    if (r < n) {
        const uint32_t op = opc[r];
        const uint4 xv = reinterpret_cast<const uint4*>(xs)[r], yv = reinterpret_cast<const uint4*>(ys)[r];
        const uint32_t xc[4] = {xv.x, xv.y, xv.z, xv.w}, yc[4] = {yv.x, yv.y, yv.z, yv.w};
        bool ok = op <= 3;
#pragma unroll
        for (int i = 0; i < 4; i++) ok = ok && xc[i] < P && yc[i] < P;
        if (ok && op == 3 && !(yc[0] | yc[1] | yc[2] | yc[3])) ok = false;
        if (!ok) {
            atomicAdd(bad, 1u);
        } else {
            Ext x, y, z, inv = ext_zero();
#pragma unroll
            for (int i = 0; i < 4; i++) x.c[i] = to_monty(xc[i]), y.c[i] = to_monty(yc[i]);
            if (op == 0) z = ext_add(x, y);
            else if (op == 1) z = ext_sub(x, y);
            else if (op == 2) z = ext_mul(x, y);
            else inv = ext_inv(y), z = ext_mul(x, inv);
#pragma unroll
            for (int i = 0; i < 4; i++) col[i] = x.c[i], col[4 + i] = y.c[i], col[8 + i] = z.c[i], col[16 + i] = inv.c[i];
            col[12 + op] = MONTY_ONE;
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_FIELD_EXT_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_field_arith_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_b, const uint32_t* d_c, size_t n, unsigned log_height,
                                          uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27 || (n && (!d_opcode || !d_b || !d_c))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "field_arith_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "field_arith_tracegen");
    hipLaunchKernelGGL(k_field_arith, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_opcode, d_b, d_c, n, N, d_trace, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "field_arith_tracegen (opcode > 3, operand not a field element, or division by zero)");
}

extern "C" int zkhip_field_ext_tracegen(zkhip_ctx* ctx, const uint32_t* d_opcode, const uint32_t* d_x, const uint32_t* d_y, size_t n, unsigned log_height,
                                        uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27 || (n && (!d_opcode || !d_x || !d_y))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "field_ext_tracegen: more records than rows");
    if (((uintptr_t)d_x | (uintptr_t)d_y) & 15u) return set_error(ctx, ZKHIP_ERR_INVALID, "field_ext_tracegen: operands must be 16-byte aligned");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "field_ext_tracegen");
    hipLaunchKernelGGL(k_field_ext, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_opcode, d_x, d_y, n, N, d_trace, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "field_ext_tracegen (opcode > 3, operand not a field element, or division by zero)");
}

// ---- variable range checker (OpenVM VariableRangeCheckerChip): one table for x < 2^bits, bits <= max_bits ------------------------------
// Row 2^bits - 1 + value.  The requesting columns are Montgomery words where they lie; `bits` may be a column or one constant.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_var_range_counts(const uint32_t* __restrict__ values, const uint32_t* __restrict__ bits, uint32_t const_bits,
                                                          size_t n, unsigned max_bits, uint32_t* __restrict__ hist, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk[HOT_SLOTS], hc[HOT_SLOTS];   // hot entries counted in LDS, merged once per workgroup (csrc/hist.hpp)
    hot_init(hk, hc);
    uint32_t n_bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t b = bits ? from_monty(bits[i]) : const_bits, v = from_monty(values[i]);
        if (b > max_bits || v >= (1u << b)) {
            n_bad++;
            continue;
        }
        hot_add(hk, hc, hist, (1u << b) - 1 + v);
    }
    if (n_bad) atomicAdd(bad, n_bad);
    hot_flush(hk, hc, hist);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_var_range_counts_tracegen(zkhip_ctx* ctx, const uint32_t* d_values, const uint32_t* d_bits, uint32_t const_bits, size_t n,
                                               unsigned max_bits, uint32_t* d_counts, int accumulate) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_counts || (n && !d_values) || max_bits > 26) return ZKHIP_ERR_INVALID;
    const size_t T = (size_t)1 << (max_bits + 1);
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "var_range_counts_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!accumulate) ZK_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, T * 4, ctx->stream));
    else if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 0);
    if (n) {
        const unsigned blocks = (unsigned)std::min<size_t>((n + 256 * 8 - 1) / (256 * 8), HOT_MAX_BLOCKS);
        hipLaunchKernelGGL(k_var_range_counts, dim3(blocks), dim3(256), 0, ctx->stream, d_values, d_bits, const_bits, n, max_bits, d_counts, (uint32_t*)flag);
    }
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_counts, T, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "var_range_counts_tracegen (bits > max_bits or value >= 2^bits)");
}

// ---- native CASTF core (OpenVM native CastFCoreAir): a field element below 2^30 to limbs of 8, 8, 8, 6 bits -----------------------------
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_castf(const uint32_t* __restrict__ xs, size_t n, size_t N, uint32_t* __restrict__ trace,
                                               uint32_t* __restrict__ var_range_counts, uint32_t* __restrict__ bad) {
    __shared__ uint32_t hk_v[HOT_SLOTS], hc_v[HOT_SLOTS];
    hot_init(hk_v, hc_v);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (size_t)gridDim.x * 256) {
    uint32_t col[ZKHIP_CASTF_WIDTH] = {};
    if (r < n) {
        const uint32_t x = xs[r];
        if (x >> 30) {
            atomicAdd(bad, 1u);
        } else {
            col[0] = to_monty(x), col[5] = MONTY_ONE;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t limb = (x >> (8 * i)) & 255u;
                col[1 + i] = to_monty(limb);
                hot_add(hk_v, hc_v, var_range_counts, (i < 3 ? 255u : 63u) + limb);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_CASTF_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
    }
    hot_flush(hk_v, hc_v, var_range_counts);
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_castf_tracegen(zkhip_ctx* ctx, const uint32_t* d_x, size_t n, unsigned log_height, uint32_t* d_trace, uint32_t* d_var_range_counts,
                                    unsigned max_bits) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || !d_var_range_counts || log_height > 27 || max_bits < 8 || max_bits > 26 || (n && !d_x)) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height, T = (size_t)1 << (max_bits + 1);
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "castf_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "castf_tracegen");
    const unsigned tb = (unsigned)((T + 255) / 256);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_var_range_counts, T, 0);
    hipLaunchKernelGGL(k_castf, dim3((unsigned)std::min<size_t>((N + 255) / 256, HOT_MAX_BLOCKS)), dim3(256), 0, ctx->stream, d_x, n, N, d_trace, d_var_range_counts, (uint32_t*)flag);
    if (!ctx->tables_canonical) hipLaunchKernelGGL(k_tab_repr, dim3(tb), dim3(256), 0, ctx->stream, d_var_range_counts, T, 1);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "castf_tracegen (value >= 2^30)");
}

// ---- FRI fold chip (air.py fri_fold_air): one arity-2 folding step per row, generated from the step's data -----------------------------
// Record = (e0[4], e1[4], beta[4] canonical, pair index k, log_n_out): x = g^bitrev(k) in the subgroup of order 2^(log_n_out + 1),
// the row holds x^-1 and folded = (e0 + e1) / 2 + beta (e0 - e1) x^-1 / 2 -- the arithmetic a verifier circuit re-does for every
// query and layer of a child proof.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_fri_fold_chip(const uint32_t* __restrict__ e0s, const uint32_t* __restrict__ e1s, const uint32_t* __restrict__ betas,
                                                       const uint32_t* __restrict__ ks, const uint32_t* __restrict__ log_n_outs, size_t n, size_t N,
                                                       uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    uint32_t col[ZKHIP_FRI_FOLD_WIDTH] = {};
    if (r < n) {
        const uint32_t k = ks[r], lo = log_n_outs[r];
        bool ok = lo <= 26 && (lo == 32 || (k >> lo) == 0);
        Ext e0, e1, beta;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t a = e0s[4 * r + i], b = e1s[4 * r + i], c = betas[4 * r + i];
            ok = ok && a < P && b < P && c < P;
            e0.c[i] = to_monty(a), e1.c[i] = to_monty(b), beta.c[i] = to_monty(c);
        }
        if (!ok) {
            atomicAdd(bad, 1u);
        } else {
            const uint32_t xinv = minv(mpow(two_adic_generator(lo + 1), bitrev32(k, lo)));
            const uint32_t half = to_monty((P + 1) / 2);
            const Ext folded = ext_mul_base(ext_add(ext_add(e0, e1), ext_mul_base(ext_mul(beta, ext_sub(e0, e1)), xinv)), half);
#pragma unroll
            for (int i = 0; i < 4; i++) col[i] = e0.c[i], col[4 + i] = e1.c[i], col[8 + i] = beta.c[i], col[13 + i] = folded.c[i];
            col[12] = xinv, col[17] = MONTY_ONE, col[18] = to_monty(k);
        }
    }
#pragma unroll
    for (int q = 0; q < ZKHIP_FRI_FOLD_WIDTH; q++) trace[(size_t)q * N + r] = col[q];
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_fri_fold_chip_tracegen(zkhip_ctx* ctx, const uint32_t* d_e0, const uint32_t* d_e1, const uint32_t* d_beta, const uint32_t* d_k,
                                            const uint32_t* d_log_n_out, size_t n, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27 || (n && (!d_e0 || !d_e1 || !d_beta || !d_k || !d_log_n_out))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "fri_fold_chip_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "fri_fold_chip_tracegen");
    hipLaunchKernelGGL(k_fri_fold_chip, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_e0, d_e1, d_beta, d_k, d_log_n_out, n, N, d_trace,
                       (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "fri_fold_chip_tracegen (a value that is not a field element, or an index outside its layer)");
}

// ---- domain-point chip (air.py domain_point_air): x^-1 of a FRI pair from the bits of its index ------------------------------------------
// Row = (k, its 26 bits, the running product acc_j = prod_{i <= j} (W_i^-1)^bit_i, multiplicity); W_i generates the subgroup of
// order 2^(i + 2), so the product is 1 / g^bitrev(k) for the layer of ANY size that holds pair k.
namespace zk {
namespace {
__global__ __launch_bounds__(256) void k_domain_point(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ mults, size_t n, size_t N,
                                                      uint32_t* __restrict__ trace, uint32_t* __restrict__ bad) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    const bool real = r < n;
    const uint32_t k = real ? ks[r] : 0u, m = real ? mults[r] : 0u;
    if (real && ((k >> ZKHIP_DOMAIN_POINT_BITS) || m >= P)) atomicAdd(bad, 1u);
    trace[r] = to_monty(k & ((1u << ZKHIP_DOMAIN_POINT_BITS) - 1));
    uint32_t acc = MONTY_ONE;
#pragma unroll 1
    for (unsigned j = 0; j < ZKHIP_DOMAIN_POINT_BITS; j++) {
        const uint32_t bit = (k >> j) & 1u;
        if (bit) acc = mmul(acc, minv(two_adic_generator(j + 2)));
        trace[(size_t)(1 + j) * N + r] = bit ? MONTY_ONE : 0u;
        trace[(size_t)(1 + ZKHIP_DOMAIN_POINT_BITS + j) * N + r] = acc;
    }
    trace[(size_t)(1 + 2 * ZKHIP_DOMAIN_POINT_BITS) * N + r] = m < P ? to_monty(m) : 0u;
}
}  // namespace
}  // namespace zk

extern "C" int zkhip_domain_point_tracegen(zkhip_ctx* ctx, const uint32_t* d_k, const uint32_t* d_mult, size_t n, unsigned log_height, uint32_t* d_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_trace || log_height > 27 || (n && (!d_k || !d_mult))) return ZKHIP_ERR_INVALID;
    const size_t N = (size_t)1 << log_height;
    if (n > N) return set_error(ctx, ZKHIP_ERR_INVALID, "domain_point_tracegen: more records than rows");
    void* flag = nullptr;
    ZK_TRY(tracegen_flag(ctx, &flag));
    KernelScope ks(ctx, "domain_point_tracegen");
    hipLaunchKernelGGL(k_domain_point, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_k, d_mult, n, N, d_trace, (uint32_t*)flag);
    ZK_HIP_CHECK(ctx, hipGetLastError());
    return finish_counts(ctx, flag, "domain_point_tracegen (a pair index of more than 26 bits or a multiplicity that is not a field element)");
}
