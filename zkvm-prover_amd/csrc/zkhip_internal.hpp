// zkhip_internal.hpp -- context, error handling, launch/profiling helpers shared by the
// translation units of libzkhip.so.  Not part of the ABI (include/zkhip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "../../include/zkhip.h"
#include "babybear.hpp"

namespace zk {

struct ProfileEntry {
    uint64_t launches = 0;
    double total_ms = 0;
};

struct PendingEvent {
    std::string name;
    hipEvent_t start, stop;
};

}  // namespace zk

struct zkhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // where all work is issued: the context's own stream unless zkhip_set_stream gave another
    hipStream_t own_stream = nullptr;   // created with the context (non-blocking): several contexts on one GPU overlap by default
    std::string last_error;
    // the shared lookup-count tables handed to trace generators are CANONICAL counts for now (zkhip_tables_canonical): no generator converts
    // them from / to Montgomery form around its increments; the caller converts once when every chip of the segment has counted
    bool tables_canonical = false;
    // twiddle tables: w^e (fwd) and w^-e (inv) for w = two_adic_generator(tw_log), e < 2^(tw_log-1)
    unsigned tw_log = 0;
    uint32_t* d_tw_fwd = nullptr;
    uint32_t* d_tw_inv = nullptr;
    // grow-only scratch buffers (the pool never returns pages to the driver while the ctx lives,
    // like the VPMM pool it replaces -- AGENTS.md:136)
    void* scratch[8] = {};
    size_t scratch_bytes[8] = {};
    // profiling
    bool profiling = false;
    std::vector<zk::PendingEvent> pending;
    std::map<std::string, zk::ProfileEntry> stats;
    int cu_count = 256;
    // trace commit as a pipeline (prover.hip): the LDE of column block k+1 runs on `side_stream` while the row sponge
    // absorbs block k on the main stream.  commit_parts = number of column blocks (0/1 = off).
    hipStream_t side_stream = nullptr;
    hipEvent_t pipe_ev[12] = {};
    unsigned commit_parts = 0;
    // CU partition of the pipelined commit (zkhip_set_cu_partition): with side_cus > 0 the side stream is created with a CU
    // mask of `side_cus` CUs (the memory-bound LDE) and the row sponge of the pipeline runs on `hash_stream`, masked to the
    // remaining CUs, so that neither queue waits behind the other's workgroups.  0 = unmasked streams.
    unsigned side_cus = 0;
    hipStream_t hash_stream = nullptr;
    // the constraint kernels of one proof are independent of each other: with zkhip_config.quot_streams = k > 0 the compiled ones go round-robin
    // over k further streams of the context (forked from and joined to the proof's stream with events; created at first use)
    hipStream_t quot_streams[4] = {};
    hipEvent_t quot_fork = nullptr, quot_join[4] = {};
    // trace generators check their records on the device and normally report at once (one stream synchronisation per call); with
    // deferred checks (zkhip_tracegen_defer_checks) the bad-record counts are summed on the device and read once (zkhip_tracegen_check)
    bool defer_tracegen_checks = false;
    uint32_t* d_deferred_bad = nullptr;
    zkhip_config cfg{};   // every switch of the library (include/zkhip.h zkhip_config); the environment is read once, in zkhip_config_default
    // pinned staging of the transcript's host sponge (csrc/transcript.hip: long absorptions run on the host's vector unit)
    void* h_sponge = nullptr;
    size_t h_sponge_bytes = 0;
    // the scale tables of a four-step LDE (ntt.hip: column / row / rho powers of the coset shifts) depend on (log_n, added_bits, shift) only:
    // made once per context and read-only afterwards (a segment proof extends ~20 heights: a launch each, every proof, before round 5)
    std::map<uint64_t, uint32_t*> lde_tables;
};

namespace zk {

int set_error(zkhip_ctx* ctx, int code, const std::string& msg);
zkhip_config process_config();   // what contexts start from, and what the context-less entry points (the circuit's witness) use
int ensure_twiddles(zkhip_ctx* ctx, unsigned log_n);
int get_scratch(zkhip_ctx* ctx, int slot, size_t bytes, void** out);
void profile_begin(zkhip_ctx* ctx, const char* name);
void profile_end(zkhip_ctx* ctx);
int profile_flush(zkhip_ctx* ctx);
// end of a trace generator: `flag` = device counter of bad records of this call; reports now, or adds it to the deferred total
int tracegen_flag(zkhip_ctx* ctx, void** flag);
int tracegen_finish(zkhip_ctx* ctx, void* flag, const std::string& what);

// every entry point that takes a context runs on the context's device, whatever device the calling thread had current
// (two contexts on different GPUs in one process; a context handed to another thread)
#define ZK_BIND_DEVICE(ctx)                            \
    do {                                               \
        if (ctx) (void)hipSetDevice((ctx)->device);    \
    } while (0)

#define ZK_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return zk::set_error((ctx), ZKHIP_ERR_HIP,                                       \
                                 std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)

#define ZK_TRY(expr)               \
    do {                           \
        int _rc = (expr);          \
        if (_rc != ZKHIP_OK) return _rc; \
    } while (0)

// kernel attributes are per DEVICE (hipFuncSetAttribute acts on the current one): set once per device, by whichever thread gets there first
// (two threads racing set them twice, which is harmless; the bit is published after the calls)
struct DeviceOnce {
    std::atomic<uint64_t> done{0};
    bool need(int device) const { return !((done.load(std::memory_order_acquire) >> (device & 63)) & 1); }
    void mark(int device) { done.fetch_or(1ull << (device & 63), std::memory_order_release); }
};

// RAII-ish scope for per-kernel timing: records HIP events on the ctx stream when profiling is on
struct KernelScope {
    zkhip_ctx* ctx;
    KernelScope(zkhip_ctx* c, const char* name) : ctx(c) { profile_begin(ctx, name); }
    ~KernelScope() { profile_end(ctx); }
};

// ---- kernels / stages implemented across the .hip files --------------------------------------
// ntt.hip
int ntt_dif_inplace(zkhip_ctx* ctx, const uint32_t* src, size_t src_stride, uint32_t* dst, size_t dst_stride,
                    unsigned log_n, size_t width, unsigned log_sub, bool inverse);
int ntt_batch(zkhip_ctx* ctx, uint32_t* d_mat, unsigned log_n, size_t width, size_t stride, bool inverse,
              bool bitrev_out);
int lde_batch(zkhip_ctx* ctx, const uint32_t* d_in, size_t in_stride, uint32_t* d_out, size_t out_stride,
              unsigned log_n, unsigned added_bits, size_t width, uint32_t shift_monty);
// the same for n_cols columns given by per-column device pointer tables (columns of several matrices, log_n >= 12)
int lde_batch_cols(zkhip_ctx* ctx, const uint32_t* const* d_src_cols, uint32_t* const* d_dst_cols, size_t n_cols,
                   unsigned log_n, unsigned added_bits, uint32_t shift_monty);
int convert_repr(zkhip_ctx* ctx, uint32_t* d, size_t n, bool to_monty);
struct BitrevSeg {
    const uint32_t* src;
    uint32_t* dst;
    size_t src_stride, dst_stride;
    uint32_t log_n, width, first_block, pad;
};
uint32_t ntt_bitrev_copy_blocks(unsigned log_n, uint32_t width);
int ntt_bitrev_copy_multi(zkhip_ctx* ctx, const BitrevSeg* d_tiled, uint32_t n_tiled, uint32_t blocks_tiled,
                          const BitrevSeg* d_small, uint32_t n_small, uint32_t blocks_small);
int ntt_bitrev_copy(zkhip_ctx* ctx, const uint32_t* src, size_t src_stride, uint32_t* dst, size_t dst_stride,
                    unsigned log_n, size_t width);

// merkle.hip
struct TreeLevelInject {
    // matrices injected at a level (or hashed at the leaf level): column pointer table on device
    const uint32_t** d_cols = nullptr;  // device array of column base pointers
    uint32_t n_cols = 0;
};

}  // namespace zk

struct zkhip_tree {
    unsigned log_height = 0;
    std::vector<zkhip_matrix> mats;        // caller order
    uint32_t* d_digests = nullptr;         // all layers, layer l at offset layer_off[l] (in digests)
    std::vector<size_t> layer_off;         // in units of 8-word digests
    void* d_colptrs = nullptr;             // backing store of the column pointer tables
    bool owns_digests = true;
    size_t total_width = 0;
    std::vector<size_t> level_off, level_cnt;  // per log-height: slice of the pointer table
    size_t shifts_off = 0;                     // byte offset of the u32 shift table in d_colptrs
};

namespace zk {
int merkle_commit(zkhip_ctx* ctx, const zkhip_matrix* mats, size_t n_mats, zkhip_tree** out);
int merkle_plan(zkhip_ctx* ctx, const zkhip_matrix* mats, size_t n_mats, uint32_t* d_digests, zkhip_tree** out);
int merkle_plan_leaves(zkhip_ctx* ctx, unsigned log_height, uint32_t* d_digests, zkhip_tree** out);
int merkle_build(zkhip_ctx* ctx, zkhip_tree* t, bool leaves_ready);
// one step of the leaf sponge over columns [col_begin, col_end) of the tallest matrices (col_begin a multiple of 8):
// `first` starts from the zero state, otherwise the 16-word state of every row is read from d_state ([16][rows]);
// `last` writes the leaf digests, otherwise the state goes back to d_state
int merkle_leaves_part(zkhip_ctx* ctx, zkhip_tree* t, size_t col_begin, size_t col_end, bool first, bool last, uint32_t* d_state);
size_t merkle_digest_count(unsigned log_height);
// diagnosis: recomputes every plain layer of the tree; d_report[0] += mismatching nodes, d_report[1] = min(layer << 24 | index)
int merkle_check_tree(zkhip_ctx* ctx, const zkhip_tree* t, uint32_t* d_report);
// gathers openings for n leaf indices (device array of u32 indices) into a device buffer (canonical)
int merkle_open_device(zkhip_ctx* ctx, const zkhip_tree* tree, const uint32_t* d_indices, unsigned index_shift,
                       size_t n, uint32_t* d_out, size_t out_pitch_words);

// fri.hip
int permute_batch(zkhip_ctx* ctx, uint32_t* d_states, size_t n);
int fri_fold(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, unsigned log_n_out, const uint32_t* d_beta,
             const uint32_t* d_add, /* optional: out[i] += beta^2 * add[i] */ bool has_add);

// sumcheck.hip
// out[i] = (num ? num[i] : 1) / in[i] over n extension elements; in == out allowed
int launch_batch_inverse(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n, const uint32_t* d_num);
// in-place inclusive prefix sum of n extension elements
int ext_inclusive_scan(zkhip_ctx* ctx, uint32_t* d_data, size_t n);
// several independent scans in three launches: segment i = n extension elements at `data`, its workgroup totals at `totals`
// (scan_blocks_of(n) extension elements), occupying blocks [first_block, first_block + n_blocks) of the flattened grid
struct ScanSeg {
    uint32_t* data;
    uint32_t* totals;
    uint64_t n;
    uint32_t first_block, n_blocks;
};
uint32_t scan_blocks_of(size_t n);
int ext_inclusive_scan_multi(zkhip_ctx* ctx, const ScanSeg* d_segs, uint32_t n_seg, uint32_t total_blocks, bool any_multi_block);
}  // namespace zk
