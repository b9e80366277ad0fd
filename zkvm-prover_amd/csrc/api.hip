#include <sched.h>
#include <cstring>
#include <thread>
#include <dlfcn.h>
#include <sys/stat.h>
#include <mutex>
// api.hip -- the extern "C" surface of libzkhip.so (include/zkhip.h): context, memory pool,
// profiling, and thin argument-checking wrappers around the stage launchers.
#include <string.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "transcript.hpp"
#include "zkhip_internal.hpp"

namespace zk {

int set_error(zkhip_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return code;
}

int get_scratch(zkhip_ctx* ctx, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->scratch[slot]) {
            ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            ZK_HIP_CHECK(ctx, hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        if (hipMalloc(&ctx->scratch[slot], bytes) != hipSuccess)
            return set_error(ctx, ZKHIP_ERR_NOMEM, "scratch allocation of " + std::to_string(bytes) + " bytes failed");
        ctx->scratch_bytes[slot] = bytes;
    }
    *out = ctx->scratch[slot];
    return ZKHIP_OK;
}

void profile_begin(zkhip_ctx* ctx, const char* name) {
    if (!ctx->profiling) return;
    PendingEvent pe;
    pe.name = name;
    if (hipEventCreate(&pe.start) != hipSuccess || hipEventCreate(&pe.stop) != hipSuccess) return;
    (void)hipEventRecord(pe.start, ctx->stream);
    ctx->pending.push_back(pe);
}
void profile_end(zkhip_ctx* ctx) {
    if (!ctx->profiling || ctx->pending.empty()) return;
    (void)hipEventRecord(ctx->pending.back().stop, ctx->stream);
}
int profile_flush(zkhip_ctx* ctx) {
    for (auto& pe : ctx->pending) {
        (void)hipEventSynchronize(pe.stop);
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) {
            auto& st = ctx->stats[pe.name];
            st.launches++;
            st.total_ms += ms;
        }
        (void)hipEventDestroy(pe.start);
        (void)hipEventDestroy(pe.stop);
    }
    ctx->pending.clear();
    return ZKHIP_OK;
}

int permute_batch(zkhip_ctx* ctx, uint32_t* d_states, size_t n);

// Power-on self-test of the hand-scheduled field primitives.  red_2p / msub issue
// v_sub_co + v_cndmask back to back from inline asm (no compiler-inserted wait states between the
// VCC write and its use: measurably faster in the LDS-bound NTT kernels); this kernel checks them on
// the running device against plain C arithmetic for boundary and pseudo-random operands, so a part
// that needed software wait states there would be refused at context creation, not mis-prove.
__global__ void k_selftest(uint32_t* bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u;
    const uint32_t edge[8] = {0u, 1u, P - 1u, P, P + 1u, 2u * P - 1u, 0x7fffffffu, 0x80000000u};
    if (t < 8) x = edge[t];
    const uint32_t x2 = x % (2u * P);
    const uint32_t want_red = x2 >= P ? x2 - P : x2;
    const uint32_t a = x % P, b = (x * 40503u + 77u) % P;
    const uint32_t want_sub = a >= b ? a - b : a + P - b;
    const uint32_t want_add = (uint32_t)(((uint64_t)a + b) % P);
    const uint32_t want_mul = (uint32_t)(((uint64_t)from_monty(a) * from_monty(b)) % P);
    if (red_2p(x2) != want_red || msub(a, b) != want_sub || madd(a, b) != want_add ||
        from_monty(mmul(a, b)) != want_mul)
        atomicAdd(bad, 1u);
}

static int selftest(zkhip_ctx* ctx) {
    uint32_t* d_bad = nullptr;
    uint32_t h_bad = 0;
    if (hipMalloc(&d_bad, 4) != hipSuccess) return ZKHIP_ERR_NOMEM;
    // everything on the context's stream (a non-blocking stream is not ordered against the legacy default stream)
    hipError_t e = hipMemsetAsync(d_bad, 0, 4, ctx->stream);
    hipLaunchKernelGGL(k_selftest, dim3(64), dim3(256), 0, ctx->stream, d_bad);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return ZKHIP_ERR_HIP;
    return h_bad == 0 ? ZKHIP_OK : ZKHIP_ERR_HIP;
}

__global__ void k_accumulate_flag(const uint32_t* flag, uint32_t* total) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && *flag) atomicAdd(total, *flag);
}
// The error counter a trace generator's kernel increments.  While the checks are deferred (zkhip_defer_tracegen_checks: a segment's ~20
// generators, one verdict at the end) it IS the context's accumulator: no memset before the kernel, no accumulate launch after it -- two tiny
// launches per generator less (40 - 100 of a segment proof's 490 - 830, round 6).  Otherwise a scratch word, zeroed here.
int tracegen_flag(zkhip_ctx* ctx, void** flag) {
    if (ctx->defer_tracegen_checks && ctx->d_deferred_bad) {
        *flag = ctx->d_deferred_bad;
        return ZKHIP_OK;
    }
    ZK_TRY(get_scratch(ctx, 2, 16, flag));
    ZK_HIP_CHECK(ctx, hipMemsetAsync(*flag, 0, 4, ctx->stream));
    return ZKHIP_OK;
}
int tracegen_finish(zkhip_ctx* ctx, void* flag, const std::string& what) {
    if (ctx->defer_tracegen_checks && ctx->d_deferred_bad && flag == (void*)ctx->d_deferred_bad) return ZKHIP_OK;   // (counted in place)
    if (ctx->defer_tracegen_checks && ctx->d_deferred_bad) {
        hipLaunchKernelGGL(k_accumulate_flag, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)flag, ctx->d_deferred_bad);
        ZK_HIP_CHECK(ctx, hipGetLastError());
        return ZKHIP_OK;
    }
    uint32_t h_bad = 0;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(&h_bad, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_bad) return set_error(ctx, ZKHIP_ERR_INVALID, what + ": " + std::to_string(h_bad) + " bad records");
    return ZKHIP_OK;
}

}  // namespace zk

using namespace zk;

extern "C" {

uint32_t zkhip_version(void) { return (0u << 16) | 2u; }  // 0.2: zkhip_air gained prep_trace / prep_commit

int zkhip_ctx_create(int device, zkhip_ctx** out) {
    if (!out) return ZKHIP_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ZKHIP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return ZKHIP_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ZKHIP_ERR_NO_DEVICE;
    zkhip_ctx* ctx = new zkhip_ctx();
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        // the code object only carries gfx950 ISA: fail loudly instead of at first launch
        delete ctx;
        return ZKHIP_ERR_NO_DEVICE;
    }
    // one stream per context: work of different contexts on one GPU (several proofs in flight) must not serialise on the
    // legacy default stream
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return ZKHIP_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (selftest(ctx) != ZKHIP_OK) {
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return ZKHIP_ERR_HIP;
    }
    ctx->cfg = zk::process_config();
    // pipelined trace commit (prover.hip): k column blocks, 0 / 1 = off; zkhip_set_commit_pipeline overrides
    ctx->commit_parts = std::min(ctx->cfg.commit_parts, 8u);
    ctx->side_cus = ctx->cfg.side_cus >= (unsigned)ctx->cu_count ? 0u : ctx->cfg.side_cus;
    *out = ctx;
    return ZKHIP_OK;
}

unsigned zkhip_host_cpus(void) {
    static const unsigned cached = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
        auto read2 = [](const char* path, long long* a, long long* b) {   // "<a> [<b>]"; a may be the word "max" (-> -1)
            FILE* f = std::fopen(path, "r");
            if (!f) return 0;
            char w[32] = {};
            long long y = 0;
            const int got = std::fscanf(f, "%31s %lld", w, &y);
            std::fclose(f);
            if (got < 1) return 0;
            *a = std::strcmp(w, "max") == 0 ? -1 : std::atoll(w), *b = y;
            return got;
        };
        long long q = -1, per = 0;
        if (read2("/sys/fs/cgroup/cpu.max", &q, &per) == 2) {
            if (q > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, q / per));
        } else if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q, &per) >= 1 && q > 0) {
            long long p1 = 0, dummy = 0;
            if (read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &p1, &dummy) >= 1 && p1 > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, q / p1));
        }
        return n;
    }();
    return cached;
}

// ---- configuration: the ONE place the library reads its ZKHIP_* environment overrides ----
void zkhip_config_default(zkhip_config* c) {
    if (!c) return;
    // (an override outside the range zkhip_ctx_set_config accepts is clamped into it here, so that a get_config / set_config round trip of
    // the defaults never fails: ADVICE round 5)
    auto num = [](const char* name, uint32_t dflt, uint32_t lo = 0, uint32_t hi = 0xffffffffu) {
        if (!getenv(name)) return dflt;
        const long long v = atoll(getenv(name));
        return (uint32_t)std::min<long long>(hi, std::max<long long>(lo, v));
    };
    memset(c, 0, sizeof *c);
    c->host_sponge = getenv("ZKHIP_NO_HOST_SPONGE") ? 0 : 1;
    c->host_sponge_min_words = num("ZKHIP_HOST_SPONGE_MIN_WORDS", 8192u);
    c->jit = getenv("ZKHIP_NO_JIT") ? 0 : getenv("ZKHIP_FORCE_JIT") ? 2 : 1;
    c->jit_min_log_work = num("ZKHIP_JIT_MIN_LOG_WORK", 26u, 0, 62);
    c->quot_slices = getenv("ZKHIP_NO_QUOT_SLICES") ? 0 : 1;
    c->grind_sweep_shift = num("ZKHIP_GRIND_SWEEP_SHIFT", 0u, 0, 8);
    c->coop_max_log = num("ZKHIP_COOP_MAX_LOG", 15u, 0, 27), c->coop_inj_max_log = num("ZKHIP_COOP_INJ_MAX_LOG", 15u, 0, 27);
    c->top_max_log = num("ZKHIP_TOP_MAX_LOG", 6u, 0, 8);   // (measured 8 / 7 / 6 / 5 / 4 on the guest flow: 987 / 936 / 911 / 924 / 931 ms of segment proving with one lane)
    c->commit_parts = num("ZKHIP_COMMIT_PARTS", 0u, 0, 8), c->side_cus = num("ZKHIP_SIDE_CUS", 0u);
    c->witness_threads = num("ZKHIP_WITNESS_THREADS", 0u);
    c->pin_witness = getenv("ZKHIP_NO_PIN_WITNESS") ? 0 : 1;
    c->parallel_queries = getenv("ZKHIP_RECURSION_SERIAL_QUERIES") ? 0 : 1;
    c->self_check = getenv("ZKHIP_SELF_CHECK") ? 1 : 0;
#ifdef ZKHIP_TEST_KERNELS
    c->tree_store_early = getenv("ZKHIP_TREE_STORE_EARLY") ? 1 : 0;   // (libzkhip_test.so only: the shipped library has no such kernels)
#endif
    c->hash_block = num("ZKHIP_HASH_BLOCK", 256u, 64, 768) / 64 * 64;
    c->coop_fused = getenv("ZKHIP_NO_COOP_FUSED") ? 0 : 1;   // (test only: the round-4 bodies of the fused tree kernels)
    c->rows_in_bulk = getenv("ZKHIP_NO_ROWS_IN_BULK") ? 0 : 1;
    c->rows_coop_max_log = num("ZKHIP_ROWS_COOP_MAX_LOG", 15u, 0, 27);
    c->ntt_log_lanes = num("ZKHIP_NTT_LOG_LANES", 10u, 8, 10);
    c->quot_streams = num("ZKHIP_QUOT_STREAMS", 0u, 0, 4);   // (measured: + 7 % for ONE lane of the mixed guest, nothing with three lanes: docs/round5_b.md 6)
    // compiled constraint kernels across processes: the variable if set (empty = none), else `jit_cache` beside this library if it exists
    std::string dir;
    if (const char* e = getenv("ZKHIP_JIT_CACHE_DIR")) {
        dir = e;
    } else {
        Dl_info info;
        if (dladdr((const void*)&zkhip_config_default, &info) && info.dli_fname) {
            std::string p = info.dli_fname;
            const size_t slash = p.rfind('/');
            p = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/jit_cache";
            struct stat st;
            if (stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) dir = p;
        }
    }
    if (dir.size() < sizeof c->jit_cache_dir) memcpy(c->jit_cache_dir, dir.c_str(), dir.size() + 1);
}
extern "C++" {
namespace zk {
static zkhip_config g_process_config;
static std::once_flag g_process_config_once;
static std::mutex g_process_config_mu;
// (by value, under the lock: zkhip_set_process_config may run on another thread -- ADVICE round 4)
zkhip_config process_config() {
    std::call_once(g_process_config_once, [] { zkhip_config_default(&g_process_config); });
    std::lock_guard<std::mutex> lk(g_process_config_mu);
    return g_process_config;
}
}  // namespace zk
}  // extern "C++"
int zkhip_set_process_config(const zkhip_config* cfg) {
    if (!cfg) return ZKHIP_ERR_INVALID;
    (void)zk::process_config();
    std::lock_guard<std::mutex> lk(zk::g_process_config_mu);
    zk::g_process_config = *cfg;
    return ZKHIP_OK;
}
// While on, the trace generators take the shared lookup-count tables handed to them (8-bit bitwise table, range-tuple table, range table)
// as CANONICAL counts and leave them so: a segment's ~20 generators each converted those tables from Montgomery form and back around their
// increments (45 launches per segment).  The caller zeroes the tables, switches this on, runs the generators, switches it off and converts
// each table once (zkhip_to_monty).
int zkhip_tables_canonical(zkhip_ctx* ctx, int on) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ctx->tables_canonical = on != 0;
    return ZKHIP_OK;
}

int zkhip_has_test_kernels(void) {
#ifdef ZKHIP_TEST_KERNELS
    return 1;
#else
    return 0;
#endif
}

int zkhip_ctx_get_config(zkhip_ctx* ctx, zkhip_config* out) {
    if (!ctx || !out) return ZKHIP_ERR_INVALID;
    *out = ctx->cfg;
    return ZKHIP_OK;
}
int zkhip_ctx_set_config(zkhip_ctx* ctx, const zkhip_config* cfg) {
    if (!ctx || !cfg) return ZKHIP_ERR_INVALID;
    // every field is applied or refused: nothing is silently clamped at its place of use
    if (cfg->jit < 0 || cfg->jit > 2 || cfg->coop_max_log > 27 || cfg->coop_inj_max_log > 27 || cfg->rows_coop_max_log > 27 || cfg->ntt_log_lanes < 8 || cfg->ntt_log_lanes > 10 || cfg->quot_streams > 4 || cfg->jit_min_log_work > 62 || cfg->top_max_log > 8 || cfg->grind_sweep_shift > 8 ||
        cfg->commit_parts > 8 || cfg->side_cus >= (unsigned)ctx->cu_count || cfg->hash_block < 64 || cfg->hash_block > 768 || cfg->hash_block % 64)
        return set_error(ctx, ZKHIP_ERR_INVALID, "zkhip_ctx_set_config: field out of range (jit 0..2, coop_* <= 27, rows_coop_max_log <= 27, ntt_log_lanes 8..10, quot_streams <= 4, jit_min_log_work <= 62, top_max_log <= 8, grind_sweep_shift <= 8, commit_parts <= 8, side_cus < CUs, hash_block a multiple of 64 in 64..768)");
#ifndef ZKHIP_TEST_KERNELS
    if (cfg->tree_store_early)
        return set_error(ctx, ZKHIP_ERR_INVALID, "zkhip_ctx_set_config: tree_store_early names TEST kernels that this library was built without (libzkhip_test.so holds them)");
#endif
    ZK_TRY(zkhip_set_cu_partition(ctx, cfg->side_cus));   // (drops the masked side streams if the partition changes)
    ctx->cfg = *cfg;
    ctx->cfg.jit_cache_dir[sizeof ctx->cfg.jit_cache_dir - 1] = 0;
    ctx->commit_parts = cfg->commit_parts;
    return ZKHIP_OK;
}

void zkhip_ctx_destroy(zkhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    profile_flush(ctx);
    if (ctx->d_tw_fwd) (void)hipFree(ctx->d_tw_fwd);
    if (ctx->d_tw_inv) (void)hipFree(ctx->d_tw_inv);
    for (auto& kv : ctx->lde_tables) (void)hipFree(kv.second);
    for (int i = 0; i < 8; i++)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->hash_stream) (void)hipStreamDestroy(ctx->hash_stream);
    for (auto& q : ctx->quot_streams)
        if (q) (void)hipStreamDestroy(q);
    if (ctx->quot_fork) (void)hipEventDestroy(ctx->quot_fork);
    for (auto& e : ctx->quot_join)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_deferred_bad) (void)hipFree(ctx->d_deferred_bad);
    if (ctx->h_sponge) (void)hipHostFree(ctx->h_sponge);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    for (auto& e : ctx->pipe_ev)
        if (e) (void)hipEventDestroy(e);
    delete ctx;
}

int zkhip_set_commit_pipeline(zkhip_ctx* ctx, unsigned parts) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || parts > 8) return ZKHIP_ERR_INVALID;
    ctx->commit_parts = parts;
    return ZKHIP_OK;
}

int zkhip_tracegen_defer_checks(zkhip_ctx* ctx, int on) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (on && !ctx->d_deferred_bad) {
        ZK_HIP_CHECK(ctx, hipMalloc(&ctx->d_deferred_bad, 4));
        ZK_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_deferred_bad, 0, 4, ctx->stream));
    }
    ctx->defer_tracegen_checks = on != 0;
    return ZKHIP_OK;
}

int zkhip_tracegen_check(zkhip_ctx* ctx) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (!ctx->d_deferred_bad) return ZKHIP_OK;
    uint32_t h_bad = 0;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(&h_bad, ctx->d_deferred_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_deferred_bad, 0, 4, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_bad) return set_error(ctx, ZKHIP_ERR_INVALID, "trace generation: " + std::to_string(h_bad) + " bad records since the last check (run without deferred checks to see which generator refuses them)");
    return ZKHIP_OK;
}

int zkhip_set_cu_partition(zkhip_ctx* ctx, unsigned side_cus) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || side_cus >= (unsigned)ctx->cu_count) return ZKHIP_ERR_INVALID;
    if (side_cus != ctx->side_cus) {
        // the masked streams are created lazily by the next pipelined commit
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (hipStream_t* s : {&ctx->side_stream, &ctx->hash_stream}) {
            if (!*s) continue;
            ZK_HIP_CHECK(ctx, hipStreamSynchronize(*s));
            (void)hipStreamDestroy(*s);
            *s = nullptr;
        }
        ctx->side_cus = side_cus;
    }
    return ZKHIP_OK;
}

const char* zkhip_last_error(const zkhip_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "no context"; }

int zkhip_set_stream(zkhip_ctx* ctx, void* hip_stream) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    // work already queued on the previous stream stays ordered before what follows on the new one
    if (ctx->stream != (hipStream_t)hip_stream) ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    return ZKHIP_OK;
}
int zkhip_sync(zkhip_ctx* ctx) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_malloc(zkhip_ctx* ctx, size_t bytes, void** dptr) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !dptr) return ZKHIP_ERR_INVALID;
    if (hipMalloc(dptr, bytes ? bytes : 16) != hipSuccess)
        return set_error(ctx, ZKHIP_ERR_NOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
    return ZKHIP_OK;
}
int zkhip_free(zkhip_ctx* ctx, void* dptr) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (!dptr) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP_CHECK(ctx, hipFree(dptr));
    return ZKHIP_OK;
}
int zkhip_h2d(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // src may be pageable and freed by the caller
    return ZKHIP_OK;
}
int zkhip_host_alloc(zkhip_ctx* ctx, size_t bytes, void** hptr) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !hptr) return ZKHIP_ERR_INVALID;
    if (hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess)
        return set_error(ctx, ZKHIP_ERR_NOMEM, "hipHostMalloc of " + std::to_string(bytes) + " bytes failed");
    return ZKHIP_OK;
}
int zkhip_host_free(zkhip_ctx* ctx, void* hptr) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (!hptr) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP_CHECK(ctx, hipHostFree(hptr));
    return ZKHIP_OK;
}
int zkhip_h2d_async(zkhip_ctx* ctx, void* dst, const void* src_pinned, size_t bytes) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || (bytes && (!dst || !src_pinned))) return ZKHIP_ERR_INVALID;
    if (bytes) ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_zero(zkhip_ctx* ctx, void* dptr, size_t bytes) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || (bytes && !dptr)) return ZKHIP_ERR_INVALID;
    if (bytes) ZK_HIP_CHECK(ctx, hipMemsetAsync(dptr, 0, bytes, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_d2h(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}
int zkhip_to_monty(zkhip_ctx* ctx, uint32_t* d, size_t n) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    return convert_repr(ctx, d, n, true);
}
int zkhip_from_monty(zkhip_ctx* ctx, uint32_t* d, size_t n) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    return convert_repr(ctx, d, n, false);
}

int zkhip_ntt_batch(zkhip_ctx* ctx, uint32_t* d_mat, unsigned log_n, size_t width, size_t stride, int inverse,
                    int bitrev_out) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_mat) return ZKHIP_ERR_INVALID;
    return ntt_batch(ctx, d_mat, log_n, width, stride, inverse != 0, bitrev_out != 0);
}
int zkhip_lde_batch(zkhip_ctx* ctx, const uint32_t* d_in, size_t in_stride, uint32_t* d_out, size_t out_stride,
                    unsigned log_n, unsigned added_bits, size_t width, uint32_t shift) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_in || !d_out) return ZKHIP_ERR_INVALID;
    if (shift == 0 || shift >= P) return set_error(ctx, ZKHIP_ERR_INVALID, "shift must be in [1,p)");
    return lde_batch(ctx, d_in, in_stride, d_out, out_stride, log_n, added_bits, width, to_monty(shift));
}

int zkhip_poseidon2_permute_batch(zkhip_ctx* ctx, uint32_t* d_states, size_t n) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_states) return ZKHIP_ERR_INVALID;
    return permute_batch(ctx, d_states, n);
}

int zkhip_merkle_commit(zkhip_ctx* ctx, const zkhip_matrix* mats, size_t n_mats, zkhip_tree** tree,
                        uint32_t* root_out) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !mats || !tree) return ZKHIP_ERR_INVALID;
    ZK_TRY(merkle_commit(ctx, mats, n_mats, tree));
    if (root_out) {
        uint32_t tmp[8];
        ZK_TRY(zkhip_d2h(ctx, tmp, zkhip_tree_root_device(*tree), sizeof tmp));
        for (int i = 0; i < 8; i++) root_out[i] = from_monty(tmp[i]);
    }
    return ZKHIP_OK;
}
// Rebuilds a committed tree in place: the same launches as the commit on the same digest store -- no allocation, no synchronisation.
int zkhip_merkle_rebuild(zkhip_ctx* ctx, zkhip_tree* tree) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !tree || tree->mats.empty()) return ZKHIP_ERR_INVALID;
    return merkle_build(ctx, tree, false);
}
// Every plain layer of the tree against the compression of its children, recomputed on the device through the plain (one lane per node)
// permutation: *n_bad = nodes that differ, *first = (layer << 24 | index) of the first one (0xffffffff if none).  Synchronises.
int zkhip_tree_check(zkhip_ctx* ctx, const zkhip_tree* tree, uint32_t* n_bad, uint32_t* first) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !tree || !n_bad) return ZKHIP_ERR_INVALID;
    void* d_rep = nullptr;
    ZK_TRY(get_scratch(ctx, 2, 16, &d_rep));
    const uint32_t init[2] = {0, 0xffffffffu};
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_rep, init, 8, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // (`init` is a local)
    ZK_TRY(merkle_check_tree(ctx, tree, (uint32_t*)d_rep));
    uint32_t rep[2];
    ZK_TRY(zkhip_d2h(ctx, rep, d_rep, 8));
    *n_bad = rep[0];
    if (first) *first = rep[1];
    return ZKHIP_OK;
}
const uint32_t* zkhip_tree_root_device(const zkhip_tree* t) {
    return t ? t->d_digests + t->layer_off[t->log_height] * 8 : nullptr;
}
unsigned zkhip_tree_log_height(const zkhip_tree* t) { return t ? t->log_height : 0; }
int zkhip_tree_layer(zkhip_ctx* ctx, const zkhip_tree* t, unsigned layer, uint32_t* out) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !t || !out || layer > t->log_height) return ZKHIP_ERR_INVALID;
    size_t words = (size_t)8 << (t->log_height - layer);
    ZK_TRY(zkhip_d2h(ctx, out, t->d_digests + t->layer_off[layer] * 8, words * 4));
    for (size_t i = 0; i < words; i++) out[i] = from_monty(out[i]);
    return ZKHIP_OK;
}
size_t zkhip_merkle_opening_words(const zkhip_tree* t) { return t ? t->total_width + 8 * (size_t)t->log_height : 0; }
int zkhip_merkle_open(zkhip_ctx* ctx, const zkhip_tree* t, const uint64_t* indices, size_t n, uint32_t* out,
                      size_t cap_words) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !t || !indices || !out) return ZKHIP_ERR_INVALID;
    size_t pitch = zkhip_merkle_opening_words(t);
    if (cap_words < pitch * n) return set_error(ctx, ZKHIP_ERR_SMALL_BUFFER, "merkle_open: output too small");
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; i++) {
        if (indices[i] >> t->log_height) return set_error(ctx, ZKHIP_ERR_INVALID, "merkle_open: index out of range");
        idx[i] = (uint32_t)indices[i];
    }
    void *d_idx, *d_out;
    ZK_TRY(get_scratch(ctx, 2, n * 4, &d_idx));
    ZK_TRY(get_scratch(ctx, 3, pitch * n * 4, &d_out));
    ZK_TRY(zkhip_h2d(ctx, d_idx, idx.data(), n * 4));
    ZK_TRY(merkle_open_device(ctx, t, (const uint32_t*)d_idx, 0, n, (uint32_t*)d_out, pitch));
    return zkhip_d2h(ctx, out, d_out, pitch * n * 4);
}
void zkhip_tree_destroy(zkhip_ctx* ctx, zkhip_tree* t) {
    ZK_BIND_DEVICE(ctx);
    if (!t) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (t->d_colptrs) (void)hipFree(t->d_colptrs);
    if (t->d_digests && t->owns_digests) (void)hipFree(t->d_digests);
    delete t;
}

int zkhip_fri_fold(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, unsigned log_n_out, const uint32_t beta[4]) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !d_in || !d_out || !beta) return ZKHIP_ERR_INVALID;
    uint32_t bm[4];
    for (int i = 0; i < 4; i++) {
        if (beta[i] >= P) return set_error(ctx, ZKHIP_ERR_INVALID, "beta not canonical");
        bm[i] = to_monty(beta[i]);
    }
    void* d_beta;
    ZK_TRY(get_scratch(ctx, 2, 16, &d_beta));
    ZK_TRY(zkhip_h2d(ctx, d_beta, bm, 16));
    return fri_fold(ctx, d_in, d_out, log_n_out, (const uint32_t*)d_beta, nullptr, false);
}

// ---- transcript handle ------------------------------------------------------------------------
struct zkhip_transcript {
    DevTranscript* d = nullptr;
    uint32_t* d_buf = nullptr;  // staging for observe/sample
    size_t buf_words = 0;
};

static int tr_buf(zkhip_ctx* ctx, zkhip_transcript* t, size_t words) {
    if (t->buf_words >= words) return ZKHIP_OK;
    if (t->d_buf) {
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        ZK_HIP_CHECK(ctx, hipFree(t->d_buf));
    }
    size_t w = std::max<size_t>(words, 1024);
    ZK_HIP_CHECK(ctx, hipMalloc(&t->d_buf, w * 4));
    t->buf_words = w;
    return ZKHIP_OK;
}

int zkhip_transcript_create(zkhip_ctx* ctx, zkhip_transcript** out) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !out) return ZKHIP_ERR_INVALID;
    zkhip_transcript* t = new zkhip_transcript();
    if (hipMalloc(&t->d, sizeof(DevTranscript)) != hipSuccess) {
        delete t;
        return set_error(ctx, ZKHIP_ERR_NOMEM, "transcript alloc");
    }
    int rc = transcript_init(ctx, t->d);
    if (rc != ZKHIP_OK) {
        (void)hipFree(t->d);
        delete t;
        return rc;
    }
    *out = t;
    return ZKHIP_OK;
}
void zkhip_transcript_destroy(zkhip_ctx* ctx, zkhip_transcript* t) {
    ZK_BIND_DEVICE(ctx);
    if (!t) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (t->d) (void)hipFree(t->d);
    if (t->d_buf) (void)hipFree(t->d_buf);
    delete t;
}
int zkhip_transcript_observe(zkhip_ctx* ctx, zkhip_transcript* t, const uint32_t* vals, size_t n) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !t || (!vals && n)) return ZKHIP_ERR_INVALID;
    if (n == 0) return ZKHIP_OK;
    for (size_t i = 0; i < n; i++)
        if (vals[i] >= P) return set_error(ctx, ZKHIP_ERR_INVALID, "observe: value not canonical");
    ZK_TRY(tr_buf(ctx, t, n));
    ZK_TRY(zkhip_h2d(ctx, t->d_buf, vals, n * 4));
    return transcript_observe(ctx, t->d, t->d_buf, (uint32_t)n, true);
}
int zkhip_transcript_sample(zkhip_ctx* ctx, zkhip_transcript* t, uint32_t* out, size_t n) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !t || (!out && n)) return ZKHIP_ERR_INVALID;
    if (n == 0) return ZKHIP_OK;
    ZK_TRY(tr_buf(ctx, t, n));
    ZK_TRY(transcript_sample(ctx, t->d, nullptr, t->d_buf, (uint32_t)n));
    return zkhip_d2h(ctx, out, t->d_buf, n * 4);
}
int zkhip_transcript_grind(zkhip_ctx* ctx, zkhip_transcript* t, unsigned bits, uint32_t* witness) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !t || !witness) return ZKHIP_ERR_INVALID;
    ZK_TRY(tr_buf(ctx, t, 1));
    ZK_TRY(transcript_grind(ctx, t->d, bits, t->d_buf));
    ZK_TRY(zkhip_d2h(ctx, witness, t->d_buf, 4));
    DevTranscript h;
    ZK_TRY(zkhip_d2h(ctx, &h, t->d, sizeof h));
    if (h.error) return set_error(ctx, ZKHIP_ERR_POW_FAILED, "proof-of-work search failed");
    return ZKHIP_OK;
}

// ---- profiling ----------------------------------------------------------------------------------
int zkhip_profile_enable(zkhip_ctx* ctx, int on) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (!on) profile_flush(ctx);
    ctx->profiling = on != 0;
    return ZKHIP_OK;
}
int zkhip_profile_read(zkhip_ctx* ctx, zkhip_kernel_stat* out, size_t cap) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    profile_flush(ctx);
    size_t i = 0;
    for (auto& kv : ctx->stats) {
        if (out && i < cap) {
            memset(&out[i], 0, sizeof out[i]);
            strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
            out[i].launches = kv.second.launches;
            out[i].total_ms = kv.second.total_ms;
        }
        i++;
    }
    return (int)i;
}
int zkhip_profile_reset(zkhip_ctx* ctx) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx) return ZKHIP_ERR_INVALID;
    profile_flush(ctx);
    ctx->stats.clear();
    return ZKHIP_OK;
}

}  // extern "C"
