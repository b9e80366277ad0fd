// merkle_stress -- the Merkle tree kernels under the concurrency of the guest flow, WITHOUT anything that serialises the device in the loop
// (round 4's version allocated, synchronised and freed per commit and was clean even with the miscompiled kernel: docs/stale_node.md).
// T tree threads, each with a context (stream) of its own on ONE GPU, rebuild their tree in place again and again (zkhip_merkle_rebuild: the
// commit's launches on the same digest store, no allocation, no synchronisation) and have the device recompute every plain layer from its
// stored children every `batch` rebuilds (zkhip_tree_check); K noise threads run LDS-heavy transforms (zkhip_ntt_batch, the four-step NTT
// stages its tiles through the LDS) on streams of their own: the lost wait of the round-4 kernel only shows when the LDS queues of a CU are
// backed up by other workgroups.
// usage: merkle_stress <tree threads> <rebuilds per thread> <log_height> <width> [noise threads = 2] [batch = 8]
// prints one JSON line; exit code 1 if any node ever differed.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "zkhip.h"

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 3, iters = argc > 2 ? atoi(argv[2]) : 2000;
    const unsigned lh = argc > 3 ? atoi(argv[3]) : 12, width = argc > 4 ? atoi(argv[4]) : 8;
    const int K = argc > 5 ? atoi(argv[5]) : 2, batch = argc > 6 ? atoi(argv[6]) : 8;
    std::mutex mu;
    std::atomic<long> bad_nodes{0}, bad_checks{0}, checks{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> noise, th;
    for (int k = 0; k < K; k++)
        noise.emplace_back([&, k] {
            zkhip_ctx* ctx = nullptr;
            if (zkhip_ctx_create(0, &ctx) != ZKHIP_OK) exit(2);
            const unsigned nl = 16 + (k % 2), w = 32;
            void* d = nullptr;
            zkhip_malloc(ctx, ((size_t)w << nl) * 4, &d);
            zkhip_zero(ctx, d, ((size_t)w << nl) * 4);
            while (!stop.load()) {
                for (int r = 0; r < 8; r++) zkhip_ntt_batch(ctx, (uint32_t*)d, nl, w, (size_t)1 << nl, r & 1, 0);
                zkhip_sync(ctx);
            }
            zkhip_free(ctx, d);
            zkhip_ctx_destroy(ctx);
        });
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            zkhip_ctx* ctx = nullptr;
            if (zkhip_ctx_create(0, &ctx) != ZKHIP_OK) { std::fprintf(stderr, "no device\n"); exit(2); }
            const unsigned my_lh = lh - (t % 2);   // two tree sizes side by side
            std::vector<uint32_t> host((size_t)width << my_lh);
            uint64_t s = 88172645463325252ull + t;
            for (auto& v : host) { s ^= s << 13, s ^= s >> 7, s ^= s << 17; v = (uint32_t)(s % 2013265921ull); }
            void* d = nullptr;
            zkhip_malloc(ctx, host.size() * 4, &d);
            zkhip_h2d(ctx, d, host.data(), host.size() * 4);
            zkhip_to_monty(ctx, (uint32_t*)d, host.size());
            zkhip_matrix m{(const uint32_t*)d, (size_t)1 << my_lh, my_lh, width};
            zkhip_tree* tree = nullptr;
            uint32_t root[8];
            if (zkhip_merkle_commit(ctx, &m, 1, &tree, root) != ZKHIP_OK) { std::fprintf(stderr, "commit: %s\n", zkhip_last_error(ctx)); exit(2); }
            if (t == 0) std::printf("root of thread 0's tree: %08x %08x %08x %08x %08x %08x %08x %08x\n", root[0], root[1], root[2], root[3], root[4], root[5], root[6], root[7]);
            for (int it = 0; it < iters; it += batch) {
                for (int b = 0; b < batch; b++)
                    if (zkhip_merkle_rebuild(ctx, tree) != ZKHIP_OK) { std::fprintf(stderr, "rebuild: %s\n", zkhip_last_error(ctx)); exit(2); }
                // (a wrong node of rebuild k < batch is overwritten by rebuild k + 1: the check sees the last one -- 1 / batch of the launches)
                uint32_t n_bad = 0, first = 0;
                if (zkhip_tree_check(ctx, tree, &n_bad, &first) != ZKHIP_OK) { std::fprintf(stderr, "check: %s\n", zkhip_last_error(ctx)); exit(2); }
                checks++;
                if (n_bad) {
                    bad_nodes += n_bad, bad_checks++;
                    std::lock_guard<std::mutex> lk(mu);
                    if (bad_checks.load() <= 10) std::printf("thread %d rebuild %d: %u nodes of a 2^%u tree differ from the hash of their children, first layer %u index %u\n", t, it, n_bad, my_lh, first >> 24, first & 0xffffffu);
                }
            }
            zkhip_tree_destroy(ctx, tree);
            zkhip_free(ctx, d);
            zkhip_ctx_destroy(ctx);
        });
    for (auto& x : th) x.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    for (auto& x : noise) x.join();
    std::printf("{\"tree_threads\": %d, \"noise_threads\": %d, \"rebuilds_per_thread\": %d, \"log_height\": %u, \"width\": %u, \"checks\": %ld, \"checks_with_a_wrong_node\": %ld, "
                "\"wrong_nodes\": %ld, \"seconds\": %.1f, \"early_form\": %s}\n",
                T, K, iters, lh, width, checks.load(), bad_checks.load(), bad_nodes.load(), secs, (zkhip_has_test_kernels() && getenv("ZKHIP_TREE_STORE_EARLY")) ? "true" : "false");
    return bad_checks.load() ? 1 : 0;
}
