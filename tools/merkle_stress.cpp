// merkle_stress -- T threads, each with a context of its own on ONE GPU, commit the same matrices again and again and compare every
// digest layer with the first result: the Merkle path under the concurrency of the guest flow (lanes + node slots share the device).
// usage: merkle_stress <threads> <iterations> <log_height> <width>     (build: g++ -O2 -std=c++17 -I include ... -lzkhip)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "zkhip.h"

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 3, iters = argc > 2 ? atoi(argv[2]) : 2000;
    const unsigned lh = argc > 3 ? atoi(argv[3]) : 15, width = argc > 4 ? atoi(argv[4]) : 24;
    std::mutex mu;
    long bad = 0;
    std::vector<std::thread> th;
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
            std::vector<std::vector<uint32_t>> ref(my_lh + 1), got(my_lh + 1);
            for (int it = 0; it < iters; it++) {
                zkhip_matrix m{(const uint32_t*)d, (size_t)1 << my_lh, my_lh, width};
                zkhip_tree* tree = nullptr;
                uint32_t root[8];
                if (zkhip_merkle_commit(ctx, &m, 1, &tree, root) != ZKHIP_OK) { std::fprintf(stderr, "commit: %s\n", zkhip_last_error(ctx)); exit(2); }
                if (it == 0 && t == 0) std::printf("root of thread 0's tree: %08x %08x %08x %08x %08x %08x %08x %08x\n", root[0], root[1], root[2], root[3], root[4], root[5], root[6], root[7]);
                for (unsigned l = 0; l <= my_lh; l++) {
                    auto& dst = it == 0 ? ref[l] : got[l];
                    dst.resize((size_t)8 << (my_lh - l));
                    zkhip_tree_layer(ctx, tree, l, dst.data());
                    if (it && dst != ref[l]) {
                        size_t first = 0, n_diff = 0;
                        for (size_t i = 0; i < dst.size(); i += 8)
                            if (memcmp(&dst[i], &ref[l][i], 32)) { if (!n_diff) first = i / 8; n_diff++; }
                        std::lock_guard<std::mutex> lk(mu);
                        bad++;
                        std::printf("thread %d iteration %d: layer %u of a 2^%u tree differs in %zu nodes (first %zu)\n", t, it, l, my_lh, n_diff, first);
                    }
                }
                zkhip_tree_destroy(ctx, tree);
            }
            zkhip_free(ctx, d);
            zkhip_ctx_destroy(ctx);
        });
    for (auto& x : th) x.join();
    std::printf("merkle_stress: %d threads x %d commits, %ld differing layers\n", T, iters, bad);
    return bad ? 1 : 0;
}
