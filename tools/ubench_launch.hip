// ubench_launch.hip -- how many kernel launches per second a process gets out of T host threads with a stream each (empty kernels, and
// dependent chains of short kernels as a proof issues them).  hipcc --offload-arch=gfx950 -O2 tools/ubench_launch.hip -o ubench_launch -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_empty() {}
__global__ void k_short(unsigned* p, unsigned iters) {
    unsigned v = threadIdx.x;
    for (unsigned i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 0xdeadbeefu) p[0] = v;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned* d = nullptr;
    hipMalloc(&d, 4096);
    for (int mode = 0; mode < 2; mode++)
        for (int T : {1, 2, 3, 4, 6, 8, 12}) {
            std::vector<hipStream_t> st(T);
            for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    for (int i = 0; i < N; i++) {
                        if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[t]);
                        else hipLaunchKernelGGL(k_short, dim3(4), dim3(256), 0, st[t], d, 2000u);   // ~10 us of work on 4 CUs
                    }
                    hipStreamSynchronize(st[t]);
                });
            for (auto& x : th) x.join();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("{\"mode\": \"%s\", \"threads\": %d, \"launches\": %d, \"seconds\": %.3f, \"launches_per_s\": %.0f, \"us_per_launch_per_thread\": %.2f}\n",
                   mode ? "short kernels (4 blocks, ~10 us)" : "empty kernels", T, T * N, s, T * N / s, 1e6 * s / N);
            for (auto& s2 : st) hipStreamDestroy(s2);
        }
    hipFree(d);
    return 0;
}
