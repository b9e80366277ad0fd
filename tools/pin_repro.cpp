// pin_repro.cpp -- what goes wrong when host memory that lives INSIDE the shared heap is page-locked in place (round 3's witness
// generator registered the storage of a std::vector: DESIGN.md 5 "the abort hunted in the full GPU suite").  Each hypothesis runs in a
// child process of its own (the parent never touches the GPU), many trials with varying sizes, and the parent reports how the child
// ended: exit 0, a HIP error count, or a signal (SIGABRT = what the suite saw).
//   h1: register a heap block, then copy host -> device FROM A NEIGHBOUR allocation that starts in the registered block's last page
//   h2: the same, the copy's source straddling the END of the registered range (last bytes of the block + first bytes of the neighbour)
//   h3: register a heap block, free() it WITHOUT unregistering, allocate again (same address, other size), copy from the new block
//   h4: register / copy / unregister a block of a mapping of its own (what the product does now), same trial count
//   h5: TWO live heap blocks that share a page (two circuits' small value arrays), both registered, copied from, both unregistered
//   h6: h5, the first block unregistered and freed while the second stays registered and in use
// build: hipcc -O1 tools/pin_repro.cpp -o gpurun_out/pin_repro ; run: gpurun_out/pin_repro [trials]
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

static int run(int h, int trials) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);   // everything below 1 GiB comes from the brk heap, next to its neighbours
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    void* d = nullptr;
    if (hipMalloc(&d, 64 << 20) != hipSuccess) return 99;
    int errors = 0;
    unsigned seed = 12345u + (unsigned)h;
    auto rnd = [&](unsigned lo, unsigned hi) {
        seed = seed * 1664525u + 1013904223u;
        return lo + (seed >> 8) % (hi - lo);
    };
    for (int t = 0; t < trials; t++) {
        const size_t sa = rnd(100 << 10, 3 << 20) + rnd(0, 4096), sb = rnd(4 << 10, 2 << 20) + rnd(0, 4096);
        if (h == 4) {
            const size_t bytes = (sa + 4095) / 4096 * 4096;
            void* a = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            memset(a, 1, bytes);
            errors += hipHostRegister(a, bytes, hipHostRegisterDefault) != hipSuccess;
            errors += hipMemcpy(d, a, sa, hipMemcpyHostToDevice) != hipSuccess;
            char* b = (char*)malloc(sb);
            memset(b, 2, sb);
            errors += hipMemcpy(d, b, sb, hipMemcpyHostToDevice) != hipSuccess;
            free(b);
            errors += hipHostUnregister(a) != hipSuccess;
            munmap(a, bytes);
            continue;
        }
        char* a = (char*)malloc(sa);
        char* b = (char*)malloc(sb);   // the next chunk of the heap: starts 16 bytes behind a's end
        memset(a, 1, sa), memset(b, 2, sb);
        if (h == 5 || h == 6) {
            errors += hipHostRegister(a, sa, hipHostRegisterDefault) != hipSuccess;
            errors += hipHostRegister(b, sb, hipHostRegisterDefault) != hipSuccess;   // shares a's last page
            errors += hipMemcpy(d, a, sa, hipMemcpyHostToDevice) != hipSuccess;
            errors += hipMemcpy(d, b, sb, hipMemcpyHostToDevice) != hipSuccess;
            errors += hipHostUnregister(a) != hipSuccess;
            if (h == 6) {
                free(a);
                a = (char*)malloc(sa / 2);
                memset(a, 4, sa / 2);
                errors += hipMemcpy(d, a, sa / 2, hipMemcpyHostToDevice) != hipSuccess;
            }
            errors += hipMemcpy(d, b, sb, hipMemcpyHostToDevice) != hipSuccess;
            errors += hipHostUnregister(b) != hipSuccess;
            (void)hipGetLastError();
            free(b), free(a);
            continue;
        }
        errors += hipHostRegister(a, sa, hipHostRegisterDefault) != hipSuccess;
        errors += hipMemcpy(d, a, sa, hipMemcpyHostToDevice) != hipSuccess;
        if (h == 1) errors += hipMemcpy(d, b, sb, hipMemcpyHostToDevice) != hipSuccess;
        if (h == 2) errors += hipMemcpy(d, a + sa - 64, 4096, hipMemcpyHostToDevice) != hipSuccess;
        if (h == 3) {
            free(a);   // still registered
            a = (char*)malloc(sa + rnd(4096, 1 << 20));
            memset(a, 3, sa);
            errors += hipMemcpy(d, a, sa + 2048, hipMemcpyHostToDevice) != hipSuccess;
            (void)hipHostUnregister(a);
            (void)hipGetLastError();
        } else {
            errors += hipHostUnregister(a) != hipSuccess;
        }
        (void)hipGetLastError();
        free(b), free(a);
    }
    (void)hipFree(d);
    return errors > 90 ? 90 : errors;
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 300;
    printf("{");
    for (int h = 1; h <= 6; h++) {
        fflush(stdout);
        const pid_t pid = fork();   // (the parent has not initialised the GPU)
        if (pid == 0) _exit(run(h, trials));
        int st = 0;
        waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) printf("%s\"h%d\": \"signal %d\"", h > 1 ? ", " : "", h, WTERMSIG(st));
        else printf("%s\"h%d\": \"exit %d (HIP errors)\"", h > 1 ? ", " : "", h, WEXITSTATUS(st));
    }
    printf(", \"trials\": %d}\n", trials);
    return 0;
}
