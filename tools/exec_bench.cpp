// exec_bench.cpp -- the segmenting executor alone (include/zkhip_vm_exec.hpp), no GPU needed (libzkhip for the host Poseidon2): instructions per second of `run_segment` over a guest,
// the stage that bounds the guest flow once frames are 2^19 instructions (docs/round5.md).  Host only.
//   g++ -O2 -std=c++17 -pthread -I include tools/exec_bench.cpp -o /tmp/exec_bench -Lzkvm-prover_amd -lzkhip -Wl,-rpath,$PWD/zkvm-prover_amd
//   /tmp/exec_bench <guest.elf> <stdin.bin | -> [log_frame = 19] [repeats = 3] [record threads = 0: the serial executor; n: metered pass + n record passes]
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>

#include "zkhip_vm_exec.hpp"

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) {
        std::fprintf(stderr, "cannot read %s\n", p);
        std::exit(2);
    }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const unsigned log_frame = argc > 3 ? (unsigned)atoi(argv[3]) : 19, reps = argc > 4 ? (unsigned)atoi(argv[4]) : 3, threads = argc > 5 ? (unsigned)atoi(argv[5]) : 0;
    const zkhip_vm::Exe exe = zkhip_vm::parse_exe(slurp(argv[1]));
    zkhip_vm::StdIn in;
    if (std::string(argv[2]) != "-") in.bytes = slurp(argv[2]);
    unsigned log_program = 0;
    while (((size_t)1 << log_program) < exe.program.size()) log_program++;
    const zkhip_vm::SegmentCaps caps = zkhip_vm::SegmentCaps::for_frame(log_frame, log_program);
    {   // the interpreter alone (no records, no memory bookkeeping): the floor under run_segment
        zkhip_vm::Machine m(exe, in);
        m.set_memory_log(false);
        const auto t0 = std::chrono::steady_clock::now();
        while (!m.done()) m.step_one(nullptr);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"interpreter_alone_instr_per_s\": %.0f}\n", m.instret() / s);
    }
    if (threads) {   // the metered pass alone: the ceiling of the parallel executor
        zkhip_vm::SegmentExecutor ex(exe, in, caps);
        ex.machine().track_dirty(true);
        const auto t0 = std::chrono::steady_clock::now();
        size_t k = 0;
        for (bool done = false; !done; k++) {
            zkhip_vm::SegmentPlan plan;
            done = ex.meter_segment(plan);
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"metered_pass_alone_instr_per_s\": %.0f, \"segments\": %zu}\n", ex.instret() / s, k);
    }
    for (unsigned rep = 0; rep < reps && threads; rep++) {
        zkhip_vm::ParallelSegmentExecutor ex(exe, in, caps, threads);
        zkhip_vm::SegmentRecords r;
        size_t k = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (bool done = false; !done; k++) done = ex.run_segment(r);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"segments\": %zu, \"instructions\": %llu, \"seconds\": %.3f, \"instr_per_s\": %.0f, \"record_threads\": %u, \"metered_pass_busy_s\": %.3f, \"record_passes_busy_s_sum\": %.3f}\n", k,
                    (unsigned long long)ex.instret(), s, ex.instret() / s, threads, ex.metered_seconds(), ex.record_seconds());
    }
    for (unsigned rep = 0; rep < reps && !threads; rep++) {
        zkhip_vm::SegmentExecutor ex(exe, in, caps);
        zkhip_vm::SegmentRecords r;
        size_t k = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (bool done = false; !done; k++) done = ex.run_segment(r);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"segments\": %zu, \"instructions\": %llu, \"seconds\": %.3f, \"instr_per_s\": %.0f, \"memory_close_seconds\": %.3f}\n", k, (unsigned long long)ex.instret(), s, ex.instret() / s, ex.close_seconds());
    }
    return 0;
}
