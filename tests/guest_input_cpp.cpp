// Builds ProvingTask::build_guest_input() for a fixed task and runs a guest over it (tests/test_vm_cpu.py compares with the
// Python model): prints the stream bytes in hex, then the guest's public values.
#include <cstdio>

#include "zkhip_prover.hpp"

using namespace scroll_zkvm_hip;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    ProvingTask task;
    task.serialized_witness = {{1, 2, 3, 4, 5}, {}, {0xff, 0xfe, 0xfd, 0xfc, 0xfb, 0xfa, 0xf9, 0xf8}};
    std::array<uint8_t, 32> c{};
    for (int i = 0; i < 32; i++) c[i] = (uint8_t)(3 * i + 1);
    task.input_commits = {c, c};
    const zkhip_vm::StdIn in = task.build_guest_input();
    for (uint8_t b : in.bytes) printf("%02x", b);
    printf("\n");
    std::ifstream f(argv[1], std::ios::binary);
    const std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const zkhip_vm::ExecutionResult r = zkhip_vm::execute_guest(zkhip_vm::parse_exe(raw), in);
    for (uint8_t b : r.public_values) printf("%02x", b);
    printf("\n%llu\n", (unsigned long long)r.total_cycle);
    return 0;
}
