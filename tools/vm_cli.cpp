// vm_cli -- the guest execution step (include/zkhip_vm.hpp: the mirror of crates/prover/src/utils/vm.rs execute_guest) as a
// command-line tool: runs an RV32IM guest and writes the per-chip execution records the device trace generators take.
//   vm_cli exec <exe.bin> <stdin.bin | -> <out_dir | -> [max_cost]
// exe.bin: an RV32 ELF file, or u32 words  [0x58455A4B "KZEX", pc_base, n_program, program..., data_base, memory_bytes, n_data_bytes, data bytes...]
// Prints one JSON line {"total_cycle", "public_values", "records": {...counts}, "trace_cells"}; with an out_dir, writes
// pc_index.u32, alu_{op,b,c}.u32, lt_{op,b,c}.u32, mul_{b,c}.u32, the memory log acc_{as,ptr,prev_data,prev_ts,data,ts,is_read}.u32 and
// the boundary records bnd_{as,ptr,initial,final,final_ts}.u32 (raw little-endian arrays).
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "zkhip_vm.hpp"

static std::vector<uint8_t> read_file(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) throw zkhip_vm::Error("cannot read " + p);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void write_u32(const std::string& p, const std::vector<uint32_t>& v) {
    std::ofstream f(p, std::ios::binary);
    f.write((const char*)v.data(), (std::streamsize)(v.size() * 4));
}
static void write_records(const std::string& out, const zkhip_vm::ExecRecords& rec) {
    write_u32(out + "/pc_index.u32", rec.pc_index);
    write_u32(out + "/alu_op.u32", rec.alu_op), write_u32(out + "/alu_b.u32", rec.alu_b), write_u32(out + "/alu_c.u32", rec.alu_c);
    write_u32(out + "/lt_op.u32", rec.lt_op), write_u32(out + "/lt_b.u32", rec.lt_b), write_u32(out + "/lt_c.u32", rec.lt_c);
    write_u32(out + "/mul_b.u32", rec.mul_b), write_u32(out + "/mul_c.u32", rec.mul_c);
    write_u32(out + "/shift_op.u32", rec.shift_op), write_u32(out + "/shift_b.u32", rec.shift_b), write_u32(out + "/shift_c.u32", rec.shift_c);
    write_u32(out + "/beq_op.u32", rec.beq_op), write_u32(out + "/beq_a.u32", rec.beq_a), write_u32(out + "/beq_b.u32", rec.beq_b);
    write_u32(out + "/beq_imm.u32", rec.beq_imm);
    write_u32(out + "/blt_op.u32", rec.blt_op), write_u32(out + "/blt_a.u32", rec.blt_a), write_u32(out + "/blt_b.u32", rec.blt_b);
    write_u32(out + "/blt_imm.u32", rec.blt_imm);
    write_u32(out + "/mulh_op.u32", rec.mulh_op), write_u32(out + "/mulh_b.u32", rec.mulh_b), write_u32(out + "/mulh_c.u32", rec.mulh_c);
    write_u32(out + "/div_op.u32", rec.div_op), write_u32(out + "/div_b.u32", rec.div_b), write_u32(out + "/div_c.u32", rec.div_c);
    write_u32(out + "/ls_case.u32", rec.ls_case), write_u32(out + "/ls_read.u32", rec.ls_read), write_u32(out + "/ls_prev.u32", rec.ls_prev);
    write_u32(out + "/jal_op.u32", rec.jal_op), write_u32(out + "/jal_pc.u32", rec.jal_pc), write_u32(out + "/jal_imm.u32", rec.jal_imm);
    write_u32(out + "/auipc_pc.u32", rec.auipc_pc), write_u32(out + "/auipc_imm.u32", rec.auipc_imm);
    write_u32(out + "/jalr_pc.u32", rec.jalr_pc), write_u32(out + "/jalr_rs1.u32", rec.jalr_rs1), write_u32(out + "/jalr_imm.u32", rec.jalr_imm);
    write_u32(out + "/acc_as.u32", rec.acc_as), write_u32(out + "/acc_ptr.u32", rec.acc_ptr);
    write_u32(out + "/acc_prev_data.u32", rec.acc_prev_data), write_u32(out + "/acc_prev_ts.u32", rec.acc_prev_ts);
    write_u32(out + "/acc_data.u32", rec.acc_data), write_u32(out + "/acc_ts.u32", rec.acc_ts), write_u32(out + "/acc_is_read.u32", rec.acc_is_read);
    write_u32(out + "/bnd_as.u32", rec.bnd_as), write_u32(out + "/bnd_ptr.u32", rec.bnd_ptr), write_u32(out + "/bnd_initial.u32", rec.bnd_initial);
    write_u32(out + "/bnd_final.u32", rec.bnd_final), write_u32(out + "/bnd_final_ts.u32", rec.bnd_final_ts);
}

int main(int argc, char** argv) {
    try {
        if (argc >= 6 && std::string(argv[1]) == "exec-segments") {
            // vm_cli exec-segments <exe> <stdin|-> <out_dir> <segment_instr>: the continuation split; out_dir/seg-<k>/ must exist
            const zkhip_vm::Exe exe = zkhip_vm::parse_exe(read_file(argv[2]));
            zkhip_vm::StdIn in;
            if (std::string(argv[3]) != "-") in.bytes = read_file(argv[3]);
            const zkhip_vm::SegmentedExecution se = zkhip_vm::execute_segments(exe, in, strtoull(argv[5], nullptr, 10));
            std::ostringstream js;
            js << "{\"total_cycle\": " << se.result.total_cycle << ", \"segments\": [";
            for (size_t k = 0; k < se.segments.size(); k++) {
                write_records(std::string(argv[4]) + "/seg-" + std::to_string(k), se.segments[k]);
                js << (k ? ", " : "") << se.segments[k].pc_index.size();
            }
            js << "], \"public_values\": [";
            for (size_t i = 0; i < se.result.public_values.size(); i++) js << (i ? ", " : "") << (unsigned)se.result.public_values[i];
            js << "]}";
            std::cout << js.str() << std::endl;
            return 0;
        }
        if (argc < 5 || std::string(argv[1]) != "exec") {
            fprintf(stderr, "usage: vm_cli exec <exe.bin> <stdin.bin|-> <out_dir|-> [max_cost]\n");
            return 2;
        }
        const std::vector<uint8_t> raw = read_file(argv[2]);
        const zkhip_vm::Exe exe = zkhip_vm::parse_exe(raw);
        zkhip_vm::StdIn in;
        if (std::string(argv[3]) != "-") in.bytes = read_file(argv[3]);
        const uint64_t max_cost = argc > 5 ? strtoull(argv[5], nullptr, 10) : 0;
        zkhip_vm::ExecRecords rec;
        const uint64_t max_instr = getenv("ZKHIP_VM_MAX_INSTR") ? strtoull(getenv("ZKHIP_VM_MAX_INSTR"), nullptr, 10) : 0;
        const zkhip_vm::ExecutionResult r = zkhip_vm::execute_guest(exe, in, max_cost, &rec, max_instr);
        const std::string out = argv[4];
        if (out != "-") write_records(out, rec);
        std::ostringstream js;
        js << "{\"total_cycle\": " << r.total_cycle << ", \"public_values\": [";
        for (size_t i = 0; i < r.public_values.size(); i++) js << (i ? ", " : "") << (unsigned)r.public_values[i];
        js << "], \"records\": {\"executed\": " << rec.pc_index.size() << ", \"alu\": " << rec.alu_op.size() << ", \"lt\": " << rec.lt_op.size()
           << ", \"mul\": " << rec.mul_b.size() << ", \"shift\": " << rec.n_shift << ", \"branch\": " << rec.n_branch << ", \"jump\": " << rec.n_jump
           << ", \"load_store\": " << rec.n_load_store << ", \"mulh\": " << rec.n_mulh << ", \"divrem\": " << rec.n_divrem
           << ", \"lui_auipc\": " << rec.n_lui_auipc << ", \"ecall\": " << rec.n_ecall << ", \"memory_accesses\": " << rec.acc_ts.size() << ", \"cells_touched\": " << rec.bnd_ptr.size()
           << "}, \"trace_cells\": "
           << zkhip_vm::trace_cells(rec, exe.program.size()) << "}";
        std::cout << js.str() << std::endl;
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "vm_cli: %s\n", e.what());
        return 1;
    }
}
