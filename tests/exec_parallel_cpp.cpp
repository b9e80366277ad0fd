// exec_parallel_cpp.cpp -- the parallel executor (metered pass + record passes, include/zkhip_vm_exec.hpp) against the serial one: the same
// segments, cut at the same instructions, with the same records word for word, whatever the number of record threads.  Host only.
//   exec_parallel_cpp <guest.elf> <stdin.bin | -> <log_frame> <threads> [openvm.toml | -] [parallel-only [abandon_after]]
// parallel-only: the serial run is skipped (errors then come from the parallel executor's threads); abandon_after = n: the caller stops
// taking segments after n and destroys the executor with its threads mid-run.
// prints one JSON line {"segments", "instructions", "equal", "first_difference"}; exit code 1 if anything differs.
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include "zkhip_vm_exec.hpp"
#include "zkhip_vm_flow.hpp"

using namespace zkhip_vm;

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) std::fprintf(stderr, "cannot read %s\n", p), std::exit(2);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// every array of a segment's records, by name
template <class F>
static void each_array(const SegmentRecords& r, F&& f) {
    const ExecRecords& c = r.core;
#define A(x) f(#x, c.x)
    A(pc_index); A(alu_op); A(alu_b); A(alu_c); A(lt_op); A(lt_b); A(lt_c); A(mul_b); A(mul_c); A(mulh_op); A(mulh_b); A(mulh_c); A(div_op); A(div_b); A(div_c);
    A(shift_op); A(shift_b); A(shift_c); A(beq_op); A(beq_a); A(beq_b); A(beq_imm); A(blt_op); A(blt_a); A(blt_b); A(blt_imm); A(jal_op); A(jal_pc); A(jal_imm);
    A(auipc_pc); A(auipc_imm); A(jalr_pc); A(jalr_rs1); A(jalr_imm); A(ls_case); A(ls_read); A(ls_prev); A(acc_as); A(acc_ptr); A(acc_prev_data); A(acc_prev_ts);
    A(acc_data); A(acc_ts); A(acc_is_read); A(bnd_as); A(bnd_ptr); A(bnd_initial); A(bnd_final); A(bnd_final_ts);
#undef A
#define B(x) f(#x, r.x)
    B(f_x); B(f_y); B(f_z); B(f_rdprev); B(f_pcinc); B(f_pts1); B(f_pts2); B(f_pts3); B(ls_ts); B(ls_base); B(ls_imm); B(ls_pts); B(ecall_rows); B(leaf_rows);
    B(merkle_rows); B(p2_inputs); B(kk_states); B(kk_ts); B(kio_rows); B(sha_blocks); B(sha_ts); B(shaio_rows); B(i256_records); B(i256_ts); B(i256io_rows);
    B(mul256_records); B(mul256_ts); B(cmp256_records); B(cmp256_ts); B(sh256_records); B(sh256_ts); B(nat_records); B(next_records); B(castf_records);
#undef B
    for (unsigned i = 0; i < vmc::MAX_MODULI; i++) f("mm_records", r.mm_records[i]), f("mm_ts", r.mm_ts[i]), f("mmio_rows", r.mmio_rows[i]);
    for (unsigned i = 0; i < vmc::MAX_CURVES; i++) f("ec_records", r.ec_records[i]), f("ec_ts", r.ec_ts[i]), f("ecio_rows", r.ecio_rows[i]);
    for (unsigned i = 0; i < vmc::MAX_FP2; i++) f("fp2_records", r.fp2_records[i]), f("fp2_ts", r.fp2_ts[i]), f("fp2io_rows", r.fp2io_rows[i]);
}
static std::vector<uint32_t> flatten(const SegmentRecords& r) {
    std::vector<uint32_t> out;
    each_array(r, [&](const char*, const std::vector<uint32_t>& v) { out.push_back((uint32_t)v.size()), out.insert(out.end(), v.begin(), v.end()); });
    const ExecRecords& c = r.core;
    for (uint64_t n : {c.n_shift, c.n_branch, c.n_jump, c.n_load_store, c.n_mulh, c.n_divrem, c.n_lui_auipc, c.n_ecall, (uint64_t)r.n_instr}) out.push_back((uint32_t)n);
    out.push_back(r.pc_start), out.push_back(r.pc_end), out.push_back(r.ts_end);
    out.insert(out.end(), r.root_init.begin(), r.root_init.end()), out.insert(out.end(), r.root_final.begin(), r.root_final.end());
    return out;
}
static std::string first_difference(const SegmentRecords& a, const SegmentRecords& b) {
    std::vector<std::pair<std::string, std::vector<uint32_t>>> va, vb;
    each_array(a, [&](const char* n, const std::vector<uint32_t>& v) { va.push_back({n, v}); });
    each_array(b, [&](const char* n, const std::vector<uint32_t>& v) { vb.push_back({n, v}); });
    for (size_t i = 0; i < va.size(); i++)
        if (va[i].second != vb[i].second) return va[i].first + " (sizes " + std::to_string(va[i].second.size()) + " / " + std::to_string(vb[i].second.size()) + ")";
    return "a scalar (counts, pcs, roots)";
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    try {
        const Exe exe = parse_exe(slurp(argv[1]));
        StdIn in;
        if (std::string(argv[2]) != "-") in.bytes = slurp(argv[2]);
        const unsigned log_frame = (unsigned)atoi(argv[3]), threads = (unsigned)atoi(argv[4]);
        const std::string cfg = argc > 5 && std::string(argv[5]) != "-" ? argv[5] : "";
        if (argc > 6 && std::string(argv[6]) == "parallel-only") {
            const size_t abandon = argc > 7 ? (size_t)atoi(argv[7]) : (size_t)-1;
            const SegmentCaps caps = cfg.empty() ? SegmentCaps::for_frame(log_frame, vm_log2_ceil(exe.program.size())) : config_caps(cfg, log_frame, vm_log2_ceil(exe.program.size()));
            size_t k = 0;
            {
                ParallelSegmentExecutor px(exe, in, caps, threads);
                SegmentRecords r;
                for (bool done = false; !done && k < abandon; k++) done = px.run_segment(r);
            }   // (the destructor joins the metered pass, the tree thread and the record passes wherever they are)
            std::printf("{\"segments_taken\": %zu, \"parallel_only\": true}\n", k);
            return 0;
        }
        const SegmentCaps caps = cfg.empty() ? SegmentCaps::for_frame(log_frame, vm_log2_ceil(exe.program.size())) : config_caps(cfg, log_frame, vm_log2_ceil(exe.program.size()));
        std::vector<std::vector<uint32_t>> serial;
        std::vector<SegmentRecords> serial_recs;
        SegmentExecutor ex(exe, in, caps);
        for (bool done = false; !done;) {
            SegmentRecords r;
            done = ex.run_segment(r);
            serial.push_back(flatten(r));
            serial_recs.push_back(std::move(r));
        }
        ParallelSegmentExecutor px(exe, in, caps, threads);
        size_t k = 0;
        bool equal = true;
        std::string diff;
        SegmentRecords r;   // (recycled, as the flow does)
        for (bool done = false; !done; k++) {
            done = px.run_segment(r);
            if (k >= serial.size() || flatten(r) != serial[k]) {
                if (equal) diff = "segment " + std::to_string(k) + ": " + (k < serial.size() ? first_difference(serial_recs[k], r) : std::string("beyond the serial run's last segment"));
                equal = false;
            }
        }
        equal = equal && k == serial.size() && px.instret() == ex.instret() && px.public_values() == ex.public_values() && px.tree().root() == ex.tree().root() &&
                px.image_root() == ex.image_root();
        std::printf("{\"segments\": %zu, \"segments_serial\": %zu, \"instructions\": %llu, \"threads\": %u, \"equal\": %s, \"first_difference\": \"%s\"}\n", k, serial.size(),
                    (unsigned long long)px.instret(), threads, equal ? "true" : "false", diff.c_str());
        return equal ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("{\"error\": \"%s\"}\n", e.what());
        return 3;
    }
}
