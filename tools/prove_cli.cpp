// prove_cli -- command-line driver of the C++ prover mirror (include/zkhip_prover.hpp).
//   prove_cli prove  <app.zkair> <openvm.toml> <task.bin> <out_proof.json>
//   prove_cli verify <app.zkair> <openvm.toml> <proof.json>
//   prove_cli prove-guest <guest.elf | exe.bin> <stdin.bin | -> <out_proof.json | out_dir> [openvm.toml | -] [max_cost] [segment_instr] [inflight]
//       guest image in, proof out (include/zkhip_vm_prover.hpp): execute, device trace generation, prove, self-verify
// task.bin: [u32 id_len][id][u32 n_witness]{[u64 len][bytes]}  (the fields of ProvingTask that
// a leaf task uses, crates/types/src/task.rs:7-23)
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iostream>

#include "zkhip_aggregation.hpp"
#include "zkhip_vm_flow.hpp"
#include "zkhip_vm_prover.hpp"

using namespace scroll_zkvm_hip;

static std::vector<uint8_t> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(Error::Io, "cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// the retried segments of a flow as JSON: [{"segment": i, "message": "..."}, ...] (GuestStark::segments_retried)
static std::string retried_json(const std::vector<std::pair<size_t, std::string>>& v) {
    std::string out = "[";
    for (size_t i = 0; i < v.size(); i++) {
        out += (i ? ", {\"segment\": " : "{\"segment\": ") + std::to_string(v[i].first) + ", \"message\": \"";
        for (char ch : v[i].second) {
            if (ch == '"' || ch == '\\') out += '\\', out += ch;
            else if ((unsigned char)ch < 0x20) out += ' ';
            else out += ch;
        }
        out += "\"}";
    }
    return out + "]";
}
static std::string list_of(const std::vector<size_t>& v) {
    std::string o = "[";
    for (size_t i = 0; i < v.size(); i++) o += (i ? ", " : "") + std::to_string(v[i]);
    return o + "]";
}
static ProvingTask read_task(const std::string& path) {
    auto tb = slurp(path);
    ProvingTask task;
    size_t p = 0;
    auto rd32 = [&]() { uint32_t v; if (p + 4 > tb.size()) throw Error(Error::Io, "short task"); memcpy(&v, &tb[p], 4); p += 4; return v; };
    auto rd64 = [&]() { uint64_t v; if (p + 8 > tb.size()) throw Error(Error::Io, "short task"); memcpy(&v, &tb[p], 8); p += 8; return v; };
    uint32_t idl = rd32();
    if (p + idl > tb.size()) throw Error(Error::Io, "short task");
    task.identifier.assign((const char*)&tb[p], idl);
    p += idl;
    uint32_t n = rd32();
    for (uint32_t i = 0; i < n; i++) {
        uint64_t len = rd64();
        if (p + len > tb.size()) throw Error(Error::Io, "short task");
        task.serialized_witness.emplace_back(tb.begin() + p, tb.begin() + p + len);
        p += len;
    }
    task.fork_name = "hip";
    return task;
}

int main(int argc, char** argv) {
    try {
        if (argc >= 5 && std::string(argv[1]) == "jit-prewarm") {
            // jit-prewarm <openvm.toml | -> <log_frame> <cache_dir> [part n_parts]: compiles (hipRTC: no GPU needed) the constraint kernels key
            // generation would compile for a guest flow under this configuration -- the segment chips of the full shape at their heights and
            // the two wide chips of the aggregation node circuits -- into <cache_dir> (zkhip_config.jit_cache_dir; `jit_cache` beside
            // libzkhip.so is found by default).  __graft_entry__.build() runs it, in parts side by side.
            const std::string cfg_path = std::string(argv[2]) == "-" ? "" : argv[2];
            const unsigned log_frame = (unsigned)atoi(argv[3]);
            const size_t part = argc >= 7 ? (size_t)atoi(argv[5]) : 0, n_parts = argc >= 7 ? std::max(1, atoi(argv[6])) : 1;
            zkhip_params params{1, 0, 100, 16, 16};
            if (!cfg_path.empty()) params = read_app_config(cfg_path);
            zkhip_vm::Exe exe;
            exe.program.assign(64, 0x00000013u);   // (the chips' constraint programs do not depend on the guest)
            const zkhip_vm::SegmentCaps caps = cfg_path.empty() ? zkhip_vm::SegmentCaps::for_frame(log_frame, 6) : zkhip_vm::config_caps(cfg_path, log_frame, 6);
            const zkhip_vm::SegmentAirs sa = zkhip_vm::segment_airs(exe, caps);
            std::vector<zkhip_air> airs = sa.airs;
            // the node circuits' gate and Poseidon2 chips (their programs do not depend on the child key; heights of a real leaf circuit)
            zkhip_recursion* toy = nullptr;
            {
                using namespace zkhip::air;
                AirBuilder b(1, 18);
                b.assert_zero(b.var(0) - b.var(0));
                const std::vector<uint32_t> prog = b.program();
                zkhip_air child{prog.data(), prog.size(), 3, 1, 18, nullptr, nullptr};
                std::vector<uint32_t> sa_, si_, ea_, ei_;
                for (uint32_t k = 0; k < 9; k++) sa_.push_back(0), si_.push_back(k), ea_.push_back(0), ei_.push_back(9 + k);
                zkhip_recursion_stmt st{};
                st.n_state = 9, st.start_air = sa_.data(), st.start_idx = si_.data(), st.end_air = ea_.data(), st.end_idx = ei_.data(), st.uniform = 1;
                const zkhip_params toy_params{1, 0, 2, 1, 1};
                if (zkhip_recursion_build(&toy_params, &child, 1, 1, &st, &toy) == ZKHIP_OK)
                    for (size_t i = 0; i < 2; i++) {
                        zkhip_air a{};
                        if (zkhip_recursion_air(toy, i, &a) == ZKHIP_OK)
                            for (unsigned h : (i == 0 ? std::vector<unsigned>{20, 21} : std::vector<unsigned>{17, 18, 19}))   // common nodes 2^20 / 2^17; a wide shape's own leaf circuit 2^21 / 2^19
                                a.log_height = h, a.prep_trace = nullptr, airs.push_back(a);
                    }
            }
            size_t total = 0, mine = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t a = 0; a < airs.size(); a++) {
                if (a % n_parts != part) continue;
                size_t ok = 0;
                if (zkhip_jit_prewarm(&airs[a], 1, params.log_blowup, argv[4], &ok) != ZKHIP_OK) throw Error(Error::Setup, "zkhip_jit_prewarm failed for AIR " + std::to_string(a));
                total += ok, mine++;
            }
            if (toy) zkhip_recursion_destroy(toy);
            std::printf("{\"airs\": %zu, \"of\": %zu, \"kernels_in_cache\": %zu, \"seconds\": %.1f}\n", mine, airs.size(), total,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            return 0;
        }
        if (argc >= 3 && std::string(argv[1]) == "chips") {
            // chips <openvm.toml> [log_frame = 20]: the chips the app's configuration asks for -- every `[app_vm_config.*]` section of the file, none
            // ignored (the reference's chunk / batch / bundle circuits: crates/circuits/*/openvm.toml) -- as a JSON line: name, columns, log height
            const unsigned log_frame = argc >= 4 ? (unsigned)atoi(argv[3]) : 20;
            const zkhip_vm::SegmentCaps caps = zkhip_vm::config_caps(argv[2], log_frame, 10);
            zkhip_vm::Exe exe;
            exe.program.assign(64, 0x00000013u);
            const zkhip_vm::SegmentAirs sa = zkhip_vm::segment_airs(exe, caps);
            namespace vmc = zkhip::vmc;
            auto name_of = [&](unsigned id) -> std::string {
                static const char* base[] = {"program", "frame", "alu", "lt", "mul", "mulh", "divrem", "shift", "beq", "blt", "jal_lui", "auipc", "jalr", "loadstore", "ecall", "leaf",
                                             "merkle", "poseidon2", "connector", "bitwise", "range_tuple", "range", "keccak", "keccak_io", "sha256", "sha256_io", "int256", "int256_io",
                                             "mul256", "cmp256", "shift256", "native_arith", "native_ext", "castf"};
                if (id < vmc::N_STATIC_AIRS) return base[id];
                if (id >= vmc::A_FP2(0)) return std::string((id - vmc::A_FP2(0)) & 1 ? "fp2_io_" : "fp2_") + std::to_string((id - vmc::A_FP2(0)) / 2);
                if (id >= vmc::A_EC(0)) return std::string((id - vmc::A_EC(0)) & 1 ? "ec_io_" : "ec_") + std::to_string((id - vmc::A_EC(0)) / 2);
                return std::string((id - vmc::N_STATIC_AIRS) & 1 ? "modmul_io_" : "modmul_") + std::to_string((id - vmc::N_STATIC_AIRS) / 2);
            };
            std::printf("{\"config\": \"%s\", \"log_frame\": %u, \"n_chips\": %u, \"phantom_hints\": [%s], \"chips\": [", argv[2], log_frame, caps.n_airs,
                        caps.ext.pairing ? "\"modular non-residue / square root\", \"pairing final exponentiation\"" : caps.moduli.empty() ? "" : "\"modular non-residue / square root\"");
            size_t cells = 0;
            for (unsigned p = 0; p < caps.n_airs; p++) {
                std::printf("%s{\"name\": \"%s\", \"columns\": %zu, \"log_height\": %u}", p ? ", " : "", name_of(caps.ids[p]).c_str(), sa.airs[p].width, sa.airs[p].log_height);
                cells += sa.airs[p].width << sa.airs[p].log_height;
            }
            std::printf("], \"main_cells\": %zu}\n", cells);
            return 0;
        }
        if (argc >= 3 && std::string(argv[1]) == "leaf-stats") {
            // leaf-stats <openvm.toml | -> [log_frame = 20] [children = 1]: builds, on the host, the leaf verifier circuit of every shape of the configuration
            // (dummy preprocessed commitments: the wiring does not depend on them) and prints its size -- wires, gate rows, permutations -- per shape;
            // with ZKHIP_RECURSION_TIMING=1 the builder also prints where the rows go.  No GPU.
            const unsigned log_frame = argc >= 4 ? (unsigned)atoi(argv[3]) : 20;
            const size_t children = argc >= 5 ? (size_t)atoi(argv[4]) : 1;
            const zkhip_vm::SegmentCaps full = zkhip_vm::config_caps(argv[2], log_frame, 10);
            const zkhip_vm::SegmentShapes shapes = zkhip_vm::SegmentShapes::of(full, log_frame, 10);
            zkhip_vm::Exe exe;
            exe.program.assign(64, 0x00000013u);
            const zkhip_params prm{1, 0, 100, 16, 16};
            static const uint32_t zero_commit[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t sh = 0; sh < shapes.caps.size(); sh++) {
                zkhip_vm::SegmentAirs sa = zkhip_vm::segment_airs(exe, shapes.caps[sh]);
                size_t width = 0, prog_words = 0;
                for (size_t p = 0; p < sa.airs.size(); p++) {
                    if (sa.prep_width[p]) sa.airs[p].prep_commit = zero_commit;
                    sa.airs[p].prep_trace = nullptr;
                    width += sa.airs[p].width, prog_words += sa.airs[p].program_len;
                }
                zkhip_recursion_stmt st{};
                zkhip_recursion* rc = nullptr;
                if (zkhip_recursion_build(&prm, sa.airs.data(), sa.airs.size(), children, &st, &rc) != ZKHIP_OK) {
                    std::fprintf(stderr, "shape %zu: %s\n", sh, zkhip_recursion_last_error(nullptr));
                    return 1;
                }
                size_t stt[4];
                zkhip_recursion_stats(rc, stt);
                std::printf("{\"shape\": %zu, \"chips\": %u, \"main_columns\": %zu, \"program_words\": %zu, \"children\": %zu, \"wires\": %zu, \"gate_rows\": %zu, \"permutations\": %zu, \"gate_rows_per_column_per_child\": %.1f}\n",
                            sh, shapes.caps[sh].n_airs, width, prog_words, children, stt[0], stt[1], stt[2], (double)stt[1] / (double)width / (double)children);
                zkhip_recursion_destroy(rc);
            }
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "dump-segments") {
            // dump-segments <guest.elf | exe.bin> <stdin.bin | -> <out_dir> <log_frame>: the segmenting executor's records (include/zkhip_vm_exec.hpp)
            // as raw u32 arrays, one directory per segment -- what the tests' CPU twins of the trace generators read
            const std::vector<uint8_t> raw = slurp(argv[2]);
            zkhip_vm::StdIn in;
            if (std::string(argv[3]) != "-") in.bytes = slurp(argv[3]);
            try {
                const zkhip_vm::Exe exe = zkhip_vm::parse_exe(raw);
                const zkhip_vm::SegmentCaps caps = zkhip_vm::SegmentCaps::for_frame((unsigned)atoi(argv[5]), zkhip_vm::vm_log2_ceil(exe.program.size()),
                                                                                    argc >= 7 ? (unsigned)atoi(argv[6]) : 0u, argc >= 8 ? (unsigned)atoi(argv[7]) : 0u,
                                                                                    argc >= 9 ? zkhip_vm::config_moduli(argv[8]) : std::vector<zkhip::modular::U256>(), 3,
                                                                                    argc >= 9 ? zkhip_vm::int256_log_rows(argv[8], 11) : 0u,
                                                                                    argc >= 9 ? zkhip_vm::config_curves(argv[8]) : std::vector<zkhip::ecc::Curve>(), 2,
                                                                                    argc >= 9 ? zkhip_vm::config_fp2_moduli(argv[8]) : std::vector<zkhip::modular::U256>(),
                                                                                    argc >= 9 ? zkhip_vm::config_native(argv[8]) : zkhip::native::Enabled());
                zkhip_vm::SegmentExecutor ex(exe, in, caps);
                zkhip_vm::SegmentRecords r;
                size_t k = 0;
                std::string roots;
                auto put = [](const std::string& path, const std::vector<uint32_t>& v) {
                    std::ofstream f(path, std::ios::binary);
                    f.write((const char*)v.data(), (std::streamsize)(v.size() * 4));
                };
                for (bool done = false; !done; k++) {
                    done = ex.run_segment(r);
                    const std::string d = std::string(argv[4]) + "/seg-" + std::to_string(k);
                    std::error_code ec;
                    std::filesystem::create_directories(d, ec);
                    if (ec) throw Error(Error::Io, "cannot create " + d + ": " + ec.message());
                    const zkhip_vm::ExecRecords& c = r.core;
#define PUT(name, vec) put(d + "/" name ".u32", vec)
                    PUT("pc_index", c.pc_index), PUT("alu_op", c.alu_op), PUT("alu_b", c.alu_b), PUT("alu_c", c.alu_c), PUT("lt_op", c.lt_op), PUT("lt_b", c.lt_b), PUT("lt_c", c.lt_c);
                    PUT("mul_b", c.mul_b), PUT("mul_c", c.mul_c), PUT("mulh_op", c.mulh_op), PUT("mulh_b", c.mulh_b), PUT("mulh_c", c.mulh_c);
                    PUT("div_op", c.div_op), PUT("div_b", c.div_b), PUT("div_c", c.div_c), PUT("shift_op", c.shift_op), PUT("shift_b", c.shift_b), PUT("shift_c", c.shift_c);
                    PUT("beq_op", c.beq_op), PUT("beq_a", c.beq_a), PUT("beq_b", c.beq_b), PUT("beq_imm", c.beq_imm), PUT("blt_op", c.blt_op), PUT("blt_a", c.blt_a), PUT("blt_b", c.blt_b), PUT("blt_imm", c.blt_imm);
                    PUT("jal_op", c.jal_op), PUT("jal_pc", c.jal_pc), PUT("jal_imm", c.jal_imm), PUT("auipc_pc", c.auipc_pc), PUT("auipc_imm", c.auipc_imm);
                    PUT("jalr_pc", c.jalr_pc), PUT("jalr_rs1", c.jalr_rs1), PUT("jalr_imm", c.jalr_imm), PUT("ls_case", c.ls_case), PUT("ls_read", c.ls_read), PUT("ls_prev", c.ls_prev);
                    PUT("f_x", r.f_x), PUT("f_y", r.f_y), PUT("f_z", r.f_z), PUT("f_rdprev", r.f_rdprev), PUT("f_pcinc", r.f_pcinc), PUT("ls_ts", r.ls_ts), PUT("ls_base", r.ls_base), PUT("ls_imm", r.ls_imm);
                    PUT("ecall_rows", r.ecall_rows), PUT("leaf_rows", r.leaf_rows), PUT("merkle_rows", r.merkle_rows), PUT("p2_inputs", r.p2_inputs);
                    PUT("f_pts1", r.f_pts1), PUT("f_pts2", r.f_pts2), PUT("f_pts3", r.f_pts3), PUT("ls_pts", r.ls_pts);
                    PUT("kk_states", r.kk_states), PUT("kk_ts", r.kk_ts), PUT("kio_rows", r.kio_rows);
                    PUT("sha_blocks", r.sha_blocks), PUT("sha_ts", r.sha_ts), PUT("shaio_rows", r.shaio_rows);
                    PUT("i256_records", r.i256_records), PUT("i256_ts", r.i256_ts), PUT("i256io_rows", r.i256io_rows);
                    PUT("nat_records", r.nat_records), PUT("next_records", r.next_records), PUT("castf_records", r.castf_records);
                    PUT("mul256_records", r.mul256_records), PUT("mul256_ts", r.mul256_ts), PUT("cmp256_records", r.cmp256_records), PUT("cmp256_ts", r.cmp256_ts), PUT("sh256_records", r.sh256_records), PUT("sh256_ts", r.sh256_ts);
                    for (unsigned i = 0; i < caps.moduli.size(); i++) {
                        const std::string t = std::to_string(i);
                        put(d + "/mm_records_" + t + ".u32", r.mm_records[i]), put(d + "/mm_ts_" + t + ".u32", r.mm_ts[i]), put(d + "/mmio_rows_" + t + ".u32", r.mmio_rows[i]);
                    }
                    for (unsigned i = 0; i < caps.fp2_moduli.size(); i++) {
                        const std::string t = std::to_string(i);
                        put(d + "/fp2_records_" + t + ".u32", r.fp2_records[i]), put(d + "/fp2_ts_" + t + ".u32", r.fp2_ts[i]), put(d + "/fp2io_rows_" + t + ".u32", r.fp2io_rows[i]);
                    }
                    for (unsigned i = 0; i < caps.curves.size(); i++) {
                        const std::string t = std::to_string(i);
                        put(d + "/ec_records_" + t + ".u32", r.ec_records[i]), put(d + "/ec_ts_" + t + ".u32", r.ec_ts[i]), put(d + "/ecio_rows_" + t + ".u32", r.ecio_rows[i]);
                    }
#undef PUT
                    std::vector<uint32_t> meta{r.pc_start, r.pc_end, r.ts_end, (uint32_t)r.n_instr};
                    meta.insert(meta.end(), r.root_init.begin(), r.root_init.end());
                    meta.insert(meta.end(), r.root_final.begin(), r.root_final.end());
                    put(d + "/meta.u32", meta);
                }
                std::vector<uint32_t> heights;
                for (unsigned id : caps.ids) heights.push_back(caps.log_height[id]);
                put(std::string(argv[4]) + "/air_ids.u32", std::vector<uint32_t>(caps.ids.begin(), caps.ids.end()));
                put(std::string(argv[4]) + "/heights.u32", heights);
                std::vector<uint32_t> fin(ex.image_root().begin(), ex.image_root().end());
                const std::vector<uint32_t> op = zkhip_vm::open_public_values(ex.tree());
                fin.insert(fin.end(), op.begin(), op.end());
                put(std::string(argv[4]) + "/image_root_and_pv_openings.u32", fin);
                std::printf("{\"segments\": %zu, \"total_cycle\": %llu, \"public_values\": \"", k, (unsigned long long)ex.instret());
                for (uint8_t b : ex.public_values()) std::printf("%02x", b);
                std::printf("\"}\n");
            } catch (const zkhip_vm::Error& e) {
                throw Error(Error::GenProof, e.what());
            }
            return 0;
        }
        if (argc >= 5 && (std::string(argv[1]) == "prove-elf" || std::string(argv[1]) == "prove-task")) {
            // prove-elf  <guest.elf | exe.bin> <stdin.bin | -> <out_dir> [openvm.toml | -] [log_frame]
            // prove-task <guest.elf | exe.bin> <task.bin>     <out_dir> [openvm.toml | -] [log_frame]
            // ONE FLOW (include/zkhip_vm_flow.hpp UniversalProver = the reference's Prover::gen_proof_universal, mod.rs:287-309): the
            // task's witnesses become the guest's input stream (ProvingTask::build_guest_input; prove-elf: the raw bytes are one
            // input stream), the guest is executed in segments, every segment is ONE statement, the aggregation tree folds them
            // into root.json (+ root.vk, the root circuit's verifying key); self-verified.  Prints a JSON line.
            const bool is_task = std::string(argv[1]) == "prove-task";
            std::string cfg_path = argc >= 6 && std::string(argv[5]) != "-" ? argv[5] : "";
            const std::string dir = argv[4];
            if (cfg_path.empty()) {
                cfg_path = dir + "/openvm.toml";
                std::ofstream(cfg_path) << "[app_fri_params.fri_params]\nlog_blowup = 1\nlog_final_poly_len = 0\nnum_queries = 100\n"
                                           "commit_proof_of_work_bits = 16\nquery_proof_of_work_bits = 16\n";
            }
            const unsigned log_frame = argc >= 7 ? (unsigned)atoi(argv[6]) : 20;
            UniversalProver up = UniversalProver::setup(ProverConfig{argv[2], cfg_path}, nullptr, 0, log_frame);
            ProvingTask task;
            if (is_task) {
                task = read_task(argv[3]);
            } else if (std::string(argv[3]) != "-") {
                // a raw input stream: handed to the guest as it is (one "witness" that IS the stream has no framing: bypass the task)
                task.identifier = "raw-stdin";
            }
            const auto t0 = std::chrono::steady_clock::now();
            StarkProof sp;
            if (is_task) {
                sp = up.gen_proof_universal(task);
            } else {
                // same flow, the stream given directly
                zkhip_vm::StdIn in;
                if (std::string(argv[3]) != "-") in.bytes = slurp(argv[3]);
                struct RawTask : ProvingTask {};
                try {
                    const zkhip_vm::SegmentCaps caps = zkhip_vm::SegmentCaps::for_frame(log_frame, zkhip_vm::vm_log2_ceil(up.exe().program.size()));
                    const zkhip_params params = read_app_config(cfg_path);
                    const unsigned log_keccak = zkhip_vm::keccak_log_rows(cfg_path, log_frame), log_sha256 = zkhip_vm::sha256_log_rows(cfg_path, log_frame);
                    const FlowOptions flow = FlowOptions::from_env();
                    const unsigned lanes = flow.lanes;   // segment provers in flight (measured 1 / 2 / 3: DESIGN.md 5)
                    const zkhip_vm::GuestStark g = zkhip_vm::prove_guest_universal(params, up.exe(), in, log_frame, 0, false, lanes, log_keccak, log_sha256, zkhip_vm::config_moduli(cfg_path),
                                                                                          zkhip_vm::int256_log_rows(cfg_path, log_frame), zkhip_vm::config_curves(cfg_path),
                                                                                          zkhip_vm::config_fp2_moduli(cfg_path), flow, zkhip_vm::config_native(cfg_path));
                    (void)caps;
                    sp = UniversalProver::encode(g);
                    std::string why;
                    if (!UniversalProver::verify_guest_stark(sp, g.root_vk, up.exe(), &why)) throw Error(Error::VerifyProof, "failed to verify proof: " + why);
                    const std::vector<uint8_t> vkb = g.root_vk.to_app_exe();
                    std::ofstream(dir + "/root.vk", std::ios::binary).write((const char*)vkb.data(), (std::streamsize)vkb.size());
                    std::ofstream(dir + "/root.json") << sp.to_json();
                    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    std::printf("{\"total_cycles\": %llu, \"segments\": %zu, \"levels\": %zu, \"lanes\": %u, \"execution_ms\": %llu, \"segment_tracegen_and_proving_ms\": %llu, "
                                "\"sum_over_segments_tracegen_ms\": %llu, \"sum_over_segments_prove_ms\": %llu, "
                                "\"aggregation_setup_wait_ms\": %llu, \"aggregation_ms\": %llu, \"wall_s\": %.3f, \"root_proof_bytes\": %zu, \"chips_per_shape\": %s, "
                                "\"segments_per_shape\": %s, \"instructions_per_shape\": %s, \"sum_prove_ms_per_shape\": %s, \"sum_tracegen_ms_per_shape\": %s, \"tree_nodes_per_device_slot\": %s, \"aggregation_circuits_build_s\": %.2f, \"aggregation_keygen_s\": %.2f, \"leaf_circuits_at_setup\": %zu, \"leaf_circuits_on_demand\": %zu, "
                                "\"segments_retried\": %zu, \"segments_retried_detail\": %s, \"retry_enabled\": %s, \"library_has_test_kernels\": %d, "
                                "\"segments_per_lane\": %s, \"devices\": %s, \"executor_record_threads\": %u, \"executor_metered_pass_busy_ms\": %llu, \"executor_record_passes_busy_ms_sum\": %llu, \"executor_memory_tree_busy_ms\": %llu, "
                                "\"node_log_heights\": [%u, %u], \"verified\": true}\n",
                                (unsigned long long)g.exec.total_cycle, g.segments, g.levels, lanes, (unsigned long long)g.execution_mills,
                                (unsigned long long)g.segment_proving_mills, (unsigned long long)g.sum_segment_tracegen_mills,
                                (unsigned long long)g.sum_segment_prove_mills, (unsigned long long)g.aggregation_setup_wait_mills,
                                (unsigned long long)g.aggregation_mills, wall, g.root.proof.size(), list_of(g.chips_per_shape).c_str(), list_of(g.segments_per_shape).c_str(), list_of(g.instr_per_shape).c_str(), list_of(g.prove_ms_per_shape).c_str(), list_of(g.tracegen_ms_per_shape).c_str(), list_of(g.nodes_per_slot).c_str(), g.agg_build_seconds, g.agg_keygen_seconds, g.leaf_circuits_at_setup, g.leaf_circuits_on_demand,
                                g.segments_retried.size(), retried_json(g.segments_retried).c_str(), flow.retry_segments ? "true" : "false", zkhip_has_test_kernels(),
                                list_of(g.segments_per_lane).c_str(), list_of(std::vector<size_t>(flow.devices.begin(), flow.devices.end())).c_str(),
                                g.executor_threads, (unsigned long long)g.executor_metered_mills, (unsigned long long)g.executor_record_mills_sum, (unsigned long long)g.executor_tree_mills,
                                g.node_log_heights.at(0), g.node_log_heights.at(1));
                } catch (const zkhip_vm::Error& e) {
                    throw Error(Error::GenProof, e.what());
                }
                return 0;
            }
            const zkhip_vm::GuestStark& g = up.last();
            const std::vector<uint8_t> vkb = g.root_vk.to_app_exe();
            std::ofstream(dir + "/root.vk", std::ios::binary).write((const char*)vkb.data(), (std::streamsize)vkb.size());
            std::ofstream(dir + "/root.json") << sp.to_json();
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("{\"identifier\": \"%s\", \"total_cycles\": %llu, \"segments\": %zu, \"levels\": %zu, \"execution_ms\": %llu, "
                        "\"segment_tracegen_and_proving_ms\": %llu, \"aggregation_ms\": %llu, \"wall_s\": %.3f, \"root_proof_bytes\": %zu, "
                        "\"segments_retried\": %zu, \"segments_retried_detail\": %s, \"verified\": true}\n",
                        task.identifier.c_str(), (unsigned long long)g.exec.total_cycle, g.segments, g.levels, (unsigned long long)g.execution_mills,
                        (unsigned long long)g.segment_proving_mills, (unsigned long long)g.aggregation_mills, wall, g.root.proof.size(),
                        g.segments_retried.size(), retried_json(g.segments_retried).c_str());
            return 0;
        }
        if (argc >= 5 && std::string(argv[1]) == "agg-vk") {
            // agg-vk <guest.elf> <openvm.toml> <out root.vk> [log_frame = 20]: THE aggregation key of this app for someone else (a verifier, a
            // parent guest's program commitment) -- crates/prover/src/prover/mod.rs:147-170 `get_agg_vk`.  Unlike a flow, which takes the
            // commitments of shapes it does not use from the key cache, this builds every shape and refuses a cache entry that does not match.
            const unsigned log_frame = argc >= 6 ? (unsigned)atoi(argv[5]) : 20;
            UniversalProver up = UniversalProver::setup(ProverConfig{argv[2], argv[3]}, nullptr, 0, log_frame);
            const std::vector<uint8_t> vkb = up.get_agg_vk().to_app_exe();
            std::ofstream(argv[4], std::ios::binary).write((const char*)vkb.data(), (std::streamsize)vkb.size());
            std::printf("{\"root_vk_bytes\": %zu}\n", vkb.size());
            return 0;
        }
        if (argc >= 5 && std::string(argv[1]) == "program-commit") {
            // program-commit <guest.elf | exe.bin> <root.vk> <openvm.toml>: what a PARENT guest holds about this app (crates/types/circuit/src/lib.rs
            // ProgramCommitment { exe, vm }; the reference generates it into crates/circuits/*-circuit/*_commit.rs): host only
            const zkhip_vm::Exe exe = zkhip_vm::parse_exe(slurp(argv[2]));
            const VerifyingKey vk = VerifyingKey::read(argv[3], read_app_config(argv[4]));
            const ProgramCommitment pc = ProgramCommitment::of(vk, exe.entry ? exe.entry : exe.pc_base, zkhip_vm::guest_image_root(exe));
            std::printf("{\"exe\": [");
            for (int k = 0; k < 8; k++) std::printf("%s%u", k ? ", " : "", pc.exe[k]);
            std::printf("], \"vm\": [");
            for (int k = 0; k < 8; k++) std::printf("%s%u", k ? ", " : "", pc.vm[k]);
            std::printf("], \"deferral_base\": %u}\n", zkhip_vm::deferral_base(exe));
            return 0;
        }
        if (argc >= 10 && std::string(argv[1]) == "prove-deferral") {
            // prove-deferral <guest.elf> <openvm.toml> <out_dir> <log_frame> <child root.vk> <child openvm.toml> <witness.bin | -> <child root.json>...
            // A guest that DEFERS the verification of its children (crates/prover/src/prover/mod.rs:200-282 enable_deferral; crates/integration/
            // src/lib.rs:461-514 compute_deferral_data, :556-571 prove_task_with_deferral): the child app's aggregation key + the child root
            // proofs in, ONE StarkProof out -- the guest's root joined with the proof of the deferral node that verified the children.  The
            // task's input stream = [witness.bin as one item][the input commitments].  Writes root.json and root.vk (the JOIN key).
            const unsigned log_frame = (unsigned)atoi(argv[5]);
            UniversalProver up = UniversalProver::setup(ProverConfig{argv[2], argv[3]}, "deferral", 0, log_frame);
            // (<child root.vk>@<child guest.elf>: a child app that itself defers -- its key is a join key, a BUNDLE over batches -- whose
            // deferral region's place in its memory the deferral node hard-wires)
            std::string child_vk_path = argv[6], child_elf;
            if (const size_t at = child_vk_path.find('@'); at != std::string::npos) child_elf = child_vk_path.substr(at + 1), child_vk_path.resize(at);
            const VerifyingKey child_key = VerifyingKey::read(child_vk_path, read_app_config(argv[7]));
            if (child_key.join && child_elf.empty()) throw Error(Error::Setup, "the child key is a join key (the child app defers): give <child root.vk>@<child guest.elf>");
            const uint32_t child_region = child_elf.empty() ? 0u : zkhip_vm::deferral_region_index(zkhip_vm::parse_exe(slurp(child_elf)));
            const auto t0 = std::chrono::steady_clock::now();
            up.enable_deferral(child_key, up.config.flow.deferral_children, child_key.join ? child_region : 0u, up.config.flow.deferral_nodes);
            std::vector<StarkProof> kids;
            for (int i = 9; i < argc; i++) {
                const auto js = slurp(argv[i]);
                kids.push_back(StarkProof::from_json(std::string(js.begin(), js.end())));
            }
            std::vector<const StarkProof*> kp;
            for (const auto& k : kids) kp.push_back(&k);
            const DeferralProver::Data data = up.compute_deferral_data(kp);
            up.warm_up();   // (the guest flow's own keys and circuits: setup)
            ProvingTask task;
            task.identifier = "deferral";
            if (std::string(argv[8]) != "-") task.serialized_witness.push_back(slurp(argv[8]));
            task.input_commits = data.input_commits;
            const auto t1 = std::chrono::steady_clock::now();
            const StarkProof sp = up.gen_proof_stark(task, data.inputs);
            const auto t2 = std::chrono::steady_clock::now();
            const std::string dir = argv[4];
            const std::vector<uint8_t> vkb = up.get_agg_vk().to_app_exe();
            std::ofstream(dir + "/root.vk", std::ios::binary).write((const char*)vkb.data(), (std::streamsize)vkb.size());
            std::ofstream(dir + "/root.json") << sp.to_json();
            {   // the deferral node's proof and key (what the join verified): for inspection and the oracle comparison of the tests
                const VerifyingKey& dvk = up.deferral()->join_child_vk();   // (the deferral node's key, or the fold's)
                std::ofstream(dir + "/deferral.json") << AggregationProver::to_stark_proof(up.last_deferral_proof(), dvk).to_json();
                const std::vector<uint8_t> db = dvk.to_app_exe();
                std::ofstream(dir + "/deferral.vk", std::ios::binary).write((const char*)db.data(), (std::streamsize)db.size());
            }
            const zkhip_vm::GuestStark& g = up.last();
            std::printf("{\"children\": %zu, \"total_cycles\": %llu, \"segments\": %zu, \"levels\": %zu, \"setup_s\": %.3f, \"prove_s\": %.3f, \"root_proof_bytes\": %zu, "
                        "\"deferral_state\": [", kids.size(), (unsigned long long)g.exec.total_cycle, g.segments, g.levels,
                        std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), g.root.proof.size());
            for (int k = 0; k < 8; k++) std::printf("%s%u", k ? ", " : "", data.state[k]);
            std::printf("], \"verified\": true}\n");
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "verify-guest") {
            // verify-guest <guest.elf | exe.bin> <root.vk> <openvm.toml> <root.json>: the root proof under the root verifying key AND the
            // statement about this guest: entry pc, memory image, exit code 0, public values opened in the final memory root (host only)
            const zkhip_vm::Exe exe = zkhip_vm::parse_exe(slurp(argv[2]));
            const VerifyingKey vk = VerifyingKey::read(argv[3], read_app_config(argv[4]));   // (an aggregation key: + leaf commitment, app digest)
            std::ifstream f(argv[5]);
            std::stringstream ss;
            ss << f.rdbuf();
            const StarkProof sp = StarkProof::from_json(ss.str());
            std::string why;
            if (!UniversalProver::verify_guest_stark(sp, vk, exe, &why)) throw Error(Error::VerifyProof, "failed to verify proof: " + why);
            std::printf("verified: public values ");
            for (size_t i = 0; i < 32; i++) std::printf("%02x", sp.user_pvs_proof[4 * vk.airs.back().n_pvs + i]);
            std::printf("\n");
            return 0;
        }
        if (argc >= 5 && std::string(argv[1]) == "prove-guest") {
            // prove-guest <guest.elf | exe.bin> <stdin.bin | -> <out.json> [openvm.toml] [max_cost]: the whole gen_proof_stark flow
            // (mod.rs:342-413) from a guest image: execute, generate the chips' traces on the device, prove, self-verify
            std::ifstream f(argv[2], std::ios::binary);
            if (!f) throw Error(Error::Io, std::string("cannot read ") + argv[2]);
            const std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            zkhip_vm::StdIn in;
            if (std::string(argv[3]) != "-") {
                std::ifstream g(argv[3], std::ios::binary);
                if (!g) throw Error(Error::Io, std::string("cannot read ") + argv[3]);
                in.bytes.assign((std::istreambuf_iterator<char>(g)), std::istreambuf_iterator<char>());
            }
            zkhip_params params{1, 0, 100, 16, 16};
            if (argc >= 6 && std::string(argv[5]) != "-") params = read_app_config(argv[5]);
            const uint64_t max_cost = argc >= 7 ? strtoull(argv[6], nullptr, 10) : 0;
            const uint64_t segment_instr = argc >= 8 ? strtoull(argv[7], nullptr, 10) : 0;
            const unsigned inflight = argc >= 9 ? (unsigned)atoi(argv[8]) : 1;
            if (segment_instr) {
                // continuation: <out_proof.json> is a directory; one StarkProof per segment, proven over `inflight` lanes
                try {
                    const zkhip_vm::SegmentedProof sg = zkhip_vm::prove_segments(params, zkhip_vm::parse_exe(raw), in, segment_instr, inflight);
                    uint64_t cycles = 0, ms_sum = 0;
                    for (size_t k = 0; k < sg.segments.size(); k++) {
                        const zkhip_vm::GuestProof& gp = sg.segments[k];
                        StarkProof sp;
                        sp.proof = gp.proof;
                        if (k + 1 == sg.segments.size()) sp.user_pvs_proof = sg.exec.public_values;
                        for (unsigned h : gp.log_heights) sp.baseline.push_back((uint8_t)h);
                        sp.stat.total_cycles = gp.exec.total_cycle;
                        sp.stat.proving_time_mills = gp.proving_time_mills;
                        std::ofstream(std::string(argv[4]) + "/segment-" + std::to_string(k) + ".json") << sp.to_json();
                        cycles += gp.exec.total_cycle, ms_sum += gp.proving_time_mills + gp.tracegen_time_mills;
                    }
                    std::string per;
                    for (const auto& gp : sg.segments) per += (per.empty() ? "[" : ", [") + std::to_string(gp.tracegen_time_mills) + ", " + std::to_string(gp.proving_time_mills) + "]";
                    printf("{\"total_cycles\": %llu, \"segments\": %zu, \"inflight\": %u, \"execution_ms\": %llu, \"proving_wall_ms\": %llu, "
                           "\"sum_of_segment_ms\": %llu, \"segment_tracegen_and_proving_ms\": [%s], \"verified\": true}\n",
                           (unsigned long long)cycles, sg.segments.size(), inflight, (unsigned long long)sg.execution_time_mills,
                           (unsigned long long)sg.proving_wall_mills, (unsigned long long)ms_sum, per.c_str());
                    if (cycles != sg.exec.total_cycle) throw Error(Error::GenProof, "segment cycle counts do not add up");
                } catch (const zkhip_vm::Error& e) {
                    throw Error(Error::GenProof, e.what());
                }
                return 0;
            }
            try {
                zkhip_vm::VmProver vp(0);
                const zkhip_vm::GuestProof gp = vp.prove_guest(params, zkhip_vm::parse_exe(raw), in, max_cost);
                StarkProof sp;
                sp.proof = gp.proof;
                sp.user_pvs_proof = gp.exec.public_values;
                for (unsigned h : gp.log_heights) sp.baseline.push_back((uint8_t)h);
                sp.stat.total_cycles = gp.exec.total_cycle;
                sp.stat.execution_time_mills = gp.execution_time_mills;
                sp.stat.proving_time_mills = gp.proving_time_mills;
                std::ofstream(argv[4]) << sp.to_json();
                printf("{\"total_cycles\": %llu, \"proof_bytes\": %zu, \"execution_ms\": %llu, \"tracegen_ms\": %llu, \"proving_ms\": %llu, \"verified\": true}\n",
                       (unsigned long long)gp.exec.total_cycle, gp.proof.size(), (unsigned long long)gp.execution_time_mills,
                       (unsigned long long)gp.tracegen_time_mills, (unsigned long long)gp.proving_time_mills);
            } catch (const zkhip_vm::Error& e) {
                throw Error(Error::GenProof, e.what());
            }
            return 0;
        }
        if (argc >= 5 && std::string(argv[1]) == "verify") {
            UniversalVerifier verifier = UniversalVerifier::setup(argv[2], argv[3]);  // no device needed
            auto js = slurp(argv[4]);
            StarkProof sp = StarkProof::from_json(std::string(js.begin(), js.end()));
            verifier.verify_stark_proof(sp);
            // an AGGREGATION key (root.vk of prove-agg / prove-elf) also says what the root must state beneath it: the tree's leaves are
            // proofs of the key's leaf circuit, its nodes of the key's own circuit, the app is the key's
            const VerifyingKey vk = VerifyingKey::read(argv[2], read_app_config(argv[3]));
            if (vk.is_aggregation_key()) {
                const size_t n = vk.airs.back().n_pvs;
                std::vector<uint32_t> stmt(n);
                if (sp.user_pvs_proof.size() < 4 * n) throw Error(Error::VerifyProof, "failed to verify proof: short root statement");
                memcpy(stmt.data(), sp.user_pvs_proof.data(), 4 * n);
                std::string why;
                if (!vk.root_statement_matches(stmt, &why)) throw Error(Error::VerifyProof, "failed to verify proof: " + why);
            }
            std::printf("verified: %zu proof bytes, proving_time_mills=%llu\n", sp.proof.size(),
                        (unsigned long long)sp.stat.proving_time_mills);
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "prove") {
            Prover prover = Prover::setup(ProverConfig{argv[2], argv[3]}, "cli");
            auto tb = slurp(argv[4]);
            ProvingTask task;
            size_t p = 0;
            auto rd32 = [&]() { uint32_t v; if (p + 4 > tb.size()) throw Error(Error::Io, "short task"); memcpy(&v, &tb[p], 4); p += 4; return v; };
            auto rd64 = [&]() { uint64_t v; if (p + 8 > tb.size()) throw Error(Error::Io, "short task"); memcpy(&v, &tb[p], 8); p += 8; return v; };
            uint32_t idl = rd32();
            task.identifier.assign((const char*)&tb[p], idl);
            p += idl;
            uint32_t n = rd32();
            for (uint32_t i = 0; i < n; i++) {
                uint64_t len = rd64();
                if (p + len > tb.size()) throw Error(Error::Io, "short task");
                task.serialized_witness.emplace_back(tb.begin() + p, tb.begin() + p + len);
                p += len;
            }
            task.fork_name = "hip";
            StarkProof sp = prover.gen_proof_universal(task, false);
            std::ofstream(argv[5]) << sp.to_json();
            std::printf("proved %s: %zu proof bytes, execution %llu ms, proving %llu ms\n", task.identifier.c_str(),
                        sp.proof.size(), (unsigned long long)sp.stat.execution_time_mills,
                        (unsigned long long)sp.stat.proving_time_mills);
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "prove-many") {
            // prove-many <app> <cfg> <out_dir> <task.bin>...  : one Prover, keys reused across tasks, like the
            // reference's multi-chunk test (crates/integration/tests/chunk_circuit.rs:198-208); reset() in between
            // every second task exercises Prover::reset (mod.rs:106-108)
            Prover prover = Prover::setup(ProverConfig{argv[2], argv[3]}, "cli");
            for (int i = 5; i < argc; i++) {
                auto tb = slurp(argv[i]);
                ProvingTask task;
                size_t p = 0;
                auto rd32 = [&]() { uint32_t v; memcpy(&v, &tb[p], 4); p += 4; return v; };
                auto rd64 = [&]() { uint64_t v; memcpy(&v, &tb[p], 8); p += 8; return v; };
                uint32_t idl = rd32();
                task.identifier.assign((const char*)&tb[p], idl);
                p += idl;
                uint32_t n = rd32();
                for (uint32_t k = 0; k < n; k++) {
                    uint64_t len = rd64();
                    task.serialized_witness.emplace_back(tb.begin() + p, tb.begin() + p + len);
                    p += len;
                }
                StarkProof sp = prover.gen_proof_universal(task, false);
                std::ofstream(std::string(argv[4]) + "/" + task.identifier + ".json") << sp.to_json();
                std::printf("proved %s: proving %llu ms\n", task.identifier.c_str(), (unsigned long long)sp.stat.proving_time_mills);
                if ((i - 5) % 2 == 1) prover.reset();
            }
            return 0;
        }
        if (argc >= 7 && std::string(argv[1]) == "prove-batch") {
            // prove-batch <app> <cfg> <out_dir> <inflight> <task.bin>... : BatchProver::prove_many -- the tasks are queued
            // over `inflight` Provers on GPU 0 (the replacement of the sequential chunk loop, batch.rs:97-107)
            // ZKHIP_BATCH_DEVICES="0,0,1": the device of every lane group (default "0").  Listing one device several times maps the
            // multi-device queue (one lane group per entry, tasks spread over all of them) onto a single GPU: the self-test of the
            // cross-device path on a one-GPU box (tests/test_gpu_prover_mirror.py).
            std::vector<int> devs{0};
            if (const char* e = getenv("ZKHIP_BATCH_DEVICES")) {
                devs.clear();
                std::stringstream ss(e);
                std::string item;
                while (std::getline(ss, item, ',')) devs.push_back(std::stoi(item));
                if (devs.empty()) throw Error(Error::Setup, "ZKHIP_BATCH_DEVICES is empty");
            }
            BatchProver bp = BatchProver::setup(ProverConfig{argv[2], argv[3]}, (unsigned)std::stoul(argv[5]), devs);
            std::vector<ProvingTask> tasks;
            for (int i = 6; i < argc; i++) tasks.push_back(read_task(argv[i]));
            BatchProver::Stats st;
            std::vector<StarkProof> proofs = bp.prove_many(tasks, &st);
            for (size_t i = 0; i < tasks.size(); i++) {
                std::ofstream(std::string(argv[4]) + "/" + tasks[i].identifier + ".json") << proofs[i].to_json();
                std::printf("proved %s: proving %llu ms\n", tasks[i].identifier.c_str(), (unsigned long long)proofs[i].stat.proving_time_mills);
            }
            std::printf("batch: %zu proofs on %zu lanes in %.3f s\n", st.proofs, bp.lanes(), st.seconds);
            return 0;
        }
        if (argc >= 3 && std::string(argv[1]) == "decode-v1") {
            // decode-v1 <proof.json>: a proof file as the reference stores them (crates/verifier/testdata/proofs/*.json:
            // {"proof": {"proofs": base64(bincode Vec<Proof<SC>>), "public_values": base64(bincode Vec<BabyBear>)}, "vk": ..}):
            // decodes the OpenVM-v1 container (include/zkhip_codec.hpp), checks the byte-exact round trip, prints its shape.  No device.
            auto js = slurp(argv[2]);
            std::string text(js.begin(), js.end());
            auto field = [&](const char* key) -> std::string {
                std::string k = std::string("\"") + key + "\"";
                size_t p = text.find(k);
                if (p == std::string::npos) throw Error(Error::Custom, std::string("no field ") + key);
                p = text.find('"', text.find(':', p + k.size()) + 1);
                size_t e = text.find('"', p + 1);
                if (p == std::string::npos || e == std::string::npos) throw Error(Error::Custom, "malformed json");
                return text.substr(p + 1, e - p - 1);
            };
            std::vector<uint8_t> blob = base64_decode(field("proofs")), pvs = base64_decode(field("public_values"));
            zkhip_v1_summary sm;
            int rc = zkhip_proof_decode_v1(blob.data(), blob.size(), ZKHIP_V1_VEC, &sm);
            if (rc != ZKHIP_OK) throw Error(Error::Custom, "not a well-formed v1 proof container (zkhip_proof_decode_v1 returned " + std::to_string(rc) + ")");
            std::vector<uint8_t> again(blob.size() + 64);
            size_t n = 0;
            rc = zkhip_proof_reencode_v1(blob.data(), blob.size(), ZKHIP_V1_VEC, again.data(), again.size(), &n);
            const bool same = rc == ZKHIP_OK && n == blob.size() && memcmp(again.data(), blob.data(), n) == 0;
            uint64_t n_pv = 0;
            if (pvs.size() >= 8) memcpy(&n_pv, pvs.data(), 8);
            std::printf("{\"bytes\": %zu, \"roundtrip_exact\": %s, \"n_proofs\": %zu, \"n_airs\": %zu, \"n_queries\": %zu, \"n_fri_layers\": %zu, "
                        "\"n_main_commits\": %zu, \"n_after_challenge_commits\": %zu, \"n_preprocessed\": %zu, \"n_input_batches\": %zu, "
                        "\"log_max_height\": %u, \"log_blowup\": %u, \"user_public_values\": %llu, \"log_degrees\": [",
                        blob.size(), same ? "true" : "false", sm.n_proofs, sm.n_airs, sm.n_queries, sm.n_fri_layers, sm.n_main_commits,
                        sm.n_after_challenge_commits, sm.n_preprocessed, sm.n_input_batches, sm.log_max_height, sm.log_blowup,
                        (unsigned long long)n_pv);
            for (size_t a = 0; a < sm.n_airs && a < ZKHIP_V1_MAX_AIRS; a++) std::printf("%s%u", a ? ", " : "", sm.log_degree[a]);
            std::printf("]}\n");
            return same ? 0 : 1;
        }
        if (argc >= 3 && std::string(argv[1]) == "agg-plan") {
            // agg-plan <n_segments> [leaf_arity internal_arity]: prints the aggregation tree (mod.rs:57-60 defaults 4 / 3), no device
            AggregationTreeConfig cfg;
            if (argc >= 5) cfg.num_children_leaf = (unsigned)std::stoul(argv[3]), cfg.num_children_internal = (unsigned)std::stoul(argv[4]);
            AggregationPlan plan = AggregationPlan::build(std::stoul(argv[2]), cfg);
            std::printf("{\"n_segments\": %zu, \"n_nodes\": %zu, \"levels\": [", plan.n_segments, plan.n_nodes());
            for (size_t l = 0; l < plan.levels.size(); l++) {
                std::printf("%s[", l ? ", " : "");
                for (size_t n = 0; n < plan.levels[l].size(); n++) {
                    std::printf("%s[", n ? ", " : "");
                    for (size_t k = 0; k < plan.levels[l][n].children.size(); k++) std::printf("%s%zu", k ? ", " : "", plan.levels[l][n].children[k]);
                    std::printf("]");
                }
                std::printf("]");
            }
            std::printf("]}\n");
            return 0;
        }
        if (argc >= 8 && std::string(argv[1]) == "prove-agg") {
            // prove-agg <seg_app> <cfg> <out_dir> <inflight> <state> <segment task.bin>...: the segments through a BatchProver,
            // then the aggregation tree (leaf arity 4, internal 3: mod.rs:57-60) through REAL node circuits -- every node proves
            // that its children verify under the child verifying key (zkhip_aggregation.hpp).  <state> = where a segment's chained
            // state lives in its public values: "air:idx,air:idx/air:idx,air:idx" (start words / end words), or "-" for none.
            // Writes root.json, root.vk (the root circuit's verifying key in app-file form: `prove_cli verify root.vk cfg root.json`)
            // and every node proof.
            BatchProver seg = BatchProver::setup(ProverConfig{argv[2], argv[3]}, (unsigned)std::stoul(argv[5]));
            StatementSpec spec;
            {
                const std::string st = argv[6];
                if (st != "-") {
                    const size_t slash = st.find('/');
                    if (slash == std::string::npos) throw Error(Error::Setup, "state spec: start/end");
                    auto parse = [](const std::string& part) {
                        std::vector<std::pair<uint32_t, uint32_t>> v;
                        std::stringstream ss(part);
                        std::string item;
                        while (std::getline(ss, item, ',')) {
                            const size_t c = item.find(':');
                            if (c == std::string::npos) throw Error(Error::Setup, "state spec: air:idx");
                            v.push_back({(uint32_t)std::stoul(item.substr(0, c)), (uint32_t)std::stoul(item.substr(c + 1))});
                        }
                        return v;
                    };
                    spec.start = parse(st.substr(0, slash)), spec.end = parse(st.substr(slash + 1));
                }
            }
            std::vector<ProvingTask> tasks;
            for (int i = 7; i < argc; i++) tasks.push_back(read_task(argv[i]));
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<StarkProof> seg_proofs = seg.prove_many(tasks);
            const auto t1 = std::chrono::steady_clock::now();
            // the app verifying key: the segment app's AIRs with the (fixed) heights of the segment proofs
            // (a lane that has proven holds the commitments of the app's preprocessed tables, adopted at its keygen)
            VerifyingKey avk;
            {
                size_t best = 0;
                for (size_t l = 0; l < seg.lanes(); l++) {
                    bool ok = true;
                    for (const auto& a : seg.lane(l).airs()) ok = ok && (!a.has_prep || a.prep_commit.size() == 8);
                    if (ok) best = l;
                }
                avk.params = seg.lane(best).params(), avk.airs = seg.lane(best).airs();
            }
            for (uint8_t h : seg_proofs.at(0).baseline) avk.heights.push_back(h);
            for (auto& a : avk.airs) {
                if (a.has_prep && a.prep_commit.size() != 8) throw Error(Error::Setup, "no lane holds the commitments of the app's preprocessed tables");
                a.prep.clear();
            }
            seg.reset();
            AggregationTreeConfig tcfg;
            tcfg.one_key = !FlowOptions::from_env().per_depth_keys;   // (round 3's keys: every level hard-wires the level below)
            AggregationProver agg = AggregationProver::setup(avk, spec, tcfg);
            AggregationPlan plan = AggregationPlan::build(seg_proofs.size(), tcfg);
            std::vector<ChildProof> segs;
            for (const auto& sp : seg_proofs) segs.push_back(AggregationProver::from_stark_proof(sp, avk));
            (void)agg.node_vk(plan.levels.size() - 1);  // circuits + keys of every level: setup, not proving
            const auto t2 = std::chrono::steady_clock::now();
            std::vector<std::vector<ChildProof>> all;
            const bool greedy = tcfg.one_key && FlowOptions::from_env().agg_cli_greedy;
            ChildProof root = agg.prove_tree(plan, segs, &all, nullptr, greedy);
            const auto t3 = std::chrono::steady_clock::now();
            const VerifyingKey& rvk = agg.root_vk(plan.levels.size());   // one key: THE aggregation key, whatever the depth
            if (rvk.is_aggregation_key()) {
                std::string why;
                if (!rvk.root_statement_matches(root.pvs[2], &why)) throw Error(Error::VerifyProof, "failed to verify proof: " + why);
            }
            const std::string dir = argv[4];
            std::ofstream(dir + "/root.json") << AggregationProver::to_stark_proof(root, rvk).to_json();
            {
                const std::vector<uint8_t> vkb = rvk.to_app_exe();
                std::ofstream(dir + "/root.vk", std::ios::binary).write((const char*)vkb.data(), (std::streamsize)vkb.size());
            }
            for (size_t l = 0; l < all.size(); l++)
                for (size_t n = 0; n < all[l].size(); n++)
                    std::ofstream(dir + "/agg-" + std::to_string(l) + "-" + std::to_string(n) + ".json") << AggregationProver::to_stark_proof(all[l][n], agg.node_vk(l)).to_json();
            const size_t n_nodes = greedy ? all[0].size() + all[1].size() : plan.n_nodes(), n_levels = greedy ? 2 : plan.levels.size();
            auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
            std::printf("{\"segments\": %zu, \"nodes\": %zu, \"levels\": %zu, \"segment_seconds\": %.4f, \"setup_seconds\": %.4f, \"tree_seconds\": %.4f, "
                        "\"proofs_folded_per_s\": %.2f, \"witness_seconds\": %.4f, \"tracegen_prove_seconds\": %.4f, \"self_verify_seconds\": %.4f, "
                        "\"root_proof_bytes\": %zu, \"root_public_values\": [",
                        seg_proofs.size(), n_nodes, n_levels, secs(t0, t1), secs(t1, t2), secs(t2, t3), seg_proofs.size() / secs(t2, t3),
                        agg.stats.witness_seconds, agg.stats.tracegen_prove_seconds, agg.stats.verify_seconds, root.proof.size());
            for (size_t i = 0; i < root.pvs[2].size(); i++) std::printf("%s%u", i ? ", " : "", root.pvs[2][i]);
            std::printf("]}\n");
            return 0;
        }
        if (argc >= 7 && std::string(argv[1]) == "bench-many") {
            // bench-many <app> <cfg> <task.bin> <n_proofs> <inflight> [n_gpus]: BatchProver::prove_repeated -- the witness is
            // uploaded once per lane (traces resident in HBM), then n proofs are proven and self-verified through the
            // Prover API with `inflight` Provers per GPU; prints one JSON line.  The API-level twin of bench.py.
            const size_t n = std::stoul(argv[5]);
            const unsigned inflight = (unsigned)std::stoul(argv[6]);
            const int n_gpus = argc >= 8 ? std::stoi(argv[7]) : 1;
            std::vector<int> devs;
            for (int d = 0; d < n_gpus; d++) devs.push_back(d);
            BatchProver bp = BatchProver::setup(ProverConfig{argv[2], argv[3]}, inflight, devs);
            ProvingTask task = read_task(argv[4]);
            std::vector<uint8_t> last;
            BatchProver::Stats st = bp.prove_repeated(task, n, &last);
            std::printf("{\"api\": \"BatchProver::prove_repeated\", \"proofs\": %zu, \"seconds\": %.4f, \"proofs_per_s\": %.4f, "
                        "\"ms_per_proof\": %.2f, \"inflight_per_gpu\": %u, \"n_gpus\": %d, \"proof_bytes\": %zu, \"self_verified\": true}\n",
                        st.proofs, st.seconds, st.proofs_per_second, st.proofs ? 1e3 * st.seconds / st.proofs : 0.0, inflight, n_gpus,
                        last.size());
            return 0;
        }
        std::fprintf(stderr, "usage: prove_cli prove|prove-many|prove-batch|prove-agg|agg-plan|bench-many|decode-v1|verify ...\n");
        return 2;
    } catch (const Error& e) {
        std::fprintf(stderr, "error(kind %d): %s\n", (int)e.kind, e.what());
        return 1;
    } catch (const std::exception& e) {   // a malformed number on the command line, an allocation failure: an error, not a crash
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
