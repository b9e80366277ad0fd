// zkhip_vm_prover.hpp -- guest in, proof out: the reference's Prover::gen_proof_stark flow (crates/prover/src/prover/mod.rs:342-413:
// execute the guest, prove, encode, self-verify) over this backend, in C++ on the C ABI alone:
//   1. zkhip_vm::execute_guest (include/zkhip_vm.hpp)            -> cycle count, public values, per-chip execution records
//   2. records -> device buffers -> the device trace generators   (include/zkhip.h: zkhip_*_tracegen; no trace ever exists on the host)
//   3. zkhip_keygen over the chips' AIRs (include/zkhip_chips.hpp), zkhip_prove, zkhip_verify
// The AIR set (order fixed, it is part of the verifying key): program (cached program + execution frequencies), execution
// frames, RV32 base ALU / less-than / multiplication / high multiplication / division / shift / branch-equal / branch-less-than / JAL-LUI / AUIPC / JALR / load-store cores, 8-bit bitwise lookup, range-tuple checker, memory access rows, memory
// boundary, 16-bit range checker.  What these nineteen AIRs prove together is stated in DESIGN.md 8: every bus balances (program,
// memory, lookups) -- the chips are not yet tied to each other by an execution bus.
#pragma once
#include <algorithm>
#include <chrono>
#include <mutex>
#include <deque>
#include <condition_variable>
#include <cstring>
#include <map>
#include <numeric>
#include <stdexcept>
#include <array>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "zkhip.h"
#include "zkhip_chips.hpp"
#include "zkhip_vm.hpp"

namespace zkhip_vm {

constexpr uint32_t TUPLE_SIZE_X = 256, TUPLE_SIZE_Y = 2048;  // covers (limb < 256, carry < 1024): a 2^19-row table
constexpr unsigned N_VM_AIRS = 19;

struct GuestProof {
    ExecutionResult exec;
    uint64_t execution_time_mills = 0, tracegen_time_mills = 0, proving_time_mills = 0;
    std::vector<uint8_t> proof;               // zkhip proof bytes
    std::vector<unsigned> log_heights;        // per AIR (the proof's baseline)
    std::vector<std::vector<uint32_t>> programs;
    std::vector<std::array<uint32_t, 8>> prep_commits;  // for the AIRs with a preprocessed trace (else zeros)
    std::vector<uint8_t> has_prep;
};

inline unsigned log2_ceil_min1(size_t n) {
    unsigned l = 1;
    while (((size_t)1 << l) < n) l++;
    return l;
}

// The program chip's cached partition: 9 fields per instruction (row index, opcode, rd, funct3, rs1, rs2, funct7, low and high
// half of the instruction word), column-major over 2^log_program rows; rows beyond the program are zero (never executed).
inline std::vector<uint32_t> program_table(const std::vector<uint32_t>& words, unsigned log_program) {
    const size_t n = (size_t)1 << log_program;
    std::vector<uint32_t> t(9 * n, 0);
    for (size_t k = 0; k < words.size(); k++) {
        const uint32_t w = words[k];
        const uint32_t f[9] = {(uint32_t)k, w & 0x7f, (w >> 7) & 31, (w >> 12) & 7, (w >> 15) & 31, (w >> 20) & 31, w >> 25, w & 0xffff, w >> 16};
        for (size_t q = 0; q < 9; q++) t[q * n + k] = f[q];
    }
    return t;
}

class VmProver {
public:
    explicit VmProver(int device = 0) {
        if (zkhip_ctx_create(device, &ctx_) != ZKHIP_OK) throw Error("zkhip_ctx_create failed (needs a gfx950 device)");
    }
    ~VmProver() {
        for (auto& kv : keys_)
            if (kv.second.pk) zkhip_pk_destroy(ctx_, kv.second.pk);
        for (Chunk& c : dev_) zkhip_free(ctx_, c.base);
        for (Chunk& c : pin_) zkhip_host_free(ctx_, c.base);
        if (ctx_) zkhip_ctx_destroy(ctx_);
    }
    VmProver(const VmProver&) = delete;
    VmProver& operator=(const VmProver&) = delete;

    GuestProof prove_guest(const zkhip_params& params, const Exe& exe, const StdIn& in, uint64_t max_cost = 0) {
        // ---- 1. execute
        const auto t0 = std::chrono::steady_clock::now();
        ExecRecords rec;
        const ExecutionResult er = execute_guest(exe, in, max_cost, &rec);
        const uint64_t exec_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        GuestProof gp = prove_records(params, exe, rec);
        gp.exec = er, gp.execution_time_mills = exec_ms;
        return gp;
    }

    // One segment's (or a whole run's) records -> device traces -> proof.  gp.exec.total_cycle = the segment's instruction count.
    GuestProof prove_records(const zkhip_params& params, const Exe& exe, const ExecRecords& rec, bool self_verify = true) {
        using clk = std::chrono::steady_clock;
        auto ms = [](clk::time_point a, clk::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(b - a).count(); };
        GuestProof gp;
        gp.exec.total_cycle = rec.pc_index.size();
        if (!dev_.empty() && dev_[0].used) {   // a previous segment ended in an exception: let its copies drain, then reuse the arenas
            zkhip_sync(ctx_);
            release_segment_buffers();
        }
        const auto t1 = clk::now();
        // ---- 2. heights, device records, device traces
        const unsigned lp = log2_ceil_min1(exe.program.size()), lf = log2_ceil_min1(rec.pc_index.size());
        const unsigned la = log2_ceil_min1(rec.alu_op.size()), ll = log2_ceil_min1(rec.lt_op.size()), lm = log2_ceil_min1(rec.mul_b.size());
        const unsigned lacc = log2_ceil_min1(rec.acc_ts.size()), lbnd = log2_ceil_min1(rec.bnd_ptr.size());
        const unsigned lsh = log2_ceil_min1(rec.shift_op.size()), lmh = log2_ceil_min1(rec.mulh_op.size()), ldv = log2_ceil_min1(rec.div_op.size());
        const unsigned lbe = log2_ceil_min1(rec.beq_op.size()), lbl = log2_ceil_min1(rec.blt_op.size());
        const unsigned lls = log2_ceil_min1(rec.ls_case.size());
        const unsigned ljl = log2_ceil_min1(rec.jal_op.size()), lau = log2_ceil_min1(rec.auipc_pc.size()), ljr = log2_ceil_min1(rec.jalr_pc.size());
        unsigned ltup = 0;
        while ((1u << ltup) < TUPLE_SIZE_X * TUPLE_SIZE_Y) ltup++;
        gp.log_heights = {lp, lf, la, ll, lm, lmh, ldv, lsh, lbe, lbl, ljl, lau, ljr, lls, 16, ltup, lacc, lbnd, 16};
        const std::vector<uint32_t> prog = program_table(exe.program, lp);
        const size_t NP = (size_t)1 << lp;
        uint32_t* d_program_trace = dmalloc(10 * NP);  // [9 cached columns | frequency]
        h2d(d_program_trace, prog.data(), 9 * NP);
        check(zkhip_to_monty(ctx_, d_program_trace, 9 * NP));
        const uint32_t* d_idx = up(rec.pc_index);
        check(zkhip_program_freq_tracegen(ctx_, d_idx, rec.pc_index.size(), lp, d_program_trace + 9 * NP));
        uint32_t* d_frames = dmalloc((size_t)10 << lf);
        check(zkhip_exec_frame_tracegen(ctx_, d_idx, rec.pc_index.size(), d_program_trace, NP, lf, d_frames));
        uint32_t* d_bw = dzeros(2u << 16);
        uint32_t* d_alu = dmalloc((size_t)ZKHIP_RV32_ALU_WIDTH << la);
        check(zkhip_rv32_alu_tracegen(ctx_, up(rec.alu_op), up(rec.alu_b), up(rec.alu_c), rec.alu_op.size(), la, d_alu, d_bw));
        uint32_t* d_lt = dmalloc((size_t)ZKHIP_RV32_LT_WIDTH << ll);
        check(zkhip_rv32_lt_tracegen(ctx_, up(rec.lt_op), up(rec.lt_b), up(rec.lt_c), rec.lt_op.size(), ll, d_lt, d_bw));
        uint32_t* d_shift = dmalloc((size_t)ZKHIP_RV32_SHIFT_WIDTH << lsh);
        check(zkhip_rv32_shift_tracegen(ctx_, up(rec.shift_op), up(rec.shift_b), up(rec.shift_c), rec.shift_op.size(), lsh, d_shift, d_bw));
        uint32_t* d_beq = dmalloc((size_t)ZKHIP_RV32_BRANCH_EQ_WIDTH << lbe);
        check(zkhip_rv32_branch_eq_tracegen(ctx_, up(rec.beq_op), up(rec.beq_a), up(rec.beq_b), up(rec.beq_imm), rec.beq_op.size(), lbe, d_beq));
        uint32_t* d_blt = dmalloc((size_t)ZKHIP_RV32_BRANCH_LT_WIDTH << lbl);
        check(zkhip_rv32_branch_lt_tracegen(ctx_, up(rec.blt_op), up(rec.blt_a), up(rec.blt_b), up(rec.blt_imm), rec.blt_op.size(), lbl, d_blt, d_bw));
        uint32_t* d_jal = dmalloc((size_t)ZKHIP_RV32_JAL_LUI_WIDTH << ljl);
        check(zkhip_rv32_jal_lui_tracegen(ctx_, up(rec.jal_op), up(rec.jal_pc), up(rec.jal_imm), rec.jal_op.size(), ljl, d_jal, d_bw));
        uint32_t* d_auipc = dmalloc((size_t)ZKHIP_RV32_AUIPC_WIDTH << lau);
        check(zkhip_rv32_auipc_tracegen(ctx_, up(rec.auipc_pc), up(rec.auipc_imm), rec.auipc_pc.size(), lau, d_auipc, d_bw));
        uint32_t* d_jalr = dmalloc((size_t)ZKHIP_RV32_JALR_WIDTH << ljr);
        check(zkhip_rv32_jalr_tracegen(ctx_, up(rec.jalr_pc), up(rec.jalr_rs1), up(rec.jalr_imm), rec.jalr_pc.size(), ljr, d_jalr, d_bw));
        uint32_t* d_ls = dmalloc((size_t)ZKHIP_RV32_LOADSTORE_WIDTH << lls);
        check(zkhip_rv32_loadstore_tracegen(ctx_, up(rec.ls_case), up(rec.ls_read), up(rec.ls_prev), rec.ls_case.size(), lls, d_ls, d_bw));
        uint32_t* d_tup = dzeros((size_t)1 << ltup);
        uint32_t* d_mul = dmalloc((size_t)ZKHIP_RV32_MUL_WIDTH << lm);
        check(zkhip_rv32_mul_tracegen(ctx_, up(rec.mul_b), up(rec.mul_c), rec.mul_b.size(), lm, d_mul, d_tup, TUPLE_SIZE_X, TUPLE_SIZE_Y));
        uint32_t* d_mulh = dmalloc((size_t)ZKHIP_RV32_MULH_WIDTH << lmh);
        check(zkhip_rv32_mulh_tracegen(ctx_, up(rec.mulh_op), up(rec.mulh_b), up(rec.mulh_c), rec.mulh_op.size(), lmh, d_mulh, d_tup, TUPLE_SIZE_X, TUPLE_SIZE_Y, d_bw));
        uint32_t* d_div = dmalloc((size_t)ZKHIP_RV32_DIVREM_WIDTH << ldv);
        check(zkhip_rv32_divrem_tracegen(ctx_, up(rec.div_op), up(rec.div_b), up(rec.div_c), rec.div_op.size(), ldv, d_div, d_tup, TUPLE_SIZE_X, TUPLE_SIZE_Y, d_bw));
        uint32_t* d_acc = dmalloc((size_t)ZKHIP_MEMORY_ACCESS_WIDTH << lacc);
        const size_t n_acc = rec.acc_ts.size(), n_bnd = rec.bnd_ptr.size();
        check(zkhip_memory_access_tracegen(ctx_, up(rec.acc_as), up(rec.acc_ptr), up(rec.acc_prev_data), up(rec.acc_prev_ts), up(rec.acc_data),
                                           up(rec.acc_ts), up(rec.acc_is_read), n_acc, lacc, d_acc));
        uint32_t* d_bnd = dmalloc((size_t)ZKHIP_MEMORY_BOUNDARY_WIDTH << lbnd);
        uint32_t* d_init = const_cast<uint32_t*>(up(rec.bnd_initial));
        uint32_t* d_fin = const_cast<uint32_t*>(up(rec.bnd_final));
        if (n_bnd) check(zkhip_to_monty(ctx_, d_init, n_bnd)), check(zkhip_to_monty(ctx_, d_fin, n_bnd));
        check(zkhip_memory_boundary_tracegen(ctx_, up(rec.bnd_as), up(rec.bnd_ptr), d_init, d_fin, up(rec.bnd_final_ts), n_bnd, 2, 27, lbnd, d_bnd));
        // the 16-bit range checker's multiplicities, counted where the requesting columns lie (valid rows only)
        uint32_t* d_rng = dzeros(1u << 16);
        const size_t NA = (size_t)1 << lacc, NB = (size_t)1 << lbnd;
        bool first = true;
        auto count = [&](const uint32_t* col, size_t n) {
            check(zkhip_range_counts_tracegen(ctx_, col, n, 16, d_rng, first ? 0 : 1));
            first = false;
        };
        count(d_acc + 8 * NA, n_acc), count(d_acc + 9 * NA, n_acc), count(d_acc + 4 * NA, n_acc);
        count(d_bnd + 6 * NB, n_bnd), count(d_bnd + 7 * NB, n_bnd);
        // 8 gap_hi of both chips: the gaps stay below 2^29 (a request outside the table fails the generator)
        if (n_acc) check(zkhip_range_counts_scaled_tracegen(ctx_, d_acc + 9 * NA, n_acc, 8, 16, d_rng, 1));
        if (n_bnd) check(zkhip_range_counts_scaled_tracegen(ctx_, d_bnd + 7 * NB, n_bnd, 8, 16, d_rng, 1));
        const auto t2 = clk::now();
        gp.tracegen_time_mills = ms(t1, t2);
        // ---- 3. the AIR set and its preprocessed tables
        namespace ch = zkhip::chips;
        using zkhip::air::AirBuilder;
        const size_t widths[N_VM_AIRS] = {10, 10, 18, 18, 13, 21, 41, 32, 17, 23, 9, 14, 20, 33, 2, 1, 10, 8, 1};
        const size_t prep_w[N_VM_AIRS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 2, 0, 0, 1};
        gp.programs.resize(N_VM_AIRS);
        for (unsigned a = 0; a < N_VM_AIRS; a++) {
            AirBuilder b(widths[a], 0, prep_w[a]);
            switch (a) {
                case 0: b.set_cached_width(9); ch::program_air(b); break;
                case 1: ch::exec_frame_air(b); break;
                case 2: ch::rv32_alu_core_air(b); break;
                case 3: ch::rv32_lt_core_air(b); break;
                case 4: ch::rv32_mul_core_air(b); break;
                case 5: ch::rv32_mulh_core_air(b); break;
                case 6: ch::rv32_divrem_core_air(b); break;
                case 7: ch::rv32_shift_core_air(b); break;
                case 8: ch::rv32_branch_eq_core_air(b); break;
                case 9: ch::rv32_branch_lt_core_air(b); break;
                case 10: ch::rv32_jal_lui_core_air(b); break;
                case 11: ch::rv32_auipc_core_air(b); break;
                case 12: ch::rv32_jalr_core_air(b); break;
                case 13: ch::rv32_loadstore_core_air(b); break;
                case 14: ch::bitwise_lookup_air(b); break;
                case 15: ch::range_tuple_table_air(b); break;
                case 16: ch::memory_access_air(b); break;
                case 17: ch::memory_boundary_air(b); break;
                default: ch::range_table_air(b); break;
            }
            gp.programs[a] = b.program();
        }
        std::vector<uint32_t> prep_bw(3u << 16), prep_tup((size_t)2 << ltup), prep_rng(1u << 16);
        for (uint32_t i = 0; i < (1u << 16); i++) prep_bw[i] = i >> 8, prep_bw[(1u << 16) + i] = i & 255u, prep_bw[(2u << 16) + i] = (i >> 8) ^ (i & 255u);
        for (uint32_t i = 0; i < (1u << ltup); i++) prep_tup[i] = i / TUPLE_SIZE_Y, prep_tup[((size_t)1 << ltup) + i] = i % TUPLE_SIZE_Y;
        std::iota(prep_rng.begin(), prep_rng.end(), 0u);
        const uint32_t* preps[N_VM_AIRS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, prep_bw.data(), prep_tup.data(), nullptr, nullptr, prep_rng.data()};
        std::vector<zkhip_air> airs(N_VM_AIRS);
        for (unsigned a = 0; a < N_VM_AIRS; a++)
            airs[a] = zkhip_air{gp.programs[a].data(), gp.programs[a].size(), gp.log_heights[a], widths[a], 0, preps[a], nullptr};
        // proving keys are kept per height set: consecutive segments of one run have the same heights, and a key holds the
        // committed lookup tables and the compiled constraint kernels (keys live until the prover is destroyed, like the
        // reference's keys until Prover::reset)
        if (have_params_ && memcmp(&params_, &params, sizeof params) != 0) throw Error("one VmProver proves with one set of FRI parameters");
        params_ = params, have_params_ = true;
        Key& key = keys_[gp.log_heights];
        if (!key.pk) {
            check(zkhip_keygen(ctx_, &params, airs.data(), airs.size(), &key.pk));
            key.prep_commits.assign(N_VM_AIRS, std::array<uint32_t, 8>{});
            for (unsigned a = 0; a < N_VM_AIRS; a++)
                if (preps[a]) check(zkhip_pk_prep_commitment(ctx_, key.pk, a, key.prep_commits[a].data()));
        }
        zkhip_pk* pk = key.pk;
        gp.prep_commits = key.prep_commits;
        gp.has_prep.assign(N_VM_AIRS, 0);
        for (unsigned a = 0; a < N_VM_AIRS; a++) gp.has_prep[a] = preps[a] ? 1 : 0;
        // ---- 4. prove from the device-resident traces, then the mandatory self-check (mod.rs:407-411)
        const uint32_t* d_traces[N_VM_AIRS] = {d_program_trace, d_frames, d_alu, d_lt, d_mul, d_mulh, d_div, d_shift, d_beq, d_blt, d_jal, d_auipc, d_jalr, d_ls, d_bw, d_tup, d_acc, d_bnd, d_rng};
        const uint32_t* pvs[N_VM_AIRS] = {};
        gp.proof.resize(zkhip_proof_size(pk));
        size_t len = 0;
        const int rc = zkhip_prove(ctx_, pk, d_traces, pvs, gp.proof.data(), gp.proof.size(), &len);
        if (rc != ZKHIP_OK) throw Error(std::string("zkhip_prove: ") + zkhip_last_error(ctx_));
        gp.proof.resize(len);
        gp.proving_time_mills = ms(t2, clk::now());
        if (self_verify && verify(params, gp) != ZKHIP_OK) throw Error("the proof does not verify");
        release_segment_buffers();  // the records and traces of this segment (the proof is out: nothing in flight reads them)
        return gp;
    }

    // host only: what a verifier needs is the programs, heights and preprocessed commitments of the proof
    static int verify(const zkhip_params& params, const GuestProof& gp) {
        const size_t widths[N_VM_AIRS] = {10, 10, 18, 18, 13, 21, 41, 32, 17, 23, 9, 14, 20, 33, 2, 1, 10, 8, 1};
        std::vector<zkhip_air> airs(N_VM_AIRS);
        for (unsigned a = 0; a < N_VM_AIRS; a++)
            airs[a] = zkhip_air{gp.programs[a].data(), gp.programs[a].size(), gp.log_heights[a], widths[a], 0, nullptr,
                                gp.has_prep[a] ? gp.prep_commits[a].data() : nullptr};
        const uint32_t* pvs[N_VM_AIRS] = {};
        return zkhip_verify(&params, airs.data(), airs.size(), pvs, gp.proof.data(), gp.proof.size());
    }

private:
    zkhip_ctx* ctx_ = nullptr;
    // Device memory and pinned staging of a segment come from chunked arenas that live as long as the prover: after the first
    // segment a lane neither allocates nor frees (hipFree synchronises the whole device, which serialised the lanes), and the
    // ~50 record arrays of a segment go up as asynchronous copies from ONE pinned buffer instead of 50 synchronous pageable ones.
    struct Chunk {
        char* base = nullptr;
        size_t cap = 0, used = 0;
    };
    std::vector<Chunk> dev_, pin_;
    static constexpr size_t DEV_CHUNK = (size_t)256 << 20, PIN_CHUNK = (size_t)64 << 20;
    struct Key {
        zkhip_pk* pk = nullptr;
        std::vector<std::array<uint32_t, 8>> prep_commits;
    };
    std::map<std::vector<unsigned>, Key> keys_;
    zkhip_params params_{};
    bool have_params_ = false;

    void check(int rc) {
        if (rc != ZKHIP_OK) throw Error(std::string("zkhip: ") + zkhip_last_error(ctx_));
    }
    template <typename Alloc>
    char* carve(std::vector<Chunk>& chunks, size_t bytes, size_t chunk_bytes, Alloc&& alloc) {
        bytes = (std::max<size_t>(bytes, 4) + 255) & ~(size_t)255;
        for (Chunk& c : chunks)
            if (c.cap - c.used >= bytes) {
                char* p = c.base + c.used;
                c.used += bytes;
                return p;
            }
        Chunk c;
        c.cap = std::max(bytes, chunk_bytes);
        void* p = nullptr;
        check(alloc(c.cap, &p));
        c.base = (char*)p, c.used = bytes;
        chunks.push_back(c);
        return c.base;
    }
    void release_segment_buffers() {
        for (Chunk& c : dev_) c.used = 0;
        for (Chunk& c : pin_) c.used = 0;
    }
    uint32_t* dmalloc(size_t words) {
        return (uint32_t*)carve(dev_, words * 4, DEV_CHUNK, [&](size_t n, void** p) { return zkhip_malloc(ctx_, n, p); });
    }
    void h2d(uint32_t* d, const uint32_t* h, size_t words) {   // through the pinned staging buffer, stream-ordered
        if (!words) return;
        char* stage = carve(pin_, words * 4, PIN_CHUNK, [&](size_t n, void** p) { return zkhip_host_alloc(ctx_, n, p); });
        memcpy(stage, h, words * 4);
        check(zkhip_h2d_async(ctx_, d, stage, words * 4));
    }
    uint32_t* dzeros(size_t words) {
        uint32_t* d = dmalloc(words);
        check(zkhip_zero(ctx_, d, words * 4));
        return d;
    }
    const uint32_t* up(const std::vector<uint32_t>& v) {  // plain integers, as the generators take them
        uint32_t* d = dmalloc(v.size());
        h2d(d, v.data(), v.size());
        return d;
    }
};

// Continuation: the run is cut into segments of at most `segment_instr` instructions (zkhip_vm::execute_segments) and every
// segment is proven on its own -- independent STARKs spread over `inflight` provers per listed GPU, each
// with a context of its own, like BatchProver's lanes (include/zkhip_prover.hpp).  What chains the segment proofs is the memory
// boundary: a cell's final value in one segment is its initial value in the next segment that touches it (the aggregation layer
// of the reference checks that link; here tests do).
struct SegmentedProof {
    ExecutionResult exec;
    uint64_t execution_time_mills = 0;   // time the executor spent executing (it overlaps the proving)
    uint64_t proving_wall_mills = 0;     // wall time from the start of the execution to the last segment proof verified
    std::vector<GuestProof> segments;
};
inline SegmentedProof prove_segments(const zkhip_params& params, const Exe& exe, const StdIn& in, uint64_t segment_instr,
                                     unsigned inflight = 1, std::vector<int> devices = {0}) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::duration d) { return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(d).count(); };
    if (segment_instr == 0) throw Error("segment length must be positive");
    if (devices.empty() || inflight == 0) throw Error("no prover lanes");
    const auto t0 = clk::now();
    SegmentedProof sp;
    // The executor (this thread) streams segments into a bounded queue while the lanes prove: execution overlaps proving, and
    // at most 2 * lanes segments of records (168 B per instruction) are alive at once instead of the whole run's.
    const size_t n_lanes = devices.size() * inflight;
    struct Item {
        size_t index;
        ExecRecords rec;
    };
    std::deque<Item> queue;
    std::vector<ExecRecords> pool;   // proven segments' records, emptied: the executor refills them instead of faulting in fresh pages
    std::mutex mu;
    std::condition_variable cv_push, cv_pop;
    bool closed = false;
    std::vector<std::string> errors(n_lanes);
    std::atomic<bool> lane_failed{false};
    std::vector<std::thread> lanes;
    for (size_t l = 0; l < n_lanes; l++)
        lanes.emplace_back([&, l] {
            try {
                VmProver vp(devices[l % devices.size()]);
                for (;;) {
                    Item it;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_pop.wait(lk, [&] { return !queue.empty() || closed; });
                        if (queue.empty()) return;
                        it = std::move(queue.front());
                        queue.pop_front();
                    }
                    cv_push.notify_one();
                    // the mandatory self-check (mod.rs:407-411) runs on host threads after the lanes are done, not on a lane
                    GuestProof gp = vp.prove_records(params, exe, it.rec, false);
                    it.rec.clear();
                    std::lock_guard<std::mutex> lk(mu);
                    if (sp.segments.size() <= it.index) sp.segments.resize(it.index + 1);
                    sp.segments[it.index] = std::move(gp);
                    pool.push_back(std::move(it.rec));
                }
            } catch (const std::exception& e) {
                {
                    std::lock_guard<std::mutex> lk(mu);   // under the queue's mutex: the executor cannot miss the notification
                    errors[l] = e.what();
                    lane_failed = true;
                }
                cv_push.notify_all();
            }
        });
    auto close_queue = [&] {
        {
            std::lock_guard<std::mutex> lk(mu);
            closed = true;
        }
        cv_pop.notify_all();
        for (auto& t : lanes)
            if (t.joinable()) t.join();
    };
    clk::duration exec_time{};
    size_t n_segments = 0;
    try {
        Machine m(exe, in);
        for (bool done = false; !done && !lane_failed;) {
            Item it;
            it.index = n_segments++;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!pool.empty()) it.rec = std::move(pool.back()), pool.pop_back();
            }
            const auto a = clk::now();
            done = m.run_segment(segment_instr, &it.rec);
            exec_time += clk::now() - a;
            std::unique_lock<std::mutex> lk(mu);
            cv_push.wait(lk, [&] { return queue.size() < 2 * n_lanes || lane_failed; });
            queue.push_back(std::move(it));
            lk.unlock();
            cv_pop.notify_one();
        }
        if (lane_failed) {   // the run was cut short by a failed lane: that error, not a verdict on an unfinished guest
            close_queue();
            for (const auto& e : errors)
                if (!e.empty()) throw Error("segment prover: " + e);
            throw Error("segment prover: a lane failed");
        }
        if (m.exit_code()) throw Error("guest exited with code " + std::to_string(m.exit_code()));
        bool all_zero = true;
        for (uint8_t b : m.public_values()) all_zero = all_zero && b == 0;
        if (all_zero) throw Error("public_values are all 0s for unexpected reason");
        sp.exec = ExecutionResult{m.instret(), m.public_values()};
    } catch (...) {
        close_queue();
        throw;
    }
    close_queue();
    for (const auto& e : errors)
        if (!e.empty()) throw Error("segment prover: " + e);
    if (sp.segments.size() != n_segments) throw Error("segment prover: proofs missing");
    sp.execution_time_mills = ms(exec_time);
    {
        const size_t n_ver = std::min<size_t>(std::max(1u, zkhip_host_cpus()), sp.segments.size());
        std::vector<std::thread> vt;
        std::atomic<size_t> nv{0};
        std::atomic<int> bad{0};
        for (size_t t = 0; t < n_ver; t++)
            vt.emplace_back([&] {
                for (size_t k = nv++; k < sp.segments.size(); k = nv++)
                    if (VmProver::verify(params, sp.segments[k]) != ZKHIP_OK) bad++;
            });
        for (auto& t : vt) t.join();
        if (bad) throw Error("a segment proof does not verify");
    }
    sp.proving_wall_mills = ms(clk::now() - t0);   // execution included: it runs under the proving
    return sp;
}

}  // namespace zkhip_vm
