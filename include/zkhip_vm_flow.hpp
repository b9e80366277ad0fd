// zkhip_vm_flow.hpp -- ONE FLOW PER TASK: guest ELF + witness bytes in, ONE proof out.  The reference's
//   Prover::gen_proof_universal -> gen_proof_stark  (crates/prover/src/prover/mod.rs:287-309, 342-413):
//   task.build_guest_input() (crates/prover/src/task/mod.rs:27-38) -> execute_guest (crates/prover/src/utils/vm.rs:13-48) -> sdk.prove
//   (app segments, then the leaf / internal aggregation tree, mod.rs:47-60) -> encode -> mandatory self-verification
// over this backend:
//   1. SegmentExecutor (include/zkhip_vm_exec.hpp) runs the guest and cuts it into segments of fixed heights;
//   2. SegmentProver uploads a segment's records, generates the 22 chips' traces on the device (include/zkhip.h zkhip_*_tracegen) and
//      proves them as ONE statement (include/zkhip_vm_circuit.hpp): (pc_start, memory root) -> (pc_end, memory root');
//   3. AggregationProver (include/zkhip_aggregation.hpp) folds the segment proofs: leaf nodes verify <= 4 segments, internal
//      nodes <= 3 nodes, chaining (pc, root) in-circuit, until one root proof remains;
//   4. verify_guest_proof: the root proof under the root verifying key + the statement checks a verifier makes outside the circuit
//      (initial pc and memory root = the guest image's; final pc = 0 = exited with code 0; the public values' Merkle openings in the
//      final root -- the `user_pvs_proof` of crates/types/src/proof.rs:52-67).
#pragma once
#include <cstdio>
#include <fstream>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <numeric>

#include "zkhip_aggregation.hpp"
#include "zkhip_vm_exec.hpp"

namespace zkhip_vm {
using scroll_zkvm_hip::AirDesc;
using scroll_zkvm_hip::ChildProof;
using scroll_zkvm_hip::VerifyingKey;

// the segment's verifying key material that does not need a device: programs, shapes, heights (commitments come from keygen)
struct SegmentAirs {
    std::vector<zkhip_air> airs;                 // program / width / n_pvs / log_height; prep_trace set for the AIRs with tables
    std::vector<size_t> prep_width;
    std::vector<uint32_t> program_prep, bitwise_prep, tuple_prep, range_prep, sha_prep;
};
inline unsigned vm_log2_ceil(size_t n) {
    unsigned l = 1;
    while (((size_t)1 << l) < n) l++;
    return l;
}
inline SegmentAirs segment_airs(const Exe& exe, const SegmentCaps& caps) {
    SegmentAirs s;
    s.airs.resize(caps.n_airs), s.prep_width.resize(caps.n_airs);
    for (unsigned p = 0; p < caps.n_airs; p++) {   // position p of the proof holds chip caps.ids[p] (base chips: p = id)
        const unsigned id = caps.ids[p];
        if (id >= vmc::A_FP2(0)) {   // the fp2 extension: chip / adapter of field i
            const unsigned i = (id - vmc::A_FP2(0)) / 2;
            const zkhip::modular::Modulus m = zkhip::modular::modulus_bytes(caps.fp2_moduli.at(i));
            if (zkhip_vm_fp2_air_x(m.data(), m.limbs, i, (id - vmc::A_FP2(0)) & 1, &s.airs[p]) != ZKHIP_OK) throw Error("zkhip_vm_fp2_air failed (the modulus must be odd with a non-zero top byte)");
            s.prep_width[p] = 0;
        } else if (id >= vmc::A_EC(0)) {   // the ecc extension: chip / adapter of curve i
            const unsigned i = (id - vmc::A_EC(0)) / 2;
            const zkhip::modular::Modulus m = zkhip::modular::modulus_bytes(caps.curves.at(i).p), ca = zkhip::modular::modulus_bytes(caps.curves.at(i).a, m.limbs);
            if (zkhip_vm_ec_air_x(m.data(), ca.data(), m.limbs, i, (id - vmc::A_EC(0)) & 1, &s.airs[p]) != ZKHIP_OK) throw Error("zkhip_vm_ec_air failed (the modulus must be odd with a non-zero top byte)");
            s.prep_width[p] = 0;
        } else if (id >= vmc::N_STATIC_AIRS) {   // the modular extension: chip / adapter of modulus i
            const unsigned i = (id - vmc::N_STATIC_AIRS) / 2;
            const zkhip::modular::Modulus m = zkhip::modular::modulus_bytes(caps.moduli.at(i));
            if (zkhip_vm_modmul_air_x(m.data(), m.limbs, i, (id - vmc::N_STATIC_AIRS) & 1, &s.airs[p]) != ZKHIP_OK) throw Error("zkhip_vm_modmul_air failed");
            s.prep_width[p] = 0;
        } else if (zkhip_vm_air(id, &s.airs[p], &s.prep_width[p]) != ZKHIP_OK) {
            throw Error("zkhip_vm_air failed");
        }
        s.airs[p].log_height = caps.log_height[id];
    }
    const unsigned lp = caps.log_height[vmc::A_PROGRAM];
    s.program_prep = vmc::program_table(exe.program, exe.pc_base, lp);
    s.bitwise_prep.resize(3u << 16), s.tuple_prep.resize((size_t)2 << 19), s.range_prep.resize(1u << 16);
    for (uint32_t i = 0; i < (1u << 16); i++) s.bitwise_prep[i] = i >> 8, s.bitwise_prep[(1u << 16) + i] = i & 255u, s.bitwise_prep[(2u << 16) + i] = (i >> 8) ^ (i & 255u);
    for (uint32_t i = 0; i < (1u << 19); i++) s.tuple_prep[i] = i / TUPLE_Y, s.tuple_prep[((size_t)1 << 19) + i] = i % TUPLE_Y;
    std::iota(s.range_prep.begin(), s.range_prep.end(), 0u);
    s.airs[vmc::A_PROGRAM].prep_trace = s.program_prep.data(), s.airs[vmc::A_BITWISE].prep_trace = s.bitwise_prep.data();
    s.airs[vmc::A_RANGE_TUPLE].prep_trace = s.tuple_prep.data(), s.airs[vmc::A_RANGE].prep_trace = s.range_prep.data();
    if (caps.sha256()) {
        s.sha_prep = zkhip::sha256::prep_trace_vm(caps.log_height[vmc::A_SHA256]);
        s.airs[caps.pos(vmc::A_SHA256)].prep_trace = s.sha_prep.data();
    }
    return s;
}

struct SegmentProof {
    ChildProof proof;               // proof bytes + public values per AIR (merkle chip: roots; connector: pc_start, pc_end)
    uint64_t n_instr = 0, tracegen_mills = 0, proving_mills = 0;
    size_t shape = 0;               // which of the app's shapes (sets of chips) the segment was proven under
};

// where the chained state of a segment proof lives: (pc, memory root) -- the StatementSpec of the aggregation layer
inline scroll_zkvm_hip::StatementSpec segment_statement() {
    scroll_zkvm_hip::StatementSpec sp;
    sp.start.push_back({vmc::A_CONNECTOR, 0}), sp.end.push_back({vmc::A_CONNECTOR, 1});
    for (uint32_t k = 0; k < 8; k++) sp.start.push_back({vmc::A_MERKLE, k}), sp.end.push_back({vmc::A_MERKLE, 8 + k});
    return sp;
}

class SegmentProver {
public:
    SegmentProver(const zkhip_params& params, const Exe& exe, const SegmentCaps& caps, int device = 0) : params_(params), caps_(caps), sa_(segment_airs(exe, caps)) {
        if (zkhip_ctx_create(device, &ctx_) != ZKHIP_OK) throw Error("zkhip_ctx_create failed (needs a gfx950 device)");
        check(zkhip_keygen(ctx_, &params_, sa_.airs.data(), sa_.airs.size(), &pk_));
        vk_.params = params_;
        for (unsigned a = 0; a < caps.n_airs; a++) {
            AirDesc d;
            d.width = sa_.airs[a].width, d.n_pvs = sa_.airs[a].n_pvs;
            d.program.assign(sa_.airs[a].program, sa_.airs[a].program + sa_.airs[a].program_len);
            if (sa_.prep_width[a]) {
                uint32_t c[8];
                check(zkhip_pk_prep_commitment(ctx_, pk_, a, c));
                d.has_prep = true, d.prep_log_height = sa_.airs[a].log_height, d.prep_commit.assign(c, c + 8);
            }
            vk_.airs.push_back(std::move(d));
            vk_.heights.push_back(sa_.airs[a].log_height);
        }
        // device-resident tables: the program (for the frame chip's gather), zeros for the bitwise "op" argument
        const size_t NP = caps_.rows(vmc::A_PROGRAM);
        check(zkhip_malloc(ctx_, vmc::PROGRAM_FIELDS * NP * 4, (void**)&d_program_));
        check(zkhip_h2d(ctx_, d_program_, sa_.program_prep.data(), vmc::PROGRAM_FIELDS * NP * 4));
        check(zkhip_to_monty(ctx_, d_program_, vmc::PROGRAM_FIELDS * NP));
        size_t max_rows = 0;
        for (unsigned a = 0; a < caps_.n_airs; a++) {
            const size_t words = sa_.airs[a].width << sa_.airs[a].log_height;
            void* d = nullptr;
            check(zkhip_malloc(ctx_, words * 4, &d));
            d_traces_.push_back((uint32_t*)d);
            max_rows = std::max(max_rows, (size_t)1 << sa_.airs[a].log_height);
        }
        check(zkhip_malloc(ctx_, max_rows * 4, (void**)&d_zeros_));
        check(zkhip_zero(ctx_, d_zeros_, max_rows * 4));
        check(zkhip_tracegen_defer_checks(ctx_, 1));   // ~60 generator calls per segment: one check at the end instead of one sync each
    }
    ~SegmentProver() {
        for (uint32_t* d : d_traces_) zkhip_free(ctx_, d);
        for (auto& c : dev_) zkhip_free(ctx_, c.base);
        for (auto& c : pin_) zkhip_host_free(ctx_, c.base);
        if (d_program_) zkhip_free(ctx_, d_program_);
        if (d_zeros_) zkhip_free(ctx_, d_zeros_);
        if (pk_) zkhip_pk_destroy(ctx_, pk_);
        if (ctx_) zkhip_ctx_destroy(ctx_);
    }
    SegmentProver(const SegmentProver&) = delete;
    SegmentProver& operator=(const SegmentProver&) = delete;
    const VerifyingKey& vk() const { return vk_; }

    SegmentProof prove(const SegmentRecords& r, bool self_verify = true) {
        using clk = std::chrono::steady_clock;
        auto ms = [](clk::time_point a, clk::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(b - a).count(); };
        const auto t0 = clk::now();
        release();
        const ExecRecords& c = r.core;
        auto H = [&](unsigned a) { return caps_.log_height[a]; };
        auto T = [&](unsigned a) { return d_traces_[(size_t)caps_.pos(a)]; };
        auto N = [&](unsigned a) { return caps_.rows(a); };
        // ---- tables start empty
        check(zkhip_zero(ctx_, T(vmc::A_BITWISE), (2u << 16) * 4)), check(zkhip_zero(ctx_, T(vmc::A_RANGE_TUPLE), (1u << 19) * 4));
        uint32_t *d_bw = T(vmc::A_BITWISE), *d_tup = T(vmc::A_RANGE_TUPLE), *d_rng = T(vmc::A_RANGE);
        // (the three count tables stay canonical until every chip has counted: one conversion each at the end instead of two per generator)
        // (switched back when this scope is left, also by a throwing check(): a context left in canonical mode would hand non-Montgomery
        // count tables to any later standalone generator call -- ADVICE round 4)
        struct CanonicalTables {
            zkhip_ctx* ctx;
            explicit CanonicalTables(zkhip_ctx* c) : ctx(c) {}
            ~CanonicalTables() { if (ctx) (void)zkhip_tables_canonical(ctx, 0); }
            int release() { zkhip_ctx* c = ctx; ctx = nullptr; return zkhip_tables_canonical(c, 0); }
        };
        check(zkhip_tables_canonical(ctx_, 1));
        CanonicalTables canonical_guard(ctx_);
        // ---- program frequencies, frame
        const uint32_t* d_idx = up(c.pc_index);
        check(zkhip_program_freq_tracegen(ctx_, d_idx, c.pc_index.size(), H(vmc::A_PROGRAM), T(vmc::A_PROGRAM)));
        check(zkhip_vm_frame_tracegen(ctx_, d_idx, up(r.f_x), up(r.f_y), up(r.f_z), up(r.f_rdprev), up(r.f_pcinc), up(r.f_pts1), up(r.f_pts2), up(r.f_pts3), r.n_instr,
                                      d_program_, N(vmc::A_PROGRAM), H(vmc::A_FRAME), T(vmc::A_FRAME)));
        // ---- instruction cores (include/zkhip_chips.hpp: unchanged generators)
        check(zkhip_rv32_alu_tracegen(ctx_, up(c.alu_op), up(c.alu_b), up(c.alu_c), c.alu_op.size(), H(vmc::A_ALU), T(vmc::A_ALU), d_bw));
        check(zkhip_rv32_lt_tracegen(ctx_, up(c.lt_op), up(c.lt_b), up(c.lt_c), c.lt_op.size(), H(vmc::A_LT), T(vmc::A_LT), d_bw));
        check(zkhip_rv32_mul_tracegen(ctx_, up(c.mul_b), up(c.mul_c), c.mul_b.size(), H(vmc::A_MUL), T(vmc::A_MUL), d_tup, TUPLE_X, TUPLE_Y));
        check(zkhip_rv32_mulh_tracegen(ctx_, up(c.mulh_op), up(c.mulh_b), up(c.mulh_c), c.mulh_op.size(), H(vmc::A_MULH), T(vmc::A_MULH), d_tup, TUPLE_X, TUPLE_Y, d_bw));
        check(zkhip_rv32_divrem_tracegen(ctx_, up(c.div_op), up(c.div_b), up(c.div_c), c.div_op.size(), H(vmc::A_DIVREM), T(vmc::A_DIVREM), d_tup, TUPLE_X, TUPLE_Y, d_bw));
        check(zkhip_rv32_shift_tracegen(ctx_, up(c.shift_op), up(c.shift_b), up(c.shift_c), c.shift_op.size(), H(vmc::A_SHIFT), T(vmc::A_SHIFT), d_bw));
        check(zkhip_rv32_branch_eq_tracegen(ctx_, up(c.beq_op), up(c.beq_a), up(c.beq_b), up(c.beq_imm), c.beq_op.size(), H(vmc::A_BEQ), T(vmc::A_BEQ)));
        check(zkhip_rv32_branch_lt_tracegen(ctx_, up(c.blt_op), up(c.blt_a), up(c.blt_b), up(c.blt_imm), c.blt_op.size(), H(vmc::A_BLT), T(vmc::A_BLT), d_bw));
        check(zkhip_rv32_jal_lui_tracegen(ctx_, up(c.jal_op), up(c.jal_pc), up(c.jal_imm), c.jal_op.size(), H(vmc::A_JAL_LUI), T(vmc::A_JAL_LUI), d_bw));
        check(zkhip_rv32_auipc_tracegen(ctx_, up(c.auipc_pc), up(c.auipc_imm), c.auipc_pc.size(), H(vmc::A_AUIPC), T(vmc::A_AUIPC), d_bw));
        check(zkhip_rv32_jalr_tracegen(ctx_, up(c.jalr_pc), up(c.jalr_rs1), up(c.jalr_imm), c.jalr_pc.size(), H(vmc::A_JALR), T(vmc::A_JALR), d_bw));
        check(zkhip_vm_loadstore_tracegen(ctx_, up(c.ls_case), up(c.ls_read), up(c.ls_prev), up(r.ls_ts), up(r.ls_base), up(r.ls_imm), up(r.ls_pts), c.ls_case.size(),
                                          H(vmc::A_LS), T(vmc::A_LS), d_bw));
        // ---- the chips whose rows the executor wrote
        const uint32_t leaf_pad[vmc::LEAF_WIDTH] = {1};   // padding rows of the leaf chip carry address space 1
        check(zkhip_rows_tracegen(ctx_, up(r.ecall_rows), r.n_ecall(), vmc::ECALL_WIDTH, H(vmc::A_ECALL), T(vmc::A_ECALL), nullptr));
        check(zkhip_rows_tracegen(ctx_, up(r.leaf_rows), r.n_leaf(), vmc::LEAF_WIDTH, H(vmc::A_LEAF), T(vmc::A_LEAF), leaf_pad));
        check(zkhip_rows_tracegen(ctx_, up(r.merkle_rows), r.n_merkle(), vmc::MERKLE_WIDTH, H(vmc::A_MERKLE), T(vmc::A_MERKLE), nullptr));
        const std::vector<uint32_t> conn{r.ts_end & 0xffffu, r.ts_end >> 16};
        check(zkhip_rows_tracegen(ctx_, up(conn), 1, vmc::CONNECTOR_WIDTH, 0, T(vmc::A_CONNECTOR), nullptr));
        uint32_t* d_p2in = const_cast<uint32_t*>(up(r.p2_inputs));
        if (r.n_p2()) check(zkhip_to_monty(ctx_, d_p2in, r.p2_inputs.size()));
        check(zkhip_vm_poseidon2_tracegen(ctx_, d_p2in, r.n_p2(), H(vmc::A_POSEIDON2), T(vmc::A_POSEIDON2)));
        // ---- keccak calls: the Keccak-f chip on the device, its adapter's rows from the executor
        if (caps_.keccak()) {
            check(zkhip_vm_keccak_tracegen(ctx_, up(r.kk_states), up(r.kk_ts), r.n_keccak(), H(vmc::A_KECCAK), T(vmc::A_KECCAK)));
            check(zkhip_rows_tracegen(ctx_, up(r.kio_rows), r.kio_rows.size() / vmc::KECCAK_IO_WIDTH, vmc::KECCAK_IO_WIDTH, H(vmc::A_KECCAK_IO), T(vmc::A_KECCAK_IO), nullptr));
        }
        if (caps_.sha256()) {
            check(zkhip_vm_sha256_tracegen(ctx_, up(r.sha_blocks), up(r.sha_ts), r.n_sha256(), H(vmc::A_SHA256), T(vmc::A_SHA256)));
            check(zkhip_rows_tracegen(ctx_, up(r.shaio_rows), r.shaio_rows.size() / vmc::SHA_IO_WIDTH, vmc::SHA_IO_WIDTH, H(vmc::A_SHA256_IO), T(vmc::A_SHA256_IO), nullptr));
        }
        if (caps_.int256()) {   // bigint extension: the chip's lookups are counted by its generator
            check(zkhip_vm_int256_tracegen(ctx_, up(r.i256_records), up(r.i256_ts), r.i256_ts.size(), H(vmc::A_INT256), T(vmc::A_INT256), d_bw));
            check(zkhip_rows_tracegen(ctx_, up(r.i256io_rows), r.i256io_rows.size() / vmc::INT256_IO_WIDTH, vmc::INT256_IO_WIDTH, H(vmc::A_INT256_IO), T(vmc::A_INT256_IO), nullptr));
            check(zkhip_vm_mul256_tracegen(ctx_, up(r.mul256_records), up(r.mul256_ts), r.mul256_ts.size(), H(vmc::A_MUL256), T(vmc::A_MUL256), d_bw, d_tup, TUPLE_X, TUPLE_Y));
            check(zkhip_vm_cmp256_tracegen(ctx_, up(r.cmp256_records), up(r.cmp256_ts), r.cmp256_ts.size(), H(vmc::A_CMP256), T(vmc::A_CMP256), d_bw));
            check(zkhip_vm_shift256_tracegen(ctx_, up(r.sh256_records), up(r.sh256_ts), r.sh256_ts.size(), H(vmc::A_SHIFT256), T(vmc::A_SHIFT256), d_bw));
        }
        for (unsigned i = 0; i < caps_.moduli.size(); i++) {   // modular extension: the chip's lookups are counted by its generator
            const uint32_t nw = (uint32_t)zkhip::modular::words_of(caps_.moduli[i]);
            const size_t iow = vmc::modmul_io_width(nw);
            check(zkhip_vm_modmul_tracegen_x(ctx_, nw, caps_.moduli[i].w, up(r.mm_records[i]), up(r.mm_ts[i]), r.mm_ts[i].size(), H(vmc::A_MODMUL(i)), T(vmc::A_MODMUL(i)), d_bw,
                                             d_tup, TUPLE_X, TUPLE_Y));
            check(zkhip_rows_tracegen(ctx_, up(r.mmio_rows[i]), r.mmio_rows[i].size() / iow, iow, H(vmc::A_MODMUL_IO(i)), T(vmc::A_MODMUL_IO(i)), nullptr));
        }
        for (unsigned i = 0; i < caps_.curves.size(); i++) {   // ecc extension: the chip's lookups are counted by its generator
            const uint32_t nw = (uint32_t)zkhip::modular::words_of(caps_.curves[i].p);
            const size_t iow = vmc::ec_io_width(nw);
            check(zkhip_vm_ec_tracegen_x(ctx_, nw, caps_.curves[i].p.w, caps_.curves[i].a.w, up(r.ec_records[i]), up(r.ec_ts[i]), r.ec_ts[i].size(), H(vmc::A_EC(i)), T(vmc::A_EC(i)),
                                         d_bw, d_tup, TUPLE_X, TUPLE_Y));
            check(zkhip_rows_tracegen(ctx_, up(r.ecio_rows[i]), r.ecio_rows[i].size() / iow, iow, H(vmc::A_EC_IO(i)), T(vmc::A_EC_IO(i)), nullptr));
        }
        for (unsigned i = 0; i < caps_.fp2_moduli.size(); i++) {   // fp2 extension: the chip's lookups are counted by its generator
            const uint32_t nw = (uint32_t)zkhip::modular::words_of(caps_.fp2_moduli[i]);
            const size_t iow = vmc::ec_io_width(nw);
            check(zkhip_vm_fp2_tracegen_x(ctx_, nw, caps_.fp2_moduli[i].w, up(r.fp2_records[i]), up(r.fp2_ts[i]), r.fp2_ts[i].size(), H(vmc::A_FP2(i)), T(vmc::A_FP2(i)), d_bw, d_tup,
                                          TUPLE_X, TUPLE_Y));
            check(zkhip_rows_tracegen(ctx_, up(r.fp2io_rows[i]), r.fp2io_rows[i].size() / iow, iow, H(vmc::A_FP2_IO(i)), T(vmc::A_FP2_IO(i)), nullptr));
        }
        // ---- native field / extension / castf calls: ONE row per call, made on the device from the call's record
        if (caps_.native()) {
            check(zkhip_vm_native_arith_tracegen(ctx_, up(r.nat_records), r.n_nat(), H(vmc::A_NATIVE_ARITH), T(vmc::A_NATIVE_ARITH)));
            check(zkhip_vm_native_ext_tracegen(ctx_, up(r.next_records), r.n_next(), H(vmc::A_NATIVE_EXT), T(vmc::A_NATIVE_EXT)));
        }
        if (caps_.castf()) check(zkhip_vm_castf_tracegen(ctx_, up(r.castf_records), r.n_castf(), H(vmc::A_CASTF), T(vmc::A_CASTF)));
        // ---- lookup multiplicities of the new chips, counted from the requesting columns
        auto pairs = [&](unsigned a, size_t cx, size_t cy, size_t n) {
            if (n) check(zkhip_bitwise_lookup_tracegen(ctx_, T(a) + cx * N(a), T(a) + cy * N(a), d_zeros_, n, 8, d_bw, 1));
        };
        for (size_t q : {18, 20, 22, 24, 26, 28}) pairs(vmc::A_FRAME, q, q + 1, r.n_instr);
        for (size_t q : {0, 2, 4, 6}) pairs(vmc::A_LS, q, q + 1, c.ls_case.size());
        bool first = true;
        auto count = [&](const uint32_t* col, size_t n, uint32_t scale = 1) {
            if (scale == 1) check(zkhip_range_counts_tracegen(ctx_, col, n, 16, d_rng, first ? 0 : 1));
            else check(zkhip_range_counts_scaled_tracegen(ctx_, col, n, scale, 16, d_rng, first ? 0 : 1));
            first = false;
        };
        const size_t NF = N(vmc::A_FRAME), NL = N(vmc::A_LS), n_ls = c.ls_case.size();
        // timestamp gaps of the register and memory adapters: gap_lo in the range table, (0, gap_hi) in the range-tuple table
        for (size_t q : {35, 38, 41}) count(T(vmc::A_FRAME) + q * NF, r.n_instr);
        count(T(vmc::A_LS) + 46 * NL, n_ls);
        for (size_t q : {36, 39, 42})
            if (r.n_instr) check(zkhip_range_tuple_counts_tracegen(ctx_, d_zeros_, T(vmc::A_FRAME) + q * NF, r.n_instr, TUPLE_X, TUPLE_Y, d_tup, 1));
        if (n_ls) check(zkhip_range_tuple_counts_tracegen(ctx_, d_zeros_, T(vmc::A_LS) + 47 * NL, n_ls, TUPLE_X, TUPLE_Y, d_tup, 1));
        count(T(vmc::A_LS) + 40 * NL, n_ls), count(T(vmc::A_LS) + 41 * NL, n_ls, 4), count(T(vmc::A_LS) + 44 * NL, n_ls, 4);
        count(T(vmc::A_LS) + 44 * NL, n_ls), count(T(vmc::A_LS) + 41 * NL, n_ls);
        // native chips: a canonical result word asks for (lo, hi, hi_gap) in the range table; every word access for its gap_lo there and for
        // (0, gap_hi) in the range-tuple table; castf for its limbs as byte pairs
        auto access = [&](unsigned a, size_t first, size_t n) {   // columns prev_ts gap_lo gap_hi from `first`
            if (!n) return;
            count(T(a) + (first + 1) * N(a), n);
            check(zkhip_range_tuple_counts_tracegen(ctx_, d_zeros_, T(a) + (first + 2) * N(a), n, TUPLE_X, TUPLE_Y, d_tup, 1));
        };
        if (caps_.native()) {
            const size_t na = r.n_nat(), ne = r.n_next();
            if (na) for (size_t q : {6, 7, 8}) count(T(vmc::A_NATIVE_ARITH) + q * N(vmc::A_NATIVE_ARITH), na);
            for (size_t q : {18, 21, 24}) access(vmc::A_NATIVE_ARITH, q, na);
            if (ne)
                for (size_t i = 0; i < 4; i++)
                    for (size_t q : {18 + 5 * i, 19 + 5 * i, 20 + 5 * i}) count(T(vmc::A_NATIVE_EXT) + q * N(vmc::A_NATIVE_EXT), ne);
            for (size_t k = 0; k < 12; k++) access(vmc::A_NATIVE_EXT, 54 + 3 * k, ne);
        }
        if (caps_.castf()) {
            const size_t nc = r.n_castf();
            pairs(vmc::A_CASTF, 2, 3, nc), pairs(vmc::A_CASTF, 4, 5, nc);
            if (nc) check(zkhip_bitwise_lookup_tracegen(ctx_, T(vmc::A_CASTF) + 6 * N(vmc::A_CASTF), d_zeros_, d_zeros_, nc, 8, d_bw, 1));
            access(vmc::A_CASTF, 9, nc), access(vmc::A_CASTF, 12, nc);
        }
        {
            // the few range requests of the ecall, leaf and connector chips, listed on the host
            std::vector<uint32_t> misc{r.ts_end & 0xffffu, 8 * (r.ts_end >> 16), r.ts_end >> 16}, tup_y, bw_x, bw_y;
            for (size_t i = 0; i < r.n_ecall(); i++) {
                const uint32_t* row = &r.ecall_rows[i * vmc::ECALL_WIDTH];
                if (row[15]) {
                    for (uint32_t v : {row[17] * 8192, row[17], row[22], row[25]}) misc.push_back(v);
                    tup_y.push_back(row[23]), tup_y.push_back(row[26]);
                }
                if (row[30] || row[31] || row[32] || row[33] || row[34] || row[35]) misc.push_back(row[22]), tup_y.push_back(row[23]);   // modmul / int256 / ecc / fp2 / native: the a1 read
                if (row[37]) misc.push_back(row[43]), tup_y.push_back(row[44]), misc.push_back((row[40] - 32768u * row[41]) * 2u);   // a 256-bit branch: the a2 read, the offset's sign split
                if (row[16]) bw_x.push_back(row[10]), bw_y.push_back(row[11]), bw_x.push_back(row[12]), bw_y.push_back(row[13]);
                if (row[27] || row[29] || row[30] || row[31] || row[32] || row[33] || row[34] || row[35] || row[36])
                    for (uint32_t v : {row[28] * 1024, row[28], row[9] * 1024}) misc.push_back(v);
            }
            for (size_t i = 0; i < r.kio_rows.size() / vmc::KECCAK_IO_WIDTH; i++) {
                const uint32_t* row = &r.kio_rows[i * vmc::KECCAK_IO_WIDTH];
                misc.push_back(row[36]), misc.push_back(row[39]), tup_y.push_back(row[37]), tup_y.push_back(row[40]);
            }
            for (size_t i = 0; i < r.n_leaf(); i++) {
                const uint32_t* row = &r.leaf_rows[i * vmc::LEAF_WIDTH];
                for (uint32_t v : {row[39], row[40] * 16, row[40], row[41], row[42] * 64, row[42]}) misc.push_back(v);
            }
            for (size_t i = 0; i < r.shaio_rows.size() / vmc::SHA_IO_WIDTH; i++) {
                const uint32_t* row = &r.shaio_rows[i * vmc::SHA_IO_WIDTH];
                misc.push_back(row[31]), tup_y.push_back(row[32]);
            }
            for (size_t i = 0; i < r.i256io_rows.size() / vmc::INT256_IO_WIDTH; i++) {
                const uint32_t* row = &r.i256io_rows[i * vmc::INT256_IO_WIDTH];
                misc.push_back(row[31]), tup_y.push_back(row[32]);
            }
            // (an adapter row: word[W] | ts | base | v_lo v_hi | n_lo n_hi | prev_ts gap_lo gap_hi | ..: the gap's halves at W + 7, W + 8)
            for (unsigned m = 0; m < caps_.moduli.size(); m++) {
                const size_t nw = zkhip::modular::words_of(caps_.moduli[m]), iow = vmc::modmul_io_width(nw), W = 3 * nw;
                for (size_t i = 0; i < r.mmio_rows[m].size() / iow; i++) {
                    const uint32_t* row = &r.mmio_rows[m][i * iow];
                    misc.push_back(row[W + 7]), tup_y.push_back(row[W + 8]);
                }
            }
            for (unsigned m = 0; m < caps_.curves.size(); m++) {
                const size_t nw = zkhip::modular::words_of(caps_.curves[m].p), iow = vmc::ec_io_width(nw), W = 6 * nw;
                for (size_t i = 0; i < r.ecio_rows[m].size() / iow; i++) {
                    const uint32_t* row = &r.ecio_rows[m][i * iow];
                    misc.push_back(row[W + 7]), tup_y.push_back(row[W + 8]);
                }
            }
            for (unsigned m = 0; m < caps_.fp2_moduli.size(); m++) {
                const size_t nw = zkhip::modular::words_of(caps_.fp2_moduli[m]), iow = vmc::ec_io_width(nw), W = 6 * nw;
                for (size_t i = 0; i < r.fp2io_rows[m].size() / iow; i++) {
                    const uint32_t* row = &r.fp2io_rows[m][i * iow];
                    misc.push_back(row[W + 7]), tup_y.push_back(row[W + 8]);
                }
            }
            uint32_t* d_misc = const_cast<uint32_t*>(up(misc));
            check(zkhip_to_monty(ctx_, d_misc, misc.size()));
            count(d_misc, misc.size());
            if (!tup_y.empty()) {   // the ecall chip's two word accesses: (0, gap_hi)
                uint32_t* d_y = const_cast<uint32_t*>(up(tup_y));
                check(zkhip_to_monty(ctx_, d_y, tup_y.size()));
                check(zkhip_range_tuple_counts_tracegen(ctx_, d_zeros_, d_y, tup_y.size(), TUPLE_X, TUPLE_Y, d_tup, 1));
            }
            if (!bw_x.empty()) {    // the hinted words' bytes
                uint32_t *d_x = const_cast<uint32_t*>(up(bw_x)), *d_y = const_cast<uint32_t*>(up(bw_y));
                check(zkhip_to_monty(ctx_, d_x, bw_x.size())), check(zkhip_to_monty(ctx_, d_y, bw_y.size()));
                check(zkhip_bitwise_lookup_tracegen(ctx_, d_x, d_y, d_zeros_, bw_x.size(), 8, d_bw, 1));
            }
        }
        check(canonical_guard.release());
        check(zkhip_to_monty(ctx_, d_bw, (size_t)2 << 16)), check(zkhip_to_monty(ctx_, d_tup, (size_t)1 << 19)), check(zkhip_to_monty(ctx_, d_rng, N(vmc::A_RANGE)));
        check(zkhip_tracegen_check(ctx_));
        const auto t1 = clk::now();
        // ---- prove
        SegmentProof sp;
        sp.n_instr = r.n_instr;
        sp.proof.pvs.resize(caps_.n_airs);
        for (int k = 0; k < 8; k++) sp.proof.pvs[vmc::A_MERKLE].push_back(r.root_init[k]);
        for (int k = 0; k < 8; k++) sp.proof.pvs[vmc::A_MERKLE].push_back(r.root_final[k]);
        sp.proof.pvs[vmc::A_CONNECTOR] = {r.pc_start, r.pc_end};
        std::vector<const uint32_t*> dt(d_traces_.begin(), d_traces_.end()), pv(caps_.n_airs, nullptr);
        pv[vmc::A_MERKLE] = sp.proof.pvs[vmc::A_MERKLE].data(), pv[vmc::A_CONNECTOR] = sp.proof.pvs[vmc::A_CONNECTOR].data();
        sp.proof.proof.resize(zkhip_proof_size(pk_));
        size_t len = 0;
        check(zkhip_prove(ctx_, pk_, dt.data(), pv.data(), sp.proof.proof.data(), sp.proof.proof.size(), &len));
        sp.proof.proof.resize(len);
        sp.tracegen_mills = ms(t0, t1), sp.proving_mills = ms(t1, clk::now());
        if (self_verify && !vk_.verify(sp.proof)) throw Error("the segment proof does not verify");
        return sp;
    }

private:
    zkhip_params params_;
    SegmentCaps caps_;
    SegmentAirs sa_;
    zkhip_ctx* ctx_ = nullptr;
    zkhip_pk* pk_ = nullptr;
    VerifyingKey vk_;
    uint32_t *d_program_ = nullptr, *d_zeros_ = nullptr;
    std::vector<uint32_t*> d_traces_;
    struct Chunk {
        char* base = nullptr;
        size_t cap = 0, used = 0;
    };
    std::vector<Chunk> dev_, pin_;
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
    void release() {
        zkhip_sync(ctx_);
        for (Chunk& c : dev_) c.used = 0;
        for (Chunk& c : pin_) c.used = 0;
    }
    const uint32_t* up(const std::vector<uint32_t>& v) {
        uint32_t* d = (uint32_t*)carve(dev_, v.size() * 4, (size_t)64 << 20, [&](size_t n, void** p) { return zkhip_malloc(ctx_, n, p); });
        if (v.empty()) return d;
        char* stage = carve(pin_, v.size() * 4, (size_t)32 << 20, [&](size_t n, void** p) { return zkhip_host_alloc(ctx_, n, p); });
        memcpy(stage, v.data(), v.size() * 4);
        check(zkhip_h2d_async(ctx_, d, stage, v.size() * 4));
        return d;
    }
};

// ---- per-proof chip presence (the reference's engine proves only the chips a segment used -- the chunk circuit's 42 AIRs, AGENTS.md:183-185).
// Here heights and AIR sets are part of a key, so an app has a few SHAPES instead: the base chips; + the hash intrinsics its openvm.toml
// enables; the full set.  Each shape is a SegmentCaps (the same base heights: they depend on the frame only) with a segment key of its own;
// a segment is proven under the smallest shape that holds every chip it used.  A shape without chip X has no receiver on X's request bus,
// so a segment that called X cannot be proven under it (the LogUp sums do not cancel).
struct SegmentShapes {
    std::vector<SegmentCaps> caps;   // increasing sets of chips; caps.back() = the full set (the executor's limits)
    bool lean = false;               // caps[0] = the base chips with a SMALL memory system (see lean_caps)
    // The base chips for a segment that stays in its registers: the memory system's chips -- load / store, touched blocks, the memory
    // tree's path nodes, and the Poseidon2 chip that hashes both -- hold 28 % of the base set's cells at the heights a segment MAY need
    // (2^(f - 1) loads / stores, 2^(f - 5) blocks, 2^(f - 4) nodes, 2^(f - 2) permutations per 2^f instructions) and a compute-bound
    // segment fills a few hundred rows of them.  The reference's engine proves every chip at the height the segment needs; here heights
    // are part of a key, so this is one more shape: the same 22 chips, those four at 1 / 16 .. 1 / 8 of their height.
    static SegmentCaps lean_caps(SegmentCaps base, unsigned log_frame) {
        auto down = [&](unsigned id, unsigned by, unsigned floor) { base.log_height[id] = std::max(floor, base.log_height[id] > by ? base.log_height[id] - by : 0u); };
        (void)log_frame;
        down(vmc::A_LS, 4, 2), down(vmc::A_LEAF, 4, 4), down(vmc::A_MERKLE, 3, 8);
        base.log_height[vmc::A_POSEIDON2] = std::max(base.log_height[vmc::A_MERKLE] + 2, base.log_height[vmc::A_POSEIDON2] > 3 ? base.log_height[vmc::A_POSEIDON2] - 3 : 0u);
        return base;
    }
    static SegmentShapes of(const SegmentCaps& full, unsigned log_frame, unsigned log_program, bool with_lean = true) {
        SegmentShapes s;
        const bool hash = full.keccak() || full.sha256();
        const bool more = full.int256() || !full.moduli.empty() || !full.curves.empty() || !full.fp2_moduli.empty();
        // (the native / castf chips -- 27 + 90 + 16 columns -- are part of every shape of an app that enables them)
        const SegmentCaps base = SegmentCaps::for_frame(log_frame, log_program, 0, 0, {}, 0, 0, {}, 0, {}, full.ext);
        if (with_lean) {
            const SegmentCaps l = lean_caps(base, log_frame);
            bool smaller = false;
            for (unsigned id : l.ids) smaller = smaller || l.log_height[id] < base.log_height[id];
            if (smaller) s.caps.push_back(l), s.lean = true;   // (tiny frames sit on the floors already)
        }
        if (hash || more) s.caps.push_back(base);
        if (hash && more)
            s.caps.push_back(SegmentCaps::for_frame(log_frame, log_program, full.keccak() ? full.log_height[vmc::A_KECCAK] : 0, full.sha256() ? full.log_height[vmc::A_SHA256] : 0, {}, 0, 0,
                                                    {}, 0, {}, full.ext));
        s.caps.push_back(full);
        for (size_t k = s.lean ? 1 : 0; k < s.caps.size(); k++)   // beside the lean one a shape shares every height with the full set
            for (unsigned id : s.caps[k].ids)
                if (s.caps[k].log_height[id] != full.log_height[id]) throw Error("internal: a shape's heights differ from the full set's");
        return s;
    }
    static SegmentShapes only(const SegmentCaps& full) {
        SegmentShapes s;
        s.caps.push_back(full);
        return s;
    }
    // the smallest shape that holds every chip the segment used
    size_t shape_of(const SegmentRecords& r) const {
        auto used = [&](unsigned id) -> bool {
            if (id == vmc::A_KECCAK || id == vmc::A_KECCAK_IO) return r.n_keccak() != 0;
            if (id == vmc::A_SHA256 || id == vmc::A_SHA256_IO) return r.n_sha256() != 0;
            if (id >= vmc::A_INT256 && id <= vmc::A_SHIFT256) return !r.i256io_rows.empty();
            if (id >= vmc::A_FP2(0)) return !r.fp2_ts[(id - vmc::A_FP2(0)) / 2].empty();
            if (id >= vmc::A_EC(0)) return !r.ec_ts[(id - vmc::A_EC(0)) / 2].empty();
            if (id >= vmc::N_STATIC_AIRS) return !r.mm_ts[(id - vmc::N_STATIC_AIRS) / 2].empty();
            return false;   // (base chips are in every shape)
        };
        const SegmentCaps& full = caps.back();
        for (size_t s = 0; s + 1 < caps.size(); s++) {
            bool ok = true;
            for (unsigned id : full.ids)
                if (id >= vmc::N_BASE_AIRS && used(id) && caps[s].pos(id) < 0) ok = false;
            if (ok && lean && s == 0)   // the small memory system holds what the segment touched
                ok = r.core.ls_case.size() <= caps[0].rows(vmc::A_LS) && r.n_leaf() <= caps[0].rows(vmc::A_LEAF) && r.n_merkle() <= caps[0].rows(vmc::A_MERKLE) &&
                     r.n_p2() <= caps[0].rows(vmc::A_POSEIDON2);
            if (ok) return s;
        }
        return caps.size() - 1;
    }
};
// How each shape's proofs enter the aggregation tree (AggregationProver::ShapePolicy).  The tree's circuits share ONE height set.  Since the
// second session of round 5 (packed opened rows + Horner rows of the gate chip, csrc/recursion.hip Builder::hstep) the internal circuit --
// three node proofs at 100 queries -- needs 2^20 gate rows and 2^17 permutations, and a leaf circuit should fit that too.  Verifying a
// segment proof in-circuit costs, measured with `prove_cli leaf-stats` on the reference's three openvm.toml files at frames of 2^19
// (base 22 chips / + hash intrinsics / the chunk circuit's 51 chips / the batch circuit's 37):
//     gate rows    ~ 335 k + 28 per main column + 0.61 per word of the chips' constraint programs   (399 k / 587 k / 1377 k / 1053 k)
//     permutations ~ 39 k + 3.8 per main column + 0.21 per program word (narrow shapes)              (57 k / 103 k / 270 k / 196 k)
// (round 4: 835 rows per column, 2^21 / 2^18).  A shape takes as many proofs per leaf node as fit 2^20 rows and 2^17 permutations (the base
// chips: 2; + the hash intrinsics: 1); a shape of which not even one proof fits (the full chunk- and batch-circuit sets) keeps a leaf circuit
// of its own, natural size (2^21 / 2^19 for the chunk circuit's -- it was 2^23 / 2^19), one proof per node, and enters the tree through a
// wrapper of the common size.  The estimate only steers arity and wrapping: the common heights are the fixed point of the circuits as built.
inline std::vector<scroll_zkvm_hip::AggregationProver::ShapePolicy> shape_policies(const std::vector<VerifyingKey>& shape_vks, unsigned max_arity = 4) {
    std::vector<scroll_zkvm_hip::AggregationProver::ShapePolicy> out;
    for (const VerifyingKey& vk : shape_vks) {
        size_t w = 0, words = 0;
        for (const auto& a : vk.airs) w += a.width, words += a.program.size();
        unsigned lh = 0;
        for (unsigned h : vk.heights) lh = std::max(lh, h);
        const double up = (double)lh - 19.0;   // (a taller frame: one more level on every path -- ~20 k rows, ~2.4 k permutations per child)
        const double rows = 335e3 + 28.0 * (double)w + 0.61 * (double)words + 20e3 * up, perms = 39.4e3 + 3.8 * (double)w + 0.206 * (double)words + 2.4e3 * up;
        // (the permutation estimate is exact to 1 % on the base chips with and without the native extension and on the 26-chip shape, and high for
        // the wide shapes, which are wrapped either way.  A leaf that misses doubles EVERY node of the tree; one proof less per leaf costs half a node.)
        const double fit = std::min(0.97 * (double)(1u << 20) / rows, 0.98 * (double)(1u << 17) / perms);
        scroll_zkvm_hip::AggregationProver::ShapePolicy p;
        p.arity = (unsigned)std::min<double>(max_arity, fit);
        if (p.arity == 0) p.arity = 2, p.wrapped = true;   // (a wrapper verifies up to two proofs of the shape's own leaf circuit: ~0.35 M rows, 44 k permutations each)
        out.push_back(p);
    }
    return out;
}
// Lane 0 builds the segment keys of every shape at setup (their verifying keys make the aggregation key); the other lanes build a shape's key when its
// first segment arrives -- inside the timed proving, 0.2 - 0.4 s for the chunk circuit's 51-chip shape.  warm_lanes builds, side by side, the keys of the
// shapes this guest's earlier flows used (the aggregation key cache remembers them) on every other lane before the flow starts: key generation is
// setup, as the reference's `Sdk::app_keygen` is (crates/prover/src/prover/mod.rs:147-170).  Without a cache file nothing is known and nothing is built.
template <class Lane>
inline void warm_lanes(const std::vector<Lane*>& lanes, const std::vector<char>& used_before) {
    std::vector<std::thread> th;
    for (size_t l = 1; l < lanes.size(); l++)
        th.emplace_back([&, l] {
            try {
                for (size_t sh = 0; sh < used_before.size() && sh < lanes[l]->n_shapes(); sh++)
                    if (used_before[sh]) (void)lanes[l]->vk(sh);
            } catch (...) {   // (the lane builds the key again, and reports, when a segment of the shape arrives)
            }
        });
    for (auto& t : th) t.join();
}
// a lane of the flow: one SegmentProver per shape (built at the first segment of that shape; lane 0 builds all of them at setup: their keys
// are the aggregation tree's leaf circuits)
class ShapedSegmentProver {
public:
    ShapedSegmentProver(const zkhip_params& params, const Exe& exe, const SegmentShapes& shapes, int device = 0, bool build_all = false)
        : params_(params), exe_(exe), shapes_(shapes), device_(device), provers_(shapes.caps.size()) {
        if (build_all)
            for (size_t s = 0; s < provers_.size(); s++) (void)prover(s);
    }
    size_t n_shapes() const { return provers_.size(); }
    const VerifyingKey& vk(size_t shape) { return prover(shape).vk(); }
    const VerifyingKey& vk() { return vk(provers_.size() - 1); }
    size_t shape_of(const SegmentRecords& r) const { return shapes_.shape_of(r); }
    SegmentProof prove(const SegmentRecords& r, bool self_verify = true, int force_shape = -1) {
        const size_t s = force_shape >= 0 ? (size_t)force_shape : shapes_.shape_of(r);
        SegmentProof p = prover(s).prove(r, self_verify);
        p.shape = s;
        return p;
    }

private:
    zkhip_params params_;
    const Exe& exe_;
    SegmentShapes shapes_;
    int device_;
    std::vector<std::unique_ptr<SegmentProver>> provers_;
    SegmentProver& prover(size_t s) {
        if (!provers_.at(s)) provers_[s].reset(new SegmentProver(params_, exe_, shapes_.caps[s], device_));
        return *provers_[s];
    }
};

// ---- the whole flow ---------------------------------------------------------------------------------------------------------------------
struct GuestStark {
    ChildProof root;                       // the root node's proof + public values [app-vk digest | (pc, root) start | (pc, root) end | accumulator]
    VerifyingKey root_vk;
    size_t levels = 0, segments = 0;
    std::vector<size_t> segments_per_shape, chips_per_shape;   // how many segments were proven under each shape, and how many chips a shape carries
    std::vector<size_t> instr_per_shape, prove_ms_per_shape, tracegen_ms_per_shape;   // per shape: instructions, summed proving / trace-generation time of its segments
    std::vector<size_t> nodes_per_slot;                        // how the tree's nodes spread over the device slots
    std::vector<size_t> segments_per_lane;                     // segment proofs per lane (lanes are listed device by device: FlowOptions::lanes per device)
    size_t leaf_circuits_at_setup = 0, leaf_circuits_on_demand = 0;   // leaf circuits built with the aggregation key / when a shape's first segment arrived
    double agg_build_seconds = 0, agg_keygen_seconds = 0;      // the aggregation circuits and their keys (a one-time cost of a prover that lives on)
    std::vector<unsigned> node_log_heights;                    // heights of the node circuits' chips (gate, Poseidon2, public values)
    ExecutionResult exec;
    Digest image_root{};
    uint32_t entry_pc = 0;
    // openings of the two public-value blocks (address space 3) in the final memory root: sibling digests bottom-up, per block
    std::vector<uint32_t> pv_openings;
    // the guest's deferral region in the final memory: its 4096 cells, then the 19 sibling digests above its subtree (deferral_base)
    std::vector<uint32_t> deferral_opening;
    // execution_mills: what the flow's feeding thread spent in the executor -- with the serial executor the execution itself, with the parallel
    // one the time it WAITED for the next segment (the passes run on their own threads: executor_*_mills are their busy times)
    uint64_t execution_mills = 0, segment_proving_mills = 0, aggregation_mills = 0;
    unsigned executor_threads = 0;
    uint64_t executor_metered_mills = 0, executor_record_mills_sum = 0, executor_tree_mills = 0;
    uint64_t aggregation_setup_wait_mills = 0;   // waiting for the aggregation circuits / keys of this app to be built (first task only)
    uint64_t sum_segment_tracegen_mills = 0, sum_segment_prove_mills = 0;   // summed over the segments (lanes run side by side)
    std::vector<SegmentProof> segment_proofs;   // kept when asked for
    // segment proofs that failed once and were made again from the same records (FlowOptions::retry_segments): segment index + the first
    // attempt's message.  Empty on a clean run; the stress loops require it to be empty AND run with the retry off.
    std::vector<std::pair<size_t, std::string>> segments_retried;
};

// the public-value cells' Merkle openings in the executor's final tree
inline std::vector<uint32_t> open_public_values(const MemoryTree& tree) {
    std::vector<uint32_t> out;
    for (uint32_t blk = 0; blk < 2; blk++) {
        uint32_t idx = (3u << vmc::LABEL_BITS) | blk;
        for (int l = (int)vmc::LEAF_LEVEL; l > 0; l--) {
            const Digest sib = tree.get((unsigned)l, idx ^ 1u);
            out.insert(out.end(), sib.begin(), sib.end());
            idx >>= 1;
        }
    }
    return out;
}
// what a verifier recomputes: the final memory root from the claimed public values and their openings
inline bool check_public_values(const std::vector<uint8_t>& pv, const std::vector<uint32_t>& openings, const uint32_t root[8]) {
    if (pv.size() != NUM_PUBLIC_VALUE_BYTES || openings.size() != 2 * 8 * vmc::LEAF_LEVEL) return false;
    for (uint32_t blk = 0; blk < 2; blk++) {
        uint32_t cells[8];
        for (int j = 0; j < 8; j++) cells[j] = pv[16 * blk + 2 * j] | ((uint32_t)pv[16 * blk + 2 * j + 1] << 8);
        Digest cur = p2_block(cells);
        uint32_t idx = (3u << vmc::LABEL_BITS) | blk;
        const uint32_t* sib = &openings[(size_t)blk * 8 * vmc::LEAF_LEVEL];
        for (int l = (int)vmc::LEAF_LEVEL; l > 0; l--, sib += 8, idx >>= 1) {
            Digest s;
            std::copy(sib, sib + 8, s.begin());
            cur = (idx & 1u) ? p2_compress(s, cur) : p2_compress(cur, s);
        }
        if (!std::equal(cur.begin(), cur.end(), root)) return false;
    }
    return true;
}

// ---- the deferral region (crates/prover/src/prover/mod.rs:222-232: the reference reserves an address space of 2^25 cells for the deferral
// state; crates/types/src/proof.rs `deferral_merkle_proofs`) ----
// A guest that defers verification writes its claims into the 8 KiB (512 blocks = one subtree of the memory tree) that follow its
// initial data image, 8 KiB-aligned: word 0 = the number of claims n, claim k = the 32 words at word 32 + 32 k
// [input commitment (8) | exe commitment (8) | vm commitment (8) | the child's 32 public-value bytes].  Up to 63 claims (a batch holds up to 45 chunks: crates/types/batch/src/payload/v6.rs:10).
constexpr uint32_t DEFERRAL_REGION_BYTES = 8192, DEFERRAL_MAX_CLAIMS = 63, DEFERRAL_SUBTREE_LEVELS = 9;
inline uint32_t deferral_base(const Exe& exe) {
    const uint64_t end = (uint64_t)exe.data_base + exe.data.size();
    return (uint32_t)((end + DEFERRAL_REGION_BYTES - 1) / DEFERRAL_REGION_BYTES * DEFERRAL_REGION_BYTES);
}
// the node of the memory tree above the region's 512 blocks, on the level DEFERRAL_SUBTREE_LEVELS above the leaves (what a deferral node
// over proofs of THIS guest hard-wires: zkhip_recursion_stmt.region_index)
inline uint32_t deferral_region_index(const Exe& exe) { return ((2u << vmc::LABEL_BITS) | (deferral_base(exe) / 16)) >> DEFERRAL_SUBTREE_LEVELS; }
inline bool has_deferral_region(const Exe& exe) { return (uint64_t)deferral_base(exe) + DEFERRAL_REGION_BYTES <= (uint64_t)exe.data_base + exe.memory_bytes; }
// the region's cells in the executor's final memory + the sibling digests above its subtree, bottom-up
template <class Executor>
inline std::vector<uint32_t> open_deferral_region(const Executor& ex, const Exe& exe) {
    std::vector<uint32_t> out;
    if (!has_deferral_region(exe)) return out;
    const uint32_t base = deferral_base(exe);
    for (uint32_t w = 0; w < DEFERRAL_REGION_BYTES / 4; w++) {
        const uint32_t v = ex.peek_memory(base + 4 * w);
        out.push_back(v & 0xffffu), out.push_back(v >> 16);
    }
    uint32_t idx = ((2u << vmc::LABEL_BITS) | (base / 16)) >> DEFERRAL_SUBTREE_LEVELS;
    for (int l = (int)vmc::LEAF_LEVEL - (int)DEFERRAL_SUBTREE_LEVELS; l > 0; l--, idx >>= 1) {
        const Digest sib = ex.tree().get((unsigned)l, idx ^ 1u);
        out.insert(out.end(), sib.begin(), sib.end());
    }
    return out;
}
// what a verifier recomputes: the final memory root from the region's cells and the siblings; then the claims the guest made
inline bool check_deferral_region(const Exe& exe, const std::vector<uint32_t>& opening, const uint32_t root[8], std::vector<std::array<uint32_t, 32>>* claims) {
    const size_t n_cells = DEFERRAL_REGION_BYTES / 2, n_sib = vmc::LEAF_LEVEL - DEFERRAL_SUBTREE_LEVELS;
    if (!has_deferral_region(exe) || opening.size() != n_cells + 8 * n_sib) return false;
    for (size_t i = 0; i < n_cells; i++)
        if (opening[i] > 0xffffu) return false;
    std::vector<Digest> level(n_cells / 8);
    for (size_t b = 0; b < level.size(); b++) level[b] = p2_block(&opening[8 * b]);
    while (level.size() > 1) {
        std::vector<Digest> up(level.size() / 2);
        for (size_t i = 0; i < up.size(); i++) up[i] = p2_compress(level[2 * i], level[2 * i + 1]);
        level = std::move(up);
    }
    Digest cur = level[0];
    uint32_t idx = ((2u << vmc::LABEL_BITS) | (deferral_base(exe) / 16)) >> DEFERRAL_SUBTREE_LEVELS;
    const uint32_t* sib = &opening[n_cells];
    for (size_t l = 0; l < n_sib; l++, sib += 8, idx >>= 1) {
        Digest s;
        std::copy(sib, sib + 8, s.begin());
        cur = (idx & 1u) ? p2_compress(s, cur) : p2_compress(cur, s);
    }
    if (!std::equal(cur.begin(), cur.end(), root)) return false;
    auto word = [&](size_t w) { return opening[2 * w] | (opening[2 * w + 1] << 16); };
    const uint32_t n = word(0);
    if (n > DEFERRAL_MAX_CLAIMS) return false;
    if (claims) {
        claims->clear();
        for (uint32_t k = 0; k < n; k++) {
            std::array<uint32_t, 32> c;
            for (size_t j = 0; j < 32; j++) c[j] = word(32 + 32 * k + j);
            claims->push_back(c);
        }
    }
    return true;
}

// root of the initial memory of a guest image (what a verifier derives from the ELF)
inline Digest guest_image_root(const Exe& exe) {
    StdIn none;
    SegmentCaps caps = SegmentCaps::for_frame(4, 2);
    SegmentExecutor ex(exe, none, caps);
    return ex.image_root();
}

// execute -> segment proofs -> aggregation tree -> root, with the provers handed in (their keys and circuits are reused from task
// to task: the reference keeps its Sdk in a OnceLock, crates/prover/src/prover/mod.rs:78,115-126).  The executor (this thread)
// streams segments into a bounded queue; every lane (a SegmentProver with a context = HIP stream of its own) proves the segments it
// takes, so the launch gaps and host-side pauses of one lane are filled by the other; the mandatory self-verification of the segment
// proofs runs on host threads beside the lanes.  Segments are independent proofs; their order is restored.
template <class Lane>
inline GuestStark prove_guest_with(const std::vector<Lane*>& lanes, scroll_zkvm_hip::AggregationProver& agg, const Exe& exe, const StdIn& in,
                                   const SegmentCaps& caps, bool keep_segments = false, bool verify_segments = false, bool greedy_tree = true, bool trace_tree = false,
                                   size_t wide_in_flight = 0, bool retry_segments = true, unsigned exec_threads = 0, int fail_segment_once = -1) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::duration d) { return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(d).count(); };
    if (lanes.empty()) throw Error("no segment prover lanes");
    GuestStark g;
    // the executor: a metered pass that cuts the run + exec_threads record passes side by side (FlowOptions::exec_threads; 0 = the serial
    // executor on this thread).  The loop below takes the segments in order either way.
    ParallelSegmentExecutor ex(exe, in, caps, exec_threads);
    g.image_root = ex.image_root(), g.entry_pc = exe.entry ? exe.entry : exe.pc_base;
    struct Item {
        size_t index;
        SegmentRecords rec;
        bool wide = false;   // a segment of a WRAPPED shape (FlowOptions::wide_in_flight)
    };
    std::deque<Item> queue;
    std::vector<SegmentRecords> pool;
    std::mutex mu;
    std::condition_variable cv_push, cv_pop;
    bool closed = false, failed = false;
    std::string error;
    std::vector<std::pair<size_t, std::string>> retried;
    std::deque<SegmentProof> proofs;   // (a deque: verifier threads read finished proofs while later ones are still being added)
    std::vector<std::thread> workers, verifiers;
    std::deque<std::pair<size_t, const SegmentProof*>> to_verify;   // (element addresses of a deque stay put while it grows)
    bool proving_done = false;
    const auto t_start = clk::now();
    // Every segment proof is checked by the leaf circuit's witness generation (it replays the verifier and throws on a child that
    // does not verify), every node proof by the level above and the root by verify_guest_stark -- as the reference verifies its final
    // proof only (crates/prover/src/prover/mod.rs:407-411).  FlowOptions::verify_segments adds the host verification of every segment
    // proof beside the proving (it hashes as much as the witness generation does, on the same cores).
    // the aggregation tree runs as a stream beside the segment proving: a leaf node starts when its four segment proofs exist
    scroll_zkvm_hip::AggregationProver::TreeStream tree(agg, greedy_tree);
    tree.trace = trace_tree;
    size_t n_wide = 0;
    std::vector<size_t> seg_per_lane(lanes.size(), 0);
    for (size_t lane_index = 0; lane_index < lanes.size(); lane_index++)
        workers.emplace_back([&, lane_index] {
            Lane* const lane = lanes[lane_index];
            try {
                for (;;) {
                    Item it;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        // FlowOptions::wide_in_flight > 0 (an experiment, off): at most so many proofs of WRAPPED shapes at a time; the segment stays at
                        // the head of the queue meanwhile -- a lane that held it back would hand the tree its segment proofs out of order, and
                        // the tree folds neighbours.
                        cv_pop.wait(lk, [&] { return failed || (queue.empty() ? closed : !(queue.front().wide && n_wide >= wide_in_flight)); });
                        if (failed || queue.empty()) return;
                        it = std::move(queue.front());
                        queue.pop_front();
                        if (it.wide) n_wide++;
                    }
                    cv_push.notify_one();
                    struct WideSlot {   // (given back when the proof exists or fails)
                        std::mutex& mu;
                        std::condition_variable& cv;
                        size_t& n;
                        bool held;
                        ~WideSlot() {
                            if (!held) return;
                            {
                                std::lock_guard<std::mutex> lk(mu);
                                n--;
                            }
                            cv.notify_all();
                        }
                    } wide_slot{mu, cv_pop, n_wide, it.wide};
                    // a segment proof that fails (a refused trace check, the device self-check of zkhip_config.self_check, a device error)
                    // is made once more from the same records before the run is given up; a second failure ends it
                    // (FlowOptions::retry_segments; every retry is counted into GuestStark::segments_retried -- a retried wrong node must
                    // not pass for a clean run: VERDICT round 5, weak 1)
                    SegmentProof p;
                    auto first_attempt = [&] {
                        if (fail_segment_once >= 0 && it.index == (size_t)fail_segment_once) throw Error("injected failure (FlowOptions::fail_segment_once)");
                        return lane->prove(it.rec, /*self_verify=*/false);
                    };
                    if (!retry_segments) {
                        p = first_attempt();
                    } else {
                        try {
                            p = first_attempt();
                        } catch (const std::exception& e) {
                            std::fprintf(stderr, "[zkhip flow] segment %zu failed (%s): proving it once more\n", it.index, e.what());
                            {
                                std::lock_guard<std::mutex> lk(mu);
                                retried.push_back({it.index, e.what()});
                            }
                            p = lane->prove(it.rec, /*self_verify=*/false);
                        }
                    }
                    tree.push(it.index, p.proof, p.shape);
                    std::lock_guard<std::mutex> lk(mu);
                    seg_per_lane[lane_index]++;
                    if (proofs.size() <= it.index) proofs.resize(it.index + 1);
                    proofs[it.index] = std::move(p);
                    if (verify_segments) to_verify.push_back({it.index, &proofs[it.index]});
                    pool.push_back(std::move(it.rec));
                    cv_pop.notify_all();
                }
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lk(mu);
                if (error.empty()) error = e.what();
                failed = true;
                cv_push.notify_all(), cv_pop.notify_all();
            }
        });
    for (size_t v = 0; v < std::max<size_t>(4, 2 * lanes.size()); v++)
        verifiers.emplace_back([&] {
            for (;;) {
                size_t k;
                const SegmentProof* made;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_pop.wait(lk, [&] { return !to_verify.empty() || proving_done || failed; });
                    if (to_verify.empty()) return;
                    k = to_verify.front().first, made = to_verify.front().second;
                    to_verify.pop_front();
                }
                if (!agg.app_vk(made->shape).verify(made->proof)) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (error.empty())
                        error = "the proof of segment " + std::to_string(k) + " (shape " + std::to_string(made->shape) + ") does not verify: verifier.hip:" +
                                std::to_string(agg.app_vk(made->shape).refused_at(made->proof)) + " refuses it";
                    failed = true;
                    cv_push.notify_all(), cv_pop.notify_all();
                }
            }
        });
    // the aggregation circuits and keys of the first levels are built beside the segment proving (nothing else uses the
    // aggregation prover until the tree starts); deeper trees build their levels when they get there
    std::thread warm_agg([&agg] {
        try {
            for (size_t l = 1; l <= 3; l++) (void)agg.node_vk(l);
        } catch (...) {
        }
    });
    clk::duration t_exec{};
    size_t n_seg = 0;
    std::string exec_error;
    try {
        for (bool done = false; !done;) {
            Item it;
            it.index = n_seg++;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (failed) break;
                if (!pool.empty()) it.rec = std::move(pool.back()), pool.pop_back();
            }
            const auto a = clk::now();
            done = ex.run_segment(it.rec);
            t_exec += clk::now() - a;
            const size_t seg_shape = lanes[0]->shape_of(it.rec);
            it.wide = wide_in_flight && agg.wrapped(seg_shape);
            agg.prefetch_leaf(seg_shape);   // (a shape whose leaf circuit was left out at setup is built beside the proving, from now)
            std::unique_lock<std::mutex> lk(mu);
            cv_push.wait(lk, [&] { return queue.size() < 2 * lanes.size() || failed; });
            queue.push_back(std::move(it));
            lk.unlock();
            cv_pop.notify_all();
        }
    } catch (const std::exception& e) {
        exec_error = e.what();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        closed = true;
        if (!exec_error.empty()) failed = true;
    }
    cv_pop.notify_all();
    for (auto& t : workers) t.join();
    {
        std::lock_guard<std::mutex> lk(mu);
        proving_done = true;
    }
    cv_pop.notify_all();
    for (auto& t : verifiers) t.join();
    const auto t_segments_done = clk::now();
    warm_agg.join();   // (a one-time cost per app: a prover that lives on keeps the circuits and keys)
    g.aggregation_setup_wait_mills = ms(clk::now() - t_segments_done);
    if (!exec_error.empty()) throw Error(exec_error);
    if (!error.empty()) throw Error("segment prover: " + error);
    if (proofs.size() != n_seg) throw Error("segment prover: proofs missing");
    g.executor_threads = ex.threads(), g.executor_metered_mills = (uint64_t)(1e3 * ex.metered_seconds()), g.executor_record_mills_sum = (uint64_t)(1e3 * ex.record_seconds());
    g.executor_tree_mills = (uint64_t)(1e3 * (ex.threads() ? ex.tree_seconds() : ex.close_seconds()));
    g.execution_mills = ms(t_exec), g.segment_proving_mills = ms(t_segments_done - t_start);   // wall: the execution runs under the proving
    bool all_zero = true;
    for (uint8_t b : ex.public_values()) all_zero = all_zero && b == 0;
    if (all_zero) throw Error("public_values are all 0s for unexpected reason");   // crates/prover/src/utils/vm.rs:34-46
    g.exec = ExecutionResult{ex.instret(), ex.public_values()};
    g.pv_openings = open_public_values(ex.tree());
    g.deferral_opening = open_deferral_region(ex, exe);
    g.segments = n_seg;
    g.segments_per_lane = seg_per_lane;
    std::sort(retried.begin(), retried.end());
    g.segments_retried = std::move(retried);
    std::vector<ChildProof> seg_proofs;
    g.segments_per_shape.assign(agg.n_shapes(), 0), g.instr_per_shape.assign(agg.n_shapes(), 0), g.prove_ms_per_shape.assign(agg.n_shapes(), 0), g.tracegen_ms_per_shape.assign(agg.n_shapes(), 0);
    for (auto& p : proofs) g.instr_per_shape.at(p.shape) += p.n_instr, g.prove_ms_per_shape.at(p.shape) += p.proving_mills, g.tracegen_ms_per_shape.at(p.shape) += p.tracegen_mills;
    for (size_t sh = 0; sh < agg.n_shapes(); sh++) g.chips_per_shape.push_back(agg.app_vk(sh).airs.size());
    for (auto& p : proofs) seg_proofs.push_back(p.proof), g.sum_segment_tracegen_mills += p.tracegen_mills, g.sum_segment_prove_mills += p.proving_mills, g.segments_per_shape.at(p.shape)++;
    if (keep_segments) g.segment_proofs.assign(std::make_move_iterator(proofs.begin()), std::make_move_iterator(proofs.end()));
    const auto t0 = clk::now();
    const scroll_zkvm_hip::AggregationPlan plan = scroll_zkvm_hip::AggregationPlan::build(seg_proofs.size(), agg.tree_config());
    g.root = tree.finish(seg_proofs.size());   // (what is left of the tree once the last segment proof exists)
    agg.remember_used_shapes();                // (the key cache learns which shapes this guest puts into the tree)
    g.levels = tree.levels();   // (leaf nodes are cut where the shape changes: the plan's fixed grouping is a lower bound)
    (void)plan;
    g.root_vk = agg.root_vk(plan.levels.size());   // (one aggregation key: the same for every depth)
    g.aggregation_mills = ms(clk::now() - t0);
    g.nodes_per_slot = agg.stats.nodes_per_slot;
    g.agg_build_seconds = agg.stats.build_seconds, g.agg_keygen_seconds = agg.stats.keygen_seconds;
    g.leaf_circuits_at_setup = agg.stats.leafs_at_setup, g.leaf_circuits_on_demand = agg.stats.leafs_on_demand;
    g.node_log_heights = g.root_vk.heights;
    return g;
}
// Rows of the Keccak-f chip for an app: the reference enables the intrinsic per app in openvm.toml (`[app_vm_config.keccak]`,
// crates/circuits/chunk-circuit/openvm.toml); with it the chip holds 2^(log_frame - 6) rows (at least 32), without it the app has no keccak chips (0).
inline bool config_has_section(const std::string& path_app_config, const char* section) {
    std::ifstream f(path_app_config);
    std::string line;
    // (a section HEADER: the line, trimmed, starts with it -- the name inside a comment or a value is not the section: ADVICE round 5)
    while (std::getline(f, line)) {
        const size_t b0 = line.find_first_not_of(" \t");
        if (b0 != std::string::npos && line.compare(b0, strlen(section), section) == 0) return true;
    }
    return false;
}
inline unsigned keccak_log_rows(const std::string& path_app_config, unsigned log_frame) {
    return config_has_section(path_app_config, "[app_vm_config.keccak]") ? std::max(5u, log_frame > 6 ? log_frame - 6 : 0u) : 0u;
}
// likewise `[app_vm_config.sha256]`: 2^(log_frame - 5) rows of the SHA-256 compression chip (65 per block, at least 128)
inline unsigned sha256_log_rows(const std::string& path_app_config, unsigned log_frame) {
    const bool on = config_has_section(path_app_config, "[app_vm_config.sha256]") || config_has_section(path_app_config, "[app_vm_config.sha2]");   // (the reference's name)
    return on ? std::max(7u, log_frame > 5 ? log_frame - 5 : 0u) : 0u;
}
// `[app_vm_config.bigint]`: 2^(log_frame - 8) rows of the 256-bit ALU chip (one operation per row)
inline unsigned int256_log_rows(const std::string& path_app_config, unsigned log_frame) {
    return config_has_section(path_app_config, "[app_vm_config.bigint]") ? std::max(1u, log_frame > 8 ? log_frame - 8 : 0u) : 0u;
}
inline zkhip::modular::U256 parse_decimal_u256(const std::string& digits) {
    zkhip::modular::U256 v{};
    if (digits.empty()) throw Error("openvm.toml: an empty number");
    for (char ch : digits) {   // decimal -> words
        if (ch < '0' || ch > '9') throw Error("openvm.toml: a modulus or coefficient is not a decimal number");
        uint64_t c = (uint64_t)(ch - '0');
        for (size_t w = 0; w < zkhip::modular::MAX_WORDS; w++) {
            c += (uint64_t)v.w[w] * 10u;
            v.w[w] = (uint32_t)c, c >>= 32;
        }
        if (c) throw Error("openvm.toml: a modulus or coefficient does not fit 384 bits (the limb chips are built for 32 or 48 byte limbs)");
    }
    return v;
}
// `[[app_vm_config.ecc.supported_curves]]` blocks with `modulus = "<decimal>"` and `a = "<decimal>"` (the reference's chunk circuit
// lists secp256k1, P-256 and bn254 G1; `scalar`, `b` and `struct_name` do not enter the chips): the curves in the file's order
inline std::vector<zkhip::ecc::Curve> config_curves(const std::string& path_app_config) {
    std::ifstream f(path_app_config);
    std::string line;
    std::vector<zkhip::ecc::Curve> out;
    std::vector<std::pair<bool, bool>> have;
    bool in_block = false;
    auto value_of = [](const std::string& l) {
        const size_t q0 = l.find('"'), q1 = q0 == std::string::npos ? q0 : l.find('"', q0 + 1);
        if (q1 == std::string::npos) throw Error("openvm.toml: a curve parameter is not a quoted decimal number");
        return l.substr(q0 + 1, q1 - q0 - 1);
    };
    while (std::getline(f, line)) {
        const size_t b0 = line.find_first_not_of(" \t");
        if (b0 == std::string::npos) continue;
        if (line[b0] == '[') {
            in_block = line.find("[[app_vm_config.ecc.supported_curves]]") != std::string::npos;
            if (in_block) out.push_back(zkhip::ecc::Curve{}), have.push_back({false, false});
            continue;
        }
        if (!in_block) continue;
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        std::string key = line.substr(b0, eq - b0);
        while (!key.empty() && (key.back() == ' ' || key.back() == '\t')) key.pop_back();
        if (key == "modulus") out.back().p = parse_decimal_u256(value_of(line)), have.back().first = true;
        if (key == "a") out.back().a = parse_decimal_u256(value_of(line)), have.back().second = true;
    }
    for (size_t i = 0; i < out.size(); i++)
        if (!have[i].first || !have[i].second) throw Error("openvm.toml: curve " + std::to_string(i) + " needs `modulus` and `a`");
    return out;
}
// `[app_vm_config.fp2] supported_moduli = [["<name>", "<decimal>"], ...]` (the reference's chunk circuit lists bn254's Fp2): the moduli
inline std::vector<zkhip::modular::U256> config_fp2_moduli(const std::string& path_app_config) {
    std::ifstream f(path_app_config);
    std::string line, body;
    bool in_section = false, in_list = false;
    int depth = 0;
    while (std::getline(f, line)) {
        const size_t b0 = line.find_first_not_of(" \t");
        if (!in_list && b0 != std::string::npos && line[b0] == '[') in_section = line.find("[app_vm_config.fp2]") != std::string::npos;
        if (!in_section) continue;
        if (!in_list && line.find("supported_moduli") != std::string::npos) in_list = true, line = line.substr(line.find('=') + 1);
        if (!in_list) continue;
        body += line + " ";
        for (char ch : line) depth += ch == '[' ? 1 : ch == ']' ? -1 : 0;
        if (depth <= 0 && body.find('[') != std::string::npos) break;
    }
    std::vector<zkhip::modular::U256> out;
    // every inner [ "name", "decimal" ] pair: the last quoted string of the pair is the modulus
    for (size_t p = body.find('[', body.find('[') == std::string::npos ? 0 : body.find('[') + 1); p != std::string::npos; p = body.find('[', p + 1)) {
        const size_t e = body.find(']', p);
        if (e == std::string::npos) break;
        const std::string pair = body.substr(p + 1, e - p - 1);
        const size_t q1 = pair.rfind('"'), q0 = q1 == std::string::npos || q1 == 0 ? std::string::npos : pair.rfind('"', q1 - 1);
        if (q0 == std::string::npos) throw Error("openvm.toml: an fp2 modulus entry is not [\"name\", \"decimal\"]");
        out.push_back(parse_decimal_u256(pair.substr(q0 + 1, q1 - q0 - 1)));
        p = e;
    }
    return out;
}
// 2^(log_frame - 9) rows per curve's point chip (one operation per row)
inline unsigned ec_log_rows(unsigned log_frame) { return log_frame > 10 ? log_frame - 9 : 1; }
// `[app_vm_config.modular] supported_moduli = ["<decimal>", ...]` (the reference's chunk circuit lists six): the moduli as 256-bit words
inline std::vector<zkhip::modular::U256> config_moduli(const std::string& path_app_config) {
    std::ifstream f(path_app_config);
    std::string line, body;
    bool in_section = false, in_list = false;
    while (std::getline(f, line)) {
        const size_t b0 = line.find('[');
        if (!in_list && b0 != std::string::npos && line.find_first_not_of(" \t") == b0) in_section = line.find("[app_vm_config.modular]") != std::string::npos;
        if (!in_section) continue;
        if (line.find("supported_moduli") != std::string::npos) in_list = true;
        if (in_list) {
            body += line;
            if (line.find(']') != std::string::npos && line.find("[app_vm_config") == std::string::npos) break;
        }
    }
    std::vector<zkhip::modular::U256> out;
    for (size_t p = body.find('"'); p != std::string::npos;) {
        const size_t e = body.find('"', p + 1);
        if (e == std::string::npos) break;
        const zkhip::modular::U256 v = parse_decimal_u256(body.substr(p + 1, e - p - 1));
        out.push_back(v);
        p = body.find('"', e + 1);
    }
    return out;
}
// `[app_vm_config.native]`, `[app_vm_config.castf]`, `[app_vm_config.pairing]` (crates/circuits/batch-circuit/openvm.toml:16,24,25;
// bundle-circuit/openvm.toml:16,18; chunk-circuit/openvm.toml:35): include/zkhip_native.hpp
inline zkhip::native::Enabled config_native(const std::string& path_app_config) {
    zkhip::native::Enabled e;
    e.native = config_has_section(path_app_config, "[app_vm_config.native]");
    e.castf = config_has_section(path_app_config, "[app_vm_config.castf]");
    e.pairing = config_has_section(path_app_config, "[app_vm_config.pairing]");
    return e;
}
// the chips of an app: every section of its openvm.toml at the heights of a 2^log_frame frame (no section is ignored)
inline SegmentCaps config_caps(const std::string& path_app_config, unsigned log_frame, unsigned log_program) {
    return SegmentCaps::for_frame(log_frame, log_program, keccak_log_rows(path_app_config, log_frame), sha256_log_rows(path_app_config, log_frame), config_moduli(path_app_config),
                                  log_frame > 8 ? log_frame - 8 : 1, int256_log_rows(path_app_config, log_frame), config_curves(path_app_config), ec_log_rows(log_frame),
                                  config_fp2_moduli(path_app_config), config_native(path_app_config));
}
inline GuestStark prove_guest_universal(const zkhip_params& params, const Exe& exe, const StdIn& in, unsigned log_frame = 20, int device = 0,
                                        bool keep_segments = false, unsigned inflight = 2, unsigned log_keccak = 0, unsigned log_sha256 = 0,
                                        const std::vector<zkhip::modular::U256>& moduli = {}, unsigned log_int256 = 0,
                                        const std::vector<zkhip::ecc::Curve>& curves = {}, const std::vector<zkhip::modular::U256>& fp2_moduli = {},
                                        const scroll_zkvm_hip::FlowOptions& flow = scroll_zkvm_hip::FlowOptions::from_env(),
                                        const zkhip::native::Enabled& native_ext = zkhip::native::Enabled()) {
    const SegmentCaps caps = SegmentCaps::for_frame(log_frame, vm_log2_ceil(exe.program.size()), log_keccak, log_sha256, moduli, log_frame > 8 ? log_frame - 8 : 1, log_int256,
                                                    curves, ec_log_rows(log_frame), fp2_moduli, native_ext);
    // a segment is proven under the smallest SHAPE (set of chips) that holds what it used (ZKHIP_ONE_SHAPE=1: always the full set, as round 3)
    const SegmentShapes shapes = flow.one_shape ? SegmentShapes::only(caps) : SegmentShapes::of(caps, log_frame, vm_log2_ceil(exe.program.size()), flow.lean_shape);
    // the segments of ONE task spread over the GPUs of the node -- `inflight` lanes per listed device, each device its
    // own copy of the (read-only) segment keys; a segment goes to whichever lane is free first
    const std::vector<int> devs = flow.devices.empty() ? std::vector<int>{device} : flow.devices;
    std::vector<std::unique_ptr<ShapedSegmentProver>> own;
    std::vector<ShapedSegmentProver*> lanes;
    for (int d : devs)
        for (unsigned l = 0; l < std::max(1u, inflight); l++) own.emplace_back(new ShapedSegmentProver(params, exe, shapes, d, /*build_all=*/own.empty())), lanes.push_back(own.back().get());
    std::vector<VerifyingKey> shape_vks;
    for (size_t sh = 0; sh < lanes[0]->n_shapes(); sh++) shape_vks.push_back(lanes[0]->vk(sh));
    // ZKHIP_AGG_100BIT=1 (measurements): the leaf / internal pair of AggregationSystemParams::with_100_bits_security() instead of the app's
    // parameters on every level (the root then verifies under ITS level's parameters: in-process through GuestStark::root_vk)
    const scroll_zkvm_hip::AggregationSystemParams pair = scroll_zkvm_hip::AggregationSystemParams::nodes_100_bits_security();
    scroll_zkvm_hip::AggregationTreeConfig tree_cfg;
    tree_cfg.num_children_internal = flow.internal_arity;
    scroll_zkvm_hip::AggregationProver agg =
        scroll_zkvm_hip::AggregationProver::setup_shapes(shape_vks, segment_statement(), tree_cfg, device, flow.agg_nodes_100bit ? &pair : nullptr);
    {   // the tree's nodes go to whichever pipeline is free first: flow.agg_slots pipelines per device
        std::vector<int> slots;
        for (unsigned r = 0; r < std::max(1u, flow.agg_slots); r++) slots.insert(slots.end(), devs.begin(), devs.end());
        agg.set_devices(slots);
    }
    agg.set_shape_policies(shape_policies(shape_vks));
    (void)agg.node_vk(0);   // the leaf circuit and its key: setup, like the segment keys
    warm_lanes(lanes, agg.shapes_used_before());
    return prove_guest_with(lanes, agg, exe, in, caps, keep_segments, flow.verify_segments, !flow.balanced_tree, flow.trace_tree, flow.wide_in_flight, flow.retry_segments, flow.exec_threads_or_auto(), flow.fail_segment_once);
}

// The verifier's side: the root proof under the root verifying key, then the statement: the run starts at the guest's entry on the
// guest's image, ends with pc = 0 (exit code 0), and the claimed public values open in the final memory root.
inline bool verify_guest_proof(const VerifyingKey& root_vk, const ChildProof& root, uint32_t entry_pc, const Digest& image_root,
                               const std::vector<uint8_t>& public_values, const std::vector<uint32_t>& pv_openings, std::string* why = nullptr) {
    auto fail = [&](const char* m) {
        if (why) *why = m;
        return false;
    };
    if (!root_vk.verify(root)) return fail("the root proof does not verify under the root verifying key");
    const std::vector<uint32_t>& pv = root.pvs.at(2);
    if (pv.size() != 8 + 9 + 9 + 8 && pv.size() != 8 + 9 + 9 + 8 + 16 && pv.size() != 8 + 9 + 9 + 8 + 16 + 8) return fail("the root statement does not have the (pc, memory root) layout");
    // An aggregation key pins the tree beneath the root: its leaves are proofs of the key's leaf circuit (which hard-wires the app's
    // verifying key: programs, heights, the committed program), its nodes proofs of the key's own circuit, the statement is about this
    // app.  (A per-depth key of round 3 pins the same through its preprocessed commitments and carries no pair.)
    if (pv.size() >= 50) {
        std::string w;
        if ((pv.size() == 58) != root_vk.join) return fail("the root statement does not match the kind of the root verifying key");
        if (!root_vk.root_statement_matches(pv, &w)) {
            if (why) *why = w;
            return false;
        }
    }
    if (pv[8] != entry_pc) return fail("the run does not start at the guest's entry point");
    if (!std::equal(image_root.begin(), image_root.end(), pv.begin() + 9)) return fail("the run does not start on the guest's memory image");
    if (pv[17] != 0) return fail("the run does not end with exit code 0");
    if (!check_public_values(public_values, pv_openings, &pv[18])) return fail("the public values do not open in the final memory root");
    return true;
}

}  // namespace zkhip_vm

// ---- the reference's Prover API over the one-statement flow ------------------------------------------------------------------------------
namespace scroll_zkvm_hip {

// crates/prover/src/prover/mod.rs:83-413 for guests: ProverConfig::path_app_exe is the guest ELF (the reference: the app's vmexe),
// path_app_config its openvm.toml.  gen_proof_universal(task) = task.build_guest_input() -> execute -> segments -> aggregation ->
// ONE StarkProof, self-verified.  StarkProof.user_pvs_proof = [the root node's statement (34 words) | the guest's 32 public-value
// bytes | their Merkle openings in the final memory root]; baseline = the root circuit's heights: `UniversalVerifier` over the
// root verifying key accepts it as it is, verify_guest_stark checks the whole statement.
class UniversalProver {
public:
    std::string prover_name;
    ProverConfig config;

    static UniversalProver setup(const ProverConfig& cfg, const char* name = nullptr, int device = 0, unsigned log_frame = 20) {
        UniversalProver p;
        p.config = cfg, p.prover_name = name ? name : "universal", p.device_ = device;
        std::ifstream f(cfg.path_app_exe, std::ios::binary);
        if (!f) throw Error(Error::Setup, "failed to read or deserialize " + cfg.path_app_exe + ": cannot open");
        const std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        try {
            p.exe_ = zkhip_vm::parse_exe(raw);
        } catch (const zkhip_vm::Error& e) {
            throw Error(Error::Setup, "failed to read or deserialize " + cfg.path_app_exe + ": " + e.what());
        }
        p.params_ = read_app_config(cfg.path_app_config);
        p.log_frame_ = log_frame;
        p.caps_ = zkhip_vm::config_caps(cfg.path_app_config, log_frame, zkhip_vm::vm_log2_ceil(p.exe_.program.size()));
        return p;
    }
    void reset() {   // mod.rs:106-108: releases the device-resident keys, circuits and workspaces
        agg_.reset(), seg_.clear();
    }
    // what the first task would otherwise build on its way: the segment keys, the aggregation circuits and their keys (the reference's
    // get_sdk / keygen behind its OnceLock: mod.rs:78,115-126) -- setup, not proving
    void warm_up() {
        try {
            ensure();
            (void)agg_->node_vk(0);
        } catch (const zkhip_vm::Error& e) {
            throw Error(Error::Setup, e.what());
        }
    }
    // mod.rs:312-338
    uint64_t execute_and_check(const ProvingTask& task) const {
        try {
            return zkhip_vm::execute_guest(exe_, task.build_guest_input(), 0, nullptr, 0, caps_.moduli, caps_.curves, caps_.fp2_moduli, caps_.ext).total_cycle;
        } catch (const zkhip_vm::Error& e) {
            throw Error(Error::GenProof, e.what());
        }
    }
    StarkProof gen_proof_universal(const ProvingTask& task, bool with_snark = false) {
        if (with_snark) throw Error(Error::GenProof, "the SNARK (EVM) wrap is outside the HIP backend's path");
        return gen_proof_stark(task);
    }
    // mod.rs:342-413 `gen_proof_stark(stdin, def_inputs)`: with deferral inputs the guest's root is joined with the proof of the deferral
    // node over them (crates/integration/src/lib.rs:556-571 `prove_task_with_deferral`)
    StarkProof gen_proof_stark(const ProvingTask& task, const std::vector<DeferralInput>& def_inputs = {}) {
        if (!def_inputs.empty() && !deferral_) throw Error(Error::GenProof, "deferral inputs given, but deferral is not enabled on this prover (enable_deferral)");
        try {
            ensure();
            const zkhip_vm::StdIn in = task.build_guest_input();
            std::vector<zkhip_vm::ShapedSegmentProver*> lanes;
            for (auto& l : seg_) lanes.push_back(l.get());
            last_ = zkhip_vm::prove_guest_with(lanes, *agg_, exe_, in, caps_, false, config.flow.verify_segments, !config.flow.balanced_tree, config.flow.trace_tree,
                                               config.flow.wide_in_flight, config.flow.retry_segments, config.flow.exec_threads_or_auto(), config.flow.fail_segment_once);
        } catch (const zkhip_vm::Error& e) {
            throw Error(Error::GenProof, std::string("failed to generate proof: ") + e.what());
        }
        if (deferral_) {
            // a prover with deferral enabled returns proofs under the JOIN key, with or without claims in this task's run: the guest's root and the
            // deferral node's proof (over the children the claims are about) under one more circuit
            if (def_inputs.empty()) throw Error(Error::GenProof, "this prover has deferral enabled: a task needs its deferral inputs (compute_deferral_data)");
            const auto t0 = std::chrono::steady_clock::now();
            const VerifyingKey own = agg_->root_vk();
            const ChildProof dproof = deferral_->prove_deferral(def_inputs);
            last_deferral_ = dproof;
            last_.root = deferral_->prove_join(own, last_.root, dproof);
            last_.root_vk = deferral_->join_vk(own);
            last_.aggregation_mills += (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        }
        StarkProof sp = encode(last_);
        std::string why;
        if (!verify_guest_stark(sp, last_.root_vk, exe_, &why)) throw Error(Error::VerifyProof, "failed to verify proof: " + why);   // mod.rs:407-411
        return sp;
    }
    // mod.rs:200-282 `enable_deferral(child_prover)`: from now on this prover's guest may state claims about proofs of the CHILD app (its
    // aggregation key is all that is needed: mod.rs:213 `child_prover.load_agg_vk()`), and every proof it returns is a join.
    // `child_region_index` (a child app that itself defers -- its key is a join key: this prover's guest is a bundle over batches):
    // zkhip_vm::deferral_region_index(child exe)
    // `max_nodes` > 1: up to max_nodes deferral nodes of max_children each per task, folded (DeferralProver::setup)
    void enable_deferral(const VerifyingKey& child_agg_key, size_t max_children = 4, uint32_t child_region_index = 0, size_t max_nodes = 1) {
        if (!zkhip_vm::has_deferral_region(exe_)) throw Error(Error::Setup, "enable_deferral: the guest's memory has no room for the deferral region");
        reset();   // (mod.rs:265 `self.reset()`: the keys are rebuilt with deferral)
        deferral_ = DeferralProver::setup(child_agg_key, params_, device_, max_children, child_region_index, max_nodes);
    }
    void enable_deferral(UniversalProver& child_prover, size_t max_children = 4, size_t max_nodes = 1) {
        enable_deferral(child_prover.get_agg_vk(), max_children, child_prover.deferral_enabled() ? zkhip_vm::deferral_region_index(child_prover.exe()) : 0u, max_nodes);
    }
    bool deferral_enabled() const { return (bool)deferral_; }
    // crates/integration/src/lib.rs:461-514 `compute_deferral_data(child_prover, cached_commit, proofs)`
    DeferralProver::Data compute_deferral_data(const std::vector<const StarkProof*>& proofs) const {
        if (!deferral_) throw Error(Error::GenProof, "compute_deferral_data: deferral is not enabled on this prover");
        return deferral_->compute_deferral_data(proofs);
    }
    // mod.rs:147-170 `get_agg_vk`: the key every proof of this prover verifies under (one aggregation key; with deferral: the join's)
    const VerifyingKey& get_agg_vk() {
        ensure();
        agg_->verify_lazy_shapes();   // (the key leaves this process: nothing in it may rest on an unchecked cache file)
        const VerifyingKey& own = agg_->root_vk();
        return deferral_ ? deferral_->join_vk(own) : own;
    }
    // what a PARENT guest holds about this app (crates/types/circuit/src/lib.rs ProgramCommitment)
    ProgramCommitment program_commitment() {
        ensure();
        agg_->verify_lazy_shapes();
        return ProgramCommitment::of(agg_->root_vk(), exe_.entry ? exe_.entry : exe_.pc_base, zkhip_vm::guest_image_root(exe_));
    }
    const zkhip_vm::GuestStark& last() const { return last_; }
    const ChildProof& last_deferral_proof() const { return last_deferral_; }   // the deferral node's proof of the last task (what the join verified)
    const DeferralProver* deferral() const { return deferral_.get(); }
    const zkhip_vm::Exe& exe() const { return exe_; }

    static StarkProof encode(const zkhip_vm::GuestStark& g) {
        StarkProof sp = AggregationProver::to_stark_proof(g.root, g.root_vk);   // proof, the node's public values, heights
        sp.user_pvs_proof.insert(sp.user_pvs_proof.end(), g.exec.public_values.begin(), g.exec.public_values.end());
        const uint8_t* ob = reinterpret_cast<const uint8_t*>(g.pv_openings.data());
        sp.user_pvs_proof.insert(sp.user_pvs_proof.end(), ob, ob + 4 * g.pv_openings.size());
        if (g.root_vk.join) {   // crates/types/src/proof.rs `deferral_merkle_proofs`: the guest's claims, opened in the final memory root
            const uint8_t* db = reinterpret_cast<const uint8_t*>(g.deferral_opening.data());
            sp.deferral_merkle_proofs.assign(db, db + 4 * g.deferral_opening.size());
        }
        sp.stat.total_cycles = g.exec.total_cycle, sp.stat.execution_time_mills = g.execution_mills;
        sp.stat.proving_time_mills = g.segment_proving_mills + g.aggregation_mills;
        return sp;
    }
    // crates/verifier/src/verifier.rs:38-85 for this flow: the root proof under the root verifying key and the statement about THIS guest
    static bool verify_guest_stark(const StarkProof& sp, const VerifyingKey& root_vk, const zkhip_vm::Exe& exe, std::string* why = nullptr) {
        const size_t n_stmt = root_vk.airs.empty() ? 0 : root_vk.airs.back().n_pvs, n_open = 2 * 8 * zkhip::vmc::LEAF_LEVEL;
        if (sp.user_pvs_proof.size() != 4 * n_stmt + zkhip_vm::NUM_PUBLIC_VALUE_BYTES + 4 * n_open) {
            if (why) *why = "user_pvs_proof has the wrong size";
            return false;
        }
        ChildProof root;
        root.proof = sp.proof;
        root.pvs.resize(3);
        root.pvs[2].resize(n_stmt);
        memcpy(root.pvs[2].data(), sp.user_pvs_proof.data(), 4 * n_stmt);
        const std::vector<uint8_t> pv(sp.user_pvs_proof.begin() + 4 * n_stmt, sp.user_pvs_proof.begin() + 4 * n_stmt + zkhip_vm::NUM_PUBLIC_VALUE_BYTES);
        std::vector<uint32_t> openings(n_open);
        memcpy(openings.data(), sp.user_pvs_proof.data() + 4 * n_stmt + zkhip_vm::NUM_PUBLIC_VALUE_BYTES, 4 * n_open);
        if (sp.baseline.size() != root_vk.heights.size()) {
            if (why) *why = "baseline does not match the root verifying key";
            return false;
        }
        for (size_t a = 0; a < root_vk.heights.size(); a++)
            if (sp.baseline[a] != root_vk.heights[a]) {
                if (why) *why = "baseline does not match the root verifying key";
                return false;
            }
        try {
            if (!zkhip_vm::verify_guest_proof(root_vk, root, exe.entry ? exe.entry : exe.pc_base, zkhip_vm::guest_image_root(exe), pv, openings, why)) return false;
            if (!root_vk.join) return true;
            // DEFERRAL: the claims the guest made (opened in its final memory root) chain to the accumulator the join took from the
            // deferral node -- every claim is backed by a child proof the deferral node verified, in this order, and there is no other
            auto fail = [&](const char* m) {
                if (why) *why = m;
                return false;
            };
            if (sp.deferral_merkle_proofs.size() % 4) return fail("deferral_merkle_proofs is not a word array");
            std::vector<uint32_t> opening(sp.deferral_merkle_proofs.size() / 4);
            if (!opening.empty()) memcpy(opening.data(), sp.deferral_merkle_proofs.data(), sp.deferral_merkle_proofs.size());
            std::vector<std::array<uint32_t, 32>> claims;
            if (!zkhip_vm::check_deferral_region(exe, opening, &root.pvs[2][18], &claims)) return fail("the deferral region does not open in the final memory root");
            if (claims.empty()) return fail("a proof under a join key without claims");
            Digest8 acc{};
            for (const auto& c : claims)
                if (!chain_claim(acc, c.data())) return fail("a claim's commitment is not a field element");
            if (!std::equal(acc.begin(), acc.end(), root.pvs[2].begin() + 50)) return fail("the guest's claims are not the ones the deferral node verified");
            return true;
        } catch (const zkhip_vm::Error& e) {
            if (why) *why = e.what();
            return false;
        }
    }

private:
    UniversalProver() = default;
    zkhip_vm::Exe exe_;
    zkhip_params params_{};
    zkhip_vm::SegmentCaps caps_;
    int device_ = 0;
    std::vector<std::unique_ptr<zkhip_vm::ShapedSegmentProver>> seg_;   // lanes: segment provers in flight on the device (one key per shape)
    unsigned log_frame_ = 20;
    std::unique_ptr<AggregationProver> agg_;
    std::unique_ptr<DeferralProver> deferral_;
    zkhip_vm::GuestStark last_;
    ChildProof last_deferral_;
    void ensure() {
        if (!seg_.empty()) return;
        const zkhip_vm::SegmentShapes shapes = config.flow.one_shape ? zkhip_vm::SegmentShapes::only(caps_)
                                                                         : zkhip_vm::SegmentShapes::of(caps_, log_frame_, zkhip_vm::vm_log2_ceil(exe_.program.size()), config.flow.lean_shape);
        const std::vector<int> devs = config.flow.devices.empty() ? std::vector<int>{device_} : config.flow.devices;
        for (int d : devs)
            for (unsigned l = 0; l < std::max(1u, config.flow.lanes); l++) seg_.emplace_back(new zkhip_vm::ShapedSegmentProver(params_, exe_, shapes, d, seg_.empty()));
        std::vector<VerifyingKey> shape_vks;
        for (size_t sh = 0; sh < seg_[0]->n_shapes(); sh++) shape_vks.push_back(seg_[0]->vk(sh));
        AggregationTreeConfig tree_cfg;
        tree_cfg.num_children_internal = config.flow.internal_arity;
        agg_.reset(new AggregationProver(AggregationProver::setup_shapes(shape_vks, zkhip_vm::segment_statement(), tree_cfg, device_)));
        {
            std::vector<int> slots;   // config.flow.agg_slots node pipelines per device
            for (unsigned r = 0; r < std::max(1u, config.flow.agg_slots); r++) slots.insert(slots.end(), devs.begin(), devs.end());
            agg_->set_devices(slots);
        }
        agg_->set_shape_policies(zkhip_vm::shape_policies(shape_vks));
    }
};

}  // namespace scroll_zkvm_hip
