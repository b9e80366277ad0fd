// zkhip_vm.hpp -- host-side mirror of the reference's guest execution step:
//   crates/prover/src/utils/vm.rs:13-48  execute_guest(sdk, exe, inputs) -> ExecutionResult { total_cycle, public_values }
// a metered run first (instruction count + cost, bounded by a maximum cost), the plain executor as the fall-back when the
// metered one gives up (total_cycle = u64::MAX then, as in the reference), and the same sanity check on the public values
// ("public_values are all 0s ...").  Called by Prover::gen_proof_stark before proving (crates/prover/src/prover/mod.rs:318).
//
// What runs here is an RV32IM interpreter (the reference's guests are rv32im ELFs transpiled for OpenVM; its executor and
// transpiler live in un-vendored crates, so the instruction set is the public RISC-V one, not OpenVM's opcode numbering).
// Besides the result it produces what the step AFTER execution needs on this backend -- the execution -> trace hand-off of
// SURVEY.md 8(f) f3 -- in the layout the device trace generators take (include/zkhip.h):
//   * per-chip EXECUTION RECORDS: executed instruction indices (zkhip_program_freq_tracegen / zkhip_exec_frame_tracegen), ALU,
//     less-than, multiplication and shift operands (zkhip_rv32_{alu,lt,mul,shift}_tracegen); instruction classes without a device
//     generator yet (branches, loads / stores, jumps, division) are counted, so that the cost model sees them;
//   * the MEMORY LOG over 16-bit cells of the register file and the read-write memory (zkhip_memory_access_tracegen) and the
//     boundary records of the touched cells (zkhip_memory_boundary_tracegen): the offline memory-checking argument;
//   * CONTINUATION: execute_segments cuts the run into independently provable segments chained by their memory boundaries.
// load_elf / parse_exe read an ELF32 RISC-V image (or the flat test format).
//
// Environment calls (a7): 93 = exit(a0) (non-zero = guest failure), 1 = reveal the word a0 as public-value bytes
// [4 a1, 4 a1 + 4), 2 = read the next word of the input stream into a0.  Header-only; no GPU code.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstring>
#include <deque>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <vector>

#include "zkhip_ecc.hpp"
#include "zkhip_fp2.hpp"
#include "zkhip_int256.hpp"
#include "zkhip_keccak.hpp"
#include "zkhip_modular.hpp"
#include "zkhip_native.hpp"
#include "zkhip_pairing.hpp"
#include "zkhip_sha256.hpp"

namespace zkhip_vm {

struct Exe {
    uint32_t pc_base = 0x00200000u;       // address of program[0]
    std::vector<uint32_t> program;        // RV32IM instruction words
    uint32_t data_base = 0x00400000u;
    std::vector<uint8_t> data;            // initial read-write memory at data_base
    uint32_t memory_bytes = 1u << 20;     // size of the read-write region (stack at its top)
    uint32_t entry = 0;                   // first pc; 0 = pc_base
};
struct StdIn {
    std::vector<uint8_t> bytes;           // the hint stream, consumed one little-endian word per read
};
struct ExecutionResult {                  // crates/prover/src/utils/vm.rs:3-8
    uint64_t total_cycle = 0;
    std::vector<uint8_t> public_values;
};
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Per-chip records, in the argument layout of the device trace generators (plain integers).
// a record array of the interpreter's inner loop: std::vector<uint32_t> with an append that skips the capacity check (ExecRecords::reserved: the
// segmenting executor reserves a whole frame's rows before the segment starts; four appends per instruction were a fifth of its time)
struct U32Vec : std::vector<uint32_t> {
    using std::vector<uint32_t>::vector;
#if defined(__GLIBCXX__)
    void put_unchecked(uint32_t x) {
        assert(this->_M_impl._M_finish != this->_M_impl._M_end_of_storage && "U32Vec::put_unchecked beyond the reserved rows");
        *this->_M_impl._M_finish = x;
        ++this->_M_impl._M_finish;
    }
#else
    void put_unchecked(uint32_t x) { this->push_back(x); }
#endif
};
struct ExecRecords {
    U32Vec pc_index;                   // every executed instruction: its row in the program
    U32Vec alu_op, alu_b, alu_c;       // 0 add 1 sub 2 xor 3 or 4 and (register and immediate forms)
    U32Vec lt_op, lt_b, lt_c;          // 0 slt 1 sltu
    U32Vec mul_b, mul_c;               // mul (low word)
    U32Vec mulh_op, mulh_b, mulh_c;    // 0 mulh 1 mulhsu 2 mulhu
    U32Vec div_op, div_b, div_c;       // 0 div 1 divu 2 rem 3 remu; b = dividend, c = divisor
    U32Vec shift_op, shift_b, shift_c; // 0 sll 1 srl 2 sra; c = the shift operand (register value or shamt)
    U32Vec beq_op, beq_a, beq_b, beq_imm;  // 0 beq 1 bne; imm = the offset as a canonical BabyBear element
    U32Vec blt_op, blt_a, blt_b, blt_imm;  // 0 blt 1 bltu 2 bge 3 bgeu; imm as above
    U32Vec jal_op, jal_pc, jal_imm;        // 0 jal (imm = the offset as a field element) 1 lui (imm = the 20-bit immediate)
    U32Vec auipc_pc, auipc_imm;            // imm = the 20-bit immediate
    U32Vec jalr_pc, jalr_rs1, jalr_imm;    // imm = the raw 12-bit immediate
    // loads and stores: case 0..19 (LW, LHU@0 LHU@2, LBU@0..3, SW, SH@0 SH@2, SB@0..3, LH@0 LH@2, LB@0..3; @k = byte offset inside the
    // word, words aligned from data_base), read = the aligned memory word (loads) / the stored register (stores), prev = what the
    // destination register (loads) / the aligned memory word (stores) held before
    U32Vec ls_case, ls_read, ls_prev;
    uint64_t n_shift = 0, n_branch = 0, n_jump = 0, n_load_store = 0, n_mulh = 0, n_divrem = 0, n_lui_auipc = 0, n_ecall = 0;
    // != 0: every U32Vec above has room for the rows of a whole segment (set by the segmenting executor for the duration of run_segment).  A COPY
    // of the records starts at 0: the flag speaks about capacities, and a copied vector's capacity is its size.
    struct Reserved {
        uint64_t v = 0;
        Reserved() = default;
        Reserved(const Reserved&) {}
        Reserved(Reserved&& o) noexcept : v(o.v) {}
        Reserved& operator=(const Reserved&) { v = 0; return *this; }
        Reserved& operator=(Reserved&& o) noexcept { v = o.v; return *this; }
        Reserved& operator=(uint64_t x) { v = x; return *this; }
        explicit operator bool() const { return v != 0; }
    } reserved;
    // Memory log for the offline memory-checking argument (OpenVM's memory bus): state is kept in 16-bit CELLS -- address space
    // 1 = registers (cell 2 i + half of x_i), 2 = read-write memory (cell = halfword index from data_base) -- so that a cell's
    // value is a field element.  One entry per cell access, in execution order, timestamps 1, 2, 3, ...:
    //   (as, ptr, prev_data, prev_ts) is what the access consumes, (as, ptr, data, ts) what it leaves; reads have data == prev_data.
    std::vector<uint32_t> acc_as, acc_ptr, acc_prev_data, acc_prev_ts, acc_data, acc_ts, acc_is_read;
    // and per touched cell, once the run is over: initial value (before the first access), final value, timestamp of the last
    // access -- the records of the memory boundary chip (zkhip_memory_boundary_tracegen)
    std::vector<uint32_t> bnd_as, bnd_ptr, bnd_initial, bnd_final, bnd_final_ts;

    // empties every array but keeps its storage: a recycled ExecRecords spares the next segment ~25 MB of fresh pages
    void clear() {
        for (std::vector<uint32_t>* v : std::initializer_list<std::vector<uint32_t>*>
             {&pc_index, &alu_op, &alu_b, &alu_c, &lt_op, &lt_b, &lt_c, &mul_b, &mul_c, &mulh_op, &mulh_b, &mulh_c, &div_op, &div_b, &div_c, &shift_op,
              &shift_b, &shift_c, &beq_op, &beq_a, &beq_b, &beq_imm, &blt_op, &blt_a, &blt_b, &blt_imm, &jal_op, &jal_pc, &jal_imm, &auipc_pc, &auipc_imm,
              &jalr_pc, &jalr_rs1, &jalr_imm, &ls_case, &ls_read, &ls_prev, &acc_as, &acc_ptr, &acc_prev_data, &acc_prev_ts, &acc_data, &acc_ts,
              &acc_is_read, &bnd_as, &bnd_ptr, &bnd_initial, &bnd_final, &bnd_final_ts})
            v->clear();
        n_shift = n_branch = n_jump = n_load_store = n_mulh = n_divrem = n_lui_auipc = n_ecall = 0;
        reserved = 0;   // (whoever reserves says so again: the segmenting executor, per segment)
    }
};
static_assert(sizeof(ExecRecords) == 49 * sizeof(std::vector<uint32_t>) + 9 * sizeof(uint64_t), "ExecRecords::clear() must list every array");

constexpr unsigned NUM_PUBLIC_VALUE_BYTES = 32;
// trace widths used by the cost model (cells per record) = the widths of the device trace generators (include/zkhip.h
// ZKHIP_*_WIDTH); program chip rows are counted once per program instruction, memory rows once per cell access
constexpr uint64_t W_ALU = 18, W_LT = 18, W_MUL = 13, W_MULH = 21, W_DIVREM = 41, W_FRAME = 10, W_SHIFT = 32, W_BRANCH_EQ = 17, W_BRANCH_LT = 23,
                   W_JAL_LUI = 9, W_AUIPC = 14, W_JALR = 20, W_LOAD_STORE = 33, W_MEM_ACCESS = 10;

inline uint64_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}
// metered cost = main-trace cells of the segment (rows padded to powers of two), the quantity OpenVM's metered executor bounds
inline uint64_t trace_cells(const ExecRecords& r, size_t program_len) {
    auto cells = [](uint64_t n, uint64_t w) { return n ? next_pow2(n) * w : 0; };
    return cells(r.alu_op.size(), W_ALU) + cells(r.lt_op.size(), W_LT) + cells(r.mul_b.size(), W_MUL) + cells(r.mulh_op.size(), W_MULH) +
           cells(r.div_op.size(), W_DIVREM) + cells(r.pc_index.size(), W_FRAME) + cells(program_len, 10) + cells(r.shift_op.size(), W_SHIFT) +
           cells(r.beq_op.size(), W_BRANCH_EQ) + cells(r.blt_op.size(), W_BRANCH_LT) + cells(r.jal_op.size(), W_JAL_LUI) +
           cells(r.auipc_pc.size(), W_AUIPC) + cells(r.jalr_pc.size(), W_JALR) + cells(r.ls_case.size(), W_LOAD_STORE) +
           cells(r.acc_ts.size(), W_MEM_ACCESS);
}

// an append of the interpreter (see U32Vec)
static inline void put(ExecRecords* rec, U32Vec& v, uint32_t x) {
    if (rec->reserved) v.put_unchecked(x);
    else v.push_back(x);
}
class Machine {
public:
    Machine(const Exe& exe, const StdIn& in) : exe_(exe), in_(in), mem_(exe.memory_bytes, 0), pv_(NUM_PUBLIC_VALUE_BYTES, 0) {
        pre_.resize(exe.program.size());
        for (size_t i = 0; i < pre_.size(); i++) {
            const uint32_t w = exe.program[i];
            Pre& q = pre_[i];
            // immediates: assembled as unsigned fields, sign-extended once (shifting a negative value left is undefined before C++20)
            auto sext = [](uint32_t v, unsigned bits) { const uint32_t m = 1u << (bits - 1); return (int32_t)((v ^ m) - m); };
            q.w = w, q.op = w & 0x7f, q.rd = (w >> 7) & 31, q.f3 = (w >> 12) & 7, q.rs1 = (w >> 15) & 31, q.rs2 = (w >> 20) & 31, q.f7 = (uint8_t)(w >> 25);
            q.uses_rs1 = q.op == 0x67 || q.op == 0x63 || q.op == 0x03 || q.op == 0x23 || q.op == 0x13 || q.op == 0x33;
            q.uses_rs2 = q.op == 0x63 || q.op == 0x23 || q.op == 0x33;
            q.imm_i = sext(w >> 20, 12);
            q.imm_s = sext(((w >> 25) << 5) | ((w >> 7) & 31u), 12);
            q.imm_b = sext(((w >> 31) << 12) | (((w >> 7) & 1u) << 11) | (((w >> 25) & 63u) << 5) | (((w >> 8) & 15u) << 1), 13);
            q.imm_j = sext(((w >> 31) << 20) | (((w >> 12) & 255u) << 12) | (((w >> 20) & 1u) << 11) | (((w >> 21) & 1023u) << 1), 21);
        }
        if (exe.data.size() > mem_.size()) throw Error("initial data larger than memory");
        if (!exe.data.empty()) memcpy(mem_.data(), exe.data.data(), exe.data.size());
        memset(x_, 0, sizeof x_);
        x_[2] = exe.data_base + exe.memory_bytes;  // sp
        pc_ = exe.entry ? exe.entry : exe.pc_base;
    }
    // Runs until exit.  max_instr / max_cost == 0: unbounded.  Returns false if a bound was hit (state stays valid).
    bool run(uint64_t max_instr, uint64_t max_cost, ExecRecords* rec) {
        while (!done_) {
            if (max_instr && instret_ >= max_instr) return false;
            if (max_cost && rec && (instret_ & 1023u) == 0 && trace_cells(*rec, exe_.program.size()) > max_cost) return false;
            step(rec);
        }
        close_segment(rec);
        return true;
    }
    // Continuation: executes at most `max_instr` further instructions into `rec` and closes the SEGMENT -- boundary records of the
    // cells it touched (initial = their value when the segment began), memory log restarted at timestamp 1.  Consecutive
    // segments are independent proofs; what chains them is that a cell's final value in one segment is its
    // initial value in the next one that touches it.  Returns true once the guest has exited.
    bool run_segment(uint64_t max_instr, ExecRecords* rec) {
        if (rec && max_instr && log_memory_) {   // the memory log grows by about five entries per instruction; a hint, capped at 2^20 instructions
            const uint64_t hint = std::min<uint64_t>(max_instr, (uint64_t)1 << 20);
            rec->pc_index.reserve(hint);
            for (auto* v : {&rec->acc_as, &rec->acc_ptr, &rec->acc_prev_data, &rec->acc_prev_ts, &rec->acc_data, &rec->acc_ts, &rec->acc_is_read})
                v->reserve(5 * hint);
        }
        for (uint64_t k = 0; !done_ && (max_instr == 0 || k < max_instr); k++) step(rec);
        close_segment(rec);
        return done_;
    }
    // one instruction (the segmenting executor of include/zkhip_vm_exec.hpp drives the machine step by step and keeps the memory
    // log of its own: set_memory_log(false) stops this class from writing acc_* / bnd_* records)
    void step_one(ExecRecords* rec) {
        if (!done_) step(rec);
    }
    void set_memory_log(bool on) { log_memory_ = on; }
    void set_moduli(const std::vector<zkhip::modular::U256>& m) { moduli_ = m; }   // the modular extension's moduli (ecall 5)
    void set_curves(const std::vector<zkhip::ecc::Curve>& c) { curves_ = c; }       // the ecc extension's curves (ecall 7)
    void set_fp2_moduli(const std::vector<zkhip::modular::U256>& m) { fp2_moduli_ = m; }   // the fp2 extension's fields (ecall 8)
    // `[app_vm_config.native]` (ecalls 9, 10), `[app_vm_config.castf]` (ecall 11), `[app_vm_config.pairing]` (phantom kind 2)
    void set_native(const zkhip::native::Enabled& e) { native_ = e.native, castf_ = e.castf, pairing_ = e.pairing; }
    const zkhip::modular::U256& last_slope() const { return last_slope_; }          // of the last ecc call (the chip's record carries it)
    // ---- snapshots (the parallel executor of include/zkhip_vm_exec.hpp: a metered first pass cuts the run, record passes replay the
    // segments on other threads).  A State is everything but the memory: that travels as the 4 KiB PAGES written since the previous
    // snapshot (track_dirty + take_dirty_pages on the machine that runs ahead, apply_pages on the ones that follow). ----
    static constexpr uint32_t PAGE_BYTES = 4096;
    struct Page {
        uint32_t index;                       // byte offset / PAGE_BYTES from data_base
        std::vector<uint8_t> bytes;           // PAGE_BYTES (less for the last page of the region)
    };
    struct State {
        uint32_t x[32] = {}, pc = 0, exit_code = 0, last_val = 0;
        bool done = false;
        uint64_t instret = 0;
        size_t in_pos = 0;
        std::deque<uint32_t> hints;
        std::vector<uint8_t> pv;
        zkhip::modular::U256 last_slope{};
    };
    State save_state() const {
        State st;
        memcpy(st.x, x_, sizeof x_);
        st.pc = pc_, st.exit_code = exit_code_, st.last_val = last_val_, st.done = done_, st.instret = instret_, st.in_pos = in_pos_;
        st.hints = hints_, st.pv = pv_, st.last_slope = last_slope_;
        return st;
    }
    void restore_state(const State& st) {
        memcpy(x_, st.x, sizeof x_);
        pc_ = st.pc, exit_code_ = st.exit_code, last_val_ = st.last_val, done_ = st.done, instret_ = st.instret, in_pos_ = st.in_pos;
        hints_ = st.hints, pv_ = st.pv, last_slope_ = st.last_slope;
    }
    void track_dirty(bool on) {
        dirty_.assign(on ? (mem_.size() + PAGE_BYTES - 1) / PAGE_BYTES : 0, 0);
        dirty_list_.clear();
    }
    // the pages written since the last call (or since track_dirty(true)), with their contents NOW, in ascending order
    std::vector<Page> take_dirty_pages() {
        std::sort(dirty_list_.begin(), dirty_list_.end());
        std::vector<Page> out;
        out.reserve(dirty_list_.size());
        for (uint32_t pg : dirty_list_) {
            const size_t off = (size_t)pg * PAGE_BYTES, n = std::min<size_t>(PAGE_BYTES, mem_.size() - off);
            out.push_back(Page{pg, std::vector<uint8_t>(mem_.begin() + (std::ptrdiff_t)off, mem_.begin() + (std::ptrdiff_t)(off + n))});
            dirty_[pg] = 0;
        }
        dirty_list_.clear();
        return out;
    }
    void apply_pages(const std::vector<Page>& pages) {
        for (const Page& pg : pages) {
            const size_t off = (size_t)pg.index * PAGE_BYTES;
            if (off + pg.bytes.size() > mem_.size()) throw Error("internal: a memory page beyond the region");
            memcpy(mem_.data() + off, pg.bytes.data(), pg.bytes.size());
        }
    }
    bool done() const { return done_; }
    uint32_t pc() const { return pc_; }
    uint32_t last_result() const { return last_val_; }   // the value the last instruction computed for rd (also when rd = x0)
    uint32_t peek_word(uint32_t addr) const { return word_around(addr); }
    const Exe& exe() const { return exe_; }
    uint64_t instret() const { return instret_; }
    const std::vector<uint8_t>& public_values() const { return pv_; }
    uint32_t exit_code() const { return exit_code_; }
    uint32_t reg(unsigned i) const { return x_[i & 31]; }

private:
    const Exe& exe_;
    const StdIn& in_;
    struct Pre {   // a program word, decoded
        uint32_t w;
        int32_t imm_i, imm_s, imm_b, imm_j;
        uint8_t op, rd, f3, rs1, rs2, f7;
        bool uses_rs1, uses_rs2;
    };
    std::vector<Pre> pre_;
    std::vector<uint8_t> mem_, pv_;
    uint32_t x_[32], pc_ = 0, exit_code_ = 0, last_val_ = 0;
    bool log_memory_ = true;
    std::vector<uint8_t> dirty_;          // per page: written since the last take_dirty_pages (empty = not tracked)
    std::vector<uint32_t> dirty_list_;
    void mark_dirty(size_t off, unsigned n) {
        if (dirty_.empty()) return;
        for (size_t pg = off / PAGE_BYTES; pg <= (off + n - 1) / PAGE_BYTES; pg++)
            if (!dirty_[pg]) dirty_[pg] = 1, dirty_list_.push_back((uint32_t)pg);
    }
    std::vector<zkhip::modular::U256> moduli_;
    std::vector<zkhip::ecc::Curve> curves_;
    std::vector<zkhip::modular::U256> fp2_moduli_;
    bool native_ = false, castf_ = false, pairing_ = false;
    zkhip::modular::U256 last_slope_{};
    uint64_t instret_ = 0;
    size_t in_pos_ = 0;
    std::deque<uint32_t> hints_;   // the hint stream: words phantom instructions computed, read by the input ecall before the input stream
    bool done_ = false;
    // Per-cell state of the memory log for the CURRENT segment, kept in flat tables (register cells, then memory cells) instead
    // of a hash map: a slot belongs to the segment whose number its `seg` field holds, so closing a segment costs nothing
    // per untouched cell; `touched_` lists the slots of this segment for the boundary records.
    struct Cell {
        uint32_t initial, data, ts, seg;
    };
    std::vector<Cell> reg_cells_ = std::vector<Cell>(64, Cell{0, 0, 0, 0});
    std::vector<Cell> mem_cells_;          // memory_bytes / 2 slots, allocated at the first logged access
    std::vector<uint64_t> touched_;        // (address space << 32) | pointer
    uint32_t clock_ = 0, segment_ = 1;

    void close_segment(ExecRecords* rec) {
        if (rec && log_memory_) {  // boundary records, sorted by (address space, pointer)
            std::sort(touched_.begin(), touched_.end());
            for (uint64_t k : touched_) {
                const Cell& c = (k >> 32) == 1 ? reg_cells_[(uint32_t)k] : mem_cells_[(uint32_t)k];
                rec->bnd_as.push_back((uint32_t)(k >> 32)), rec->bnd_ptr.push_back((uint32_t)k);
                rec->bnd_initial.push_back(c.initial), rec->bnd_final.push_back(c.data), rec->bnd_final_ts.push_back(c.ts);
            }
        }
        touched_.clear();
        clock_ = 0;
        segment_++;
    }
    // one access of a 16-bit cell: `write` replaces its value by `value`
    void touch(ExecRecords* rec, uint32_t as, uint32_t ptr, uint32_t current, bool write, uint32_t value) {
        if (!rec || !log_memory_) return;
        if (as == 2 && mem_cells_.empty()) mem_cells_.assign(mem_.size() / 2 + 1, Cell{0, 0, 0, 0});
        Cell& c = as == 1 ? reg_cells_[ptr] : mem_cells_[ptr];
        if (c.seg != segment_) c = Cell{current, current, 0, segment_}, touched_.push_back(((uint64_t)as << 32) | ptr);
        if (++clock_ == 0) throw Error("memory clock overflow");
        rec->acc_as.push_back(as), rec->acc_ptr.push_back(ptr), rec->acc_prev_data.push_back(c.data), rec->acc_prev_ts.push_back(c.ts);
        c.data = write ? value : c.data, c.ts = clock_;
        rec->acc_data.push_back(c.data), rec->acc_ts.push_back(clock_), rec->acc_is_read.push_back(write ? 0u : 1u);
    }
    // (the segmenting executor keeps the memory log of its own -- log_memory_ off: nothing to record per cell, and this is its inner loop)
    uint32_t reg_read(ExecRecords* rec, unsigned i) {
        if (rec && log_memory_) touch(rec, 1, 2 * i, x_[i] & 0xffffu, false, 0), touch(rec, 1, 2 * i + 1, x_[i] >> 16, false, 0);
        return x_[i];
    }
    void reg_write(ExecRecords* rec, unsigned i, uint32_t v) {
        if (i == 0) v = 0;  // x0 is a cell that only ever holds zero
        if (rec && log_memory_) touch(rec, 1, 2 * i, x_[i] & 0xffffu, true, v & 0xffffu), touch(rec, 1, 2 * i + 1, x_[i] >> 16, true, v >> 16);
        x_[i] = v;
    }
    // Phantom sub-executors (OpenVM's algebra extension: ModularPhantom::{HintNonQr, HintSqrt}; pairing extension: HintFinalExp).  The
    // circuit proves nothing about them -- the guest checks what it reads (a square root by one multiplication and an equality test).
    //   kind 0, buffer [modulus index]:            pushes a quadratic non-residue z of that modulus (nw words)
    //   kind 1, buffer [modulus index | x (nw)]:   pushes [s | r (nw)]: s = 1 and r^2 = x if x is a square, else s = 0 and r^2 = x z
    //   kind 2, buffer [pairing curve | f]: the pairing extension's final-exponentiation witness (include/zkhip_pairing.hpp): pushes c and the
    //           scaling factor (OpenVM's SexticExtField<Fp2> layout: 96 words each for curve 0 = Bn254, 144 for curve 1 = Bls12_381) with
    //           c^lambda = f u; needs `[app_vm_config.pairing]`
    void phantom(uint32_t kind, uint32_t ptr) {
        using namespace zkhip::modular;
        if (kind > 2) throw Error("unknown phantom instruction " + std::to_string(kind));
        if (ptr & 3u) throw Error("phantom operand must be word-aligned");
        if (kind == 2) {
            if (!pairing_) throw Error("pairing hint: the app's openvm.toml does not enable the pairing extension");
            uint32_t curve;
            memcpy(&curve, at_ro(ptr, 4), 4);
            if (curve > 1) throw Error("pairing hint: curve " + std::to_string(curve) + " (0 = Bn254, 1 = Bls12_381)");
            const unsigned n_in = curve ? 144u : 96u;
            uint32_t buf[144], out[288];
            memcpy(buf, at_ro(ptr + 4, 4 * n_in), 4 * n_in);
            const char* why = "";
            const bool ok = curve ? zkhip::pairing::final_exp_hint_bls12_381(buf, out, &why) : zkhip::pairing::final_exp_hint_bn254(buf, out, &why);
            if (!ok) throw Error(std::string("pairing hint: ") + why);
            hints_.insert(hints_.end(), out, out + 2 * n_in);
            return;
        }
        uint32_t which;
        memcpy(&which, at_ro(ptr, 4), 4);
        if (which >= moduli_.size()) throw Error("phantom hint: the app's openvm.toml lists no modulus " + std::to_string(which));
        const UInt p = moduli_[which];
        const size_t nw = words_of(p);
        if (!(p.w[0] & 1u)) throw Error("phantom hint: the modulus is even");
        auto mul = [&](const UInt& a, const UInt& b) {
            UInt q, r;
            mulmod(a, b, p, &q, &r);
            return r;
        };
        auto is_zero = [](const UInt& a) {
            uint32_t o = 0;
            for (uint32_t w : a.w) o |= w;
            return o == 0;
        };
        auto equal = [](const UInt& a, const UInt& b) { return memcmp(a.w, b.w, sizeof a.w) == 0; };
        auto shr1 = [](UInt a) {
            for (size_t k = 0; k + 1 < MAX_WORDS; k++) a.w[k] = (a.w[k] >> 1) | (a.w[k + 1] << 31);
            a.w[MAX_WORDS - 1] >>= 1;
            return a;
        };
        UInt one{};
        one.w[0] = 1;
        auto pow = [&](UInt base, UInt e) {
            UInt acc = one;
            while (!is_zero(e)) {
                if (e.w[0] & 1u) acc = mul(acc, base);
                base = mul(base, base), e = shr1(e);
            }
            return acc;
        };
        UInt pm1 = p;
        pm1.w[0] -= 1;   // (p odd)
        const UInt half = shr1(pm1);
        // the smallest non-residue
        UInt z{};
        for (uint32_t c = 2;; c++) {
            z = UInt{};
            z.w[0] = c;
            if (equal(pow(z, half), pm1)) break;
            if (c > 1000) throw Error("phantom hint: no small quadratic non-residue (is the modulus prime?)");
        }
        auto push = [&](const UInt& v) {
            for (size_t k = 0; k < nw; k++) hints_.push_back(v.w[k]);
        };
        if (kind == 0) return push(z);
        UInt x = load_words(at_ro(ptr + 4, (unsigned)(4 * nw)), nw);
        {   // reduce
            UInt q, r;
            mulmod(x, one, p, &q, &r);
            x = r;
        }
        const bool square = is_zero(x) || equal(pow(x, half), one);
        const UInt target = square ? x : mul(x, z);
        // Tonelli - Shanks: p - 1 = q 2^s
        UInt q = pm1;
        unsigned s = 0;
        while (!(q.w[0] & 1u)) q = shr1(q), s++;
        UInt r{};
        if (!is_zero(target)) {
            UInt qp1 = q;   // (q + 1) / 2
            {
                uint64_t c = 1;
                for (size_t k = 0; k < MAX_WORDS; k++) c += qp1.w[k], qp1.w[k] = (uint32_t)c, c >>= 32;
            }
            UInt c = pow(z, q), t = pow(target, q);
            r = pow(target, shr1(qp1));
            unsigned m = s;
            while (!equal(t, one)) {
                unsigned i = 0;
                UInt t2 = t;
                while (!equal(t2, one)) {
                    t2 = mul(t2, t2), i++;
                    if (i >= m) throw Error("phantom hint: the modulus is not prime");
                }
                UInt b = c;
                for (unsigned k = 0; k + i + 1 < m; k++) b = mul(b, b);
                r = mul(r, b), c = mul(b, b), t = mul(t, c), m = i;
            }
        }
        hints_.push_back(square ? 1u : 0u);
        push(r);
    }
    // (at: the pointer of a range that is about to be WRITTEN -- stores, the intrinsic calls' buffers; at_ro: of one that is only read)
    uint8_t* at(uint32_t addr, unsigned n) {
        const uint64_t off = (uint64_t)addr - exe_.data_base;
        if (addr < exe_.data_base || off + n > mem_.size()) throw Error("memory access out of range at pc " + std::to_string(pc_));
        if (n) mark_dirty((size_t)off, n);
        return mem_.data() + off;
    }
    const uint8_t* at_ro(uint32_t addr, unsigned n) const {
        const uint64_t off = (uint64_t)addr - exe_.data_base;
        if (addr < exe_.data_base || off + n > mem_.size()) throw Error("memory access out of range at pc " + std::to_string(pc_));
        return mem_.data() + off;
    }
    uint32_t word_around(uint32_t addr) const {   // the aligned word (from data_base) holding addr, zero beyond the end of memory
        const size_t wo = ((size_t)addr - exe_.data_base) & ~(size_t)3;
        uint32_t v = 0;
        if (wo < mem_.size()) memcpy(&v, mem_.data() + wo, std::min<size_t>(4, mem_.size() - wo));
        return v;
    }
    uint16_t cell_value(uint32_t cidx) const {
        uint16_t v;
        memcpy(&v, mem_.data() + 2 * (size_t)cidx, 2);
        return v;
    }
    uint32_t load(ExecRecords* rec, uint32_t addr, unsigned n) {
        if (addr % n) throw Error("misaligned load");
        uint32_t v = 0;
        memcpy(&v, at_ro(addr, n), n);
        if (rec && log_memory_) {
            const uint32_t first = (addr - exe_.data_base) >> 1, last = (addr - exe_.data_base + n - 1) >> 1;
            for (uint32_t c = first; c <= last; c++) touch(rec, 2, c, cell_value(c), false, 0);
        }
        return v;
    }
    void store(ExecRecords* rec, uint32_t addr, uint32_t v, unsigned n) {
        if (addr % n) throw Error("misaligned store");
        uint8_t* p = at(addr, n);
        if (!(rec && log_memory_)) {
            memcpy(p, &v, n);
            return;
        }
        const uint32_t first = (addr - exe_.data_base) >> 1, last = (addr - exe_.data_base + n - 1) >> 1;
        const uint16_t before[2] = {cell_value(first), cell_value(last)};
        memcpy(p, &v, n);
        for (uint32_t c = first; c <= last; c++) touch(rec, 2, c, before[c - first], true, cell_value(c));  // a byte store rewrites its cell
    }
    void step(ExecRecords* rec) {
        const uint64_t idx64 = ((uint64_t)pc_ - exe_.pc_base) / 4;
        if (pc_ < exe_.pc_base || (pc_ & 3u) || idx64 >= exe_.program.size()) throw Error("pc outside the program: " + std::to_string(pc_));
        // the word's fields and immediates, decoded once per program word (pre_: the interpreter is a third of the segmenting executor's time)
        const Pre& q = pre_[idx64];
        const uint32_t w = q.w;
        const uint32_t op = q.op, rd = q.rd, f3 = q.f3, rs1 = q.rs1, rs2 = q.rs2, f7 = q.f7;
        // registers the instruction reads, in the order rs1, rs2 (logged for the memory argument)
        const uint32_t a = q.uses_rs1 ? reg_read(rec, rs1) : 0, b = q.uses_rs2 ? reg_read(rec, rs2) : 0;
        const int32_t imm_i = q.imm_i, imm_s = q.imm_s, imm_b = q.imm_b, imm_j = q.imm_j;
        uint32_t next = pc_ + 4, val = 0;
        bool wr = false;
        if (rec) put(rec, rec->pc_index, (uint32_t)idx64);
        auto alu = [&](uint32_t o, uint32_t x, uint32_t y) {
            if (rec) put(rec, rec->alu_op, o), put(rec, rec->alu_b, x), put(rec, rec->alu_c, y);
            return o == 0 ? x + y : o == 1 ? x - y : o == 2 ? (x ^ y) : o == 3 ? (x | y) : (x & y);
        };
        auto lt = [&](uint32_t o, uint32_t x, uint32_t y) {
            if (rec) put(rec, rec->lt_op, o), put(rec, rec->lt_b, x), put(rec, rec->lt_c, y);
            return (uint32_t)(o == 0 ? (int32_t)x < (int32_t)y : x < y);
        };
        auto shift = [&](uint32_t kind, uint32_t x, uint32_t s) {
            if (rec) rec->n_shift++, put(rec, rec->shift_op, kind), put(rec, rec->shift_b, x), put(rec, rec->shift_c, s);
            s &= 31;
            return kind == 0 ? x << s : kind == 1 ? x >> s : (uint32_t)((int32_t)x >> s);
        };
        switch (op) {
            case 0x37:  // LUI
                val = w & 0xfffff000u, wr = true;
                if (rec) rec->n_lui_auipc++, put(rec, rec->jal_op, 1), put(rec, rec->jal_pc, pc_), put(rec, rec->jal_imm, w >> 12);
                break;
            case 0x17:  // AUIPC
                val = pc_ + (w & 0xfffff000u), wr = true;
                if (rec) rec->n_lui_auipc++, put(rec, rec->auipc_pc, pc_), put(rec, rec->auipc_imm, w >> 12);
                break;
            case 0x6f:  // JAL
                val = pc_ + 4, wr = true, next = pc_ + (uint32_t)imm_j;
                if (rec) {
                    rec->n_jump++, put(rec, rec->jal_op, 0), put(rec, rec->jal_pc, pc_);
                    put(rec, rec->jal_imm, imm_j < 0 ? 2013265921u - (uint32_t)(-imm_j) : (uint32_t)imm_j);
                }
                break;
            case 0x67:  // JALR
                val = pc_ + 4, wr = true, next = (a + (uint32_t)imm_i) & ~1u;
                if (rec) rec->n_jump++, put(rec, rec->jalr_pc, pc_), put(rec, rec->jalr_rs1, a), put(rec, rec->jalr_imm, w >> 20);
                break;
            case 0x63: {
                bool t;
                switch (f3) {
                    case 0: t = a == b; break;
                    case 1: t = a != b; break;
                    case 4: t = (int32_t)a < (int32_t)b; break;
                    case 5: t = (int32_t)a >= (int32_t)b; break;
                    case 6: t = a < b; break;
                    case 7: t = a >= b; break;
                    default: throw Error("illegal branch");
                }
                if (t) next = pc_ + (uint32_t)imm_b;
                if (rec) rec->n_branch++;
                if (rec) {
                    const uint32_t off = imm_b < 0 ? 2013265921u - (uint32_t)(-imm_b) : (uint32_t)imm_b;
                    if (f3 < 2) {
                        put(rec, rec->beq_op, f3), put(rec, rec->beq_a, a), put(rec, rec->beq_b, b), put(rec, rec->beq_imm, off);
                    } else {   // funct3 4 blt, 5 bge, 6 bltu, 7 bgeu -> the chip's opcode order blt, bltu, bge, bgeu
                        put(rec, rec->blt_op, ((f3 & 1u) << 1) | ((f3 >> 1) & 1u));
                        put(rec, rec->blt_a, a), put(rec, rec->blt_b, b), put(rec, rec->blt_imm, off);
                    }
                }
                break;
            }
            case 0x03: {
                const uint32_t addr = a + (uint32_t)imm_i;
                switch (f3) {
                    case 0: val = (uint32_t)(int32_t)(int8_t)load(rec, addr, 1); break;
                    case 1: val = (uint32_t)(int32_t)(int16_t)load(rec, addr, 2); break;
                    case 2: val = load(rec, addr, 4); break;
                    case 4: val = load(rec, addr, 1); break;
                    case 5: val = load(rec, addr, 2); break;
                    default: throw Error("illegal load");
                }
                wr = true;
                if (rec) {
                    const uint32_t off = (addr - exe_.data_base) & 3u;
                    const uint32_t cs = f3 == 2 ? 0 : f3 == 5 ? 1 + off / 2 : f3 == 4 ? 3 + off : f3 == 1 ? 14 + off / 2 : 16 + off;
                    rec->n_load_store++, put(rec, rec->ls_case, cs), put(rec, rec->ls_read, word_around(addr)), put(rec, rec->ls_prev, x_[rd]);
                }
                break;
            }
            case 0x23: {
                const uint32_t addr = a + (uint32_t)imm_s;
                if (f3 > 2) throw Error("illegal store");
                if (rec) {
                    (void)at_ro(addr, 1u << f3);   // range check before the word is read
                    const uint32_t off = (addr - exe_.data_base) & 3u;
                    rec->n_load_store++, put(rec, rec->ls_case, f3 == 2 ? 7 : f3 == 1 ? 8 + off / 2 : 10 + off);
                    put(rec, rec->ls_read, b), put(rec, rec->ls_prev, word_around(addr));
                }
                store(rec, addr, b, 1u << f3);
                break;
            }
            case 0x13: {
                const uint32_t c = (uint32_t)imm_i;
                wr = true;
                switch (f3) {
                    case 0: val = alu(0, a, c); break;
                    case 2: val = lt(0, a, c); break;
                    case 3: val = lt(1, a, c); break;
                    case 4: val = alu(2, a, c); break;
                    case 6: val = alu(3, a, c); break;
                    case 7: val = alu(4, a, c); break;
                    case 1: if (f7) throw Error("illegal slli"); val = shift(0, a, rs2); break;
                    default: if (f7 != 0 && f7 != 0x20) throw Error("illegal shift"); val = shift(f7 ? 2 : 1, a, rs2); break;
                }
                break;
            }
            case 0x33: {
                wr = true;
                if (f7 == 1) {
                    const int64_t sa = (int32_t)a, sb = (int32_t)b;
                    const uint64_t ua = a, ub = b;
                    switch (f3) {
                        case 0: val = a * b; if (rec) put(rec, rec->mul_b, a), put(rec, rec->mul_c, b); break;
                        case 1: val = (uint32_t)((uint64_t)(sa * sb) >> 32); break;
                        case 2: val = (uint32_t)((uint64_t)(sa * (int64_t)ub) >> 32); break;
                        case 3: val = (uint32_t)((ua * ub) >> 32); break;
                        case 4: val = b == 0 ? 0xffffffffu : (a == 0x80000000u && b == 0xffffffffu) ? a : (uint32_t)((int32_t)a / (int32_t)b); break;
                        case 5: val = b == 0 ? 0xffffffffu : a / b; break;
                        case 6: val = b == 0 ? a : (a == 0x80000000u && b == 0xffffffffu) ? 0 : (uint32_t)((int32_t)a % (int32_t)b); break;
                        default: val = b == 0 ? a : a % b; break;
                    }
                    if (rec && f3 >= 4) put(rec, rec->div_op, f3 - 4), put(rec, rec->div_b, a), put(rec, rec->div_c, b);
                    if (rec && f3 >= 1 && f3 <= 3) rec->n_mulh++, put(rec, rec->mulh_op, f3 - 1), put(rec, rec->mulh_b, a), put(rec, rec->mulh_c, b);
                    if (f3 >= 4 && rec) rec->n_divrem++;
                } else if (f7 == 0 || f7 == 0x20) {
                    switch (f3) {
                        case 0: val = alu(f7 ? 1 : 0, a, b); break;
                        case 1: if (f7) throw Error("illegal sll"); val = shift(0, a, b); break;
                        case 2: if (f7) throw Error("illegal slt"); val = lt(0, a, b); break;
                        case 3: if (f7) throw Error("illegal sltu"); val = lt(1, a, b); break;
                        case 4: if (f7) throw Error("illegal xor"); val = alu(2, a, b); break;
                        case 5: val = shift(f7 ? 2 : 1, a, b); break;
                        case 6: if (f7) throw Error("illegal or"); val = alu(3, a, b); break;
                        default: if (f7) throw Error("illegal and"); val = alu(4, a, b); break;
                    }
                } else {
                    throw Error("illegal op");
                }
                break;
            }
            case 0x73: {
                if (w != 0x00000073u) throw Error("unsupported system instruction");
                if (rec) rec->n_ecall++;
                const uint32_t call = reg_read(rec, 17);
                if (call == 93) {
                    done_ = true, exit_code_ = reg_read(rec, 10);
                } else if (call == 1) {
                    const uint32_t word = reg_read(rec, 10);
                    const uint64_t off = 4ull * reg_read(rec, 11);
                    if (off + 4 > pv_.size()) throw Error("public value index out of range");
                    memcpy(pv_.data() + off, &word, 4);
                } else if (call == 2) {
                    if (!hints_.empty()) {   // what a phantom instruction left for the guest comes first
                        val = hints_.front();
                        hints_.pop_front();
                    } else {
                        if (in_pos_ + 4 > in_.bytes.size()) throw Error("input stream exhausted");
                        memcpy(&val, in_.bytes.data() + in_pos_, 4);
                        in_pos_ += 4;
                    }
                    reg_write(rec, 10, val);
                } else if (call == 3) {
                    // Keccak-f[1600] in place on the 200 bytes at a0 (25 little-endian lanes): the intrinsic behind OpenVM's keccak
                    // extension; proven by the keccak chips of the one-statement circuit (include/zkhip_vm_circuit.hpp)
                    if (rec && log_memory_) throw Error("the keccak intrinsic is proven by the one-statement flow only");
                    const uint32_t addr = reg_read(rec, 10);
                    if (addr & 3u) throw Error("keccak state must be word-aligned");
                    uint8_t* p = at(addr, 200);
                    uint64_t st[25];
                    memcpy(st, p, 200);
                    zkhip::keccak::keccak_f1600(st);
                    memcpy(p, st, 200);
                } else if (call == 4) {
                    // SHA-256 compression on the 24 words at a0: words 0..7 the state, words 8..23 the block's message words (plain
                    // 32-bit values: the guest does the byte order); the state becomes compress(state, block) (OpenVM's sha256 extension)
                    if (rec && log_memory_) throw Error("the sha256 intrinsic is proven by the one-statement flow only");
                    const uint32_t addr = reg_read(rec, 10);
                    if (addr & 3u) throw Error("sha256 buffer must be word-aligned");
                    uint8_t* p = at(addr, 96);
                    uint32_t buf[24];
                    memcpy(buf, p, 96);
                    zkhip::sha256::compress(buf, buf + 8);
                    memcpy(p, buf, 32);
                } else if (call == 5) {
                    // r = a b mod P_i on the 24 words at a0 (a[8] | b[8] | r[8], little-endian words), i = a1: one of the moduli the
                    // app's openvm.toml lists (OpenVM's modular extension)
                    if (rec && log_memory_) throw Error("the modmul intrinsic is proven by the one-statement flow only");
                    const uint32_t addr = reg_read(rec, 10), sel = reg_read(rec, 11), which = sel & 7u, mop = sel >> 3;   // a1 = index + 8 op: mul, add, sub, div, is_eq
                    if (which >= moduli_.size()) throw Error("modmul: the app's openvm.toml lists no modulus " + std::to_string(which));
                    if (mop >= zkhip::modular::N_OPS) throw Error("modular: unknown operation " + std::to_string(mop));
                    if (addr & 3u) throw Error("modmul buffer must be word-aligned");
                    const size_t nb = 4 * zkhip::modular::words_of(moduli_[which]);   // bytes per operand: 32, or 48 for a modulus above 2^256
                    uint8_t* p = at(addr, 3 * nb);
                    zkhip::modular::U256 a = zkhip::modular::load_words(p, nb / 4), b = zkhip::modular::load_words(p + nb, nb / 4), q{}, r{};
                    if (mop == zkhip::modular::OP_IS_EQ) {   // r = [a = b (mod P)]: the difference's residue, then the bit
                        if (!zkhip::modular::addsubmod(zkhip::modular::OP_SUB, a, b, moduli_[which], &q, &r))
                            throw Error("modular: operands of an equality test further apart than the modulus");
                        bool zero_r = true;
                        for (uint32_t w : r.w) zero_r = zero_r && w == 0;
                        r = zkhip::modular::U256{};
                        r.w[0] = zero_r ? 1u : 0u;
                    } else {
                        const bool ok_op = mop == zkhip::modular::OP_MUL   ? zkhip::modular::mulmod(a, b, moduli_[which], &q, &r)
                                           : mop == zkhip::modular::OP_DIV ? zkhip::modular::divmod_p(a, b, moduli_[which], &r)
                                                                           : zkhip::modular::addsubmod(mop, a, b, moduli_[which], &q, &r);
                        if (!ok_op)
                            throw Error("modular: operands too far from the modulus (a product's quotient beyond 256 bits, or a difference beyond the modulus), "
                                        "or a division with a dividend that is not reduced or a divisor that is not invertible");
                    }
                    memcpy(p + 2 * nb, r.w, nb);
                } else if (call == 6) {
                    // a = b op c on the 24 words at a0 (b[8] | c[8] | a[8], little-endian 256-bit words), op = a1: 0 add, 1 sub, 2 xor,
                    // 3 or, 4 and, 5 mul, 6 sltu, 7 slt, 8 eq, 9 sll, 10 srl, 11 sra (OpenVM's bigint extension)
                    if (rec && log_memory_) throw Error("the int256 intrinsic is proven by the one-statement flow only");
                    const uint32_t addr = reg_read(rec, 10), op = reg_read(rec, 11);
                    if (op >= zkhip::int256::N_INT256_OPS) throw Error("int256: unknown opcode " + std::to_string(op));
                    if (addr & 3u) throw Error("int256 buffer must be word-aligned");
                    uint8_t* p = at(addr, 96);
                    uint32_t b[8], c[8], a[8];
                    memcpy(b, p, 32), memcpy(c, p + 32, 32);
                    if (zkhip::int256::is_branch_op(op)) {
                        // 12 beq, 13 bne, 14 bltu, 15 blt, 16 bgeu, 17 bge (OpenVM's Rv32BranchEqual256 / Rv32BranchLessThan256): compares b and
                        // c, leaves the comparison's 0 / 1 in a like opcodes 6 .. 8, and continues at pc + a2 (a byte offset, a multiple of
                        // four, backwards when negative) if the branch is taken
                        memset(a, 0, sizeof a);
                        a[0] = zkhip::int256::cmp256(op, b, c);
                        const uint32_t off = reg_read(rec, 12);
                        if (off & 3u) throw Error("int256 branch: the offset in a2 must be a multiple of four");
                        if ((a[0] != 0) != zkhip::int256::branch_negates(op)) next = pc_ + off;
                    } else if (op >= zkhip::int256::OP_SLL) {
                        zkhip::int256::shift256(op, b, c, a);
                    } else if (op >= zkhip::int256::OP_SLTU) {
                        memset(a, 0, sizeof a);
                        a[0] = zkhip::int256::cmp256(op, b, c);
                    } else if (op == zkhip::int256::OP_MUL) zkhip::int256::mul256(b, c, a);
                    else zkhip::int256::alu256(op, b, c, a);
                    memcpy(p + 64, a, 32);
                } else if (call == 7) {
                    // (x3, y3) = (x1, y1) + (x2, y2) (op 0; x1 != x2) or 2 (x1, y1) (op 1) on the 48 words at a0 (x1 y1 | x2 y2 | x3 y3,
                    // little-endian 256-bit words) on curve i of the app's openvm.toml, a1 = i + 8 op (OpenVM's ecc extension)
                    if (rec && log_memory_) throw Error("the ecc intrinsic is proven by the one-statement flow only");
                    const uint32_t addr = reg_read(rec, 10), sel = reg_read(rec, 11), which = sel & 7u, eop = sel >> 3;
                    if (which >= curves_.size()) throw Error("ecc: the app's openvm.toml lists no curve " + std::to_string(which));
                    if (eop >= zkhip::ecc::N_OPS) throw Error("ecc: unknown operation " + std::to_string(eop));
                    if (addr & 3u) throw Error("ecc buffer must be word-aligned");
                    const size_t nb = 4 * zkhip::modular::words_of(curves_[which].p);
                    uint8_t* p = at(addr, 6 * nb);
                    zkhip::modular::U256 c[4], x3, y3;
                    for (int k = 0; k < 4; k++) c[k] = zkhip::modular::load_words(p + nb * k, nb / 4);
                    if (!zkhip::ecc::ec_op(eop, curves_[which], c[0], c[1], c[2], c[3], &last_slope_, &x3, &y3))
                        throw Error("ecc: coordinates not reduced, or no slope (equal abscissae in an addition, y = 0 in a doubling)");
                    memcpy(p + 4 * nb, x3.w, nb), memcpy(p + 5 * nb, y3.w, nb);
                } else if (call == 8) {
                    // r = a b, a + b, a - b or a / b in Fp[u] / (u^2 + 1) on the 48 words at a0 (a0 a1 | b0 b1 | r0 r1, little-endian 256-bit
                    // words), field i of the app's openvm.toml, a1 = i + 8 op (OpenVM's fp2 extension)
                    if (rec && log_memory_) throw Error("the fp2 intrinsic is proven by the one-statement flow only");
                    const uint32_t addr = reg_read(rec, 10), sel = reg_read(rec, 11), which = sel & 7u, fop = sel >> 3;
                    if (which >= fp2_moduli_.size()) throw Error("fp2: the app's openvm.toml lists no fp2 field " + std::to_string(which));
                    if (fop >= zkhip::fp2::N_OPS) throw Error("fp2: unknown operation " + std::to_string(fop));
                    if (addr & 3u) throw Error("fp2 buffer must be word-aligned");
                    const size_t nb = 4 * zkhip::modular::words_of(fp2_moduli_[which]);
                    uint8_t* p = at(addr, 6 * nb);
                    using zkhip::modular::load_words;
                    zkhip::fp2::Elem a{load_words(p, nb / 4), load_words(p + nb, nb / 4)}, b{load_words(p + 2 * nb, nb / 4), load_words(p + 3 * nb, nb / 4)}, r;
                    if (!zkhip::fp2::fp2_op(fop, fp2_moduli_[which], a, b, &r)) throw Error("fp2: components not reduced, or a division by zero");
                    memcpy(p + 4 * nb, r.c0.w, nb), memcpy(p + 5 * nb, r.c1.w, nb);
                } else if (call == zkhip::native::CALL_ARITH || call == zkhip::native::CALL_EXT) {
                    // r = a op b over BabyBear (3 words at a0) or its quartic extension (12 words: a[4] | b[4] | r[4]), a1 = op: 0 add, 1 sub,
                    // 2 mul, 3 div; a word is read as a field element (value mod p), results are canonical (OpenVM's native extension:
                    // FieldArithmetic / FieldExtension; include/zkhip_native.hpp)
                    if (rec && log_memory_) throw Error("the native field intrinsics are proven by the one-statement flow only");
                    if (!native_) throw Error("native: the app's openvm.toml does not enable the native extension");
                    const uint32_t addr = reg_read(rec, 10), op = reg_read(rec, 11);
                    if (addr & 3u) throw Error("native buffer must be word-aligned");
                    if (call == zkhip::native::CALL_ARITH) {
                        uint32_t w3[3];
                        uint8_t* p = at(addr, 12);
                        memcpy(w3, p, 12);
                        if (!zkhip::native::arith(op, w3[0], w3[1], &w3[2])) throw Error("native: unknown operation or a division by zero");
                        memcpy(p + 8, &w3[2], 4);
                    } else {
                        uint32_t w12[12];
                        uint8_t* p = at(addr, 48);
                        memcpy(w12, p, 48);
                        if (!zkhip::native::ext_arith(op, w12, w12 + 4, w12 + 8)) throw Error("native extension: unknown operation or a division by zero");
                        memcpy(p + 32, w12 + 8, 16);
                    }
                } else if (call == zkhip::native::CALL_CASTF) {
                    // the word at a0, a native field element below 2^30, as four bytes in the word at a0 + 4 (OpenVM's CastF)
                    if (rec && log_memory_) throw Error("the castf intrinsic is proven by the one-statement flow only");
                    if (!castf_) throw Error("castf: the app's openvm.toml does not enable the castf extension");
                    const uint32_t addr = reg_read(rec, 10);
                    if (addr & 3u) throw Error("castf buffer must be word-aligned");
                    uint8_t* p = at(addr, 8);
                    uint32_t v;
                    memcpy(&v, p, 4);
                    if (v >= zkhip::native::CASTF_BOUND) throw Error("castf: the field element does not lie below 2^30");
                    memcpy(p + 4, &v, 4);
                } else {
                    throw Error("unknown environment call " + std::to_string(call));
                }
                break;
            }
            case 0x0f:   // FENCE: a single hart with one memory -- no operation (ordinary rv32im toolchains emit it)
                // ... and the carrier of PHANTOM instructions (OpenVM: instructions the circuit sees as no-ops, whose sub-executors leave
                // advice in the hint stream): fm = 0101, kind in the pred / succ bits, the operand pointer in rs1
                if ((w >> 28) == 5u) phantom((w >> 20) & 0xffu, x_[(w >> 15) & 31]);
                (void)alu(0, 0, 0);   // what the circuit proves for it: add x0, x0, 0 (the frame chip hands the ALU chip that row)
                break;
            default: throw Error("illegal instruction " + std::to_string(w) + " at pc " + std::to_string(pc_));
        }
        last_val_ = val;
        if (wr) reg_write(rec, rd, val);
        pc_ = next;
        instret_++;
    }
};

// An RV32 ELF image as an Exe (the reference passes `exe: impl Into<ExecutableFormat>`, an ELF or a transpiled VmExe,
// crates/prover/src/utils/vm.rs:15): ELF32 little-endian, e_machine = EM_RISCV; the executable PT_LOAD segment becomes the
// program, the other PT_LOAD segments the initial image of the read-write region (zero-filled up to p_memsz), which extends
// `stack_bytes` beyond the highest loaded address.
inline Exe load_elf(const std::vector<uint8_t>& f, uint32_t stack_bytes = 1u << 20) {
    auto u16 = [&](size_t o) -> uint32_t {
        if (o + 2 > f.size()) throw Error("ELF truncated");
        return (uint32_t)f[o] | ((uint32_t)f[o + 1] << 8);
    };
    auto u32 = [&](size_t o) -> uint32_t { return u16(o) | (u16(o + 2) << 16); };
    if (f.size() < 52 || f[0] != 0x7f || f[1] != 'E' || f[2] != 'L' || f[3] != 'F') throw Error("not an ELF file");
    if (f[4] != 1 || f[5] != 1) throw Error("not a 32-bit little-endian ELF");
    if (u16(18) != 243) throw Error("not a RISC-V ELF");
    const uint32_t entry = u32(24), phoff = u32(28), phentsize = u16(42), phnum = u16(44);
    if (phentsize < 32) throw Error("bad program header size");
    Exe exe;
    exe.entry = entry;
    bool have_text = false;
    uint64_t lo = UINT64_MAX, hi = 0;
    struct Seg {
        uint32_t off, vaddr, filesz, memsz;
    };
    std::vector<Seg> data;
    for (uint32_t i = 0; i < phnum; i++) {
        const size_t ph = (size_t)phoff + (size_t)i * phentsize;
        if (u32(ph) != 1) continue;  // PT_LOAD
        const Seg sg{u32(ph + 4), u32(ph + 8), u32(ph + 16), u32(ph + 20)};
        const uint32_t flags = u32(ph + 24);
        if ((uint64_t)sg.off + sg.filesz > f.size() || sg.filesz > sg.memsz) throw Error("ELF segment outside the file");
        if (flags & 1u) {  // PF_X
            if (have_text) throw Error("more than one executable segment");
            if ((sg.vaddr & 3u) || (sg.filesz & 3u)) throw Error("misaligned text segment");
            have_text = true;
            exe.pc_base = sg.vaddr;
            exe.program.resize(sg.filesz / 4);
            if (sg.filesz) memcpy(exe.program.data(), f.data() + sg.off, sg.filesz);
        } else {
            data.push_back(sg);
            lo = std::min<uint64_t>(lo, sg.vaddr), hi = std::max<uint64_t>(hi, (uint64_t)sg.vaddr + sg.memsz);
        }
    }
    if (!have_text) throw Error("no executable segment");
    if (entry < exe.pc_base || entry >= exe.pc_base + 4 * (uint64_t)exe.program.size()) throw Error("entry point outside the text segment");
    // the chips hold a pc as one BabyBear element and compose pc + 4 from limbs of 30 bits (OpenVM's PC_BITS): link the guest low
    if ((uint64_t)exe.pc_base + 4 * (uint64_t)exe.program.size() >= (1ull << 30)) throw Error("text segment must lie below 2^30");
    if (data.empty()) lo = hi = exe.data_base;
    lo &= ~3ull;
    if (hi - lo + stack_bytes > (1ull << 30)) throw Error("data image too large");
    exe.data_base = (uint32_t)lo;
    exe.data.assign((size_t)(hi - lo), 0);
    for (const Seg& sg : data)
        if (sg.filesz) memcpy(exe.data.data() + (sg.vaddr - lo), f.data() + sg.off, sg.filesz);
    exe.memory_bytes = (uint32_t)(((hi - lo + 3) & ~3ull) + stack_bytes);
    return exe;
}

// The run cut into continuation segments of at most `max_instr` instructions: one ExecRecords per segment (each proven on its
// own: VmProver::prove_segments), the public values and the total instruction count of the whole run.
struct SegmentedExecution {
    ExecutionResult result;
    std::vector<ExecRecords> segments;
};
inline SegmentedExecution execute_segments(const Exe& exe, const StdIn& inputs, uint64_t max_instr) {
    if (max_instr == 0) throw Error("segment length must be positive");
    SegmentedExecution se;
    Machine m(exe, inputs);
    bool done = false;
    while (!done) {
        se.segments.emplace_back();
        done = m.run_segment(max_instr, &se.segments.back());
    }
    if (m.exit_code()) throw Error("guest exited with code " + std::to_string(m.exit_code()));
    bool all_zero = true;
    for (uint8_t b : m.public_values()) all_zero = all_zero && b == 0;
    if (all_zero) throw Error("public_values are all 0s for unexpected reason");
    se.result = ExecutionResult{m.instret(), m.public_values()};
    return se;
}

// An executable file: an RV32 ELF image, or the flat test format  u32 words [0x58455A4B "KZEX", pc_base, n_program, program...,
// data_base, memory_bytes, n_data_bytes] followed by the data bytes.
inline Exe parse_exe(const std::vector<uint8_t>& raw) {
    if (raw.size() >= 4 && raw[0] == 0x7f && raw[1] == 'E' && raw[2] == 'L' && raw[3] == 'F') return load_elf(raw);
    auto word = [&](size_t i) {
        if (4 * i + 4 > raw.size()) throw Error("exe truncated");
        uint32_t v;
        memcpy(&v, raw.data() + 4 * i, 4);
        return v;
    };
    if (word(0) != 0x58455A4Bu) throw Error("bad exe magic");
    Exe exe;
    exe.pc_base = word(1);
    if ((uint64_t)exe.pc_base + 4 * (uint64_t)word(2) >= (1ull << 30)) throw Error("text segment must lie below 2^30");
    const size_t n_prog = word(2);
    if (n_prog > raw.size() / 4) throw Error("exe truncated");
    for (size_t i = 0; i < n_prog; i++) exe.program.push_back(word(3 + i));
    exe.data_base = word(3 + n_prog), exe.memory_bytes = word(4 + n_prog);
    if ((uint64_t)exe.data_base + exe.memory_bytes > (1ull << 30)) throw Error("the read-write region must lie below 2^30");   // as load_elf
    const size_t n_data = word(5 + n_prog), off = 4 * (6 + n_prog);
    if (off + n_data > raw.size()) throw Error("exe data truncated");
    exe.data.assign(raw.begin() + (long)off, raw.begin() + (long)(off + n_data));
    return exe;
}

// crates/prover/src/utils/vm.rs:13-48.  `records`, when given, receives the per-chip execution records of the run that
// produced the result (the metered run, or the fall-back run).
// `moduli` / `curves`: the app's modular and ecc extensions (openvm.toml), for guests that call those intrinsics.
inline ExecutionResult execute_guest(const Exe& exe, const StdIn& inputs, uint64_t max_cost = 0, ExecRecords* records = nullptr,
                                     uint64_t max_instr = 0, const std::vector<zkhip::modular::U256>& moduli = {},
                                     const std::vector<zkhip::ecc::Curve>& curves = {}, const std::vector<zkhip::modular::U256>& fp2_moduli = {},
                                     const zkhip::native::Enabled& native_ext = zkhip::native::Enabled()) {
    auto all_zero = [](const std::vector<uint8_t>& v) {
        for (uint8_t b : v)
            if (b) return false;
        return true;
    };
    ExecRecords local;
    ExecRecords* rec = records ? records : &local;
    {
        *rec = ExecRecords();
        Machine m(exe, inputs);
        m.set_moduli(moduli), m.set_curves(curves), m.set_fp2_moduli(fp2_moduli), m.set_native(native_ext);
        bool finished = false;
        try {
            finished = m.run(max_instr, max_cost, rec);
        } catch (const Error&) {
            finished = false;  // "Metered execution failed: {e}, falling back to execute"
        }
        if (finished) {
            if (m.exit_code()) throw Error("guest exited with code " + std::to_string(m.exit_code()));
            if (all_zero(m.public_values())) throw Error("public_values are all 0s for unexpected reason");
            return ExecutionResult{m.instret(), m.public_values()};
        }
    }
    // the plain executor: no cost bound, and -- like the reference's -- no instruction count (u64::MAX as the sentinel)
    *rec = ExecRecords();
    Machine m(exe, inputs);
    m.set_moduli(moduli), m.set_curves(curves), m.set_fp2_moduli(fp2_moduli), m.set_native(native_ext);
    if (!m.run(max_instr, 0, rec)) throw Error("instruction limit reached");  // max_instr: a service-side bound, 0 = none
    if (m.exit_code()) throw Error("guest exited with code " + std::to_string(m.exit_code()));
    if (all_zero(m.public_values())) throw Error("public_values are all 0s upon execute");
    return ExecutionResult{UINT64_MAX, m.public_values()};
}

}  // namespace zkhip_vm
