// zkhip_vm_exec.hpp -- the SEGMENTING executor of the one-statement flow: runs the guest (the interpreter of include/zkhip_vm.hpp)
// and produces, segment by segment, everything the chips of include/zkhip_vm_circuit.hpp are generated from:
//   * the per-class core records of include/zkhip_vm.hpp (unchanged), the frame records (operands, result, pc step of every
//     instruction), the load/store adapter records, the ecall rows;
//   * for every word access of every chip the timestamp of the word's previous access (the circuit's timestamps are 1 + 16 k + slot
//     for instruction k of the segment; words: registers = address space 1, memory = 2 with word = byte address / 4, public values = 3);
//   * the PERSISTENT MEMORY: a sparse Merkle tree (Poseidon2, blocks of 8 cells) over all three address spaces that lives across
//     segments; per segment the touched blocks (leaf rows) and the nodes of their paths (merkle rows) with digests before and after.
// A segment ends when the guest exits or when one more instruction could overflow a chip's fixed height (SegmentCaps): heights are
// part of the verifying key, so every segment of every run of one app is proven under ONE key -- what the aggregation layer needs
// (include/zkhip_aggregation.hpp).  This is the role of OpenVM's metered execution + segmentation + preflight (un-vendored; reached by the
// reference through sdk.prove, crates/prover/src/prover/mod.rs:355-357, and execute_guest, crates/prover/src/utils/vm.rs:13-48).
// Host only; Poseidon2 through the library's host entry point (zkhip_poseidon2_permute_host).
#pragma once
#include <array>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>

#include "zkhip.h"
#include "zkhip_vm.hpp"
#include "zkhip_vm_circuit.hpp"

namespace zkhip_vm {
namespace vmc = zkhip::vmc;
using Digest = std::array<uint32_t, 8>;

inline Digest p2_block(const uint32_t cells[8]) {
    uint32_t st[16] = {};
    for (int i = 0; i < 8; i++) st[i] = cells[i];
    zkhip_poseidon2_permute_host(st);
    Digest d;
    for (int i = 0; i < 8; i++) d[i] = st[i];
    return d;
}
inline Digest p2_compress(const Digest& l, const Digest& r) {
    uint32_t st[16];
    for (int i = 0; i < 8; i++) st[i] = l[i], st[8 + i] = r[i];
    zkhip_poseidon2_permute_host(st);
    Digest d;
    for (int i = 0; i < 8; i++) d[i] = st[i];
    return d;
}

// Sparse Merkle tree over block labels (vmc::LEAF_LEVEL levels above the blocks); absent nodes are all-zero subtrees.
class MemoryTree {
public:
    MemoryTree() {
        const uint32_t z[8] = {};
        zero_[vmc::LEAF_LEVEL] = p2_block(z);
        for (int l = (int)vmc::LEAF_LEVEL - 1; l >= 0; l--) zero_[l] = p2_compress(zero_[l + 1], zero_[l + 1]);
    }
    Digest get(unsigned level, uint32_t idx) const {
        auto it = nodes_.find(key(level, idx));
        return it == nodes_.end() ? zero_[level] : it->second;
    }
    void set(unsigned level, uint32_t idx, const Digest& d) { nodes_[key(level, idx)] = d; }
    Digest root() const { return get(0, 0); }
    // sets the given blocks and rehashes their paths (used for the initial image)
    void set_blocks(const std::vector<std::pair<uint32_t, std::array<uint32_t, 8>>>& blocks) {
        std::vector<uint32_t> cur;
        for (const auto& b : blocks) set(vmc::LEAF_LEVEL, b.first, p2_block(b.second.data())), cur.push_back(b.first >> 1);
        for (int l = (int)vmc::LEAF_LEVEL - 1; l >= 0; l--) {
            std::sort(cur.begin(), cur.end());
            cur.erase(std::unique(cur.begin(), cur.end()), cur.end());
            std::vector<uint32_t> up;
            for (uint32_t i : cur) set(l, i, p2_compress(get(l + 1, 2 * i), get(l + 1, 2 * i + 1))), up.push_back(i >> 1);
            cur = std::move(up);
        }
    }
    static uint64_t key(unsigned level, uint32_t idx) { return ((uint64_t)level << 32) | idx; }

private:
    std::unordered_map<uint64_t, Digest> nodes_;
    Digest zero_[vmc::LEAF_LEVEL + 1];
};

// fixed trace heights of a segment = part of its verifying key
struct SegmentCaps {
    unsigned log_height[vmc::N_AIRS] = {};
    // the chips of this app, in proof order: the base chips, then the extensions its openvm.toml enables (ids of vmc::AirId)
    std::vector<unsigned> ids;
    unsigned n_airs = 0;
    int pos(unsigned id) const {
        for (size_t p = 0; p < ids.size(); p++)
            if (ids[p] == id) return (int)p;
        return -1;
    }
    bool keccak() const { return pos(vmc::A_KECCAK) >= 0; }
    bool sha256() const { return pos(vmc::A_SHA256) >= 0; }
    bool int256() const { return pos(vmc::A_INT256) >= 0; }
    std::vector<zkhip::modular::U256> moduli;   // the modular extension: chips A_MODMUL(i), A_MODMUL_IO(i) per modulus
    std::vector<zkhip::ecc::Curve> curves;      // the ecc extension: chips A_EC(i), A_EC_IO(i) per curve
    std::vector<zkhip::modular::U256> fp2_moduli;   // the fp2 extension: chips A_FP2(i), A_FP2_IO(i) per field
    // `[app_vm_config.native]`: the native field and extension chips; `[app_vm_config.castf]`: the castf chip; `[app_vm_config.pairing]`
    // brings no chip (OpenVM's pairing extension is a phantom sub-executor: the final-exponentiation hint, include/zkhip_vm.hpp phantom kind 2)
    using NativeExt = zkhip::native::Enabled;
    NativeExt ext;
    bool native() const { return pos(vmc::A_NATIVE_ARITH) >= 0; }
    bool castf() const { return pos(vmc::A_CASTF) >= 0; }
    // heights for segments of about 2^log_frame instructions of an ordinary instruction mix
    // log_keccak: 0 = the app has no keccak extension (22 chips); otherwise the rows of the Keccak-f chip (24 per permutation call,
    // at least 32).  A wide chip costs every segment proof and every leaf circuit its 2634 openings per query, used or not, so it is
    // part of the key only where openvm.toml asks for it (crates/circuits/chunk-circuit/openvm.toml `[app_vm_config.keccak]`).
    // log_sha256: likewise for the SHA-256 compression chip (65 rows per block, at least 128; `[app_vm_config.sha256]`).
    // moduli / log_modmul: one multiplication chip (one call per row) + adapter (24 rows per call) per modulus of `[app_vm_config.modular]`.
    // log_int256: rows of the 256-bit ALU chip (one operation per row; `[app_vm_config.bigint]`), 0 = no bigint extension.
    // curves / log_ec: one point chip (one call per row) + adapter (48 rows per call) per curve of `[[app_vm_config.ecc.supported_curves]]`.
    static SegmentCaps for_frame(unsigned log_frame, unsigned log_program, unsigned log_keccak = 0, unsigned log_sha256 = 0,
                                 const std::vector<zkhip::modular::U256>& moduli = {}, unsigned log_modmul = 0, unsigned log_int256 = 0,
                                 const std::vector<zkhip::ecc::Curve>& curves = {}, unsigned log_ec = 0,
                                 const std::vector<zkhip::modular::U256>& fp2_moduli = {}, const NativeExt& ext = NativeExt()) {
        SegmentCaps c;
        for (unsigned a = 0; a < vmc::N_BASE_AIRS; a++) c.ids.push_back(a);
        auto sub = [&](unsigned d) { return log_frame > d + 2 ? log_frame - d : 2u; };
        c.log_height[vmc::A_PROGRAM] = log_program, c.log_height[vmc::A_FRAME] = log_frame;
        c.log_height[vmc::A_ALU] = sub(1), c.log_height[vmc::A_LT] = sub(3), c.log_height[vmc::A_MUL] = sub(3), c.log_height[vmc::A_MULH] = sub(5);
        c.log_height[vmc::A_DIVREM] = sub(5), c.log_height[vmc::A_SHIFT] = sub(3), c.log_height[vmc::A_BEQ] = sub(2), c.log_height[vmc::A_BLT] = sub(2);
        c.log_height[vmc::A_JAL_LUI] = sub(3), c.log_height[vmc::A_AUIPC] = sub(5), c.log_height[vmc::A_JALR] = sub(4), c.log_height[vmc::A_LS] = sub(1);
        // (floors: a handful of touched blocks already needs ~28 path nodes each and two permutations per row)
        c.log_height[vmc::A_ECALL] = std::max(sub(7), 3u);
        // (touched blocks: as many rows as the path-node chip -- a DENSE footprint has about one path node per block, and the Poseidon2 chip below holds
        // two permutations for each of both; round 4 had half as many rows here, and a guest sweeping a 64 KiB array ended every segment at one
        // sweep, its frame half empty: 66 segments where 34 do)
        c.log_height[vmc::A_LEAF] = std::max(sub(4), 4u);
        c.log_height[vmc::A_MERKLE] = std::max(sub(4), 8u);
        c.log_height[vmc::A_POSEIDON2] = std::max(sub(2), c.log_height[vmc::A_MERKLE] + 2), c.log_height[vmc::A_CONNECTOR] = 0, c.log_height[vmc::A_BITWISE] = 16;
        c.log_height[vmc::A_RANGE_TUPLE] = 19, c.log_height[vmc::A_RANGE] = 16;
        if (log_keccak) {
            c.ids.push_back(vmc::A_KECCAK), c.ids.push_back(vmc::A_KECCAK_IO);
            c.log_height[vmc::A_KECCAK] = std::max(log_keccak, 5u), c.log_height[vmc::A_KECCAK_IO] = c.log_height[vmc::A_KECCAK] + 1;   // 25 adapter rows per 24 chip rows
        }
        if (log_sha256) {
            c.ids.push_back(vmc::A_SHA256), c.ids.push_back(vmc::A_SHA256_IO);
            c.log_height[vmc::A_SHA256] = std::max(log_sha256, 7u), c.log_height[vmc::A_SHA256_IO] = c.log_height[vmc::A_SHA256] - 1;       // 24 adapter rows per 65 chip rows
        }
        if (log_int256) {
            c.ids.push_back(vmc::A_INT256), c.ids.push_back(vmc::A_INT256_IO), c.ids.push_back(vmc::A_MUL256), c.ids.push_back(vmc::A_CMP256), c.ids.push_back(vmc::A_SHIFT256);
            c.log_height[vmc::A_INT256] = c.log_height[vmc::A_MUL256] = c.log_height[vmc::A_CMP256] = c.log_height[vmc::A_SHIFT256] = std::max(log_int256, 1u);
            c.log_height[vmc::A_INT256_IO] = c.log_height[vmc::A_INT256] + 6;   // 24 adapter rows per call of either chip
        }
        if (moduli.size() > vmc::MAX_MODULI) throw Error("at most " + std::to_string(vmc::MAX_MODULI) + " moduli");
        c.moduli = moduli;
        for (unsigned i = 0; i < moduli.size(); i++) {
            c.ids.push_back(vmc::A_MODMUL(i)), c.ids.push_back(vmc::A_MODMUL_IO(i));
            c.log_height[vmc::A_MODMUL(i)] = std::max(log_modmul, 1u), c.log_height[vmc::A_MODMUL_IO(i)] = c.log_height[vmc::A_MODMUL(i)] + (zkhip::modular::words_of(moduli[i]) == 12 ? 6 : 5);   // 24 (36) adapter rows per call
        }
        if (curves.size() > vmc::MAX_CURVES) throw Error("at most " + std::to_string(vmc::MAX_CURVES) + " curves");
        c.curves = curves;
        for (unsigned i = 0; i < curves.size(); i++) {
            c.ids.push_back(vmc::A_EC(i)), c.ids.push_back(vmc::A_EC_IO(i));
            c.log_height[vmc::A_EC(i)] = std::max(log_ec, 1u), c.log_height[vmc::A_EC_IO(i)] = c.log_height[vmc::A_EC(i)] + (zkhip::modular::words_of(curves[i].p) == 12 ? 7 : 6);   // 48 (72) adapter rows per call
        }
        if (fp2_moduli.size() > vmc::MAX_FP2) throw Error("at most " + std::to_string(vmc::MAX_FP2) + " fp2 fields");
        c.fp2_moduli = fp2_moduli;
        for (unsigned i = 0; i < fp2_moduli.size(); i++) {   // (the heights of the ecc chips: one operation per row, 48 adapter rows per call)
            c.ids.push_back(vmc::A_FP2(i)), c.ids.push_back(vmc::A_FP2_IO(i));
            c.log_height[vmc::A_FP2(i)] = std::max(log_ec, 1u), c.log_height[vmc::A_FP2_IO(i)] = c.log_height[vmc::A_FP2(i)] + (zkhip::modular::words_of(fp2_moduli[i]) == 12 ? 7 : 6);
        }
        // native / castf: one call per row; a call is an ecall plus the instructions that set it up (at least a0, a1, a7), so 2^(f - 2)
        // rows hold whatever a frame of 2^f instructions can ask for
        c.ext = ext;
        if (ext.native) {
            c.ids.push_back(vmc::A_NATIVE_ARITH), c.ids.push_back(vmc::A_NATIVE_EXT);
            c.log_height[vmc::A_NATIVE_ARITH] = sub(2), c.log_height[vmc::A_NATIVE_EXT] = sub(3);
        }
        if (ext.castf) c.ids.push_back(vmc::A_CASTF), c.log_height[vmc::A_CASTF] = sub(3);
        c.n_airs = (unsigned)c.ids.size();
        return c;
    }
    size_t rows(unsigned air) const { return (size_t)1 << log_height[air]; }
};
constexpr uint32_t TUPLE_X = 256, TUPLE_Y = 2048;

struct SegmentRecords {
    ExecRecords core;                                        // per-class core records; core.pc_index = the frame rows' program indices
    std::vector<uint32_t> f_x, f_y, f_z, f_rdprev, f_pcinc;  // frame: operands, result, previous rd, pc step (field element)
    std::vector<uint32_t> f_pts1, f_pts2, f_pts3;            // frame: timestamps of the previous accesses of rs1, rs2, rd (0 where unused)
    std::vector<uint32_t> ls_ts, ls_base, ls_imm, ls_pts;    // load/store adapter, aligned with core.ls_*; ls_pts = the word's previous access
    std::vector<uint32_t> ecall_rows;                        // row-major canonical rows of the ecall / leaf / merkle chips
    std::vector<uint32_t> leaf_rows, merkle_rows, p2_inputs; // p2_inputs: [n][16] canonical, one row per hash-bus request
    std::vector<uint32_t> kk_states, kk_ts, kio_rows;        // keccak calls: 50 input words and the timestamp per call; the adapter's rows
    std::vector<uint32_t> sha_blocks, sha_ts, shaio_rows;    // sha256 calls: 24 words (state, block) and the timestamp per call; the adapter's rows
    std::vector<uint32_t> i256_records, i256_ts, i256io_rows;   // int256 calls: op | b | c (17 words) and the timestamp per call; the adapter's rows
    std::vector<uint32_t> mul256_records, mul256_ts;            // ... those with opcode 5: the multiplication chip's
    std::vector<uint32_t> cmp256_records, cmp256_ts;            // ... those with opcodes 6..8: the comparison chip's
    std::vector<uint32_t> sh256_records, sh256_ts;              // ... those with opcodes 9..11: the shift chip's
    // modular calls per modulus: op | a | b (17 words) and the timestamp per call; the adapter's rows
    std::vector<uint32_t> mm_records[vmc::MAX_MODULI], mm_ts[vmc::MAX_MODULI], mmio_rows[vmc::MAX_MODULI];
    // ecc calls per curve: op | x1 y1 x2 y2 | slope (41 words) and the timestamp per call; the adapter's rows
    std::vector<uint32_t> ec_records[vmc::MAX_CURVES], ec_ts[vmc::MAX_CURVES], ecio_rows[vmc::MAX_CURVES];
    // fp2 calls per field: op | a | b (33 words; a division's a slot holds the quotient) and the timestamp per call; the adapter's rows
    std::vector<uint32_t> fp2_records[vmc::MAX_FP2], fp2_ts[vmc::MAX_FP2], fp2io_rows[vmc::MAX_FP2];
    // native field calls: op | b | c | previous result word | word pointer | ts | previous timestamps of the three words (9 words per call)
    // native extension calls: op | x[4] | y[4] | previous result words [4] | word pointer | ts | previous timestamps of the twelve words (27 words)
    // castf calls: x | previous output word | word pointer | ts | previous timestamps of the two words (6 words)
    static constexpr size_t NAT_RECORD = 9, NEXT_RECORD = 27, CASTF_RECORD = 6;
    std::vector<uint32_t> nat_records, next_records, castf_records;
    size_t n_nat() const { return nat_records.size() / NAT_RECORD; }
    size_t n_next() const { return next_records.size() / NEXT_RECORD; }
    size_t n_castf() const { return castf_records.size() / CASTF_RECORD; }
    uint32_t pc_start = 0, pc_end = 0, ts_end = 0;
    Digest root_init{}, root_final{};
    size_t n_instr = 0;
    size_t n_ecall() const { return ecall_rows.size() / vmc::ECALL_WIDTH; }
    size_t n_leaf() const { return leaf_rows.size() / vmc::LEAF_WIDTH; }
    size_t n_merkle() const { return merkle_rows.size() / vmc::MERKLE_WIDTH; }
    size_t n_p2() const { return p2_inputs.size() / 16; }
    size_t n_keccak() const { return kk_ts.size(); }
    size_t n_sha256() const { return sha_ts.size(); }
    // keep_frame: the eight frame arrays keep their size -- the executor sizes them to the frame's rows, writes by index and cuts them to
    // n_instr at the end of the segment, so a recycled record pays no zero-fill for them (2 MB each at 2^19 rows)
    void clear(bool keep_frame = false) {
        core.clear();
        if (!keep_frame)
            for (auto* v : {&f_x, &f_y, &f_z, &f_rdprev, &f_pcinc, &f_pts1, &f_pts2, &f_pts3}) v->clear();
        for (auto* v : {&ls_ts, &ls_base, &ls_imm, &ls_pts, &ecall_rows, &leaf_rows, &merkle_rows,
                        &p2_inputs, &kk_states, &kk_ts, &kio_rows, &sha_blocks, &sha_ts, &shaio_rows, &i256_records, &i256_ts, &i256io_rows, &mul256_records, &mul256_ts, &cmp256_records, &cmp256_ts, &sh256_records, &sh256_ts, &nat_records, &next_records, &castf_records})
            v->clear();
        for (unsigned i = 0; i < vmc::MAX_MODULI; i++) mm_records[i].clear(), mm_ts[i].clear(), mmio_rows[i].clear();
        for (unsigned i = 0; i < vmc::MAX_CURVES; i++) ec_records[i].clear(), ec_ts[i].clear(), ecio_rows[i].clear();
        for (unsigned i = 0; i < vmc::MAX_FP2; i++) fp2_records[i].clear(), fp2_ts[i].clear(), fp2io_rows[i].clear();
        n_instr = 0;
    }
};

// a touched block of the current segment (8 cells = 4 words) and what the close of a segment works on
struct MemBlock {
    uint32_t init[8], cur[8], ts[4];   // cells 2 j, 2 j + 1 = the halves of word j; ts[j] = the word's last access in this segment
};
struct CloseJob {
    std::vector<MemBlock> blk;      // in the order of first touch, final cells filled in
    std::vector<uint32_t> label;
    size_t n_path = 0;              // internal nodes above them
};

// What the METERED first pass knows about a segment (ParallelSegmentExecutor below; the role of OpenVM's metered execution before its
// per-segment preflight runs, reached by the reference through execute_metered_cost, crates/prover/src/utils/vm.rs:19): where it starts (the
// machine's state; the memory pages written since the previous cut), where it ends (n_instr), and everything of the segment's records that
// hangs on the PERSISTENT memory tree -- the one chain that runs through all segments: touched blocks (sorted), leaf rows (their timestamp
// columns are the record pass's), path-node rows, hash requests, the roots.
struct SegmentPlan {
    size_t index = 0;
    Machine::State start;
    std::shared_ptr<const std::vector<Machine::Page>> pages;   // written during the PREVIOUS segment (none for segment 0)
    size_t n_instr = 0;
    bool last = false;
    uint32_t pc_start = 0, pc_end = 0;
    Digest root_init{}, root_final{};
    std::vector<uint32_t> leaf_rows, merkle_rows, p2_inputs;
    CloseJob job;   // between the metered run and its close (SegmentExecutor::meter_run / meter_close)
};

class SegmentExecutor {
public:
    // replay_only: a record-pass worker of ParallelSegmentExecutor -- it never owns the persistent tree (the metered pass does)
    SegmentExecutor(const Exe& exe, const StdIn& in, const SegmentCaps& caps, bool replay_only = false) : exe_(exe), m_(exe, in), caps_(caps) {
        if (exe.data_base % 16) throw Error("the one-statement flow needs a 16-byte aligned data base");
        if ((uint64_t)exe.data_base + exe.memory_bytes > (1ull << 30) || (uint64_t)exe.pc_base + 4 * exe.program.size() > (1ull << 30))
            throw Error("guest addresses must lie below 2^30");
        m_.set_memory_log(false);
        m_.set_moduli(caps.moduli);
        m_.set_curves(caps.curves);
        m_.set_fp2_moduli(caps.fp2_moduli);
        m_.set_native(caps.ext);
        dec_.reserve(exe.program.size());
        for (size_t k = 0; k < exe.program.size(); k++) dec_.push_back(vmc::decode(exe.program[k], exe.pc_base + 4 * (uint32_t)k));
        for (const vmc::Decoded& d : dec_)   // the register BLOCKS (four registers each) an instruction touches, as a mask
            dec_regmask_.push_back((uint8_t)((d.use_rs1 ? 1u << (d.rs1 >> 2) : 0u) | (d.use_rs2 ? 1u << (d.rs2 >> 2) : 0u) | (d.wr_rd ? 1u << (d.rd >> 2) : 0u)));
        mem_lo_blk_ = exe.data_base / 16;
        mem_slot_.assign(((size_t)exe.memory_bytes + 15) / 16 + 1, 0);
        if (replay_only) return;
        // the initial memory image: registers (sp), the data segment
        std::vector<std::pair<uint32_t, std::array<uint32_t, 8>>> blocks;
        for (uint32_t blk = 0; blk < 8; blk++) push_nonzero(blocks, 1, blk);
        const uint32_t first = exe.data_base / 16, last = (uint32_t)(((uint64_t)exe.data_base + exe.data.size() + 15) / 16);
        for (uint32_t blk = first; blk < last; blk++) push_nonzero(blocks, 2, blk);
        tree_.set_blocks(blocks);
        image_root_ = tree_.root();
    }
    const Digest& image_root() const { return image_root_; }
    const MemoryTree& tree() const { return tree_; }
    bool done() const { return m_.done(); }
    uint64_t instret() const { return m_.instret(); }
    uint32_t exit_code() const { return m_.exit_code(); }
    const std::vector<uint8_t>& public_values() const { return m_.public_values(); }
    uint32_t peek_memory(uint32_t addr) const { return mem_word(addr); }
    double close_seconds() const { return close_seconds_; }   // of run_segment's time: hashing the touched blocks and their paths, leaf / merkle rows   // a word of the guest's memory as it is now (zero beyond its end)

    // Executes the next segment into `r`.  Returns true once the guest has exited (then r.pc_end = 0).
    // words of the buffer of an intrinsic call (a7 = call, a1 = index + 8 op): operands of 8 words, or 12 for a modulus above 2^256
    uint32_t call_words(uint32_t call, uint32_t a1) const {
        const uint32_t i = a1 & 7u;
        auto nw = [](const zkhip::modular::UInt& p) { return (uint32_t)zkhip::modular::words_of(p); };
        if (call == 5) return 3 * (i < caps_.moduli.size() ? nw(caps_.moduli[i]) : 8u);
        if (call == 7) return 6 * (i < caps_.curves.size() ? nw(caps_.curves[i].p) : 8u);
        if (call == 8) return 6 * (i < caps_.fp2_moduli.size() ? nw(caps_.fp2_moduli[i]) : 8u);
        if (call == zkhip::native::CALL_ARITH) return 3;
        if (call == zkhip::native::CALL_EXT) return 12;
        if (call == zkhip::native::CALL_CASTF) return 2;
        return 24;   // (int256)
    }
    bool run_segment(SegmentRecords& r) { return run_impl<SERIAL>(r, nullptr); }
    // The metered pass: the next segment's cut and its memory-tree records into `plan`, no execution records (the machine runs without a
    // record sink).  track_dirty(true) on the machine before the first call.
    bool meter_segment(SegmentPlan& plan) {
        const bool done = meter_run(plan);
        meter_close(plan);
        return done;
    }
    // ... in two halves that may run on two threads, one plan behind the other: the RUN (the machine, the touched blocks) and the CLOSE (the
    // persistent tree: hashing the touched blocks and their paths, leaf / path-node rows) -- the tree is the close's alone
    bool meter_run(SegmentPlan& plan) {
        plan.start = m_.save_state();
        plan.pages = std::make_shared<const std::vector<Machine::Page>>(m_.take_dirty_pages());
        const bool done = run_impl<METER>(meter_scratch_, &plan);
        plan.n_instr = meter_scratch_.n_instr, plan.last = done;
        plan.pc_start = meter_scratch_.pc_start, plan.pc_end = meter_scratch_.pc_end;
        return done;
    }
    void meter_close(SegmentPlan& plan) {
        const auto t0 = std::chrono::steady_clock::now();
        SegmentRecords& r = close_scratch_;
        r.leaf_rows.clear(), r.merkle_rows.clear(), r.p2_inputs.clear();
        plan.root_init = tree_.root();
        close_job(plan.job, r);
        plan.root_final = r.root_final;
        plan.leaf_rows.swap(r.leaf_rows), plan.merkle_rows.swap(r.merkle_rows), plan.p2_inputs.swap(r.p2_inputs);
        plan.job = CloseJob();
        close_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // The record pass of one planned segment: the machine is put where the plan starts (its memory must already be the memory of that cut:
    // machine().apply_pages of every plan since the one this executor last replayed) and runs plan.n_instr instructions into `r`.
    bool replay_segment(const SegmentPlan& plan, SegmentRecords& r) {
        m_.restore_state(plan.start);
        return run_impl<REPLAY>(r, const_cast<SegmentPlan*>(&plan));
    }
    Machine& machine() { return m_; }

private:
    enum Mode { SERIAL = 0, METER = 1, REPLAY = 2 };
    SegmentRecords meter_scratch_, close_scratch_;
    template <int MODE>
    bool run_impl(SegmentRecords& r, SegmentPlan* plan) {
        r.clear(/*keep_frame=*/MODE != METER);
        reset_blocks();
        const size_t frame_rows = MODE == METER ? 0 : caps_.rows(vmc::A_FRAME);
        // room for a whole frame in every array the interpreter appends to (an instruction leaves at most one entry per array): its appends
        // then skip the capacity check (ExecRecords::reserved, U32Vec::put_unchecked).  Address space only: pages are touched as rows are written.
        for (zkhip_vm::U32Vec* v : {&r.core.pc_index, &r.core.alu_op, &r.core.alu_b, &r.core.alu_c, &r.core.lt_op, &r.core.lt_b, &r.core.lt_c, &r.core.mul_b, &r.core.mul_c,
                                    &r.core.mulh_op, &r.core.mulh_b, &r.core.mulh_c, &r.core.div_op, &r.core.div_b, &r.core.div_c, &r.core.shift_op, &r.core.shift_b,
                                    &r.core.shift_c, &r.core.beq_op, &r.core.beq_a, &r.core.beq_b, &r.core.beq_imm, &r.core.blt_op, &r.core.blt_a, &r.core.blt_b,
                                    &r.core.blt_imm, &r.core.jal_op, &r.core.jal_pc, &r.core.jal_imm, &r.core.auipc_pc, &r.core.auipc_imm, &r.core.jalr_pc,
                                    &r.core.jalr_rs1, &r.core.jalr_imm, &r.core.ls_case, &r.core.ls_read, &r.core.ls_prev})
            if (MODE != METER && v->capacity() < frame_rows + 8) v->reserve(frame_rows + 8);
        r.core.reserved = MODE != METER;
        for (auto* v : {&r.f_x, &r.f_y, &r.f_z, &r.f_rdprev, &r.f_pcinc, &r.f_pts1, &r.f_pts2, &r.f_pts3}) v->resize(frame_rows);
        uint32_t *const fx = r.f_x.data(), *const fy = r.f_y.data(), *const fz = r.f_z.data(), *const frd = r.f_rdprev.data(), *const fpc = r.f_pcinc.data();
        uint32_t *const fp1 = r.f_pts1.data(), *const fp2 = r.f_pts2.data(), *const fp3 = r.f_pts3.data();
        struct CutFrame {   // (also when an instruction throws: the arrays never hold rows beyond n_instr)
            SegmentRecords& r;
            ~CutFrame() {
                for (auto* v : {&r.f_x, &r.f_y, &r.f_z, &r.f_rdprev, &r.f_pcinc, &r.f_pts1, &r.f_pts2, &r.f_pts3}) v->resize(MODE == METER ? 0 : r.n_instr);
                r.core.reserved = 0;   // (the promise of room ends with the segment: ADVICE round 5)
            }
        } cut_frame{r};
        r.pc_start = m_.pc();
        if (MODE == REPLAY) r.root_init = plan->root_init;
        if (MODE == SERIAL) r.root_init = tree_.root();   // (the metered pass leaves the tree to meter_close, which may be another thread)
        size_t n_cls[vmc::N_CLS] = {};
        size_t rows_cls[vmc::N_CLS];
        for (unsigned c = 0; c < vmc::N_CLS; c++) rows_cls[c] = caps_.rows(vmc::A_ALU + c);
        const size_t rows_frame = caps_.rows(vmc::A_FRAME);
        while (!m_.done()) {
            const uint32_t pc = m_.pc();
            const uint64_t k = ((uint64_t)pc - exe_.pc_base) / 4;
            if (pc < exe_.pc_base || (pc & 3u) || k >= exe_.program.size()) throw Error("pc outside the program: " + std::to_string(pc));
            const vmc::Decoded& d = dec_[k];
            if (!d.legal) throw Error("illegal instruction at pc " + std::to_string(pc));
            // The metered pass's common case: no environment call, every block the instruction touches is one the segment has already
            // touched (so it adds no block, no path node, no hash request -- the counts below cannot grow) and its class has a row left:
            // the instruction runs and is counted.  Everything else takes the general path, which is the serial executor's.
            if (MODE == METER && d.cls != vmc::C_ECALL && !(dec_regmask_[k] & ~reg_known_) && r.n_instr < rows_frame && n_cls[d.cls] < rows_cls[d.cls] &&
                (d.cls != vmc::C_LS || mem_block_known(((d.use_rs1 ? m_.reg(d.rs1) : 0) + (d.imm_lo | (d.imm_hi << 16))) >> 4))) {
                m_.step_one(nullptr);
                n_cls[d.cls]++, r.n_instr++;
                continue;
            }
            // would one more instruction overflow a chip?  (the blocks and path nodes it would add are counted exactly)
            // (a record pass does not ask: it ends where the metered pass cut)
            if (MODE == REPLAY && r.n_instr == plan->n_instr) break;
            size_t new_blocks = 0, new_nodes = 0;
            if (MODE != REPLAY) {
                uint32_t labels[6 + 20];
                unsigned nl = 0;
                bool all_known = true;
                auto want = [&](uint32_t as, uint32_t word) {
                    labels[nl] = (as << vmc::LABEL_BITS) | (word >> 2);
                    all_known = all_known && touched(labels[nl]);
                    nl++;
                };
                if (d.use_rs1) want(1, d.rs1);
                if (d.use_rs2) want(1, d.rs2);
                if (d.wr_rd) want(1, d.rd);
                if (d.cls == vmc::C_LS) want(2, ((d.use_rs1 ? m_.reg(d.rs1) : 0) + (d.imm_lo | (d.imm_hi << 16))) >> 2);
                if (d.cls == vmc::C_ECALL) want(1, 11), want(3, m_.reg(11) & 7u);
                if (d.cls == vmc::C_ECALL && m_.reg(17) == 6 && zkhip::int256::is_branch_op(m_.reg(11))) want(1, 12);   // (a 256-bit branch reads its offset from a2)
                if (d.cls == vmc::C_ECALL && m_.reg(17) >= 3 && m_.reg(17) <= 11) {
                    const uint32_t n_words = m_.reg(17) == 3 ? 50 : call_words(m_.reg(17), m_.reg(11));
                    for (uint32_t w = m_.reg(10) >> 2; w < (m_.reg(10) >> 2) + n_words; w += (w & 3u) ? 4 - (w & 3u) : 4) want(2, w);
                }
                if (!all_known) {   // rare: count exactly what the instruction would add
                    std::unordered_set<uint64_t> fresh;
                    for (unsigned i = 0; i < nl; i++) {
                        if (find_block(labels[i]) || !fresh.insert(MemoryTree::key(vmc::LEAF_LEVEL, labels[i])).second) continue;
                        new_blocks++;
                        uint32_t idx = labels[i];
                        for (int l = (int)vmc::LEAF_LEVEL - 1; l >= 0; l--) {
                            idx >>= 1;
                            const uint64_t kk = MemoryTree::key((unsigned)l, idx);
                            if (path_nodes_.count(kk) || !fresh.insert(kk).second) break;
                            new_nodes++;
                        }
                    }
                }
            }
            if (d.cls == vmc::C_ECALL && m_.reg(17) == 3 && !caps_.keccak())
                throw Error("the guest calls the keccak intrinsic but the app's openvm.toml does not enable the keccak extension");
            if (d.cls == vmc::C_ECALL && m_.reg(17) == 4 && !caps_.sha256())
                throw Error("the guest calls the sha256 intrinsic but the app's openvm.toml does not enable the sha256 extension");
            if (d.cls == vmc::C_ECALL && m_.reg(17) == 6 && !caps_.int256())
                throw Error("the guest calls the int256 intrinsic but the app's openvm.toml does not enable the bigint extension");
            if (d.cls == vmc::C_ECALL && m_.reg(17) == 5 && ((m_.reg(11) & 7u) >= caps_.moduli.size() || (m_.reg(11) >> 3) >= zkhip::modular::N_OPS))
                throw Error("the guest calls the modular intrinsic for modulus " + std::to_string(m_.reg(11) & 7u) + " but the app's openvm.toml lists " +
                            std::to_string(caps_.moduli.size()) + " moduli");
            if (d.cls == vmc::C_ECALL && m_.reg(17) == 7 && ((m_.reg(11) & 7u) >= caps_.curves.size() || (m_.reg(11) >> 3) >= zkhip::ecc::N_OPS))
                throw Error("the guest calls the ecc intrinsic for curve " + std::to_string(m_.reg(11) & 7u) + " but the app's openvm.toml lists " +
                            std::to_string(caps_.curves.size()) + " curves");
            if (d.cls == vmc::C_ECALL && m_.reg(17) == 8 && ((m_.reg(11) & 7u) >= caps_.fp2_moduli.size() || (m_.reg(11) >> 3) >= zkhip::fp2::N_OPS))
                throw Error("the guest calls the fp2 intrinsic for field " + std::to_string(m_.reg(11) & 7u) + " but the app's openvm.toml lists " +
                            std::to_string(caps_.fp2_moduli.size()) + " fp2 fields");
            if (d.cls == vmc::C_ECALL && (m_.reg(17) == zkhip::native::CALL_ARITH || m_.reg(17) == zkhip::native::CALL_EXT) && !caps_.native())
                throw Error("the guest calls a native field intrinsic but the app's openvm.toml does not enable the native extension");
            if (d.cls == vmc::C_ECALL && m_.reg(17) == zkhip::native::CALL_CASTF && !caps_.castf())
                throw Error("the guest calls the castf intrinsic but the app's openvm.toml does not enable the castf extension");
            if (MODE != REPLAY && (r.n_instr + 1 > caps_.rows(vmc::A_FRAME) || n_cls[d.cls] + 1 > caps_.rows(vmc::A_ALU + d.cls) ||
                blk_.size() + new_blocks > caps_.rows(vmc::A_LEAF) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == 3 && 24 * (r.n_keccak() + 1) > caps_.rows(vmc::A_KECCAK)) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == 4 && 65 * (r.n_sha256() + 1) > caps_.rows(vmc::A_SHA256)) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == 5 && r.mm_ts[m_.reg(11) & 7u].size() + 1 > caps_.rows(vmc::A_MODMUL(m_.reg(11) & 7u))) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == 7 && r.ec_ts[m_.reg(11) & 7u].size() + 1 > caps_.rows(vmc::A_EC(m_.reg(11) & 7u))) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == 8 && r.fp2_ts[m_.reg(11) & 7u].size() + 1 > caps_.rows(vmc::A_FP2(m_.reg(11) & 7u))) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == 6 && (m_.reg(11) == 5 ? r.mul256_ts.size() : m_.reg(11) >= 12 ? r.cmp256_ts.size() : m_.reg(11) > 8 ? r.sh256_ts.size() : m_.reg(11) > 5 ? r.cmp256_ts.size() : r.i256_ts.size()) + 1 > caps_.rows(vmc::A_INT256)) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == zkhip::native::CALL_ARITH && r.n_nat() + 1 > caps_.rows(vmc::A_NATIVE_ARITH)) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == zkhip::native::CALL_EXT && r.n_next() + 1 > caps_.rows(vmc::A_NATIVE_EXT)) ||
                (d.cls == vmc::C_ECALL && m_.reg(17) == zkhip::native::CALL_CASTF && r.n_castf() + 1 > caps_.rows(vmc::A_CASTF)) ||
                path_nodes_.size() + new_nodes > caps_.rows(vmc::A_MERKLE) ||
                2 * (blk_.size() + new_blocks + path_nodes_.size() + new_nodes) > caps_.rows(vmc::A_POSEIDON2))) {
                if (r.n_instr == 0) throw Error("segment heights too small for a single instruction");
                break;
            }
            const uint32_t ts = 1 + vmc::TS_STEP * (uint32_t)r.n_instr;
            const uint32_t x = d.use_rs1 ? m_.reg(d.rs1) : 0;
            const uint32_t imm32 = d.imm_lo | (d.imm_hi << 16);
            const uint32_t y = d.use_rs2 ? m_.reg(d.rs2) : d.y_is_imm ? imm32 : 0;
            const uint32_t rd_prev = m_.reg(d.rd);
            // memory cells of a load / store, before the instruction
            uint32_t addr = 0, word_before = 0;
            if (d.cls == vmc::C_LS) addr = x + imm32, word_before = mem_word(addr & ~3u);
            const uint32_t pts1 = d.use_rs1 ? read_word(1, d.rs1, x, ts) : 0;
            const uint32_t pts2 = d.use_rs2 ? read_word(1, d.rs2, y, ts + 2) : 0;
            uint32_t a1 = 0, pv_before = 0;
            if (d.cls == vmc::C_ECALL && x == 1) {
                a1 = m_.reg(11);
                if (a1 >= 8) throw Error("public value index out of range");
                memcpy(&pv_before, m_.public_values().data() + 4 * a1, 4);
            }
            // (a call's buffer: 3 nw words for the modular intrinsic, 6 nw for ecc / fp2 -- nw = 8 or 12 words per operand)
            uint32_t mm_in[36];
            const uint32_t cw = d.cls == vmc::C_ECALL && x >= 5 && x <= 8 ? call_words(x, m_.reg(11)) : 0u, nw = x == 5 || x == 6 ? cw / 3 : cw / 6;
            if (d.cls == vmc::C_ECALL && (x == 5 || x == 6)) {
                a1 = m_.reg(11);
                if ((y & 3u) || (uint64_t)y + 4 * cw > (1ull << 30)) throw Error("modmul / int256 buffer must be word-aligned and below 2^30");
                for (uint32_t j = 0; j < cw; j++) mm_in[j] = mem_word(y + 4 * j), (void)block_of(2, (y >> 2) + j);
            }
            uint32_t ec_in[72];
            if (d.cls == vmc::C_ECALL && (x == 7 || x == 8)) {
                a1 = m_.reg(11);
                if ((y & 3u) || (uint64_t)y + 4 * cw > (1ull << 30)) throw Error("ecc / fp2 buffer must be word-aligned and below 2^30");
                for (uint32_t j = 0; j < cw; j++) ec_in[j] = mem_word(y + 4 * j), (void)block_of(2, (y >> 2) + j);
            }
            uint32_t nat_in[12];
            const uint32_t nat_words = d.cls == vmc::C_ECALL && x >= zkhip::native::CALL_ARITH && x <= zkhip::native::CALL_CASTF ? call_words(x, 0) : 0u;
            if (nat_words) {
                a1 = m_.reg(11);
                if ((y & 3u) || (uint64_t)y + 4 * nat_words > (1ull << 30)) throw Error("native / castf buffer must be word-aligned and below 2^30");
                for (uint32_t j = 0; j < nat_words; j++) nat_in[j] = mem_word(y + 4 * j), (void)block_of(2, (y >> 2) + j);
            }
            // snapshot the blocks this instruction is going to change
            if (d.wr_rd) (void)block_of(1, d.rd);
            if (d.cls == vmc::C_LS) (void)block_of(2, addr >> 2);
            if (d.cls == vmc::C_ECALL && x == 1) (void)block_of(3, a1);
            uint32_t kk_in[50];
            if (d.cls == vmc::C_ECALL && x == 3) {
                if ((y & 3u) || (uint64_t)y + 200 > (1ull << 30)) throw Error("keccak state must be word-aligned and below 2^30");
                for (uint32_t j = 0; j < 50; j++) kk_in[j] = mem_word(y + 4 * j), (void)block_of(2, (y >> 2) + j);
            }
            uint32_t sha_in[24];
            if (d.cls == vmc::C_ECALL && x == 4) {
                if ((y & 3u) || (uint64_t)y + 96 > (1ull << 30)) throw Error("sha256 buffer must be word-aligned and below 2^30");
                for (uint32_t j = 0; j < 24; j++) sha_in[j] = mem_word(y + 4 * j), (void)block_of(2, (y >> 2) + j);
            }
            const size_t n_ls_before = r.core.ls_case.size();
            m_.step_one(MODE == METER ? nullptr : &r.core);
            uint32_t z = 0, pc_inc;
            if (MODE == METER && d.cls == vmc::C_ECALL) {
                // the metered pass keeps the COUNTS the cut rule reads (one entry per call), none of the rows
                if (m_.done() && m_.exit_code()) throw Error("guest exited with code " + std::to_string(m_.exit_code()));
                z = x == 2 ? m_.last_result() : y;
                pc_inc = 4;
                if (x == 3) r.kk_ts.push_back(ts);
                if (x == 4) r.sha_ts.push_back(ts);
                if (x == 5) r.mm_ts[a1 & 7u].push_back(ts);
                if (x == 6) (a1 == 5 ? r.mul256_ts : a1 >= 12 ? r.cmp256_ts : a1 > 8 ? r.sh256_ts : a1 > 5 ? r.cmp256_ts : r.i256_ts).push_back(ts);
                // (the register blocks the ecall chip's own reads touch -- a1 for the calls with a second argument, a2 for a 256-bit branch --
                // are touched here as the record pass touches them: the two passes must agree on the segment's blocks)
                if (x == 1 || (x >= 5 && x <= 8) || x == zkhip::native::CALL_ARITH || x == zkhip::native::CALL_EXT) (void)read_word(1, 11, a1, ts + 4);
                if (x == 6 && zkhip::int256::is_branch_op(a1)) (void)read_word(1, 12, 0, ts + 6);
                if (x == 7) r.ec_ts[a1 & 7u].push_back(ts);
                if (x == 8) r.fp2_ts[a1 & 7u].push_back(ts);
                if (x == zkhip::native::CALL_ARITH) r.nat_records.resize(r.nat_records.size() + SegmentRecords::NAT_RECORD);
                if (x == zkhip::native::CALL_EXT) r.next_records.resize(r.next_records.size() + SegmentRecords::NEXT_RECORD);
                if (x == zkhip::native::CALL_CASTF) r.castf_records.resize(r.castf_records.size() + SegmentRecords::CASTF_RECORD);
            } else if (d.cls == vmc::C_ECALL) {
                z = x == 2 ? m_.last_result() : y;
                pc_inc = m_.done() ? vmc::field_of(-(int32_t)pc) : vmc::field_of((int32_t)(m_.pc() - pc));   // (4, or a taken 256-bit branch's offset)
                if (m_.done() && m_.exit_code()) throw Error("guest exited with code " + std::to_string(m_.exit_code()));
                uint32_t row[vmc::ECALL_WIDTH] = {};
                row[0] = pc, row[1] = ts;
                for (int i = 0; i < 4; i++) row[2 + i] = (x >> (8 * i)) & 255u, row[6 + i] = (y >> (8 * i)) & 255u, row[10 + i] = (z >> (8 * i)) & 255u;
                row[14] = x == 93, row[15] = x == 1, row[16] = x == 2, row[20] = pc_inc, row[27] = x == 3;
                if (x == 3) {
                    row[28] = (y & 255u) >> 2;
                    r.kk_ts.push_back(ts);
                    r.kk_states.insert(r.kk_states.end(), kk_in, kk_in + 50);
                    for (uint32_t lane = 0; lane < 25; lane++) {   // the adapter's rows: both words of every lane replaced at ts + 4
                        uint32_t kr[vmc::KECCAK_IO_WIDTH] = {};
                        kr[lane] = 1, kr[25] = ts, kr[26] = y >> 2, kr[41] = 1;
                        for (uint32_t h = 0; h < 2; h++) {
                            const uint32_t w = (y >> 2) + 2 * lane + h, before = kk_in[2 * lane + h], after = mem_word(y + 4 * (2 * lane + h));
                            kr[27 + 2 * h] = before & 0xffffu, kr[28 + 2 * h] = before >> 16, kr[31 + 2 * h] = after & 0xffffu, kr[32 + 2 * h] = after >> 16;
                            const uint32_t pts = write_word(2, w, after, ts + 4), gap = ts + 4 - pts - 1;
                            kr[35 + 3 * h] = pts, kr[36 + 3 * h] = gap & 0xffffu, kr[37 + 3 * h] = gap >> 16;
                        }
                        r.kio_rows.insert(r.kio_rows.end(), kr, kr + vmc::KECCAK_IO_WIDTH);
                    }
                }
                if (x == 1) {
                    row[17] = a1, row[18] = pv_before & 0xffffu, row[19] = pv_before >> 16;
                    auto gap = [&](uint32_t* dst, uint32_t pts, uint32_t at) {
                        dst[0] = pts, dst[1] = (at - pts - 1) & 0xffffu, dst[2] = (at - pts - 1) >> 16;
                    };
                    gap(row + 21, read_word(1, 11, a1, ts + 4), ts + 4);
                    gap(row + 24, write_word(3, a1, y, ts + 5), ts + 5);
                }
                if (x == 6) {
                    row[31] = 1, row[28] = (y & 255u) >> 2, row[17] = a1;
                    auto gap3 = [&](uint32_t* dst, uint32_t pts, uint32_t at) { dst[0] = pts, dst[1] = (at - pts - 1) & 0xffffu, dst[2] = (at - pts - 1) >> 16; };
                    gap3(row + 21, read_word(1, 11, a1, ts + 4), ts + 4);
                    const bool branch = zkhip::int256::is_branch_op(a1);   // (12 .. 17: the comparison chip decides a branch)
                    auto& recs = a1 == 5 ? r.mul256_records : branch ? r.cmp256_records : a1 > 8 ? r.sh256_records : a1 > 5 ? r.cmp256_records : r.i256_records;
                    (a1 == 5 ? r.mul256_ts : branch ? r.cmp256_ts : a1 > 8 ? r.sh256_ts : a1 > 5 ? r.cmp256_ts : r.i256_ts).push_back(ts);
                    if (branch) {
                        const uint32_t off = m_.reg(12);
                        row[37] = 1, row[38] = zkhip::int256::branch256_taken(a1, mm_in, mm_in + 8) ? 1u : 0u;   // (mm_in: b | c | a as they were before the call)
                        row[39] = off & 0xffffu, row[40] = off >> 16, row[41] = off >> 31;
                        gap3(row + 42, read_word(1, 12, off, ts + 6), ts + 6);
                    }
                    recs.push_back(a1);
                    recs.insert(recs.end(), mm_in, mm_in + 16);
                    for (uint32_t k = 0; k < 24; k++) {   // the adapter's rows: b, c read, a written, all at ts + 5
                        uint32_t sr[vmc::INT256_IO_WIDTH] = {};
                        const uint32_t w = (y >> 2) + k, before = mm_in[k], after = mem_word(y + 4 * k);
                        sr[k] = 1, sr[24] = ts, sr[25] = y >> 2, sr[33] = 1, sr[34] = a1;
                        sr[26] = before & 0xffffu, sr[27] = before >> 16, sr[28] = after & 0xffffu, sr[29] = after >> 16;
                        const uint32_t pts = k >= 16 ? write_word(2, w, after, ts + 5) : read_word(2, w, before, ts + 5), gap = ts + 5 - pts - 1;
                        sr[30] = pts, sr[31] = gap & 0xffffu, sr[32] = gap >> 16;
                        r.i256io_rows.insert(r.i256io_rows.end(), sr, sr + vmc::INT256_IO_WIDTH);
                    }
                }
                if (x == 7) {
                    row[32] = 1, row[28] = (y & 255u) >> 2, row[17] = a1;
                    auto gap3 = [&](uint32_t* dst, uint32_t pts, uint32_t at) { dst[0] = pts, dst[1] = (at - pts - 1) & 0xffffu, dst[2] = (at - pts - 1) >> 16; };
                    gap3(row + 21, read_word(1, 11, a1, ts + 4), ts + 4);
                    const uint32_t ci = a1 & 7u, eop = a1 >> 3;
                    r.ec_ts[ci].push_back(ts);
                    r.ec_records[ci].push_back(eop);
                    r.ec_records[ci].insert(r.ec_records[ci].end(), ec_in, ec_in + 4 * nw);
                    r.ec_records[ci].insert(r.ec_records[ci].end(), m_.last_slope().w, m_.last_slope().w + nw);
                    const uint32_t W = cw, IOW = (uint32_t)vmc::ec_io_width(nw);
                    for (uint32_t k = 0; k < W; k++) {   // the adapter's rows: the operands read, the result written, all at ts + 5
                        uint32_t sr[vmc::ec_io_width(12)] = {};
                        const uint32_t w = (y >> 2) + k, before = ec_in[k], after = mem_word(y + 4 * k);
                        sr[k] = 1, sr[W] = ts, sr[W + 1] = y >> 2, sr[W + 9] = 1, sr[W + 10] = eop;
                        sr[W + 2] = before & 0xffffu, sr[W + 3] = before >> 16, sr[W + 4] = after & 0xffffu, sr[W + 5] = after >> 16;
                        const uint32_t pts = k >= 4 * nw ? write_word(2, w, after, ts + 5) : read_word(2, w, before, ts + 5), gap = ts + 5 - pts - 1;
                        sr[W + 6] = pts, sr[W + 7] = gap & 0xffffu, sr[W + 8] = gap >> 16;
                        r.ecio_rows[ci].insert(r.ecio_rows[ci].end(), sr, sr + IOW);
                    }
                }
                if (x == 8) {
                    row[33] = 1, row[28] = (y & 255u) >> 2, row[17] = a1;
                    auto gap3 = [&](uint32_t* dst, uint32_t pts, uint32_t at) { dst[0] = pts, dst[1] = (at - pts - 1) & 0xffffu, dst[2] = (at - pts - 1) >> 16; };
                    gap3(row + 21, read_word(1, 11, a1, ts + 4), ts + 4);
                    const uint32_t fi = a1 & 7u, fop = a1 >> 3;
                    r.fp2_ts[fi].push_back(ts);
                    r.fp2_records[fi].push_back(fop);
                    if (fop == zkhip::fp2::OP_DIV) {   // the chip's row is the product (x / y) y = x: the record holds the quotient and y
                        for (uint32_t k = 0; k < 2 * nw; k++) r.fp2_records[fi].push_back(mem_word(y + 16 * nw + 4 * k));
                        r.fp2_records[fi].insert(r.fp2_records[fi].end(), ec_in + 2 * nw, ec_in + 4 * nw);
                    } else {
                        r.fp2_records[fi].insert(r.fp2_records[fi].end(), ec_in, ec_in + 4 * nw);
                    }
                    const uint32_t W = cw, IOW = (uint32_t)vmc::ec_io_width(nw);
                    for (uint32_t k = 0; k < W; k++) {   // the adapter's rows: the operands read, the result written, all at ts + 5
                        uint32_t sr[vmc::ec_io_width(12)] = {};
                        const uint32_t w = (y >> 2) + k, before = ec_in[k], after = mem_word(y + 4 * k);
                        sr[k] = 1, sr[W] = ts, sr[W + 1] = y >> 2, sr[W + 9] = 1, sr[W + 10] = fop;
                        sr[W + 2] = before & 0xffffu, sr[W + 3] = before >> 16, sr[W + 4] = after & 0xffffu, sr[W + 5] = after >> 16;
                        const uint32_t pts = k >= 4 * nw ? write_word(2, w, after, ts + 5) : read_word(2, w, before, ts + 5), gap = ts + 5 - pts - 1;
                        sr[W + 6] = pts, sr[W + 7] = gap & 0xffffu, sr[W + 8] = gap >> 16;
                        r.fp2io_rows[fi].insert(r.fp2io_rows[fi].end(), sr, sr + IOW);
                    }
                }
                if (nat_words) {   // native field / extension / castf: one record per call; the chip's row is made on the device
                    row[x == zkhip::native::CALL_ARITH ? 34 : x == zkhip::native::CALL_EXT ? 35 : 36] = 1, row[28] = (y & 255u) >> 2;
                    if (x != zkhip::native::CALL_CASTF) {
                        row[17] = a1;
                        const uint32_t pts = read_word(1, 11, a1, ts + 4), gap = ts + 4 - pts - 1;
                        row[21] = pts, row[22] = gap & 0xffffu, row[23] = gap >> 16;
                    }
                    const uint32_t n_rd = x == zkhip::native::CALL_ARITH ? 2 : x == zkhip::native::CALL_EXT ? 8 : 1, n_wr = nat_words - n_rd;
                    auto& rec = x == zkhip::native::CALL_ARITH ? r.nat_records : x == zkhip::native::CALL_EXT ? r.next_records : r.castf_records;
                    if (x != zkhip::native::CALL_CASTF) rec.push_back(a1);
                    rec.insert(rec.end(), nat_in, nat_in + nat_words);   // operands, then the result words as they were before
                    rec.push_back(y >> 2), rec.push_back(ts);
                    for (uint32_t k = 0; k < n_rd; k++) rec.push_back(read_word(2, (y >> 2) + k, nat_in[k], ts + 5));
                    for (uint32_t k = 0; k < n_wr; k++) rec.push_back(write_word(2, (y >> 2) + n_rd + k, mem_word(y + 4 * (n_rd + k)), ts + 5));
                }
                if (x == 5) {
                    row[30] = 1, row[28] = (y & 255u) >> 2, row[17] = a1;
                    auto gap3 = [&](uint32_t* dst, uint32_t pts, uint32_t at) { dst[0] = pts, dst[1] = (at - pts - 1) & 0xffffu, dst[2] = (at - pts - 1) >> 16; };
                    gap3(row + 21, read_word(1, 11, a1, ts + 4), ts + 4);
                    const uint32_t mi = a1 & 7u, mop = a1 >> 3;
                    r.mm_ts[mi].push_back(ts);
                    r.mm_records[mi].push_back(mop);
                    if (mop == zkhip::modular::OP_DIV) {   // the chip's row is the product (x / y) y = x: the record holds the quotient and y
                        for (uint32_t k = 0; k < nw; k++) r.mm_records[mi].push_back(mem_word(y + 8 * nw + 4 * k));
                        r.mm_records[mi].insert(r.mm_records[mi].end(), mm_in + nw, mm_in + 2 * nw);
                    } else {
                        r.mm_records[mi].insert(r.mm_records[mi].end(), mm_in, mm_in + 2 * nw);
                    }
                    const uint32_t W = cw, IOW = (uint32_t)vmc::modmul_io_width(nw);
                    for (uint32_t k = 0; k < W; k++) {   // the adapter's rows: a, b read, r written, all at ts + 5
                        uint32_t sr[vmc::modmul_io_width(12)] = {};
                        const uint32_t w = (y >> 2) + k, before = mm_in[k], after = mem_word(y + 4 * k);
                        sr[k] = 1, sr[W] = ts, sr[W + 1] = y >> 2, sr[W + 9] = 1, sr[W + 10] = mop;
                        sr[W + 2] = before & 0xffffu, sr[W + 3] = before >> 16, sr[W + 4] = after & 0xffffu, sr[W + 5] = after >> 16;
                        const uint32_t pts = k >= 2 * nw ? write_word(2, w, after, ts + 5) : read_word(2, w, before, ts + 5), gap = ts + 5 - pts - 1;
                        sr[W + 6] = pts, sr[W + 7] = gap & 0xffffu, sr[W + 8] = gap >> 16;
                        r.mmio_rows[mi].insert(r.mmio_rows[mi].end(), sr, sr + IOW);
                    }
                }
                if (x == 4) {
                    row[29] = 1, row[28] = (y & 255u) >> 2;
                    r.sha_ts.push_back(ts);
                    r.sha_blocks.insert(r.sha_blocks.end(), sha_in, sha_in + 24);
                    for (uint32_t k = 0; k < 24; k++) {   // the adapter's rows: state words replaced, message words read, all at ts + 4
                        uint32_t sr[vmc::SHA_IO_WIDTH] = {};
                        const uint32_t w = (y >> 2) + k, before = sha_in[k], after = mem_word(y + 4 * k);
                        sr[k] = 1, sr[24] = ts, sr[25] = y >> 2, sr[33] = 1;
                        sr[26] = before & 0xffffu, sr[27] = before >> 16, sr[28] = after & 0xffffu, sr[29] = after >> 16;
                        const uint32_t pts = k < 8 ? write_word(2, w, after, ts + 4) : read_word(2, w, before, ts + 4), gap = ts + 4 - pts - 1;
                        sr[30] = pts, sr[31] = gap & 0xffffu, sr[32] = gap >> 16;
                        r.shaio_rows.insert(r.shaio_rows.end(), sr, sr + vmc::SHA_IO_WIDTH);
                    }
                }
                r.ecall_rows.insert(r.ecall_rows.end(), row, row + vmc::ECALL_WIDTH);
            } else {
                const bool has_result = d.cls != vmc::C_BEQ && d.cls != vmc::C_BLT && !(d.cls == vmc::C_LS && d.op >= 3 && d.op <= 5);
                z = has_result ? m_.last_result() : 0;
                pc_inc = vmc::field_of((int32_t)(m_.pc() - pc));
            }
            if (MODE != METER && d.cls == vmc::C_LS) {
                if (r.core.ls_case.size() != n_ls_before + 1) throw Error("internal: load/store record missing");
                r.ls_ts.push_back(ts), r.ls_base.push_back(x), r.ls_imm.push_back(imm32);
                const bool store = d.op >= 3 && d.op <= 5;
                r.ls_pts.push_back(store ? write_word(2, addr >> 2, mem_word(addr & ~3u), ts + 4) : read_word(2, addr >> 2, word_before, ts + 4));
            }
            const uint32_t pts3 = d.wr_rd ? write_word(1, d.rd, z, ts + 12) : 0;
            if (MODE != METER) {
                const size_t row = r.n_instr;
                fp1[row] = pts1, fp2[row] = pts2, fp3[row] = pts3;
                fx[row] = x, fy[row] = y, fz[row] = z, frd[row] = d.wr_rd ? rd_prev : 0, fpc[row] = pc_inc;
            }
            n_cls[d.cls]++;
            r.n_instr++;
        }
        if (MODE == REPLAY && (r.n_instr != plan->n_instr || m_.done() != plan->last || (m_.done() ? 0 : m_.pc()) != plan->pc_end))
            throw Error("internal: the record pass of segment " + std::to_string(plan->index) + " left the metered pass's path");
        r.pc_end = m_.done() ? 0 : m_.pc();
        r.ts_end = 1 + vmc::TS_STEP * (uint32_t)r.n_instr;
        if (MODE == METER) {   // the close is meter_close's (possibly another thread's): hand the touched blocks over
            finish_blocks();
            detach_blocks(plan->job);
            return m_.done();
        }
        {
            const auto t0 = std::chrono::steady_clock::now();
            if (MODE == REPLAY) close_replay(r, *plan);
            else close_memory(r);
            close_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        return m_.done();
    }

    const Exe& exe_;
    Machine m_;
    SegmentCaps caps_;
    MemoryTree tree_;
    Digest image_root_{};
    using Block = MemBlock;
    // touched blocks of the current segment: dense storage + direct-mapped slot tables (registers, public values, the guest's
    // read-write region) so that the common lookup is an array access; anything else (addresses outside the region) goes through a map
    std::vector<Block> blk_;
    std::vector<uint32_t> blk_label_;
    uint32_t reg_slot_[8] = {}, pv_slot_[2] = {}, mem_lo_blk_ = 0;
    uint32_t reg_ts_[32] = {};                       // last access of register j in this segment (its block's ts[] and cur[] are written at the close)
    std::vector<uint32_t> mem_slot_;                 // index + 1 into blk_, 0 = untouched
    std::unordered_map<uint32_t, uint32_t> other_slot_;
    std::unordered_set<uint64_t> path_nodes_;        // internal nodes above them
    std::vector<vmc::Decoded> dec_;
    std::vector<uint8_t> dec_regmask_;
    uint32_t reg_known_ = 0;   // register blocks touched in this segment (bit = block)
    double close_seconds_ = 0;

    uint32_t* slot_of(uint32_t label, bool create) {
        const uint32_t as = label >> vmc::LABEL_BITS, blk = label & ((1u << vmc::LABEL_BITS) - 1);
        if (as == 1 && blk < 8) return &reg_slot_[blk];
        if (as == 3 && blk < 2) return &pv_slot_[blk];
        if (as == 2 && blk >= mem_lo_blk_ && blk - mem_lo_blk_ < mem_slot_.size()) return &mem_slot_[blk - mem_lo_blk_];
        if (create) return &other_slot_[label];
        auto it = other_slot_.find(label);
        return it == other_slot_.end() ? nullptr : &it->second;
    }
    // (a block of the guest's read-write region: the direct-mapped table alone -- anything else is "not known" and takes the general path)
    bool mem_block_known(uint32_t blk) const {
        const uint32_t off = blk - mem_lo_blk_;
        return off < mem_slot_.size() && mem_slot_[off] != 0;
    }
    bool touched(uint32_t label) {
        const uint32_t* s = slot_of(label, false);
        return s && *s;
    }
    Block* find_block(uint32_t label) {
        const uint32_t* s = slot_of(label, false);
        return s && *s ? &blk_[*s - 1] : nullptr;
    }
    void reset_blocks() {
        for (uint32_t label : blk_label_) {
            uint32_t* s = slot_of(label, false);
            if (s) *s = 0;
        }
        other_slot_.clear();
        blk_.clear(), blk_label_.clear(), path_nodes_.clear();
        memset(reg_ts_, 0, sizeof reg_ts_);
        reg_known_ = 0;
    }

    uint32_t mem_word(uint32_t addr) const { return m_.peek_word(addr); }
    // value of a word in the machine's CURRENT state
    uint32_t word_now(uint32_t as, uint32_t word) const {
        if (as == 1) return word < 32 ? m_.reg(word) : 0;
        if (as == 3) {
            if (word >= 8) return 0;
            uint32_t v;
            memcpy(&v, m_.public_values().data() + 4 * word, 4);
            return v;
        }
        return mem_word(word << 2);
    }
    void push_nonzero(std::vector<std::pair<uint32_t, std::array<uint32_t, 8>>>& out, uint32_t as, uint32_t blk) const {
        std::array<uint32_t, 8> c;
        bool any = false;
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t w = word_now(as, 4 * blk + j);
            c[2 * j] = w & 0xffffu, c[2 * j + 1] = w >> 16, any = any || w;
        }
        if (any) out.push_back({(as << vmc::LABEL_BITS) | blk, c});
    }
    Block& block_of(uint32_t as, uint32_t word) {
        const uint32_t label = (as << vmc::LABEL_BITS) | (word >> 2);
        if (Block* known = find_block(label)) return *known;
        // first touch in this segment: the block's words still have their values from the segment's start
        Block b;
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t w = word_now(as, (word & ~3u) + j);
            b.init[2 * j] = b.cur[2 * j] = w & 0xffffu, b.init[2 * j + 1] = b.cur[2 * j + 1] = w >> 16, b.ts[j] = 0;
        }
        uint32_t idx = label;
        for (int l = (int)vmc::LEAF_LEVEL - 1; l >= 0; l--) {
            idx >>= 1;
            if (!path_nodes_.insert(MemoryTree::key((unsigned)l, idx)).second) break;
        }
        blk_.push_back(b), blk_label_.push_back(label);
        *slot_of(label, true) = (uint32_t)blk_.size();
        if (as == 1 && (word >> 2) < 8) reg_known_ |= 1u << (word >> 2);
        return blk_.back();
    }
    // A block is snapshotted at its first touch in the segment, which must happen BEFORE the machine executes an instruction that
    // changes it: run_segment calls block_of for every word an instruction will write before step_one.  Both return the timestamp of
    // the word's previous access (0 = untouched in this segment: the leaf chip's initial state).
    uint32_t read_word(uint32_t as, uint32_t word, uint32_t value, uint32_t ts) {
        if (as == 1 && word < 32) {   // a register: three of these per instruction -- the timestamp lives in reg_ts_, the block is completed at the close
            if (!reg_slot_[word >> 2]) (void)block_of(1, word);
            const uint32_t prev = reg_ts_[word];
            reg_ts_[word] = ts;
            return prev;
        }
        (void)value;   // (the words' values at the close come from the machine: close_memory)
        Block& b = block_of(as, word);
        const unsigned j = word & 3u;
        const uint32_t prev = b.ts[j];
        b.ts[j] = ts;
        return prev;
    }
    uint32_t write_word(uint32_t as, uint32_t word, uint32_t value, uint32_t ts) {
        if (as == 1 && word < 32) {
            if (!reg_slot_[word >> 2]) throw Error("internal: write to a block that was not snapshotted");
            const uint32_t prev = reg_ts_[word];
            reg_ts_[word] = ts;
            return prev;
        }
        Block* known = find_block((as << vmc::LABEL_BITS) | (word >> 2));
        if (!known) throw Error("internal: write to a block that was not snapshotted");
        const unsigned j = word & 3u;
        const uint32_t prev = known->ts[j];
        known->ts[j] = ts;
        return prev;
    }

    // The close of a RECORD pass: rows and roots are the metered pass's (it owns the tree); this pass contributes the words' last-access
    // timestamps (leaf-row columns 18..21) and checks that it touched the same blocks with the same initial and final cells.
    void close_replay(SegmentRecords& r, const SegmentPlan& plan) {
        const size_t nL = blk_label_.size();
        if (plan.leaf_rows.size() != nL * vmc::LEAF_WIDTH) throw Error("internal: the record pass of segment " + std::to_string(plan.index) + " touched other blocks than the metered pass");
        finish_blocks();
        std::vector<std::pair<uint32_t, uint32_t>> order(nL);
        for (size_t i = 0; i < nL; i++) order[i] = {blk_label_[i], (uint32_t)i};
        std::sort(order.begin(), order.end());
        r.leaf_rows = plan.leaf_rows, r.merkle_rows = plan.merkle_rows, r.p2_inputs = plan.p2_inputs;
        for (size_t i = 0; i < nL; i++) {
            uint32_t* row = &r.leaf_rows[i * vmc::LEAF_WIDTH];
            const Block& b = blk_[order[i].second];
            bool same = ((row[0] << vmc::LABEL_BITS) | row[1]) == order[i].first;
            for (int j = 0; j < 8; j++) same = same && row[2 + j] == b.init[j] && row[10 + j] == b.cur[j];
            if (!same) throw Error("internal: the record pass of segment " + std::to_string(plan.index) + " and the metered pass disagree on a touched block");
            for (int j = 0; j < 4; j++) row[18 + j] = b.ts[j];
        }
        r.root_final = plan.root_final;
    }

    // leaf rows, merkle rows, Poseidon2 requests; commits the segment's final memory to the tree.  A memory-bound guest touches thousands
    // of blocks per segment (4096 blocks + 4100 path nodes in the guest of tools/guest_bench2.py `mem`): the hashes go sixteen at a time
    // through zkhip_poseidon2_permute16_host, levels are walked over SORTED index arrays (a parent's touched children are neighbours),
    // rows are built side by side; the tree itself (a hash map) is only read until every level is done.
    // every touched block's FINAL cells are the machine's words now (read_word / write_word keep timestamps only); the register blocks'
    // timestamps come from reg_ts_
    void finish_blocks() {
        const size_t nL = blk_label_.size();
        for (size_t i = 0; i < nL; i++) {
            const uint32_t label = blk_label_[i], as = label >> vmc::LABEL_BITS, blk = label & ((1u << vmc::LABEL_BITS) - 1);
            Block& b = blk_[i];
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t w = word_now(as, 4 * blk + j);
                b.cur[2 * j] = w & 0xffffu, b.cur[2 * j + 1] = w >> 16;
                if (as == 1 && blk < 8) b.ts[j] = reg_ts_[4 * blk + j];
            }
        }
    }
    // the segment's touched blocks leave this executor (their slots are freed: the next segment starts clean)
    void detach_blocks(CloseJob& job) {
        for (uint32_t label : blk_label_) {
            uint32_t* s = slot_of(label, false);
            if (s) *s = 0;
        }
        other_slot_.clear();
        job.blk = std::move(blk_), job.label = std::move(blk_label_), job.n_path = path_nodes_.size();
        blk_.clear(), blk_label_.clear(), path_nodes_.clear();
        memset(reg_ts_, 0, sizeof reg_ts_);
        reg_known_ = 0;
    }
    void close_memory(SegmentRecords& r) {
        finish_blocks();
        CloseJob job;
        job.blk.swap(blk_), job.label.swap(blk_label_), job.n_path = path_nodes_.size();
        struct GiveBack {   // (reset_blocks frees the slots from the labels at the start of the next segment)
            SegmentExecutor& ex;
            CloseJob& job;
            ~GiveBack() { ex.blk_.swap(job.blk), ex.blk_label_.swap(job.label); }
        } give_back{*this, job};
        close_job(job, r);
    }
    void close_job(const CloseJob& job, SegmentRecords& r) {
        const std::vector<Block>& blk_ = job.blk;
        const std::vector<uint32_t>& blk_label_ = job.label;
        const size_t nL = blk_label_.size();
        std::vector<std::pair<uint32_t, uint32_t>> order(nL);   // (label, index into blk_)
        for (size_t i = 0; i < nL; i++) order[i] = {blk_label_[i], (uint32_t)i};
        std::sort(order.begin(), order.end());
        // n independent permutations; fill(k, st) writes the 16 input words of state k, the first 8 output words are out[k]
        auto hash_many = [&](size_t n, auto&& fill, Digest* out) {
            parallel_for((n + 15) / 16, 16, [&](size_t bt) {
                const size_t k0 = 16 * bt, k1 = std::min<size_t>(16, n - k0);
                if (k1 == 1) {
                    uint32_t st[16];
                    fill(k0, st);
                    zkhip_poseidon2_permute_host(st);
                    for (int w = 0; w < 8; w++) out[k0][w] = st[w];
                    return;
                }
                uint32_t st[256] = {};
                for (size_t k = 0; k < k1; k++) fill(k0 + k, st + 16 * k);
                zkhip_poseidon2_permute16_host(st);
                for (size_t k = 0; k < k1; k++)
                    for (int w = 0; w < 8; w++) out[k0 + k][w] = st[16 * k + w];
            });
        };
        std::vector<uint32_t> leaf_idx(nL);
        std::vector<Digest> leaf_fin(nL);
        for (size_t i = 0; i < nL; i++) leaf_idx[i] = order[i].first;
        hash_many(nL, [&](size_t i, uint32_t* st) {
            const Block& b = blk_[order[i].second];
            for (int j = 0; j < 8; j++) st[j] = b.cur[j], st[8 + j] = 0;
        }, leaf_fin.data());
        r.leaf_rows.assign(nL * vmc::LEAF_WIDTH, 0);
        const size_t p2_0 = r.p2_inputs.size();
        r.p2_inputs.resize(p2_0 + 32 * nL, 0);
        parallel_for(nL, 512, [&](size_t i) {
            const uint32_t label = leaf_idx[i];
            const Block& b = blk_[order[i].second];
            const Digest hi = tree_.get(vmc::LEAF_LEVEL, label);
            uint32_t* row = &r.leaf_rows[i * vmc::LEAF_WIDTH];
            row[0] = label >> vmc::LABEL_BITS, row[1] = label & ((1u << vmc::LABEL_BITS) - 1);
            for (int j = 0; j < 8; j++) row[2 + j] = b.init[j], row[10 + j] = b.cur[j], row[22 + j] = hi[j], row[30 + j] = leaf_fin[i][j];
            for (int j = 0; j < 4; j++) row[18 + j] = b.ts[j];
            row[38] = 1;
            const uint32_t gap = i + 1 < nL ? leaf_idx[i + 1] - label - 1 : 0;
            row[39] = gap & 0xffffu, row[40] = gap >> 16, row[41] = row[1] & 0xffffu, row[42] = row[1] >> 16;
            uint32_t* q = &r.p2_inputs[p2_0 + 32 * i];   // two requests per block: its initial and its final cells (the second half of a request is zero)
            for (int j = 0; j < 8; j++) q[j] = b.init[j], q[16 + j] = b.cur[j];
        });
        // path nodes level by level, bottom-up: the parents of a sorted index array are sorted, a parent's touched children are neighbours
        struct Level {
            std::vector<uint32_t> idx;
            std::vector<Digest> fin;
            std::vector<uint32_t> rows;
        };
        std::vector<Level> lv(vmc::LEAF_LEVEL);
        const std::vector<uint32_t>* c_idx = &leaf_idx;
        const std::vector<Digest>* c_fin = &leaf_fin;
        std::vector<uint32_t> first;
        std::vector<Digest> lf, rf;
        std::vector<uint8_t> on;
        size_t n_nodes = 0;
        for (int l = (int)vmc::LEAF_LEVEL - 1; l >= 0; l--) {
            Level& L = lv[l];
            first.clear();
            for (size_t i = 0; i < c_idx->size(); i++) {
                const uint32_t p = (*c_idx)[i] >> 1;
                if (L.idx.empty() || L.idx.back() != p) L.idx.push_back(p), first.push_back((uint32_t)i);
            }
            const size_t n = L.idx.size();
            n_nodes += n;
            L.fin.resize(n), lf.resize(n), rf.resize(n), on.resize(2 * n);
            parallel_for(n, 512, [&](size_t i) {
                const uint32_t p = L.idx[i];
                const size_t pos = first[i];
                const bool l_on = (*c_idx)[pos] == 2 * p;
                const size_t rpos = pos + (l_on ? 1 : 0);
                const bool r_on = rpos < c_idx->size() && (*c_idx)[rpos] == 2 * p + 1;
                lf[i] = l_on ? (*c_fin)[pos] : tree_.get((unsigned)l + 1, 2 * p);
                rf[i] = r_on ? (*c_fin)[rpos] : tree_.get((unsigned)l + 1, 2 * p + 1);
                on[2 * i] = l_on, on[2 * i + 1] = r_on;
            });
            hash_many(n, [&](size_t i, uint32_t* st) {
                for (int q = 0; q < 8; q++) st[q] = lf[i][q], st[8 + q] = rf[i][q];
            }, L.fin.data());
            L.rows.assign(n * vmc::MERKLE_WIDTH, 0);
            parallel_for(n, 512, [&](size_t i) {
                const uint32_t idx = L.idx[i];
                const Digest li = on[2 * i] ? tree_.get((unsigned)l + 1, 2 * idx) : lf[i], ri = on[2 * i + 1] ? tree_.get((unsigned)l + 1, 2 * idx + 1) : rf[i];
                const Digest pi = tree_.get((unsigned)l, idx);
                uint32_t* row = &L.rows[i * vmc::MERKLE_WIDTH];
                row[0] = (uint32_t)l, row[1] = idx;
                for (int q = 0; q < 8; q++) row[2 + q] = li[q], row[10 + q] = ri[q], row[18 + q] = lf[i][q], row[26 + q] = rf[i][q], row[34 + q] = pi[q], row[42 + q] = L.fin[i][q];
                row[50] = on[2 * i], row[51] = on[2 * i + 1], row[52] = 1, row[53] = l == 0;
            });
            c_idx = &L.idx, c_fin = &L.fin;
        }
        if (n_nodes != job.n_path) throw Error("internal: the touched path nodes and the touched blocks' ancestors differ");
        // the root's row goes first in the chip; two Poseidon2 requests per row (initial children, final children)
        r.merkle_rows.clear();
        r.merkle_rows.reserve(n_nodes * vmc::MERKLE_WIDTH);
        for (unsigned l = 0; l < vmc::LEAF_LEVEL; l++) r.merkle_rows.insert(r.merkle_rows.end(), lv[l].rows.begin(), lv[l].rows.end());
        const size_t p2_1 = r.p2_inputs.size();
        r.p2_inputs.resize(p2_1 + 32 * n_nodes);
        parallel_for(n_nodes, 512, [&](size_t i) {
            const uint32_t* row = &r.merkle_rows[i * vmc::MERKLE_WIDTH];
            uint32_t* q = &r.p2_inputs[p2_1 + 32 * i];
            for (int k = 0; k < 16; k++) q[k] = row[2 + k], q[16 + k] = row[18 + k];
        });
        for (size_t i = 0; i < nL; i++) tree_.set(vmc::LEAF_LEVEL, leaf_idx[i], leaf_fin[i]);
        for (unsigned l = 0; l < vmc::LEAF_LEVEL; l++)
            for (size_t i = 0; i < lv[l].idx.size(); i++) tree_.set(l, lv[l].idx[i], lv[l].fin[i]);
        r.root_final = tree_.root();
    }
    // f(i) for i < n on up to zkhip_host_cpus() threads, each taking at least `grain` indices (fewer than 2 grains: the caller's thread)
    template <class F>
    static void parallel_for(size_t n, size_t grain, F&& f) {
        const size_t nt = std::min<size_t>(std::max(1u, zkhip_host_cpus()), n / std::max<size_t>(1, grain));
        if (nt <= 1) {
            for (size_t i = 0; i < n; i++) f(i);
            return;
        }
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; t++)
            th.emplace_back([&, t]() {
                for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) f(i);
            });
        for (size_t i = 0; i < n / nt; i++) f(i);
        for (auto& t : th) t.join();
    }
};
// The PARALLEL executor: a metered first pass on one thread (SegmentExecutor::meter_segment: the interpreter without a record sink plus
// the cut rule, ~3-4 x the speed of the recording executor) decides where every segment ends, snapshots the machine there and keeps the
// persistent memory tree; `threads` record passes (SegmentExecutor::replay_segment, each on a machine of its own) replay the planned
// segments side by side and hand them out IN ORDER.  The segments, cut points and records are those of the serial SegmentExecutor byte for
// byte (tests/test_exec_parallel_cpu.py).  This is how the reference's stack executes: execute_metered_cost first
// (crates/prover/src/utils/vm.rs:19), then OpenVM replays the segments; host parallelism is a first-class tool there
// (crates/integration/src/testers/chunk.rs:352-368).  threads == 0: the serial executor on the caller's thread, as before.
// Memory: at most `window_` = 2 threads + 2 segments are planned, replayed or waiting for the caller at a time; each holds its records (what the
// serial executor holds once) and the 4 KiB pages its predecessor wrote -- at most the guest's whole memory per segment, in practice the working
// set of 2^20 instructions; pages are dropped as soon as every record pass has moved beyond them.
class ParallelSegmentExecutor {
public:
    ParallelSegmentExecutor(const Exe& exe, const StdIn& in, const SegmentCaps& caps, unsigned threads)
        : exe_(exe), in_(in), caps_(caps), meter_(exe, in, caps), n_workers_(threads), window_(2 * (size_t)threads + 2) {
        if (!n_workers_) return;
        meter_.machine().track_dirty(true);
        pos_.assign(n_workers_, 0);
        busy_.assign(n_workers_, 0.0);
        meter_thread_ = std::thread([this] { meter_loop(); });
        closer_thread_ = std::thread([this] { closer_loop(); });
        for (unsigned w = 0; w < n_workers_; w++) workers_.emplace_back([this, w] { worker_loop(w); });
    }
    ~ParallelSegmentExecutor() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (meter_thread_.joinable()) meter_thread_.join();
        if (closer_thread_.joinable()) closer_thread_.join();
        for (auto& t : workers_) t.join();
    }
    ParallelSegmentExecutor(const ParallelSegmentExecutor&) = delete;
    ParallelSegmentExecutor& operator=(const ParallelSegmentExecutor&) = delete;

    // the next segment, in order; true once the guest has exited.  `r`'s storage is recycled.
    bool run_segment(SegmentRecords& r) {
        if (!n_workers_) return meter_.run_segment(r);
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return failed_ || ready_.count(delivered_); });
        if (failed_ && !ready_.count(delivered_)) throw Error(error_);
        auto it = ready_.find(delivered_);
        std::swap(r, it->second.rec);
        const bool last = it->second.last;
        free_.push_back(std::move(it->second.rec));
        ready_.erase(it);
        delivered_++;
        lk.unlock();
        cv_.notify_all();
        return last;
    }
    // (the end state is the metered machine's: valid once run_segment has returned true)
    const Digest& image_root() const { return meter_.image_root(); }
    const MemoryTree& tree() const { return meter_.tree(); }
    bool done() const { return meter_.done(); }
    uint64_t instret() const { return meter_.instret(); }
    uint32_t exit_code() const { return meter_.exit_code(); }
    const std::vector<uint8_t>& public_values() const { return meter_.public_values(); }
    uint32_t peek_memory(uint32_t addr) const { return meter_.peek_memory(addr); }
    double close_seconds() const { return meter_.close_seconds(); }
    unsigned threads() const { return n_workers_; }
    // busy seconds of the metered pass / summed over the record passes (0 for the serial executor)
    double metered_seconds() const { return meter_busy_; }
    double tree_seconds() const { return closer_busy_; }
    double record_seconds() const {
        double s = 0;
        for (double b : busy_) s += b;
        return s;
    }

private:
    using clk = std::chrono::steady_clock;
    struct Ready {
        SegmentRecords rec;
        bool last = false;
    };
    const Exe& exe_;
    const StdIn& in_;
    SegmentCaps caps_;
    SegmentExecutor meter_;
    unsigned n_workers_;
    size_t window_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false, failed_ = false, metered_all_ = false;
    std::string error_;
    std::map<size_t, std::shared_ptr<SegmentPlan>> plans_;                                // produced, not yet replayed
    std::map<size_t, std::shared_ptr<const std::vector<Machine::Page>>> pages_;          // per plan index, until every worker is past it
    std::map<size_t, Ready> ready_;
    std::vector<SegmentRecords> free_;
    std::vector<size_t> pos_;        // per worker: the cut its machine's memory stands at
    std::vector<double> busy_;
    size_t produced_ = 0, next_claim_ = 0, delivered_ = 0;
    double meter_busy_ = 0;
    std::deque<std::shared_ptr<SegmentPlan>> to_close_;   // run by the metered pass, their memory close pending (in order)
    size_t metered_ = 0;                                  // plans the metered pass has run
    bool meter_finished_ = false;
    double closer_busy_ = 0;
    std::thread meter_thread_, closer_thread_;
    std::vector<std::thread> workers_;

    void fail(const std::string& what) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!failed_) failed_ = true, error_ = what;
        }
        cv_.notify_all();
    }
    // the machine: runs ahead of the record passes by at most `window_` segments
    void meter_loop() {
        try {
            for (bool done = false; !done;) {
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return stop_ || failed_ || metered_ - delivered_ < window_; });
                    if (stop_ || failed_) return;
                }
                auto plan = std::make_shared<SegmentPlan>();
                const auto t0 = clk::now();
                done = meter_.meter_run(*plan);
                meter_busy_ += std::chrono::duration<double>(clk::now() - t0).count();
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    plan->index = metered_++;
                    to_close_.push_back(std::move(plan));
                    meter_finished_ = done;
                }
                cv_.notify_all();
            }
        } catch (const std::exception& e) {
            fail(e.what());
        }
    }
    // the persistent memory tree: one plan behind the machine, on a thread of its own (a memory-bound guest hashes ~8 k nodes per segment)
    void closer_loop() {
        try {
            for (;;) {
                std::shared_ptr<SegmentPlan> plan;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return stop_ || failed_ || !to_close_.empty() || meter_finished_; });
                    if (stop_ || failed_) return;
                    if (to_close_.empty()) return;   // (the metered pass has finished and everything is closed)
                    plan = std::move(to_close_.front());
                    to_close_.pop_front();
                }
                const auto t0 = clk::now();
                meter_.meter_close(*plan);
                closer_busy_ += std::chrono::duration<double>(clk::now() - t0).count();
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    const bool last = plan->last;
                    pages_[produced_] = plan->pages;
                    plans_[produced_] = std::move(plan);
                    produced_++;
                    metered_all_ = last;
                }
                cv_.notify_all();
            }
        } catch (const std::exception& e) {
            fail(e.what());
        }
    }
    void worker_loop(unsigned w) {
        try {
            SegmentExecutor ex(exe_, in_, caps_, /*replay_only=*/true);
            for (;;) {
                std::shared_ptr<SegmentPlan> plan;
                std::vector<std::shared_ptr<const std::vector<Machine::Page>>> catch_up;
                size_t target;
                SegmentRecords rec;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    // a claim, or -- for a worker that has been idle for a while -- catching its memory up with the cuts the others have
                    // taken, so that the pages of old cuts can be dropped
                    cv_.wait(lk, [&] { return stop_ || failed_ || next_claim_ < produced_ || next_claim_ > pos_[w] + 8 || (metered_all_ && next_claim_ == produced_); });
                    if (stop_ || failed_) return;
                    if (next_claim_ < produced_) {
                        target = next_claim_++;
                        plan = plans_.at(target);
                        plans_.erase(target);
                        if (!free_.empty()) rec = std::move(free_.back()), free_.pop_back();
                    } else if (next_claim_ > pos_[w] + 8) {
                        target = next_claim_ - 1;   // (every segment below next_claim_ is someone else's: this worker's next claim lies at or beyond it)
                    } else {
                        return;   // every segment has been claimed
                    }
                    for (size_t i = pos_[w] + 1; i <= target; i++) catch_up.push_back(pages_.at(i));
                }
                const auto t0 = clk::now();
                for (const auto& pg : catch_up) ex.machine().apply_pages(*pg);
                bool last = false;
                if (plan) last = ex.replay_segment(*plan, rec);
                const double dt = std::chrono::duration<double>(clk::now() - t0).count();
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    busy_[w] += dt;
                    pos_[w] = plan ? target + 1 : target;
                    if (plan) {
                        Ready& slot = ready_[target];
                        slot.rec = std::move(rec), slot.last = last;
                    }
                    size_t low = pos_[0];
                    for (size_t p : pos_) low = std::min(low, p);
                    while (!pages_.empty() && pages_.begin()->first <= low) pages_.erase(pages_.begin());
                }
                cv_.notify_all();
            }
        } catch (const std::exception& e) {
            fail(e.what());
        }
    }
};

}  // namespace zkhip_vm
