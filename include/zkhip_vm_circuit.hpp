// zkhip_vm_circuit.hpp -- ONE STATEMENT PER SEGMENT: the chips that tie the instruction cores of include/zkhip_chips.hpp to the
// executed program, to each other and to memory -- the job of OpenVM's adapters, execution bus, connector and persistent-memory
// chips (un-vendored crates openvm-circuit / openvm-rv32im-circuit; the reference reaches them through
// sdk.prove, crates/prover/src/prover/mod.rs:355-357).
//
//   program chip   the decoded program as a PREPROCESSED trace (its commitment is part of the verifying key = the committed
//                  executable, crates/verifier/src/verifier.rs:77-80), one row per instruction: 17 decode fields; the main trace is
//                  the execution frequency; every executed instruction is looked up here.
//   frame chip     one row per executed instruction: receives (pc, timestamp) on the EXECUTION BUS and sends (pc + pc_inc,
//                  timestamp + 16); its REGISTER ADAPTER reads rs1 / rs2 and writes rd on the MEMORY BUS at fixed timestamp slots (a word
//                  access = receive the word's previous state, send the new one, previous timestamp strictly smaller); range-checks
//                  the operand and result bytes; hands (class, opcode, operands, result, pc_inc ...) to the instruction's core chip
//                  on the OPERAND BUS.  A core chip (ALU, less-than, mul, mulh, div, shift, branches, jumps) is the core of
//                  include/zkhip_chips.hpp plus ONE receive on that bus: its columns are unchanged, so are its trace generators.
//   load/store     the load/store core plus its memory adapter: address = rs1 + immediate, the aligned word accessed at slot 4.
//   ecall chip     exit (pc becomes 0: the end of the execution), reveal (writes a public-value word into address space 3), hint read,
//                  keccak (a7 = 3: Keccak-f[1600] in place on the 200 bytes at a0 -- the intrinsic the reference's guests reach through
//                  OpenVM's keccak extension, crates/circuits/chunk-circuit/openvm.toml).
//                  sha256 (a7 = 4: the SHA-256 compression function on the 24 words at a0 -- eight state words, then the sixteen message
//                  words of a block; the state is replaced by compress(state, block); OpenVM's sha256 extension).
//                  modular (a7 = 5: r = a b, a + b or a - b mod P_i on the 24 words at a0 -- a, b, then the result's slot; a1 = i + 8 op:
//                  i the index of one of the moduli the app's openvm.toml lists, op 0 mul, 1 add, 2 sub; OpenVM's modular extension).
//                  int256 (a7 = 6: a = b op c on the 24 words at a0 -- b, c, then the result's slot; a1 = op: 0 add, 1 sub, 2 xor, 3 or,
//                  4 and, 5 mul, 6 b < c unsigned, 7 b < c signed, 8 b == c, 9 / 10 / 11 b shifted left / right / right arithmetically by
//                  c mod 256; 256-bit words, arithmetic modulo 2^256; OpenVM's bigint extension).
//                  ecc (a7 = 7: (x3, y3) = (x1, y1) + (x2, y2) or the double of (x1, y1) on the 48 words at a0 -- the two operands, then
//                  the result's slot; a1 = i + 8 op: i the index of one of the curves the app's openvm.toml lists, op 0 add (x1 != x2),
//                  1 double; OpenVM's ecc extension).
//                  fp2 (a7 = 8: r = a b, a + b, a - b or a / b in Fp[u] / (u^2 + 1) on the 48 words at a0 -- a, b, then the result's slot,
//                  two 256-bit components each; a1 = i + 8 op: i the index of one of the fields the app's openvm.toml lists; OpenVM's
//                  fp2 extension).  Its adapter has the ecc adapter's shape on buses of its own.
//                  native field (a7 = 9), native extension (a7 = 10), castf (a7 = 11): include/zkhip_native.hpp -- OpenVM's native and
//                  castf extensions (crates/circuits/batch-circuit/openvm.toml:16,24; bundle-circuit/openvm.toml:16,18).
//   native chips   ONE row per call, core and memory adapter in one chip (like load/store): the field-arithmetic chip reads two
//                  words and writes one, the extension chip reads 8 and writes 4, the castf chip reads one and writes one.
//   ecc adapter    one per curve, one row per WORD of a call (48 rows): operands read, result written; (timestamp, index, halves, op)
//                  goes to that curve's point chip (include/zkhip_ecc.hpp, one operation per row).
//   int256 adapter one row per WORD of a call (24 rows): b, c read, a written; (timestamp, index, halves, opcode) goes to the 256-bit ALU
//                  chip or, for opcode 5, to the 256-bit multiplication chip, for 6..8 to the comparison chip, for 9..11 to the shift
//                  chip (include/zkhip_int256.hpp, one operation per row): all four listen on the word bus, each to its opcodes.
//   modular adapter one per modulus, one row per WORD of a call (24 rows): a, b read, r written; (timestamp, index, halves) goes to that
//                  modulus's multiplication chip (include/zkhip_modular.hpp, one multiplication per row).
//   sha256 adapter one row per WORD of a call (24 rows): state words are read and replaced, message words read; (timestamp, index, limbs)
//                  goes to the SHA-256 compression chip (include/zkhip_sha256.hpp, one round per row) on the message / state buses.
//   keccak adapter one row per LANE of a permutation call (25 rows): the lane's two memory words are read and replaced by the output
//                  lane's (word accesses at the call's timestamp), and (timestamp, lane, input limbs, output limbs) goes to the
//                  Keccak-f chip (include/zkhip_keccak.hpp, one round per row) on the lane bus.
//   (memory)       offline memory checking inside the segment lives in the chips that access memory (as in OpenVM's adapters): a memory
//                  word is (address space, word pointer, low half, high half, timestamp) on the memory bus; there is no separate
//                  access chip.
//   leaf chip      one row per touched memory BLOCK of 4 words (8 sixteen-bit cells): initial / final values against the memory bus,
//                  their Poseidon2 digests, strictly increasing block labels.
//   merkle chip    one row per node on the path from a touched block to the root: the same path hashed over the initial and the
//                  final blocks; untouched siblings are equal in both; the root row's digests are PUBLIC VALUES: the segment's
//                  initial and final memory root.
//   connector      public values (pc_start, pc_end): starts and ends the execution-bus chain.
// A segment proof therefore states: "the committed program, started at pc_start on the memory with root R0, runs to pc_end and
// leaves the memory with root R1".  Consecutive segments chain by (pc_end, R1) = (pc_start', R0') -- exactly the (start, end)
// state the aggregation circuit chains (include/zkhip_aggregation.hpp) -- and pc_end = 0 means the guest has exited with code 0.
// Registers are address space 1 (word i = x_i), memory address space 2 (word = byte address / 4, addresses below 2^30), the 32
// public-value bytes address space 3 (words 0..7).  Header-only, no device code.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

#include "zkhip_chips.hpp"
#include "zkhip_ecc.hpp"
#include "zkhip_fp2.hpp"
#include "zkhip_int256.hpp"
#include "zkhip_keccak.hpp"
#include "zkhip_modular.hpp"
#include "zkhip_native.hpp"
#include "zkhip_sha256.hpp"

namespace zkhip {
namespace vmc {
using air::AirBuilder;
using air::Expr;
using air::Kind;

constexpr uint32_t MEMORY_BUS = 1, EXEC_BUS = 2, OPERAND_BUS = 3, RANGE_BUS = 5, RANGE_TUPLE_BUS = 6, PROGRAM_BUS = 8, BITWISE_BUS = 9,
                   MERKLE_BUS = 10, HASH_BUS = 11, KECCAK_REQ_BUS = 12, KECCAK_LANE_BUS = 13, SHA_REQ_BUS = 14, SHA_MSG_BUS = 15, SHA_STATE_BUS = 16, MODMUL_REQ_BUS = 17, MODMUL_WORD_BUS = 18,   // modulus i: word bus 18 + i
                   INT256_REQ_BUS = 30, INT256_WORD_BUS = 31, EC_REQ_BUS = 32, EC_WORD_BUS = 33,   // curve i: word bus 33 + i
                   FP2_REQ_BUS = 37, FP2_WORD_BUS = 38,                                                // fp2 field i: word bus 38 + i
                   NATIVE_REQ_BUS = 40, NATIVE_EXT_REQ_BUS = 41, CASTF_REQ_BUS = 42,
                   BRANCH256_BUS = 43;   // (ts, taken): the comparison chip's decision of a 256-bit branch, to the ecall chip
constexpr uint32_t TS_STEP = 16;   // timestamps per instruction: rs1 at +0; rs2 at +2; the core's own accesses at +4 ..; rd at +12
constexpr uint32_t BLOCK_CELLS = 8, BLOCK_WORDS = 4, LABEL_BITS = 26, LEAF_LEVEL = LABEL_BITS + 2;  // block label = as * 2^26 + (word >> 2); the tree has 28 levels above the blocks
constexpr uint32_t GAP_HI_BITS = 11;  // a timestamp gap is gap_lo + 2^16 gap_hi with gap_hi below 2^11 (one range-tuple lookup (0, gap_hi))
enum Cls : uint32_t { C_ALU, C_LT, C_MUL, C_MULH, C_DIVREM, C_SHIFT, C_BEQ, C_BLT, C_JAL_LUI, C_AUIPC, C_JALR, C_LS, C_ECALL, N_CLS };
constexpr size_t PROGRAM_FIELDS = 17, FRAME_WIDTH = 43, LS_WIDTH = 48, ECALL_WIDTH = 45, NATIVE_ARITH_WIDTH = 27, NATIVE_EXT_WIDTH = 90, CASTF_WIDTH = 16, LEAF_WIDTH = 43, MERKLE_WIDTH = 54, KECCAK_IO_WIDTH = 42, SHA_IO_WIDTH = 34, MODMUL_IO_WIDTH = 35, INT256_IO_WIDTH = 35, EC_IO_WIDTH = 59,
                 CONNECTOR_WIDTH = 2;
constexpr uint32_t P = air::P;

// ---- decode: the 17 fields of a program row -------------------------------------------------------------------------------------
struct Decoded {
    uint32_t pc = 0, cls = 0, op = 0, rd = 0, rs1 = 0, rs2 = 0, imm_lo = 0, imm_hi = 0, imm_f = 0;
    uint32_t use_rs1 = 0, use_rs2 = 0, y_is_imm = 0, wr_rd = 0, need_pc = 0, need_immf = 0, need_ts = 0, y_byte = 0;
    bool legal = true;
    std::array<uint32_t, PROGRAM_FIELDS> fields() const {
        return {pc, cls, op, rd, rs1, rs2, imm_lo, imm_hi, imm_f, use_rs1, use_rs2, y_is_imm, wr_rd, need_pc, need_immf, need_ts, y_byte};
    }
};
inline uint32_t field_of(int32_t v) { return v < 0 ? P - (uint32_t)(-(int64_t)v) : (uint32_t)v; }
inline Decoded decode(uint32_t w, uint32_t pc) {
    Decoded d;
    d.pc = pc;
    const uint32_t opc = w & 0x7f, rd = (w >> 7) & 31, f3 = (w >> 12) & 7, rs1 = (w >> 15) & 31, rs2 = (w >> 20) & 31, f7 = w >> 25;
    auto sext = [](uint32_t v, unsigned bits) { const uint32_t m = 1u << (bits - 1); return (int32_t)((v ^ m) - m); };
    const int32_t imm_i = sext(w >> 20, 12), imm_s = sext(((w >> 25) << 5) | ((w >> 7) & 31u), 12);
    const int32_t imm_b = sext(((w >> 31) << 12) | (((w >> 7) & 1u) << 11) | (((w >> 25) & 63u) << 5) | (((w >> 8) & 15u) << 1), 13);
    const int32_t imm_j = sext(((w >> 31) << 20) | (((w >> 12) & 255u) << 12) | (((w >> 20) & 1u) << 11) | (((w >> 21) & 1023u) << 1), 21);
    auto imm32 = [&](int32_t v) { d.imm_lo = (uint32_t)v & 0xffffu, d.imm_hi = (uint32_t)v >> 16; };
    auto dest = [&]() { d.rd = rd, d.wr_rd = rd != 0; };
    switch (opc) {
        case 0x37: d.cls = C_JAL_LUI, d.op = 1, dest(), d.imm_f = w >> 12, d.need_pc = d.need_immf = 1; break;
        case 0x17: d.cls = C_AUIPC, dest(), d.imm_f = w >> 12, d.need_pc = d.need_immf = 1; break;
        case 0x6f: d.cls = C_JAL_LUI, d.op = 0, dest(), d.imm_f = field_of(imm_j), d.need_pc = d.need_immf = 1; break;
        case 0x67: d.cls = C_JALR, dest(), d.rs1 = rs1, d.use_rs1 = 1, d.imm_f = w >> 20, d.need_pc = d.need_immf = 1; break;
        case 0x63:
            if (f3 == 2 || f3 == 3) d.legal = false;
            d.cls = f3 < 2 ? C_BEQ : C_BLT, d.op = f3 < 2 ? f3 : (((f3 & 1u) << 1) | ((f3 >> 1) & 1u));
            d.rs1 = rs1, d.rs2 = rs2, d.use_rs1 = d.use_rs2 = 1, d.imm_f = field_of(imm_b), d.need_immf = 1;
            break;
        case 0x03: {
            static const int kind[8] = {7, 6, 0, -1, 2, 1, -1, -1};  // lb lh lw - lbu lhu
            if (kind[f3] < 0) d.legal = false;
            d.cls = C_LS, d.op = (uint32_t)std::max(kind[f3], 0), dest(), d.rs1 = rs1, d.use_rs1 = 1, imm32(imm_i), d.need_ts = 1;
            break;
        }
        case 0x23:
            if (f3 > 2) d.legal = false;
            d.cls = C_LS, d.op = f3 == 2 ? 3 : f3 == 1 ? 4 : 5, d.rs1 = rs1, d.rs2 = rs2, d.use_rs1 = d.use_rs2 = 1, imm32(imm_s), d.need_ts = 1;
            break;
        case 0x13:
            dest(), d.rs1 = rs1, d.use_rs1 = 1, d.y_is_imm = 1;
            switch (f3) {
                case 0: d.cls = C_ALU, d.op = 0, imm32(imm_i); break;
                case 2: d.cls = C_LT, d.op = 0, imm32(imm_i); break;
                case 3: d.cls = C_LT, d.op = 1, imm32(imm_i); break;
                case 4: d.cls = C_ALU, d.op = 2, imm32(imm_i); break;
                case 6: d.cls = C_ALU, d.op = 3, imm32(imm_i); break;
                case 7: d.cls = C_ALU, d.op = 4, imm32(imm_i); break;
                case 1: d.cls = C_SHIFT, d.op = 0, d.imm_lo = rs2, d.y_byte = 1, d.legal = f7 == 0; break;
                default: d.cls = C_SHIFT, d.op = f7 ? 2 : 1, d.imm_lo = rs2, d.y_byte = 1, d.legal = f7 == 0 || f7 == 0x20; break;
            }
            break;
        case 0x33:
            dest(), d.rs1 = rs1, d.rs2 = rs2, d.use_rs1 = d.use_rs2 = 1;
            if (f7 == 1) {
                if (f3 == 0) d.cls = C_MUL;
                else if (f3 < 4) d.cls = C_MULH, d.op = f3 - 1;
                else d.cls = C_DIVREM, d.op = f3 - 4;
            } else if (f7 == 0 || f7 == 0x20) {
                switch (f3) {
                    case 0: d.cls = C_ALU, d.op = f7 ? 1 : 0; break;
                    case 1: d.cls = C_SHIFT, d.op = 0, d.y_byte = 1, d.legal = f7 == 0; break;
                    case 2: d.cls = C_LT, d.op = 0, d.legal = f7 == 0; break;
                    case 3: d.cls = C_LT, d.op = 1, d.legal = f7 == 0; break;
                    case 4: d.cls = C_ALU, d.op = 2, d.legal = f7 == 0; break;
                    case 5: d.cls = C_SHIFT, d.op = f7 ? 2 : 1, d.y_byte = 1; break;
                    case 6: d.cls = C_ALU, d.op = 3, d.legal = f7 == 0; break;
                    default: d.cls = C_ALU, d.op = 4, d.legal = f7 == 0; break;
                }
            } else {
                d.legal = false;
            }
            break;
        case 0x73:
            d.legal = w == 0x00000073u;
            d.cls = C_ECALL, d.rs1 = 17, d.rs2 = 10, d.rd = 10, d.use_rs1 = d.use_rs2 = d.wr_rd = 1, d.need_pc = d.need_ts = 1;
            break;
        case 0x0f: d.cls = C_ALU, d.op = 0, d.y_is_imm = 1; break;  // FENCE: no operation (add x0, x0, 0)
        default: d.legal = false; break;
    }
    return d;
}

// The program chip's preprocessed trace: PROGRAM_FIELDS columns (column-major over 2^log_program rows, canonical); rows beyond the
// program are zero and never executed.  An illegal word gets a row nothing can use (class N_CLS): executing it cannot be proven.
inline std::vector<uint32_t> program_table(const std::vector<uint32_t>& words, uint32_t pc_base, unsigned log_program) {
    const size_t n = (size_t)1 << log_program;
    std::vector<uint32_t> t(PROGRAM_FIELDS * n, 0);
    for (size_t k = 0; k < words.size(); k++) {
        Decoded d = decode(words[k], pc_base + 4 * (uint32_t)k);
        if (!d.legal) d = Decoded(), d.pc = pc_base + 4 * (uint32_t)k, d.cls = N_CLS;
        const auto f = d.fields();
        for (size_t q = 0; q < PROGRAM_FIELDS; q++) t[q * n + k] = f[q];
    }
    return t;
}

// ---- chips -------------------------------------------------------------------------------------------------------------------------
// program chip: AirBuilder(1, 0, prep_width = PROGRAM_FIELDS); main column 0 = execution frequency
inline void program_air(AirBuilder& b) {
    std::vector<Expr> f;
    for (size_t c = 0; c < PROGRAM_FIELDS; c++) f.push_back(b.prep(c));
    b.push_interaction(PROGRAM_BUS, f, b.var(0), Kind::Receive);
}

// One WORD ACCESS inside a chip's row (the memory side of an adapter): the word's previous state (p_lo, p_hi, prev_ts) is received
// from the memory bus and its new state (n_lo, n_hi, ts) sent, `count` times (0 or 1).  ts - prev_ts - 1 = gap_lo + 2^16 gap_hi with
// gap_lo in the range table and (0, gap_hi) in the range-tuple table (gap_hi < 2^11): timestamps stay below 2^29 (connector), so the
// difference cannot wrap around the field and the previous access is strictly earlier.  The two lookups are made `looked_up` times
// (the row's validity flag: rows that skip the access carry a zero gap), so their multiplicities are plain column histograms.
inline void word_access(AirBuilder& b, Expr count, Expr looked_up, Expr as_, Expr ptr, Expr p_lo, Expr p_hi, Expr n_lo, Expr n_hi, Expr ts, Expr prev_ts,
                        Expr gap_lo, Expr gap_hi) {
    b.assert_zero(count * (ts - prev_ts - 1 - gap_lo - gap_hi * 65536));
    b.push_interaction(MEMORY_BUS, {as_, ptr, p_lo, p_hi, prev_ts}, count, Kind::Receive);
    b.push_interaction(MEMORY_BUS, {as_, ptr, n_lo, n_hi, ts}, count, Kind::Send);
    b.push_interaction(RANGE_BUS, {gap_lo}, looked_up, Kind::Send);
    b.push_interaction(RANGE_TUPLE_BUS, {b.constant(0), gap_hi}, looked_up, Kind::Send);
}

// frame chip: AirBuilder(FRAME_WIDTH, 0).  Columns: pc ts | cls op rd rs1 rs2 imm_lo imm_hi imm_f use_rs1 use_rs2 y_is_imm wr_rd need_pc
// need_immf need_ts y_byte | x[4] y[4] z[4] | rd_prev_lo rd_prev_hi | pc_inc | ok | (prev_ts gap_lo gap_hi) of the rs1, rs2, rd accesses
inline void frame_air(AirBuilder& b) {
    const Expr pc = b.var(0), ts = b.var(1), cls = b.var(2), op = b.var(3), rd = b.var(4), rs1 = b.var(5), rs2 = b.var(6), imm_lo = b.var(7),
               imm_hi = b.var(8), imm_f = b.var(9), use_rs1 = b.var(10), use_rs2 = b.var(11), y_is_imm = b.var(12), wr_rd = b.var(13), need_pc = b.var(14),
               need_immf = b.var(15), need_ts = b.var(16), y_byte = b.var(17);
    Expr x[4], y[4], z[4];
    for (int i = 0; i < 4; i++) x[i] = b.var(18 + i), y[i] = b.var(22 + i), z[i] = b.var(26 + i);
    const Expr rdp_lo = b.var(30), rdp_hi = b.var(31), pc_inc = b.var(32), ok = b.var(33);
    const Expr x_lo = x[0] + x[1] * 256, x_hi = x[2] + x[3] * 256, y_lo = y[0] + y[1] * 256, y_hi = y[2] + y[3] * 256, z_lo = z[0] + z[1] * 256,
               z_hi = z[2] + z[3] * 256;
    b.assert_zero(ok * (ok - 1));
    // operands that do not come from a register: zero, or the immediate
    b.assert_zero((1 - use_rs1) * x_lo);
    b.assert_zero((1 - use_rs1) * x_hi);
    b.assert_zero(y_is_imm * (y_lo - imm_lo));
    b.assert_zero(y_is_imm * (y_hi - imm_hi));
    b.assert_zero((1 - use_rs2 - y_is_imm) * y_lo);
    b.assert_zero((1 - use_rs2 - y_is_imm) * y_hi);
    std::vector<Expr> prog{pc, cls, op, rd, rs1, rs2, imm_lo, imm_hi, imm_f, use_rs1, use_rs2, y_is_imm, wr_rd, need_pc, need_immf, need_ts, y_byte};
    b.push_interaction(PROGRAM_BUS, prog, ok, Kind::Send);
    b.push_interaction(EXEC_BUS, {pc, ts}, ok, Kind::Receive);
    b.push_interaction(EXEC_BUS, {pc + pc_inc, ts + (int64_t)TS_STEP}, ok, Kind::Send);
    {
        const Expr wide = 1 - y_byte;
        std::vector<Expr> m{cls, op, x[0], x[1], x[2], x[3], y[0], wide * y[1], wide * y[2], wide * y[3], z[0], z[1], z[2], z[3],
                            need_pc * pc, need_immf * imm_f, pc_inc, need_ts * ts, need_ts * imm_lo, need_ts * imm_hi};
        b.push_interaction(OPERAND_BUS, m, ok, Kind::Send);
    }
    const Expr one = b.constant(1), zero = b.constant(0);
    // register adapter: rs1 and rs2 are read (the word stays), rd is written
    word_access(b, use_rs1, ok, one, rs1, x_lo, x_hi, x_lo, x_hi, ts, b.var(34), b.var(35), b.var(36));
    word_access(b, use_rs2, ok, one, rs2, y_lo, y_hi, y_lo, y_hi, ts + 2, b.var(37), b.var(38), b.var(39));
    word_access(b, wr_rd, ok, one, rd, rdp_lo, rdp_hi, z_lo, z_hi, ts + 12, b.var(40), b.var(41), b.var(42));
    b.push_interaction(BITWISE_BUS, {x[0], x[1], zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {x[2], x[3], zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {y[0], y[1], zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {y[2], y[3], zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {z[0], z[1], zero, zero}, ok, Kind::Send);   // what is written to a register is a word of bytes
    b.push_interaction(BITWISE_BUS, {z[2], z[3], zero, zero}, ok, Kind::Send);
}

// The operand-bus message as a core chip states it: 20 fields (class, opcode, x[4], y[4], z[4], pc, imm_f, pc_inc, ts, imm_lo, imm_hi);
// fields a class does not use are zero on both sides (the frame masks them with the program's need_* flags).
struct OperandMsg {
    Expr op, x[4], y[4], z[4], pc, imm_f, pc_inc, ts, imm_lo, imm_hi;
};
inline void receive_operands(AirBuilder& b, uint32_t cls, const OperandMsg& m, Expr count) {
    std::vector<Expr> f{b.constant(cls), m.op};
    for (int i = 0; i < 4; i++) f.push_back(m.x[i]);
    for (int i = 0; i < 4; i++) f.push_back(m.y[i]);
    for (int i = 0; i < 4; i++) f.push_back(m.z[i]);
    for (const Expr& e : {m.pc, m.imm_f, m.pc_inc, m.ts, m.imm_lo, m.imm_hi}) f.push_back(e);
    b.push_interaction(OPERAND_BUS, f, count, Kind::Receive);
}
inline OperandMsg blank_msg(AirBuilder& b) {
    OperandMsg m;
    const Expr z = b.constant(0);
    m.op = z, m.pc = z, m.imm_f = z, m.pc_inc = b.constant(4), m.ts = z, m.imm_lo = z, m.imm_hi = z;
    for (int i = 0; i < 4; i++) m.x[i] = m.y[i] = m.z[i] = z;
    return m;
}
// The instruction cores of include/zkhip_chips.hpp, each with its receive on the operand bus (columns unchanged).
inline void core_air(AirBuilder& b, uint32_t cls) {
    OperandMsg m = blank_msg(b);
    auto cols = [&](Expr* dst, size_t first) { for (int i = 0; i < 4; i++) dst[i] = b.var(first + i); };
    Expr count;
    switch (cls) {
        case C_ALU:
            chips::rv32_alu_core_air(b);
            cols(m.z, 0), cols(m.x, 4), cols(m.y, 8);
            m.op = b.var(13) + b.var(14) * 2 + b.var(15) * 3 + b.var(16) * 4, count = b.var(17);
            break;
        case C_LT:
            chips::rv32_lt_core_air(b);
            cols(m.x, 0), cols(m.y, 4), m.z[0] = b.var(8), m.op = b.var(10), count = b.var(9) + b.var(10);
            break;
        case C_MUL:
            chips::rv32_mul_core_air(b);
            cols(m.z, 0), cols(m.x, 4), cols(m.y, 8), count = b.var(12);
            break;
        case C_MULH:
            chips::rv32_mulh_core_air(b);
            cols(m.z, 0), cols(m.x, 4), cols(m.y, 8), m.op = b.var(19) + b.var(20) * 2, count = b.var(18) + b.var(19) + b.var(20);
            break;
        case C_DIVREM: {
            chips::rv32_divrem_core_air(b);
            cols(m.x, 0), cols(m.y, 4);
            const Expr quot = b.var(37) + b.var(38), rem = b.var(39) + b.var(40);
            for (int i = 0; i < 4; i++) m.z[i] = quot * b.var(8 + i) + rem * b.var(12 + i);
            m.op = b.var(38) + b.var(39) * 2 + b.var(40) * 3, count = quot + rem;
            break;
        }
        case C_SHIFT:
            chips::rv32_shift_core_air(b);
            cols(m.z, 0), cols(m.x, 4), m.y[0] = b.var(8), m.op = b.var(10) + b.var(11) * 2, count = b.var(9) + b.var(10) + b.var(11);
            break;
        case C_BEQ:
            chips::rv32_branch_eq_core_air(b);
            cols(m.x, 0), cols(m.y, 4), m.imm_f = b.var(9), m.pc_inc = b.var(16), m.op = b.var(11), count = b.var(10) + b.var(11);
            break;
        case C_BLT:
            chips::rv32_branch_lt_core_air(b);
            cols(m.x, 0), cols(m.y, 4), m.imm_f = b.var(10), m.pc_inc = b.var(22), m.op = b.var(12) + b.var(13) * 2 + b.var(14) * 3;
            count = b.var(11) + b.var(12) + b.var(13) + b.var(14);
            break;
        case C_JAL_LUI:
            chips::rv32_jal_lui_core_air(b);
            cols(m.z, 2), m.pc = b.var(0), m.imm_f = b.var(1), m.pc_inc = b.var(8), m.op = b.var(7), count = b.var(6) + b.var(7);
            break;
        case C_AUIPC:
            chips::rv32_auipc_core_air(b);
            cols(m.z, 9), m.pc = b.var(0), m.imm_f = b.var(1), count = b.var(13);
            break;
        case C_JALR:
            chips::rv32_jalr_core_air(b);
            cols(m.x, 5), cols(m.z, 9), m.pc = b.var(0), m.imm_f = b.var(1), m.pc_inc = b.var(18) - b.var(0), count = b.var(19);
            break;
        default: throw std::invalid_argument("core_air: not a plain core class");
    }
    receive_operands(b, cls, m, count);
}
inline size_t core_width(uint32_t cls) {
    static const size_t w[11] = {18, 18, 13, 21, 41, 32, 17, 23, 9, 14, 20};
    return w[cls];
}

// load/store chip: AirBuilder(LS_WIDTH, 0).  Columns 0..32 are the core's (read[4] prev[4] write[4] flag[20] sign); then
// ts | base[4] | imm_lo imm_hi | addr_lo addr_hi | carry0 carry1 | word_lo (= addr_lo >> 2) | prev_ts gap_lo gap_hi of the word access
inline void loadstore_air(AirBuilder& b) {
    chips::rv32_loadstore_core_air(b);
    Expr rdv[4], pv[4], wr[4], fl[20], base[4];
    for (int i = 0; i < 4; i++) rdv[i] = b.var(i), pv[i] = b.var(4 + i), wr[i] = b.var(8 + i), base[i] = b.var(34 + i);
    for (int i = 0; i < 20; i++) fl[i] = b.var(12 + i);
    const Expr ts = b.var(33), imm_lo = b.var(38), imm_hi = b.var(39), a_lo = b.var(40), a_hi = b.var(41), c0 = b.var(42), c1 = b.var(43), w_lo = b.var(44);
    static const int KIND[20] = {0, 1, 1, 2, 2, 2, 2, 3, 4, 4, 5, 5, 5, 5, 6, 6, 7, 7, 7, 7};
    static const int SHIFT[20] = {0, 0, 2, 0, 1, 2, 3, 0, 0, 2, 0, 1, 2, 3, 0, 2, 0, 1, 2, 3};
    Expr ok = fl[0], kind = fl[0] * KIND[0], shift = fl[0] * SHIFT[0], is_store = fl[7];
    for (int c = 1; c < 20; c++) {
        ok = ok + fl[c], kind = kind + fl[c] * KIND[c], shift = shift + fl[c] * SHIFT[c];
        if (KIND[c] >= 3 && KIND[c] <= 5 && c != 7) is_store = is_store + fl[c];
    }
    const Expr is_load = ok - is_store;
    b.assert_zero(c0 * (c0 - 1));
    b.assert_zero(c1 * (c1 - 1));
    // address = base + immediate (mod 2^32), in 16-bit halves; byte offset inside the aligned word = the case's shift
    b.assert_zero(ok * (a_lo + c0 * 65536 - (base[0] + base[1] * 256) - imm_lo));
    b.assert_zero(ok * (a_hi + c1 * 65536 - (base[2] + base[3] * 256) - imm_hi - c0));
    b.assert_zero(ok * (a_lo - w_lo * 4 - shift));
    OperandMsg m = blank_msg(b);
    m.op = kind, m.ts = ts, m.imm_lo = imm_lo, m.imm_hi = imm_hi;
    for (int i = 0; i < 4; i++) m.x[i] = base[i], m.y[i] = is_store * rdv[i], m.z[i] = is_load * wr[i];
    receive_operands(b, C_LS, m, ok);
    // the aligned word: word pointer = address / 4; a load leaves it as it is (read = the memory word), a store replaces the previous
    // word by the core's merged word (the bytes a narrow store does not touch are the previous word's, by the core's constraints)
    const Expr word = w_lo + a_hi * 16384;
    const Expr two = b.constant(2);
    const Expr r0 = rdv[0] + rdv[1] * 256, r1 = rdv[2] + rdv[3] * 256, p0 = pv[0] + pv[1] * 256, p1 = pv[2] + pv[3] * 256, n0 = wr[0] + wr[1] * 256,
               n1 = wr[2] + wr[3] * 256;
    word_access(b, ok, ok, two, word, is_load * r0 + is_store * p0, is_load * r1 + is_store * p1, is_load * r0 + is_store * n0, is_load * r1 + is_store * n1, ts + 4,
                b.var(45), b.var(46), b.var(47));
    const Expr zero = b.constant(0);
    b.push_interaction(BITWISE_BUS, {rdv[0], rdv[1], zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {rdv[2], rdv[3], zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {pv[0], pv[1], zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {pv[2], pv[3], zero, zero}, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {a_lo}, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {a_hi * 4}, ok, Kind::Send);   // address below 2^30
    b.push_interaction(RANGE_BUS, {w_lo * 4}, ok, Kind::Send);   // with the next one: word_lo below 2^14
    b.push_interaction(RANGE_BUS, {w_lo}, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {a_hi}, ok, Kind::Send);
}

// ecall chip: AirBuilder(ECALL_WIDTH, 0).  Columns: pc ts | x[4] (a7) | y[4] (a0) | z[4] (a0 afterwards) | is_exit is_reveal is_read |
// idx (a1, the public-value word index) | pv_prev_lo pv_prev_hi | pc_inc | (prev_ts gap_lo gap_hi) of the a1 read and the public-value write |
// is_keccak | q (= a0's low byte / 4: the state must be word-aligned) | is_sha | is_mod | is_int | is_ec | is_fp2 | is_nat | is_next | is_castf |
// is_br taken | off_lo off_hi off_neg | (prev_ts gap_lo gap_hi) of the a2 read
// A 256-BIT BRANCH (round 6; OpenVM's Rv32BranchEqual256 / Rv32BranchLessThan256) is an int256 call (is_int) with is_br: the comparison chip
// sends its decision `taken` on the branch bus, a2 (register 12, read at ts + 6) is the byte offset, and the row's pc_inc is
// 4 + taken (off - 4), off = off_lo + 2^16 off_hi - 2^32 off_neg (off_neg = the word's sign bit: backward branches).
inline void ecall_air(AirBuilder& b) {
    const Expr pc = b.var(0), ts = b.var(1);
    Expr x[4], y[4], z[4];
    for (int i = 0; i < 4; i++) x[i] = b.var(2 + i), y[i] = b.var(6 + i), z[i] = b.var(10 + i);
    const Expr is_exit = b.var(14), is_reveal = b.var(15), is_read = b.var(16), idx = b.var(17), pvp_lo = b.var(18), pvp_hi = b.var(19), pc_inc = b.var(20);
    const Expr is_keccak = b.var(27), q = b.var(28), is_sha = b.var(29), is_mod = b.var(30), is_int = b.var(31), is_ec = b.var(32), is_fp2 = b.var(33), is_nat = b.var(34),
               is_next = b.var(35), is_castf = b.var(36), is_hash = is_keccak + is_sha + is_mod + is_int + is_ec + is_fp2 + is_nat + is_next + is_castf;
    const Expr ok = is_exit + is_reveal + is_read + is_keccak + is_sha + is_mod + is_int + is_ec + is_fp2 + is_nat + is_next + is_castf;
    const Expr is_br = b.var(37), taken = b.var(38), off_lo = b.var(39), off_hi = b.var(40), off_neg = b.var(41);
    for (const Expr& f : {is_exit, is_reveal, is_read, is_keccak, is_sha, is_mod, is_int, is_ec, is_fp2, is_nat, is_next, is_castf, ok, is_br, taken, off_neg}) b.assert_zero(f * (f - 1));
    b.assert_zero(is_br * (1 - is_int));   // a branch is an int256 call
    b.assert_zero(taken * (1 - is_br));
    b.assert_zero(x[0] - is_exit * 93 - is_reveal - is_read * 2 - is_keccak * 3 - is_sha * 4 - is_mod * 5 - is_int * 6 - is_ec * 7 - is_fp2 * 8 - is_nat * (int64_t)native::CALL_ARITH -
                  is_next * (int64_t)native::CALL_EXT - is_castf * (int64_t)native::CALL_CASTF);
    for (int i = 1; i < 4; i++) b.assert_zero(x[i]);
    for (int i = 0; i < 4; i++) {
        b.assert_zero(is_exit * y[i]);                 // exit code 0: a failing guest has no proof
        b.assert_zero((ok - is_read) * (z[i] - y[i]));  // only the hint read changes a0
    }
    b.assert_zero(is_exit * (pc_inc + pc));             // the execution ends at pc = 0
    b.assert_zero((ok - is_exit - is_br) * (pc_inc - 4));
    {   // a 256-bit branch: pc + 4, or pc + a2 when the comparison chip says so
        const Expr off = off_lo + off_hi * 65536 - off_neg * (int64_t)(4294967296ull % 2013265921ull);
        b.assert_zero(is_br * (pc_inc - 4 - taken * (off - 4)));
    }
    OperandMsg m = blank_msg(b);
    m.pc = pc, m.pc_inc = pc_inc, m.ts = ts;
    for (int i = 0; i < 4; i++) m.x[i] = x[i], m.y[i] = y[i], m.z[i] = z[i];
    receive_operands(b, C_ECALL, m, ok);
    const Expr one = b.constant(1), zero = b.constant(0), three = b.constant(3);
    // reveal: a1 (register 11) is the word index, below 8; the word a0 goes to word idx of address space 3
    // (modmul, int256 and ecc read a1 the same way: the index of the modulus / curve, the opcode)
    // (the native field and extension calls read their opcode there too; castf has no second argument)
    const Expr reads_a1 = is_reveal + is_mod + is_int + is_ec + is_fp2 + is_nat + is_next;
    word_access(b, reads_a1, reads_a1, one, b.constant(11), idx, zero, idx, zero, ts + 4, b.var(21), b.var(22), b.var(23));
    word_access(b, is_reveal, is_reveal, three, idx, pvp_lo, pvp_hi, y[0] + y[1] * 256, y[2] + y[3] * 256, ts + 5, b.var(24), b.var(25), b.var(26));
    b.push_interaction(RANGE_BUS, {idx * 8192}, is_reveal, Kind::Send);
    b.push_interaction(RANGE_BUS, {idx}, is_reveal, Kind::Send);
    // the branch: a2 read (its halves are whatever the register holds: honest 16-bit cells by the memory argument), its sign bit split off
    // by one range lookup ((off_hi - 2^15 off_neg) 2 < 2^16), the decision from the comparison chip
    word_access(b, is_br, is_br, one, b.constant(12), off_lo, off_hi, off_lo, off_hi, ts + 6, b.var(42), b.var(43), b.var(44));
    b.push_interaction(RANGE_BUS, {(off_hi - off_neg * 32768) * 2}, is_br, Kind::Send);
    b.push_interaction(BRANCH256_BUS, {ts, taken}, is_br, Kind::Receive);
    // keccak: the state's word pointer = a0 / 4 (a0 word-aligned, below 2^30) goes to the keccak adapter with the call's timestamp
    // (sha256: the same with the 24-word state-and-block buffer and the SHA-256 adapter)
    b.assert_zero(is_hash * (y[0] - q * 4));
    const Expr wptr = q + y[1] * 64 + y[2] * 16384 + y[3] * 4194304;
    b.push_interaction(KECCAK_REQ_BUS, {ts, wptr}, is_keccak, Kind::Send);
    b.push_interaction(SHA_REQ_BUS, {ts, wptr}, is_sha, Kind::Send);
    b.push_interaction(MODMUL_REQ_BUS, {ts, wptr, idx}, is_mod, Kind::Send);
    b.push_interaction(INT256_REQ_BUS, {ts, wptr, idx}, is_int, Kind::Send);
    b.push_interaction(EC_REQ_BUS, {ts, wptr, idx}, is_ec, Kind::Send);
    b.push_interaction(FP2_REQ_BUS, {ts, wptr, idx}, is_fp2, Kind::Send);
    b.push_interaction(NATIVE_REQ_BUS, {ts, wptr, idx}, is_nat, Kind::Send);
    b.push_interaction(NATIVE_EXT_REQ_BUS, {ts, wptr, idx}, is_next, Kind::Send);
    b.push_interaction(CASTF_REQ_BUS, {ts, wptr}, is_castf, Kind::Send);
    b.push_interaction(RANGE_BUS, {q * 1024}, is_hash, Kind::Send);
    b.push_interaction(RANGE_BUS, {q}, is_hash, Kind::Send);
    b.push_interaction(RANGE_BUS, {y[3] * 1024}, is_hash, Kind::Send);
    // the hinted word is a word of bytes (the other cases copy a0, whose bytes the frame chip checks)
    b.push_interaction(BITWISE_BUS, {z[0], z[1], zero, zero}, is_read, Kind::Send);
    b.push_interaction(BITWISE_BUS, {z[2], z[3], zero, zero}, is_read, Kind::Send);
}

// keccak adapter: AirBuilder(KECCAK_IO_WIDTH, 0).  Columns: lane[25] (one-hot: the row's lane index x + 5 y) | ts | ptr (word pointer of
// the state) | in[4] | out[4] (the lane's 16-bit limbs before / after) | (prev_ts gap_lo gap_hi) of its two word accesses | ok.  A call
// occupies 25 consecutive rows: lane k is followed by lane k + 1 with the same (ts, ptr); lane 0 receives the call from the ecall chip.
inline void keccak_io_air(AirBuilder& b) {
    Expr lane[25], in[4], out[4];
    for (int k = 0; k < 25; k++) lane[k] = b.var(k);
    const Expr ts = b.var(25), ptr = b.var(26), ok = b.var(41);
    for (int l = 0; l < 4; l++) in[l] = b.var(27 + l), out[l] = b.var(31 + l);
    Expr sum = lane[0], idx = b.constant(0);
    for (int k = 1; k < 25; k++) sum = sum + lane[k], idx = idx + lane[k] * k;
    for (int k = 0; k < 25; k++) b.assert_zero(lane[k] * (lane[k] - 1));
    b.assert_zero(ok * (ok - 1));
    b.assert_zero(sum - ok);
    for (int k = 0; k < 24; k++) b.when_transition(b.var(k + 1, 1) - lane[k]);   // a started call runs through all 25 lanes
    b.when_first_row(sum - lane[0]);                                              // no call starts in the middle
    const Expr cont = b.var(41, 1) - b.var(0, 1);                                 // the next row continues this call
    b.when_transition(cont * (b.var(25, 1) - ts));
    b.when_transition(cont * (b.var(26, 1) - ptr));
    b.push_interaction(KECCAK_REQ_BUS, {ts, ptr}, lane[0], Kind::Receive);
    const Expr two = b.constant(2), w0 = ptr + idx * 2;
    word_access(b, ok, ok, two, w0, in[0], in[1], out[0], out[1], ts + 4, b.var(35), b.var(36), b.var(37));
    word_access(b, ok, ok, two, w0 + 1, in[2], in[3], out[2], out[3], ts + 4, b.var(38), b.var(39), b.var(40));
    b.push_interaction(KECCAK_LANE_BUS, {ts, idx, in[0], in[1], in[2], in[3], out[0], out[1], out[2], out[3]}, ok, Kind::Send);
}

// sha256 adapter: AirBuilder(SHA_IO_WIDTH, 0).  Columns: word[24] (one-hot: 0..7 the state words, 8..23 the message words) | ts | base
// (word pointer of the buffer) | v_lo v_hi (the word before) | n_lo n_hi (after: H_out for a state word, unchanged for a message
// word) | prev_ts gap_lo gap_hi | ok.  A call occupies 24 consecutive rows, like the keccak adapter's 25.
inline void sha256_io_air(AirBuilder& b) {
    Expr sel[24];
    for (int k = 0; k < 24; k++) sel[k] = b.var(k);
    const Expr ts = b.var(24), base = b.var(25), v_lo = b.var(26), v_hi = b.var(27), n_lo = b.var(28), n_hi = b.var(29), ok = b.var(33);
    Expr sum = sel[0], idx = b.constant(0), is_state = sel[0];
    for (int k = 1; k < 24; k++) {
        sum = sum + sel[k], idx = idx + sel[k] * k;
        if (k < 8) is_state = is_state + sel[k];
    }
    const Expr is_msg = ok - is_state;
    for (int k = 0; k < 24; k++) b.assert_zero(sel[k] * (sel[k] - 1));
    b.assert_zero(ok * (ok - 1));
    b.assert_zero(sum - ok);
    for (int k = 0; k < 23; k++) b.when_transition(b.var(k + 1, 1) - sel[k]);
    b.when_first_row(sum - sel[0]);
    const Expr cont = b.var(33, 1) - b.var(0, 1);
    b.when_transition(cont * (b.var(24, 1) - ts));
    b.when_transition(cont * (b.var(25, 1) - base));
    b.assert_zero(is_msg * (n_lo - v_lo));
    b.assert_zero(is_msg * (n_hi - v_hi));
    b.push_interaction(SHA_REQ_BUS, {ts, base}, sel[0], Kind::Receive);
    word_access(b, ok, ok, b.constant(2), base + idx, v_lo, v_hi, n_lo, n_hi, ts + 4, b.var(30), b.var(31), b.var(32));
    b.push_interaction(SHA_MSG_BUS, {ts, idx - 8, v_lo, v_hi}, is_msg, Kind::Send);
    b.push_interaction(SHA_STATE_BUS, {ts, idx, v_lo, v_hi, n_lo, n_hi}, is_state, Kind::Send);
}

// modular adapter of modulus `index`: AirBuilder(modmul_io_width(nw), 0) for a modulus of nw words (8: up to 256 bits; 12: up to 384).
// The sha256 adapter's columns: word[3 nw] (a, b: read; r: written) | ts | base | v_lo v_hi | n_lo n_hi | prev_ts gap_lo gap_hi | ok |
// op (the call's operation: a1 = index + 8 op).
constexpr size_t modmul_io_width(size_t nw) { return 3 * nw + 11; }
constexpr size_t ec_io_width(size_t nw) { return 6 * nw + 11; }
inline void modmul_io_air(AirBuilder& b, uint32_t index, size_t nw = 8) {
    const int W = (int)(3 * nw), RD = (int)(2 * nw);
    std::vector<Expr> sel(W);
    for (int k = 0; k < W; k++) sel[k] = b.var(k);
    const Expr ts = b.var(W), base = b.var(W + 1), v_lo = b.var(W + 2), v_hi = b.var(W + 3), n_lo = b.var(W + 4), n_hi = b.var(W + 5), ok = b.var(W + 9), op = b.var(W + 10);
    Expr sum = sel[0], idx = b.constant(0), is_read = sel[0];
    for (int k = 1; k < W; k++) {
        sum = sum + sel[k], idx = idx + sel[k] * k;
        if (k < RD) is_read = is_read + sel[k];
    }
    for (int k = 0; k < W; k++) b.assert_zero(sel[k] * (sel[k] - 1));
    b.assert_zero(ok * (ok - 1));
    b.assert_zero(sum - ok);
    for (int k = 0; k + 1 < W; k++) b.when_transition(b.var(k + 1, 1) - sel[k]);
    b.when_first_row(sum - sel[0]);
    const Expr cont = b.var(W + 9, 1) - b.var(0, 1);
    b.when_transition(cont * (b.var(W, 1) - ts));
    b.when_transition(cont * (b.var(W + 1, 1) - base));
    b.assert_zero(is_read * (n_lo - v_lo));
    b.assert_zero(is_read * (n_hi - v_hi));
    b.when_transition(cont * (b.var(W + 10, 1) - op));
    b.push_interaction(MODMUL_REQ_BUS, {ts, base, op * 8 + (int64_t)index}, sel[0], Kind::Receive);
    word_access(b, ok, ok, b.constant(2), base + idx, v_lo, v_hi, n_lo, n_hi, ts + 5, b.var(W + 6), b.var(W + 7), b.var(W + 8));
    b.push_interaction(MODMUL_WORD_BUS + index, {ts, idx, n_lo, n_hi, op}, ok, Kind::Send);
}

// int256 adapter: AirBuilder(INT256_IO_WIDTH, 0).  The modular adapter's columns (0..7 b, 8..15 c: read; 16..23 a: written) + op (the
// call's opcode, handed to the ALU chip with every word).
inline void int256_io_air(AirBuilder& b) {
    Expr sel[24];
    for (int k = 0; k < 24; k++) sel[k] = b.var(k);
    const Expr ts = b.var(24), base = b.var(25), v_lo = b.var(26), v_hi = b.var(27), n_lo = b.var(28), n_hi = b.var(29), ok = b.var(33), op = b.var(34);
    Expr sum = sel[0], idx = b.constant(0), is_read = sel[0];
    for (int k = 1; k < 24; k++) {
        sum = sum + sel[k], idx = idx + sel[k] * k;
        if (k < 16) is_read = is_read + sel[k];
    }
    for (int k = 0; k < 24; k++) b.assert_zero(sel[k] * (sel[k] - 1));
    b.assert_zero(ok * (ok - 1));
    b.assert_zero(sum - ok);
    for (int k = 0; k < 23; k++) b.when_transition(b.var(k + 1, 1) - sel[k]);
    b.when_first_row(sum - sel[0]);
    const Expr cont = b.var(33, 1) - b.var(0, 1);
    b.when_transition(cont * (b.var(24, 1) - ts));
    b.when_transition(cont * (b.var(25, 1) - base));
    b.when_transition(cont * (b.var(34, 1) - op));
    b.assert_zero(is_read * (n_lo - v_lo));
    b.assert_zero(is_read * (n_hi - v_hi));
    b.push_interaction(INT256_REQ_BUS, {ts, base, op}, sel[0], Kind::Receive);
    word_access(b, ok, ok, b.constant(2), base + idx, v_lo, v_hi, n_lo, n_hi, ts + 5, b.var(30), b.var(31), b.var(32));
    b.push_interaction(INT256_WORD_BUS, {ts, idx, n_lo, n_hi, op}, ok, Kind::Send);
}

// ecc adapter of curve `index`: AirBuilder(ec_io_width(nw), 0).  The modular adapter's shape with 6 nw words: word[6 nw] (x1 y1, x2 y2:
// read; x3 y3: written) | ts | base | v_lo v_hi | n_lo n_hi | prev_ts gap_lo gap_hi | ok | op (a1 = index + 8 op).
// (the fp2 adapter has the same shape on its own pair of buses: two operands of 2 nw words read, a result of 2 nw words written)
inline void ec_io_air(AirBuilder& b, uint32_t index, uint32_t req_bus = EC_REQ_BUS, uint32_t word_bus_base = EC_WORD_BUS, size_t nw = 8) {
    const int W = (int)(6 * nw), RD = (int)(4 * nw);
    std::vector<Expr> sel(W);
    for (int k = 0; k < W; k++) sel[k] = b.var(k);
    const Expr ts = b.var(W), base = b.var(W + 1), v_lo = b.var(W + 2), v_hi = b.var(W + 3), n_lo = b.var(W + 4), n_hi = b.var(W + 5), ok = b.var(W + 9), op = b.var(W + 10);
    Expr sum = sel[0], idx = b.constant(0), is_read = sel[0];
    for (int k = 1; k < W; k++) {
        sum = sum + sel[k], idx = idx + sel[k] * k;
        if (k < RD) is_read = is_read + sel[k];
    }
    for (int k = 0; k < W; k++) b.assert_zero(sel[k] * (sel[k] - 1));
    b.assert_zero(ok * (ok - 1));
    b.assert_zero(sum - ok);
    for (int k = 0; k + 1 < W; k++) b.when_transition(b.var(k + 1, 1) - sel[k]);
    b.when_first_row(sum - sel[0]);
    const Expr cont = b.var(W + 9, 1) - b.var(0, 1);
    b.when_transition(cont * (b.var(W, 1) - ts));
    b.when_transition(cont * (b.var(W + 1, 1) - base));
    b.when_transition(cont * (b.var(W + 10, 1) - op));
    b.assert_zero(is_read * (n_lo - v_lo));
    b.assert_zero(is_read * (n_hi - v_hi));
    b.push_interaction(req_bus, {ts, base, op * 8 + (int64_t)index}, sel[0], Kind::Receive);
    word_access(b, ok, ok, b.constant(2), base + idx, v_lo, v_hi, n_lo, n_hi, ts + 5, b.var(W + 6), b.var(W + 7), b.var(W + 8));
    b.push_interaction(word_bus_base + index, {ts, idx, n_lo, n_hi, op}, ok, Kind::Send);
}

// A native field element in memory: ONE word (lo, hi halves) holding a value read as lo + 2^16 hi (mod p).  A RESULT is written canonical:
// hi <= 0x7800 (both hi and hi_gap = 0x7800 - hi in the range table) and, where hi = 0x7800 (flag `top`, decided through an inverse of
// hi_gap), lo = 0 -- i.e. the word is below p = 0x78000001.  Columns of one result: lo hi hi_gap top top_inv.
inline void canonical_word(AirBuilder& b, Expr ok, Expr lo, Expr hi, Expr hi_gap, Expr top, Expr top_inv) {
    b.assert_zero(top * (top - 1));
    b.assert_zero(ok * (hi_gap + hi - (int64_t)native::P_HI));
    b.assert_zero(ok * (hi_gap * top_inv - 1 + top));
    b.assert_zero(top * hi_gap);
    b.assert_zero(top * lo);
    b.push_interaction(RANGE_BUS, {lo}, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {hi}, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {hi_gap}, ok, Kind::Send);
}

// native field-arithmetic chip: AirBuilder(NATIVE_ARITH_WIDTH, 0), ONE row per call (a7 = 9).  Columns: ts | base (word pointer of the
// 3-word buffer) | b_lo b_hi | c_lo c_hi (the operands, read) | a_lo a_hi a_hi_gap top top_inv (the result a = b op c, canonical) |
// ap_lo ap_hi (the result word before) | is_add is_sub is_mul is_div | div_inv (c^-1 on a division) | (prev_ts gap_lo gap_hi) of the
// three word accesses.  The core is OpenVM's FieldArithmeticCoreAir (air.py field_arith_air); the accesses are its native adapter's.
inline void native_arith_air(AirBuilder& b) {
    const Expr ts = b.var(0), base = b.var(1), b_lo = b.var(2), b_hi = b.var(3), c_lo = b.var(4), c_hi = b.var(5), a_lo = b.var(6), a_hi = b.var(7), ap_lo = b.var(11),
               ap_hi = b.var(12), add = b.var(13), sub = b.var(14), mul = b.var(15), div = b.var(16), inv = b.var(17);
    const Expr ok = add + sub + mul + div;
    for (const Expr& f : {add, sub, mul, div, ok}) b.assert_zero(f * (f - 1));
    const Expr bv = b_lo + b_hi * 65536, cv = c_lo + c_hi * 65536, av = a_lo + a_hi * 65536;
    b.assert_zero(add * (av - bv - cv));
    b.assert_zero(sub * (av - bv + cv));
    b.assert_zero(mul * (av - bv * cv));
    b.assert_zero(div * (bv - av * cv));
    b.assert_zero(div * (cv * inv - 1));
    canonical_word(b, ok, a_lo, a_hi, b.var(8), b.var(9), b.var(10));
    b.push_interaction(NATIVE_REQ_BUS, {ts, base, sub + mul * 2 + div * 3}, ok, Kind::Receive);
    const Expr two = b.constant(2);
    word_access(b, ok, ok, two, base, b_lo, b_hi, b_lo, b_hi, ts + 5, b.var(18), b.var(19), b.var(20));
    word_access(b, ok, ok, two, base + 1, c_lo, c_hi, c_lo, c_hi, ts + 5, b.var(21), b.var(22), b.var(23));
    word_access(b, ok, ok, two, base + 2, ap_lo, ap_hi, a_lo, a_hi, ts + 5, b.var(24), b.var(25), b.var(26));
}

// native extension chip: AirBuilder(NATIVE_EXT_WIDTH, 0), ONE row per call (a7 = 10) on F[X] / (X^4 - 11).  Columns: ts | base | x[4] (lo hi) |
// y[4] (lo hi) | z[4] (lo hi hi_gap top top_inv: the result, canonical) | zp[4] (lo hi: the result words before) | is_add is_sub is_mul
// is_div | inv[4] (y^-1 on a division) | (prev_ts gap_lo gap_hi) of the twelve word accesses.  Core: OpenVM's FieldExtensionCoreAir
// (air.py field_ext_air): z = x op y; a division is z = x inv with y inv = 1.
inline void native_ext_air(AirBuilder& b) {
    const Expr ts = b.var(0), base = b.var(1);
    Expr xl[4], xh[4], yl[4], yh[4], zl[4], zh[4], zpl[4], zph[4], x[4], y[4], z[4], inv[4];
    for (int i = 0; i < 4; i++) {
        xl[i] = b.var(2 + 2 * i), xh[i] = b.var(3 + 2 * i), yl[i] = b.var(10 + 2 * i), yh[i] = b.var(11 + 2 * i);
        zl[i] = b.var(18 + 5 * i), zh[i] = b.var(19 + 5 * i), zpl[i] = b.var(38 + 2 * i), zph[i] = b.var(39 + 2 * i), inv[i] = b.var(50 + i);
        x[i] = xl[i] + xh[i] * 65536, y[i] = yl[i] + yh[i] * 65536, z[i] = zl[i] + zh[i] * 65536;
    }
    const Expr add = b.var(46), sub = b.var(47), mul = b.var(48), div = b.var(49), ok = add + sub + mul + div;
    for (const Expr& f : {add, sub, mul, div, ok}) b.assert_zero(f * (f - 1));
    auto ext_mul = [&](const Expr* p, const Expr* q, Expr* r) {   // (p q) mod (X^4 - 11)
        for (int k = 0; k < 4; k++) {
            Expr lo = b.constant(0), hi = b.constant(0);
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    if (i + j == k) lo = lo + p[i] * q[j];
                    if (i + j == k + 4) hi = hi + p[i] * q[j];
                }
            r[k] = lo + hi * (int64_t)native::W;
        }
    };
    Expr xy[4], xi[4], yi[4];
    ext_mul(x, y, xy), ext_mul(x, inv, xi), ext_mul(y, inv, yi);
    for (int i = 0; i < 4; i++) {
        b.assert_zero(add * (z[i] - x[i] - y[i]));
        b.assert_zero(sub * (z[i] - x[i] + y[i]));
        b.assert_zero(mul * (z[i] - xy[i]));
        b.assert_zero(div * (z[i] - xi[i]));
        b.assert_zero(div * (yi[i] - (i == 0 ? 1 : 0)));
        canonical_word(b, ok, zl[i], zh[i], b.var(20 + 5 * i), b.var(21 + 5 * i), b.var(22 + 5 * i));
    }
    b.push_interaction(NATIVE_EXT_REQ_BUS, {ts, base, sub + mul * 2 + div * 3}, ok, Kind::Receive);
    const Expr two = b.constant(2);
    for (int i = 0; i < 4; i++) {
        word_access(b, ok, ok, two, base + i, xl[i], xh[i], xl[i], xh[i], ts + 5, b.var(54 + 3 * i), b.var(55 + 3 * i), b.var(56 + 3 * i));
        word_access(b, ok, ok, two, base + (4 + i), yl[i], yh[i], yl[i], yh[i], ts + 5, b.var(66 + 3 * i), b.var(67 + 3 * i), b.var(68 + 3 * i));
        word_access(b, ok, ok, two, base + (8 + i), zpl[i], zph[i], zl[i], zh[i], ts + 5, b.var(78 + 3 * i), b.var(79 + 3 * i), b.var(80 + 3 * i));
    }
}

// castf chip: AirBuilder(CASTF_WIDTH, 0), ONE row per call (a7 = 11).  Columns: ts | base | limb[4] (8, 8, 8, 6 bits: OpenVM's CastFCoreAir) |
// limb3_x4 (= 4 limb_3: a byte, so limb_3 < 64) | op_lo op_hi (the output word before) | (prev_ts gap_lo gap_hi) of the read of word `base`
// and of the write of word `base + 1` | ok.  The input word's halves ARE limb_0 + 256 limb_1 and limb_2 + 256 limb_3: it lies below 2^30.
inline void castf_vm_air(AirBuilder& b) {
    const Expr ts = b.var(0), base = b.var(1), l0 = b.var(2), l1 = b.var(3), l2 = b.var(4), l3 = b.var(5), l3x4 = b.var(6), op_lo = b.var(7), op_hi = b.var(8), ok = b.var(15);
    b.assert_zero(ok * (ok - 1));
    b.assert_zero(l3x4 - l3 * 4);
    const Expr lo = l0 + l1 * 256, hi = l2 + l3 * 256, zero = b.constant(0), two = b.constant(2);
    b.push_interaction(CASTF_REQ_BUS, {ts, base}, ok, Kind::Receive);
    word_access(b, ok, ok, two, base, lo, hi, lo, hi, ts + 5, b.var(9), b.var(10), b.var(11));
    word_access(b, ok, ok, two, base + 1, op_lo, op_hi, lo, hi, ts + 5, b.var(12), b.var(13), b.var(14));
    b.push_interaction(BITWISE_BUS, {l0, l1, zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {l2, l3, zero, zero}, ok, Kind::Send);
    b.push_interaction(BITWISE_BUS, {l3x4, zero, zero, zero}, ok, Kind::Send);
}

// leaf chip: AirBuilder(LEAF_WIDTH, 0).  Columns: as blk | init[8] | fin[8] | fin_ts[4] | h_init[8] | h_fin[8] | ok | gap_lo gap_hi |
// blk_lo blk_hi.  One row per touched block of 4 words (cells 2 j, 2 j + 1 = the halves of word j), strictly increasing
// label = as 2^26 + blk; padding rows carry as = 1.
inline void leaf_air(AirBuilder& b) {
    const Expr as_ = b.var(0), blk = b.var(1), ok = b.var(38), gap_lo = b.var(39), gap_hi = b.var(40), blk_lo = b.var(41), blk_hi = b.var(42);
    Expr init[8], fin[8], fts[4], hi[8], hf[8];
    for (int j = 0; j < 8; j++) init[j] = b.var(2 + j), fin[j] = b.var(10 + j), hi[j] = b.var(22 + j), hf[j] = b.var(30 + j);
    for (int j = 0; j < 4; j++) fts[j] = b.var(18 + j);
    const Expr ok_n = b.var(38, 1);
    b.assert_zero(ok * (ok - 1));
    b.when_transition(ok_n * (1 - ok));                              // valid rows come first
    b.assert_zero((as_ - 1) * (as_ - 2) * (as_ - 3));                // address space 1, 2 or 3 (padding rows: 1)
    const Expr label = as_ * (int64_t)(1u << LABEL_BITS) + blk, label_n = b.var(0, 1) * (int64_t)(1u << LABEL_BITS) + b.var(1, 1);
    b.when_transition(ok_n * (label_n - label - 1 - gap_lo - gap_hi * 65536));
    b.assert_zero(blk - blk_lo - blk_hi * 65536);
    const Expr zero = b.constant(0);
    for (int j = 0; j < 4; j++) {
        b.push_interaction(MEMORY_BUS, {as_, blk * 4 + j, init[2 * j], init[2 * j + 1], zero}, ok, Kind::Send);
        b.push_interaction(MEMORY_BUS, {as_, blk * 4 + j, fin[2 * j], fin[2 * j + 1], fts[j]}, ok, Kind::Receive);
    }
    std::vector<Expr> hin, hfn;
    for (int j = 0; j < 8; j++) hin.push_back(init[j]), hfn.push_back(fin[j]);
    for (int j = 0; j < 8; j++) hin.push_back(zero), hfn.push_back(zero);
    for (int j = 0; j < 8; j++) hin.push_back(hi[j]), hfn.push_back(hf[j]);
    b.push_interaction(HASH_BUS, hin, ok, Kind::Send);
    b.push_interaction(HASH_BUS, hfn, ok, Kind::Send);
    std::vector<Expr> mk{b.constant(LEAF_LEVEL), label};
    for (int j = 0; j < 8; j++) mk.push_back(hi[j]);
    for (int j = 0; j < 8; j++) mk.push_back(hf[j]);
    b.push_interaction(MERKLE_BUS, mk, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {gap_lo}, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {gap_hi * 16}, ok, Kind::Send);     // gap below 2^28: labels cannot wrap
    b.push_interaction(RANGE_BUS, {gap_hi}, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {blk_lo}, ok, Kind::Send);
    b.push_interaction(RANGE_BUS, {blk_hi * 64}, ok, Kind::Send);     // blk below 2^26
    b.push_interaction(RANGE_BUS, {blk_hi}, ok, Kind::Send);
}

// merkle chip: AirBuilder(MERKLE_WIDTH, 16).  Columns: level idx | left_init[8] right_init[8] | left_fin[8] right_fin[8] | parent_init[8]
// parent_fin[8] | left_on right_on | ok | is_root.  Public values: the initial root (8), the final root (8).
inline void merkle_air(AirBuilder& b) {
    const Expr level = b.var(0), idx = b.var(1), l_on = b.var(50), r_on = b.var(51), ok = b.var(52), is_root = b.var(53);
    Expr li[8], ri[8], lf[8], rf[8], pi[8], pf[8];
    for (int k = 0; k < 8; k++) li[k] = b.var(2 + k), ri[k] = b.var(10 + k), lf[k] = b.var(18 + k), rf[k] = b.var(26 + k), pi[k] = b.var(34 + k), pf[k] = b.var(42 + k);
    for (const Expr& f : {l_on, r_on, ok, is_root}) b.assert_zero(f * (f - 1));
    b.assert_zero((1 - ok) * l_on);
    b.assert_zero((1 - ok) * r_on);
    b.assert_zero((1 - ok) * is_root);
    b.assert_zero(ok * ((1 - l_on) * (1 - r_on)));   // at least one child lies on a touched path
    for (int k = 0; k < 8; k++) {
        b.assert_zero((ok - l_on) * (li[k] - lf[k]));   // an untouched subtree is the same before and after
        b.assert_zero((ok - r_on) * (ri[k] - rf[k]));
        b.assert_zero(is_root * (pi[k] - b.pub(k)));
        b.assert_zero(is_root * (pf[k] - b.pub(8 + k)));
    }
    b.assert_zero(is_root * level);
    b.assert_zero(is_root * idx);
    b.when_transition(b.var(53, 1));                    // only the first row can be the root
    b.when_first_row(ok - is_root);                     // and if there is any row at all, the first one is
    auto cat = [](const Expr* a, const Expr* c, const Expr* o) {
        std::vector<Expr> v(a, a + 8);
        v.insert(v.end(), c, c + 8);
        v.insert(v.end(), o, o + 8);
        return v;
    };
    b.push_interaction(HASH_BUS, cat(li, ri, pi), ok, Kind::Send);
    b.push_interaction(HASH_BUS, cat(lf, rf, pf), ok, Kind::Send);
    auto node = [&](Expr lv, Expr ix, const Expr* a, const Expr* c) {
        std::vector<Expr> v{lv, ix};
        v.insert(v.end(), a, a + 8);
        v.insert(v.end(), c, c + 8);
        return v;
    };
    b.push_interaction(MERKLE_BUS, node(level + 1, idx * 2, li, lf), l_on, Kind::Receive);
    b.push_interaction(MERKLE_BUS, node(level + 1, idx * 2 + 1, ri, rf), r_on, Kind::Receive);
    b.push_interaction(MERKLE_BUS, node(level, idx, pi, pf), ok - is_root, Kind::Send);
}

// connector: AirBuilder(CONNECTOR_WIDTH, 2), ONE row.  Public values (pc_start, pc_end); columns = the 16-bit halves of the final
// timestamp (bounded below 2^29, which bounds every timestamp of the segment).
inline void connector_air(AirBuilder& b) {
    const Expr te_lo = b.var(0), te_hi = b.var(1), one = b.constant(1);
    b.push_interaction(EXEC_BUS, {b.pub(0), one}, one, Kind::Send);
    b.push_interaction(EXEC_BUS, {b.pub(1), te_lo + te_hi * 65536}, one, Kind::Receive);
    b.push_interaction(RANGE_BUS, {te_lo}, one, Kind::Send);
    b.push_interaction(RANGE_BUS, {te_hi * 8}, one, Kind::Send);
    b.push_interaction(RANGE_BUS, {te_hi}, one, Kind::Send);
}

// ---- the segment's AIR set (order fixed: it is part of the verifying key) ----
enum AirId : unsigned {
    A_PROGRAM, A_FRAME, A_ALU, A_LT, A_MUL, A_MULH, A_DIVREM, A_SHIFT, A_BEQ, A_BLT, A_JAL_LUI, A_AUIPC, A_JALR, A_LS, A_ECALL, A_LEAF,
    A_MERKLE, A_POSEIDON2, A_CONNECTOR, A_BITWISE, A_RANGE_TUPLE, A_RANGE, A_KECCAK, A_KECCAK_IO, A_SHA256, A_SHA256_IO, A_INT256, A_INT256_IO, A_MUL256, A_CMP256, A_SHIFT256, A_NATIVE_ARITH, A_NATIVE_EXT, A_CASTF, N_STATIC_AIRS
};
// the modular extension brings two chips per configured modulus (openvm.toml `supported_moduli`): ids A_MODMUL(i), A_MODMUL_IO(i)
constexpr unsigned MAX_MODULI = 8, MAX_CURVES = 4, MAX_FP2 = 2, N_AIRS = N_STATIC_AIRS + 2 * MAX_MODULI + 2 * MAX_CURVES + 2 * MAX_FP2;
constexpr unsigned A_MODMUL(unsigned i) { return N_STATIC_AIRS + 2 * i; }
constexpr unsigned A_MODMUL_IO(unsigned i) { return N_STATIC_AIRS + 2 * i + 1; }
// the ecc extension likewise per configured curve (openvm.toml `[[app_vm_config.ecc.supported_curves]]`): A_EC(i), A_EC_IO(i)
constexpr unsigned A_EC(unsigned i) { return N_STATIC_AIRS + 2 * MAX_MODULI + 2 * i; }
constexpr unsigned A_EC_IO(unsigned i) { return N_STATIC_AIRS + 2 * MAX_MODULI + 2 * i + 1; }
// the fp2 extension per configured field (openvm.toml `[app_vm_config.fp2] supported_moduli`): A_FP2(i), A_FP2_IO(i)
constexpr unsigned A_FP2(unsigned i) { return N_STATIC_AIRS + 2 * MAX_MODULI + 2 * MAX_CURVES + 2 * i; }
constexpr unsigned A_FP2_IO(unsigned i) { return N_STATIC_AIRS + 2 * MAX_MODULI + 2 * MAX_CURVES + 2 * i + 1; }
constexpr unsigned N_BASE_AIRS = A_KECCAK;   // the extension chips come last: an app has the base chips + the extensions its openvm.toml enables
struct AirShape {
    size_t width = 0, n_pvs = 0, prep_width = 0, cached_width = 0;
    std::vector<uint32_t> program;
};
// the two chips of modulus `index`
inline AirShape build_modmul_air(const modular::Modulus& P, unsigned index, bool adapter) {
    AirShape s;
    const size_t nw = P.limbs / 4, w = adapter ? modmul_io_width(nw) : modular::Cols(P.limbs).VM_WIDTH;
    AirBuilder b(w, 0, 0);
    if (adapter) modmul_io_air(b, index, nw);
    else modular::modmul_vm_air(b, P, BITWISE_BUS, RANGE_TUPLE_BUS, MODMUL_WORD_BUS + index);
    s.width = w, s.program = b.program();
    return s;
}
// the two chips of curve `index`
inline AirShape build_ec_air(const modular::Modulus& P, const modular::Modulus& A, unsigned index, bool adapter) {
    AirShape s;
    const size_t nw = P.limbs / 4, w = adapter ? ec_io_width(nw) : ecc::Cols(P.limbs).VM_WIDTH;
    AirBuilder b(w, 0, 0);
    if (adapter) ec_io_air(b, index, EC_REQ_BUS, EC_WORD_BUS, nw);
    else ecc::ec_vm_air(b, P, A, BITWISE_BUS, RANGE_TUPLE_BUS, EC_WORD_BUS + index);
    s.width = w, s.program = b.program();
    return s;
}
// the two chips of fp2 field `index`
inline AirShape build_fp2_air(const modular::Modulus& P, unsigned index, bool adapter) {
    AirShape s;
    const size_t nw = P.limbs / 4, w = adapter ? ec_io_width(nw) : fp2::Cols(P.limbs).VM_WIDTH;
    AirBuilder b(w, 0, 0);
    if (adapter) ec_io_air(b, index, FP2_REQ_BUS, FP2_WORD_BUS, nw);
    else fp2::fp2_vm_air(b, P, BITWISE_BUS, RANGE_TUPLE_BUS, FP2_WORD_BUS + index);
    s.width = w, s.program = b.program();
    return s;
}
inline AirShape build_air(unsigned id) {
    AirShape s;
    auto make = [&](size_t width, size_t n_pvs, size_t prep, auto&& fill) {
        AirBuilder b(width, n_pvs, prep);
        fill(b);
        s.width = width, s.n_pvs = n_pvs, s.prep_width = prep, s.cached_width = b.cached_width(), s.program = b.program();
    };
    switch (id) {
        case A_PROGRAM: make(1, 0, PROGRAM_FIELDS, program_air); break;
        case A_FRAME: make(FRAME_WIDTH, 0, 0, frame_air); break;
        case A_LS: make(LS_WIDTH, 0, 0, loadstore_air); break;
        case A_ECALL: make(ECALL_WIDTH, 0, 0, ecall_air); break;
        case A_LEAF: make(LEAF_WIDTH, 0, 0, leaf_air); break;
        case A_MERKLE: make(MERKLE_WIDTH, 16, 0, merkle_air); break;
        case A_POSEIDON2: make(air::POSEIDON2_AIR_WIDTH + 1, 0, 0, [](AirBuilder& b) { air::poseidon2_air(b, (int)HASH_BUS, 8); }); break;
        case A_CONNECTOR: make(CONNECTOR_WIDTH, 2, 0, connector_air); break;
        case A_BITWISE: make(2, 0, 3, [](AirBuilder& b) { chips::bitwise_lookup_air(b, BITWISE_BUS); }); break;
        case A_RANGE_TUPLE: make(1, 0, 2, [](AirBuilder& b) { chips::range_tuple_table_air(b, RANGE_TUPLE_BUS); }); break;
        case A_RANGE: make(1, 0, 1, [](AirBuilder& b) { chips::range_table_air(b, RANGE_BUS); }); break;
        case A_KECCAK: make(keccak::VM_WIDTH, 0, 0, [](AirBuilder& b) { keccak::keccak_vm_air(b, KECCAK_LANE_BUS); }); break;
        case A_KECCAK_IO: make(KECCAK_IO_WIDTH, 0, 0, keccak_io_air); break;
        case A_SHA256: make(sha256::VM_WIDTH, 0, sha256::VM_PREP_WIDTH, [](AirBuilder& b) { sha256::compress_vm_air(b, SHA_MSG_BUS, SHA_STATE_BUS); }); break;
        case A_SHA256_IO: make(SHA_IO_WIDTH, 0, 0, sha256_io_air); break;
        case A_INT256: make(int256::VM_WIDTH, 0, 0, [](AirBuilder& b) { int256::alu256_vm_air(b, BITWISE_BUS, INT256_WORD_BUS); }); break;
        case A_INT256_IO: make(INT256_IO_WIDTH, 0, 0, int256_io_air); break;
        case A_MUL256: make(int256::MUL_VM_WIDTH, 0, 0, [](AirBuilder& b) { int256::mul256_vm_air(b, BITWISE_BUS, RANGE_TUPLE_BUS, INT256_WORD_BUS); }); break;
        case A_CMP256: make(int256::CMP_VM_WIDTH, 0, 0, [](AirBuilder& b) { int256::cmp256_vm_air(b, BITWISE_BUS, INT256_WORD_BUS, BRANCH256_BUS); }); break;
        case A_SHIFT256: make(int256::SH_VM_WIDTH, 0, 0, [](AirBuilder& b) { int256::shift256_vm_air(b, BITWISE_BUS, INT256_WORD_BUS); }); break;
        case A_NATIVE_ARITH: make(NATIVE_ARITH_WIDTH, 0, 0, native_arith_air); break;
        case A_NATIVE_EXT: make(NATIVE_EXT_WIDTH, 0, 0, native_ext_air); break;
        case A_CASTF: make(CASTF_WIDTH, 0, 0, castf_vm_air); break;
        default:
            if (id >= A_ALU && id <= A_JALR) {
                const uint32_t cls = id - A_ALU;
                make(core_width(cls), 0, 0, [cls](AirBuilder& b) { core_air(b, cls); });
            } else {
                throw std::invalid_argument("build_air: unknown AIR");
            }
    }
    return s;
}

}  // namespace vmc
}  // namespace zkhip
