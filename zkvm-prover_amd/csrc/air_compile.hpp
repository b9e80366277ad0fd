// air_compile.hpp -- host side of the constraint path: parses the AIR bytecode that crosses the
// ABI (include/zkhip.h, zkhip_air::program; format in DESIGN.md) and lowers the expression DAG
// to the flat slot program the per-row interpreter kernel executes (K5 of SURVEY.md 2.3).
//
// The reference evaluates each chip's SymbolicConstraints DAG on every row of the quotient
// domain; here the DAG is linearised once at keygen: leaves (trace cells,
// public values, constants, selectors) become direct operands, every interior node gets a slot
// chosen by a linear-scan allocator so that the live set stays small enough to sit in LDS
// ([slot][lane], bank-conflict free).
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "babybear.hpp"

namespace zk {

constexpr uint32_t AIR_MAGIC = 0x31414B5Au;
constexpr uint32_t PROOF_MAGIC = 0x31504B5Au;
constexpr uint32_t PROTO_TAG = 0x5A4B4831u;

constexpr uint32_t LOGUP_MAGIC = 0x554C4B5Au;
constexpr uint32_t PREP_MAGIC = 0x50504B5Au;  // section [PREP_MAGIC, prep_width]: the AIR has a preprocessed trace
constexpr uint32_t CACHED_MAGIC = 0x43414B5Au;  // section [CACHED_MAGIC, cached_width]: cached main partition (own commitment)
constexpr unsigned LOGUP_MAX_FIELDS = 32;
constexpr unsigned N_CHAL = 4 * (1 + LOGUP_MAX_FIELDS);  // gamma, beta^1..beta^32 as base coordinates

// A_PERM / A_CHAL / A_EXPOSED are the leaves of the after-challenge (LogUp) phase: a base column of the
// permutation matrix, a coordinate of the interaction challenges, a coordinate of the exposed sum.
// A_PREP is a cell of the preprocessed trace (committed at keygen, opened like the main trace).
enum AirOp : uint32_t { A_VAR, A_PUB, A_CONST, A_FIRST, A_LAST, A_TRANS, A_ADD, A_SUB, A_MUL, A_NEG, A_PERM, A_CHAL, A_EXPOSED, A_PREP };

// one bus interaction: phi = (sign ? -count : count) / (gamma + bus + 1 + sum_i beta^(i+1) * field_i);
// count and the fields are node indices of expressions of the CURRENT row only (main / preprocessed cells with
// rotation 0, public values, constants, + - * neg), like the bus messages of OpenVM chips
struct Interaction {
    uint32_t bus = 0, sign = 0, n_fields = 0;
    uint32_t count = 0;
    uint32_t fields[LOGUP_MAX_FIELDS] = {};
    uint32_t group = 0;  // permutation column group: phi_group = sum of the terms of its interactions
};

struct AirProgram {
    uint32_t n_nodes = 0, n_cons = 0, n_pvs = 0;
    const uint32_t* nodes = nullptr;  // 3 words each
    const uint32_t* cons = nullptr;
    unsigned max_degree = 0;
    // quotient chunks = 2^log_qd: the quotient of degree-d constraints has degree < (d - 1) N (the reference engine's rule;
    // its stored v1 proofs have 1 or 4 chunks per AIR at blow-up 4)
    unsigned log_qd() const {
        unsigned l = 0;
        while ((1u << l) + 1u < (max_degree < 2u ? 2u : max_degree)) l++;
        return l;
    }
    unsigned qd() const { return 1u << log_qd(); }
    std::vector<Interaction> ints;
    size_t prep_width = 0;
    size_t cached_width = 0;  // leading main columns committed in a tree of their own (OpenVM-v1 cached main)
    size_t n_groups() const { return ints.empty() ? 0 : (size_t)ints.back().group + 1; }
    size_t perm_width() const { return ints.empty() ? 0 : 4 * (n_groups() + 1); }
};

// LogUp soundness bound (the trace-height constraint the reference's stark-backend carries in its verifying key):
// multiplicities live in characteristic p, so a bus whose total number of interaction rows reaches p can be balanced
// by ~p copies of a bogus message.  Sum over the AIRs of (interactions on the bus) << log_height must stay below p.
// `progs[a]` / `log_heights[a]` describe the AIRs of one proof.
inline bool logup_bus_counts_bounded(const AirProgram* progs, const unsigned* log_heights, size_t n_airs) {
    std::map<uint32_t, uint64_t> rows_on_bus;
    for (size_t a = 0; a < n_airs; a++)
        for (const auto& it : progs[a].ints) {
            uint64_t& t = rows_on_bus[it.bus];
            t += (uint64_t)1 << log_heights[a];
            if (t >= P) return false;
        }
    return true;
}

inline int parse_air(const uint32_t* w, size_t len, size_t width, AirProgram* p, std::string* err) {
    auto fail = [&](const char* m) {
        if (err) *err = m;
        return -1;
    };
    if (!w || len < 4 || w[0] != AIR_MAGIC) return fail("bad AIR magic");
    p->n_nodes = w[1];
    p->n_cons = w[2];
    p->n_pvs = w[3];
    const size_t base_len = (size_t)4 + 3 * (size_t)p->n_nodes + p->n_cons;
    if (base_len > len) return fail("AIR program length mismatch");
    p->nodes = w + 4;
    p->cons = w + 4 + 3 * (size_t)p->n_nodes;
    p->ints.clear();
    p->prep_width = 0;
    p->cached_width = 0;
    size_t q0 = base_len;
    if (q0 + 2 <= len && w[q0] == PREP_MAGIC) {
        p->prep_width = w[q0 + 1];
        if (p->prep_width == 0 || p->prep_width > (1u << 20)) return fail("bad preprocessed width");
        q0 += 2;
    }
    if (q0 + 2 <= len && w[q0] == CACHED_MAGIC) {
        p->cached_width = w[q0 + 1];
        if (p->cached_width == 0 || p->cached_width >= width) return fail("bad cached main width (a common part must remain)");
        q0 += 2;
    }
    if (q0 != len) {  // trailing interactions section
        size_t q = q0;
        if (q + 2 > len || w[q] != LOGUP_MAGIC) return fail("AIR program length mismatch");
        const uint32_t n_int = w[q + 1];
        q += 2;
        if (n_int == 0 || n_int > 4096) return fail("bad interaction count");
        p->ints.resize(n_int);
        for (uint32_t j = 0; j < n_int; j++) {
            Interaction& it = p->ints[j];
            if (q + 4 > len) return fail("truncated interaction");
            it.bus = w[q], it.sign = w[q + 1], it.count = w[q + 2], it.n_fields = w[q + 3];
            q += 4;
            if (it.sign > 1 || it.bus >= P - 1 || it.n_fields < 1 || it.n_fields > LOGUP_MAX_FIELDS || q + it.n_fields > len)
                return fail("bad interaction");
            if (it.count >= p->n_nodes) return fail("interaction count node out of range");
            for (uint32_t i = 0; i < it.n_fields; i++) {
                it.fields[i] = w[q++];
                if (it.fields[i] >= p->n_nodes) return fail("interaction field node out of range");
            }
            if (q + 1 > len) return fail("truncated interaction");
            it.group = w[q++];
            const uint32_t prev = j ? p->ints[j - 1].group : 0;
            if (it.group != prev && (j == 0 || it.group != prev + 1)) return fail("interaction groups must be numbered in order");
        }
        if (q != len) return fail("AIR program length mismatch");
    }
    const size_t perm_w = p->perm_width();
    std::vector<unsigned> deg(p->n_nodes, 0);
    for (uint32_t i = 0; i < p->n_nodes; i++) {
        uint32_t op = p->nodes[3 * i], a = p->nodes[3 * i + 1], b = p->nodes[3 * i + 2];
        switch (op) {
            case A_VAR:
                if (a >= width || b > 1) return fail("VAR out of range");
                deg[i] = 1;
                break;
            case A_PUB:
                if (a >= p->n_pvs) return fail("PUB out of range");
                break;
            case A_CONST:
                if (a >= P) return fail("CONST not canonical");
                break;
            case A_FIRST:
            case A_LAST:
                deg[i] = 1;
                break;
            case A_TRANS:
                break;
            case A_ADD:
            case A_SUB:
                if (a >= i || b >= i) return fail("operand not yet defined");
                deg[i] = deg[a] > deg[b] ? deg[a] : deg[b];
                break;
            case A_MUL:
                if (a >= i || b >= i) return fail("operand not yet defined");
                deg[i] = deg[a] + deg[b];
                break;
            case A_NEG:
                if (a >= i) return fail("operand not yet defined");
                deg[i] = deg[a];
                break;
            case A_PERM:
                if (a >= perm_w || b > 1) return fail("PERM out of range");
                deg[i] = 1;
                break;
            case A_CHAL:
                if (a >= N_CHAL || perm_w == 0) return fail("CHAL out of range");
                break;
            case A_EXPOSED:
                if (a >= 4 || perm_w == 0) return fail("EXPOSED out of range");
                break;
            case A_PREP:
                if (a >= p->prep_width || b > 1) return fail("PREP out of range");
                deg[i] = 1;
                break;
            default:
                return fail("unknown AIR op");
        }
    }
    if (!p->ints.empty()) {
        // interaction operands must be expressions of the current row
        std::vector<char> mark(p->n_nodes, 0);
        for (const Interaction& it : p->ints) {
            mark[it.count] = 1;
            for (uint32_t i = 0; i < it.n_fields; i++) mark[it.fields[i]] = 1;
        }
        for (uint32_t i = p->n_nodes; i-- > 0;) {
            if (!mark[i]) continue;
            const uint32_t op = p->nodes[3 * i], a = p->nodes[3 * i + 1], b = p->nodes[3 * i + 2];
            switch (op) {
                case A_VAR:
                case A_PREP:
                    if (b != 0) return fail("interaction operand reads the next row");
                    break;
                case A_PUB:
                case A_CONST:
                    break;
                case A_ADD:
                case A_SUB:
                case A_MUL:
                    mark[a] = mark[b] = 1;
                    break;
                case A_NEG:
                    mark[a] = 1;
                    break;
                default:
                    return fail("interaction operand is not an expression of the current row");
            }
        }
    }
    p->max_degree = 0;
    for (uint32_t k = 0; k < p->n_cons; k++) {
        if (p->cons[k] >= p->n_nodes) return fail("constraint index out of range");
        if (deg[p->cons[k]] > p->max_degree) p->max_degree = deg[p->cons[k]];
    }
    return 0;
}

// ---- lowered program -------------------------------------------------------------------------
// instruction = 3 words: w0 = op | (dst << 8), w1 = operand a, w2 = operand b
// operand = kind << 28 | payload; VAR payload = rot << 27 | column
enum QOp : uint32_t { Q_ADD, Q_SUB, Q_MUL, Q_NEG, Q_ASSERT };
// PERM payload like VAR (column of the permutation LDE); CHAL / EXPO index the per-proof challenge block
enum QKind : uint32_t { K_SLOT, K_VAR, K_PUB, K_CONST, K_SEL, K_PERM, K_CHAL, K_EXPO, K_PREP };
// live intermediates of the interpreter form: one LDS word per lane each (256 lanes: 1 KiB per slot).  Up to 64 fit the default 64 KiB
// of dynamic LDS; the limb chips at 48 limbs (ecc: 61) need a few more -- the launch raises the kernel's limit (gfx950: 160 KiB per CU)
constexpr unsigned Q_MAX_SLOTS = 96;

struct CompiledAir {
    std::vector<uint32_t> code;    // 3 words per instruction
    std::vector<uint32_t> consts;  // Montgomery
    unsigned n_slots = 0;
};

// `roots` (optional): compile these nodes instead of the constraints; the Q_ASSERT of roots[k] carries index k
// (used for the per-interaction operand programs of the LogUp phase)
inline int compile_air(const AirProgram& p, CompiledAir* out, std::string* err, const std::vector<uint32_t>* roots = nullptr) {
    const uint32_t n = p.n_nodes;
    const uint32_t n_roots = roots ? (uint32_t)roots->size() : p.n_cons;
    auto root_at = [&](uint32_t k) { return roots ? (*roots)[k] : p.cons[k]; };
    auto is_leaf = [&](uint32_t i) { return p.nodes[3 * i] <= A_TRANS || p.nodes[3 * i] >= A_PERM; };
    // reachability from the constraints
    std::vector<char> reach(n, 0);
    for (uint32_t k = 0; k < n_roots; k++) reach[root_at(k)] = 1;
    for (uint32_t i = n; i-- > 0;) {
        if (!reach[i] || is_leaf(i)) continue;
        uint32_t op = p.nodes[3 * i];
        reach[p.nodes[3 * i + 1]] = 1;
        if (op != A_NEG) reach[p.nodes[3 * i + 2]] = 1;
    }
    // constraints attached to each node
    std::vector<std::vector<uint32_t>> cons_of(n);
    for (uint32_t k = 0; k < n_roots; k++) cons_of[root_at(k)].push_back(k);
    // Evaluation order: a post-order walk from each constraint in turn, so that a subexpression is computed right before its first
    // use (node-index order keeps everything a builder created early -- e.g. the denominators of ALL bus interactions, which the
    // LogUp section builds before the first group constraint -- live until its last use: 516 live values for a 127-interaction chip
    // where this order needs a dozen).  Constraints keep their numbers (an ASSERT carries its constraint index), so the quotient is
    // the same polynomial.  event order: node (if interior) then its ASSERTs; last_use in event numbering.
    std::vector<uint32_t> order;
    order.reserve(n);
    {
        std::vector<char> done(n, 0);
        std::vector<std::pair<uint32_t, unsigned>> stack;
        for (uint32_t k = 0; k < n_roots; k++) {
            if (done[root_at(k)]) continue;
            stack.push_back({root_at(k), 0});
            while (!stack.empty()) {
                auto& top = stack.back();
                const uint32_t i = top.first;
                if (done[i]) {
                    stack.pop_back();
                    continue;
                }
                if (is_leaf(i)) {
                    done[i] = 1, order.push_back(i);
                    stack.pop_back();
                    continue;
                }
                const uint32_t op = p.nodes[3 * i];
                const unsigned n_kids = op == A_NEG ? 1 : 2;
                if (top.second < n_kids) {
                    const uint32_t kid = p.nodes[3 * i + 1 + top.second];
                    top.second++;
                    if (!done[kid]) stack.push_back({kid, 0});
                    continue;
                }
                done[i] = 1, order.push_back(i);
                stack.pop_back();
            }
        }
    }
    std::vector<uint32_t> last_use(n, 0);
    {
        uint32_t ev = 0;
        for (uint32_t i : order) {
            if (!is_leaf(i)) {
                uint32_t op = p.nodes[3 * i];
                last_use[p.nodes[3 * i + 1]] = ev;
                if (op != A_NEG) last_use[p.nodes[3 * i + 2]] = ev;
                ev++;
            }
            for (size_t c = 0; c < cons_of[i].size(); c++) {
                last_use[i] = ev;
                ev++;
            }
        }
    }
    std::map<uint32_t, uint32_t> const_idx;
    auto leaf_operand = [&](uint32_t i) -> uint32_t {
        uint32_t op = p.nodes[3 * i], a = p.nodes[3 * i + 1], b = p.nodes[3 * i + 2];
        switch (op) {
            case A_VAR:
                return (K_VAR << 28) | (b << 27) | a;
            case A_PUB:
                return (K_PUB << 28) | a;
            case A_CONST: {
                auto it = const_idx.find(a);
                if (it == const_idx.end()) {
                    it = const_idx.emplace(a, (uint32_t)out->consts.size()).first;
                    out->consts.push_back(to_monty(a));
                }
                return (K_CONST << 28) | it->second;
            }
            case A_PERM:
                return (K_PERM << 28) | (b << 27) | a;
            case A_CHAL:
                return (K_CHAL << 28) | a;
            case A_EXPOSED:
                return (K_EXPO << 28) | a;
            case A_PREP:
                return (K_PREP << 28) | (b << 27) | a;
            default:
                return (K_SEL << 28) | (op - A_FIRST);
        }
    };
    std::vector<int> slot_of(n, -1);
    std::vector<uint32_t> free_slots;
    unsigned n_slots = 0;
    uint32_t ev = 0;
    auto operand = [&](uint32_t i) -> uint32_t {
        return is_leaf(i) ? leaf_operand(i) : ((K_SLOT << 28) | (uint32_t)slot_of[i]);
    };
    auto release = [&](uint32_t i) {
        if (!is_leaf(i) && slot_of[i] >= 0 && last_use[i] == ev) {
            free_slots.push_back((uint32_t)slot_of[i]);
            slot_of[i] = -2;
        }
    };
    for (uint32_t i : order) {
        if (!is_leaf(i)) {
            uint32_t op = p.nodes[3 * i], a = p.nodes[3 * i + 1], b = p.nodes[3 * i + 2];
            uint32_t wa = operand(a), wb = op != A_NEG ? operand(b) : 0;
            release(a);
            if (op != A_NEG && b != a) release(b);
            uint32_t dst;
            if (!free_slots.empty()) {
                dst = free_slots.back();
                free_slots.pop_back();
            } else {
                dst = n_slots++;
            }
            slot_of[i] = (int)dst;
            uint32_t qop = op == A_ADD ? Q_ADD : op == A_SUB ? Q_SUB : op == A_MUL ? Q_MUL : Q_NEG;
            out->code.push_back(qop | (dst << 8));
            out->code.push_back(wa);
            out->code.push_back(wb);
            ev++;
        }
        for (size_t c = 0; c < cons_of[i].size(); c++) {
            out->code.push_back(Q_ASSERT | (cons_of[i][c] << 8));
            out->code.push_back(operand(i));
            out->code.push_back(0);
            release(i);
            ev++;
        }
        // an interior node nobody uses after its definition (only possible if it is unreachable,
        // which we skipped) cannot occur; a node whose last use is its own ASSERT was released above
    }
    if (n_slots > Q_MAX_SLOTS) {
        if (err) *err = "AIR needs " + std::to_string(n_slots) + " live intermediates (max " + std::to_string(Q_MAX_SLOTS) + ")";
        return -1;
    }
    out->n_slots = n_slots ? n_slots : 1;
    if (out->consts.empty()) out->consts.push_back(0);
    return 0;
}

}  // namespace zk
