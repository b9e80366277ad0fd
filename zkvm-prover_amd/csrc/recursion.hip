// recursion.hip -- the aggregation layer's VERIFIER CIRCUIT (a5 / a6): an AIR set whose satisfying traces
// exist exactly when `zkhip_verify` accepts the child proofs it was built for.
//
// What it replaces in the reference: the leaf / internal verifier programs the SDK proves at every node of the aggregation
// tree (crates/prover/src/prover/mod.rs:47-60 tree arity 4 / 3, :200-282 `commit_child_vk` + VerifyProver; the recursion
// circuits themselves are un-vendored OpenVM crates).  There the verifier is a program for the native VM; the child
// verifying key enters as a cached-trace commitment.  Here the child verifying key (AIR programs, heights, FRI parameters,
// preprocessed commitments) is FIXED when the circuit is built, so the verifier is a static arithmetic circuit and the whole
// wiring is a PREPROCESSED trace -- its commitment is the analogue of the committed child vk:
//
//   * wire bus: every value of the computation is a WIRE (id, v0..v3), an element of F_p[X]/(X^4 - 11) (base values have
//     v1 = v2 = v3 = 0).  The one row that DEFINES a wire sends (id, v) `fanout` times, every row that USES it receives it once;
//     ids and fanouts are preprocessed, values are the main trace.  LogUp balance <=> every use sees the defined value.
//   * gate chip: one row = one extension-field gate over four wire slots a, b, c, d:
//         qM a b + qA a + qB b + qC c + qD d + qK = 0          (q* preprocessed base scalars, qK a preprocessed ext constant)
//     which covers add / sub / mul / mul-add / inverse / division / assertions / booleanity / free inputs, plus a
//     preprocessed flag forcing the row's slots into the base field (proof words that must be base elements).
//     HORNER rows (round 5, second session): slots a = acc, b = a PACKED value (four base cells of an opened row, as the sponge takes
//     them), d = alpha, c = the result; for j = 3 .. 0 with bit j of a preprocessed mask set: t <- t alpha + b[j] -- the three values
//     between the steps sit in three further slots (28 columns in all).  The reduced opening of a query is one such row per four
//     opened cells instead of an input row, three packing rows and four multiply-add rows (Builder::hstep; docs/round5_b.md).
//   * Poseidon2 chip: the 298-column Poseidon2 AIR (zkhip_poseidon2_air_tracegen) whose 16 input and 16 output lanes leave /
//     enter as 4 + 4 wires of 4 lanes each -- digests, sponge states and transcript states never get unpacked.
//   * public-value chip: one row binding wires to the node's public values.
//
// The circuit is written against a small builder (constant folding included: the transcript preamble of a fixed child vk
// costs no rows) by a SYMBOLIC twin of verifier.hip: same transcript, same openings, same quotient identity, same FRI checks,
// in the same order.  `zkhip_recursion_witness` runs the same DAG on concrete proofs (this is the "execution" of the node);
// `zkhip_recursion_tracegen` gathers the wire values into the chips' traces on the device.
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "air_compile.hpp"
#include "poseidon2.hpp"
#include "zkhip_internal.hpp"
#include "../../include/zkhip_air.hpp"

namespace zk {
int poseidon2_air_tracegen(zkhip_ctx* ctx, const uint32_t* d_inputs, size_t n_perms, unsigned log_height, uint32_t* d_trace);

namespace rec {

constexpr uint32_t WIRE_BUS = 77;
// gate chip: slots a b c d (four extension values: 16 columns) + three intermediate values of a Horner row (12 columns, zero on other rows);
// preprocessed: wire ids (4), multiplicities (4), qM qA qB qC qD (5), qK (4), base flag (1), Horner row flag qH and h0..h3 = qH * [coordinate j taken] (5)
constexpr size_t GATE_WIDTH = 28, GATE_PREP = 23, P2W_PREP = 13;

enum GateKind : uint8_t { K_INPUT, K_LIN, K_INV, K_DIV, K_ASSERT, K_HSTEP };
enum SrcKind : uint8_t { S_NONE, S_PROOF_BASE, S_PROOF_EXT, S_PV, S_FLAG, S_HINT_BIT, S_HINT_COORD,
                         S_PREP,     // uniform node: word b of the preprocessed commitment of AIR a of the child's verifying key
                         S_KIND,     // uniform node: 1 = the child is a proof of the leaf circuit, 0 = of the internal circuit
                         S_COMMIT,   // uniform node: word b of the leaf (a = 0) / internal (a = 1) circuit commitment this node states
                         S_AUX,      // deferral node: word a of the auxiliary data handed in with the child (openings of its public values)
                         S_PROOF_PACK };   // four BASE words of the proof as the coordinates of one value: word offsets Circuit::packs[a] (~0: a zero)

struct Src {
    uint8_t kind = S_NONE;
    uint32_t child = 0, a = 0, b = 0;  // PROOF_*: a = word offset; PV, PREP: a = AIR, b = index; HINT_*: a = source wire, b = bit / coordinate
};

struct Gate {
    uint32_t w[4] = {0, 0, 0, 0};
    uint8_t role[4] = {0, 0, 0, 0};  // 0 unused, 1 use, 2 definition
    uint32_t qM = 0, qA = 0, qB = 0, qC = 0, qD = 0;  // Montgomery
    Ext qK = {{0, 0, 0, 0}};
    uint8_t kind = K_LIN, base = 0;
    uint8_t hmask = 0;  // K_HSTEP: bit j = coordinate j of slot b is taken
    Src src[4];  // K_INPUT: where each defined slot's value comes from
};
struct Perm {
    uint32_t in[4], out[4];
};
struct Op {
    uint8_t is_perm;
    uint32_t idx;
};

// a symbolic value: a wire, or a constant that has not been given a wire yet
struct V {
    uint32_t wire = 0;
    bool k = false;
    Ext c = {{0, 0, 0, 0}};
};
struct Lane {  // one base-field lane of a sponge / transcript state: coordinate `coord` of v (coord < 0: v itself, a base value)
    V v;
    int coord = -1;
};

inline V cst(Ext e) {
    V v;
    v.k = true, v.c = e;
    return v;
}
inline V cst_base(uint32_t monty) { return cst(ext_from_base(monty)); }
inline V cst_u(uint32_t canon) { return cst_base(to_monty(canon)); }
inline bool is_zero(const Ext& e) { return (e.c[0] | e.c[1] | e.c[2] | e.c[3]) == 0; }
inline bool is_base(const Ext& e) { return (e.c[1] | e.c[2] | e.c[3]) == 0; }

struct Circuit {
    uint32_t n_wires = 0;
    std::vector<uint32_t> fanout{0};
    std::vector<Gate> gates;
    std::vector<Perm> perms;
    std::vector<Op> order;
    std::vector<uint32_t> pv_wires;  // groups of 4 public values bound to one wire each
    size_t n_pvs = 0;
    std::vector<std::array<uint32_t, 4>> packs;  // S_PROOF_PACK: the proof words behind the coordinates of a packed input
    // order[sections[i] .. sections[i + 1]) verifies child i and touches only that child's wires and constant wires: the
    // sections can be evaluated side by side once the constant rows are done
    std::vector<size_t> sections;
    // inside child i: order[sub[i][q] .. sub[i][q + 1]) checks query q (its openings, reduced openings, folds) and reads only the
    // child's part before the queries and constants: once that part is done the queries run side by side (a hundred per child), then
    // whatever follows them.  `query_parallel` is set by a dependency check over the finished circuit, not assumed.
    std::vector<std::vector<size_t>> sub;
    bool query_parallel = false;
};

struct BuildError {
    std::string msg;
};

class Builder {
  public:
    Circuit c;
    const uint32_t ONE = MONTY_ONE, NEG1 = mneg(MONTY_ONE);

    uint32_t new_wire() {
        c.fanout.push_back(0);
        return ++c.n_wires;
    }
    // the wire of a value as an operand (one more use); constants get their defining row on first use
    uint32_t use(const V& v) {
        uint32_t w = v.wire;
        if (v.k) {
            const std::array<uint32_t, 4> key{v.c.c[0], v.c.c[1], v.c.c[2], v.c.c[3]};
            auto it = consts_.find(key);
            if (it == consts_.end()) {
                Gate g;
                g.kind = K_LIN, g.qC = NEG1, g.qK = v.c;
                g.w[2] = new_wire(), g.role[2] = 2;
                push_gate(g);
                it = consts_.emplace(key, g.w[2]).first;
            }
            w = it->second;
        }
        if (!w) throw BuildError{"use of an undefined value"};
        c.fanout[w]++;
        return w;
    }
    void push_gate(const Gate& g) {
        c.gates.push_back(g);
        c.order.push_back({0, (uint32_t)c.gates.size() - 1});
        phase_gates[phase]++;
    }
    // where the rows go (ZKHIP_RECURSION_TIMING prints the table): the part of the verifier being built when a row / permutation is added
    const char* phase = "other";
    std::map<const char*, size_t> phase_gates, phase_perms;   // (keyed by the literal's address: a row costs a pointer lookup)

    // ---- inputs (four per row) ----
    V input(const Src& s, bool base) {
        // An input row is appended when its first input is declared and filled while it has free slots.  A hint reads a wire
        // that must have its value when the row is evaluated: wire ids grow with creation order, so the row can take the hint
        // only if the wire is older than the row's first wire.
        const int t = base ? 1 : 0;
        const bool hint = s.kind == S_HINT_BIT || s.kind == S_HINT_COORD;
        if (in_gate_[t] == SIZE_MAX || in_fill_[t] == 4 || (hint && s.a >= c.gates[in_gate_[t]].w[0])) {
            Gate g;
            g.kind = K_INPUT, g.base = base ? 1 : 0;
            push_gate(g);
            in_gate_[t] = c.gates.size() - 1, in_fill_[t] = 0;
        }
        Gate& g = c.gates[in_gate_[t]];
        V v;
        v.wire = new_wire();
        g.w[in_fill_[t]] = v.wire, g.role[in_fill_[t]] = 2, g.src[in_fill_[t]] = s;
        in_fill_[t]++;
        return v;
    }
    void close_input_rows() { in_gate_[0] = in_gate_[1] = SIZE_MAX; }  // the next input opens a fresh row (section boundary)
    V in_base(const Src& s) { return input(s, true); }
    V in_ext(const Src& s) { return input(s, false); }
    // four base words of child `child`'s proof as ONE value (offsets[k] = ~0u: coordinate k is zero)
    V in_packed(uint32_t child, const std::array<uint32_t, 4>& offsets) {
        Src s;
        s.kind = S_PROOF_PACK, s.child = child, s.a = (uint32_t)c.packs.size();
        c.packs.push_back(offsets);
        return input(s, false);
    }

    // ---- the general gate: c = qM a b + qA a + qB b + qD d + k ----
    V lin(uint32_t qM, const V& a, const V& b, uint32_t qA, uint32_t qB, const V& d, uint32_t qD, const Ext& k) {
        const bool ua = qM || qA, ub = qM || qB, ud = qD != 0;
        if ((!ua || a.k) && (!ub || b.k) && (!ud || d.k)) {
            Ext r = k;
            if (qM) r = ext_add(r, ext_mul_base(ext_mul(a.c, b.c), qM));
            if (qA) r = ext_add(r, ext_mul_base(a.c, qA));
            if (qB) r = ext_add(r, ext_mul_base(b.c, qB));
            if (qD) r = ext_add(r, ext_mul_base(d.c, qD));
            return cst(r);
        }
        Gate g;
        g.kind = K_LIN, g.qM = qM, g.qA = qA, g.qB = qB, g.qC = NEG1, g.qD = qD, g.qK = k;
        if (ua) g.w[0] = use(a), g.role[0] = 1;
        if (ub) g.w[1] = use(b), g.role[1] = 1;
        if (ud) g.w[3] = use(d), g.role[3] = 1;
        V out;
        out.wire = g.w[2] = new_wire();
        g.role[2] = 2;
        push_gate(g);
        return out;
    }
    V zero() { return cst(ext_zero()); }
    V add(const V& a, const V& b) { return lin(0, a, b, ONE, ONE, V{}, 0, ext_zero()); }
    V sub(const V& a, const V& b) { return lin(0, a, b, ONE, NEG1, V{}, 0, ext_zero()); }
    V scale(const V& a, uint32_t s) { return lin(0, a, V{}, s, 0, V{}, 0, ext_zero()); }
    V add_const(const V& a, const Ext& k) { return lin(0, a, V{}, ONE, 0, V{}, 0, k); }
    V mul(const V& a, const V& b) {
        // a constant BASE factor is a coefficient, not a wire
        if (a.k && is_base(a.c)) return scale(b, a.c.c[0]);
        if (b.k && is_base(b.c)) return scale(a, b.c.c[0]);
        return lin(ONE, a, b, 0, 0, V{}, 0, ext_zero());
    }
    V mul_add(const V& a, const V& b, const V& d) {  // a b + d
        if (a.k && is_base(a.c)) return lin(0, b, d, a.c.c[0], ONE, V{}, 0, ext_zero());
        if (b.k && is_base(b.c)) return lin(0, a, d, b.c.c[0], ONE, V{}, 0, ext_zero());
        if (d.k) return lin(ONE, a, b, 0, 0, V{}, 0, d.c);
        return lin(ONE, a, b, 0, 0, d, ONE, ext_zero());
    }
    V lin3(const V& a, uint32_t qa, const V& b, uint32_t qb, const V& d, uint32_t qd) {
        return lin(0, a, b, qa, qb, d, qd, ext_zero());
    }
    V inv(const V& a) {
        if (a.k) {
            if (is_zero(a.c)) throw BuildError{"inverse of the constant zero"};
            return cst(ext_inv(a.c));
        }
        Gate g;
        g.kind = K_INV, g.qM = ONE, g.qK = ext_from_base(NEG1);
        g.w[0] = use(a), g.role[0] = 1;
        V out;
        out.wire = g.w[1] = new_wire();
        g.role[1] = 2;
        push_gate(g);
        return out;
    }
    V div(const V& n, const V& d) {  // out d = n
        if (d.k) return mul(n, cst(ext_inv(d.c)));
        Gate g;
        g.kind = K_DIV, g.qM = ONE, g.qD = NEG1;
        g.w[1] = use(d), g.role[1] = 1;
        if (n.k) {
            g.qD = 0, g.qK = ext_neg(n.c);
        } else {
            g.w[3] = use(n), g.role[3] = 1;
        }
        V out;
        out.wire = g.w[0] = new_wire();
        g.role[0] = 2;
        push_gate(g);
        return out;
    }
    void assert_zero(const V& a) {
        if (a.k) {
            if (!is_zero(a.c)) throw BuildError{"a constant assertion fails: the child verifying key is inconsistent"};
            return;
        }
        Gate g;
        g.kind = K_ASSERT, g.qA = ONE;
        g.w[0] = use(a), g.role[0] = 1;
        push_gate(g);
    }
    void assert_eq(const V& a, const V& b) {
        if (a.k && b.k) {
            if (!ext_eq(a.c, b.c)) throw BuildError{"a constant assertion fails: the child verifying key is inconsistent"};
            return;
        }
        Gate g;
        g.kind = K_ASSERT;
        if (a.k) {
            g.qB = NEG1, g.qK = a.c, g.w[1] = use(b), g.role[1] = 1;
        } else if (b.k) {
            g.qA = ONE, g.qK = ext_neg(b.c), g.w[0] = use(a), g.role[0] = 1;
        } else {
            g.qA = ONE, g.qB = NEG1, g.w[0] = use(a), g.role[0] = 1, g.w[1] = use(b), g.role[1] = 1;
        }
        push_gate(g);
    }
    void assert_bool(const V& a) {  // a a - a = 0
        if (a.k) {
            if (!is_zero(a.c) && !ext_eq(a.c, ext_one())) throw BuildError{"constant is not a bit"};
            return;
        }
        Gate g;
        g.kind = K_ASSERT, g.qM = ONE, g.qA = NEG1;
        g.w[0] = use(a), g.role[0] = 1, g.w[1] = use(a), g.role[1] = 1;
        push_gate(g);
    }
    void assert_product_zero(const V& a, const V& b) {  // a b = 0
        if (a.k && b.k) {
            if (!is_zero(ext_mul(a.c, b.c))) throw BuildError{"constant product is not zero"};
            return;
        }
        if (a.k || b.k) {
            const V& kk = a.k ? a : b;
            if (is_zero(kk.c)) return;
            assert_zero(a.k ? b : a);
            return;
        }
        Gate g;
        g.kind = K_ASSERT, g.qM = ONE;
        g.w[0] = use(a), g.role[0] = 1, g.w[1] = use(b), g.role[1] = 1;
        push_gate(g);
    }
    V select(const V& bit, const V& x, const V& y) {  // bit ? x : y
        if (bit.k) return is_zero(bit.c) ? y : x;
        return mul_add(bit, sub(x, y), y);
    }
    // (bit ? (y, x) : (x, y)): left = x + bit (y - x), right = x + y - left
    void swap_if(const V& bit, const V& x, const V& y, V* left, V* right) {
        *left = mul_add(bit, sub(y, x), x);
        *right = lin3(x, ONE, y, ONE, *left, NEG1);
    }

    // ---- a HORNER row over the coordinates of a packed value: out = acc, then for j = 3 .. 0 with bit j of `mask` set: out = out alpha + W_j ----
    // (W_j = coordinate j of w, a BASE value.  One row where four rows of mul_add and three of pack stood: the reduced openings of a query
    // walk every opened row once, and the opened rows are handed in packed, four to a value, as the sponge takes them.)
    V hstep(const V& acc, const V& w, unsigned mask, const V& alpha) {
        if (!(mask & 15u)) return acc;
        if (acc.k && w.k && alpha.k) {
            Ext t = acc.c;
            for (int j = 3; j >= 0; j--)
                if (mask >> j & 1u) t = ext_add(ext_mul(t, alpha.c), ext_from_base(w.c.c[j]));
            return cst(t);
        }
        Gate g;
        g.kind = K_HSTEP, g.hmask = (uint8_t)(mask & 15u);
        g.w[0] = use(acc), g.role[0] = 1;
        g.w[1] = use(w), g.role[1] = 1;
        g.w[3] = use(alpha), g.role[3] = 1;
        V out;
        out.wire = g.w[2] = new_wire();
        g.role[2] = 2;
        push_gate(g);
        return out;
    }

    // ---- Poseidon2 on four wires of four lanes ----
    std::array<V, 4> permute(const std::array<V, 4>& in) {
        if (in[0].k && in[1].k && in[2].k && in[3].k) {
            uint32_t s[16];
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 4; k++) s[4 * j + k] = in[j].c.c[k];
            poseidon2_permute_host(s);
            std::array<V, 4> out;
            for (int j = 0; j < 4; j++) out[j] = cst(Ext{{s[4 * j], s[4 * j + 1], s[4 * j + 2], s[4 * j + 3]}});
            return out;
        }
        Perm p;
        for (int j = 0; j < 4; j++) p.in[j] = use(in[j]);
        std::array<V, 4> out;
        for (int j = 0; j < 4; j++) out[j].wire = p.out[j] = new_wire();
        c.perms.push_back(p);
        c.order.push_back({1, (uint32_t)c.perms.size() - 1});
        phase_perms[phase]++;
        return out;
    }

    // ---- lanes <-> wires ----
    const Ext X1 = Ext{{0, MONTY_ONE, 0, 0}};
    V pack(const std::array<V, 4>& b) {  // b0 + b1 X + b2 X^2 + b3 X^3 for BASE values b_i
        if (b[0].k && b[1].k && b[2].k && b[3].k) return cst(Ext{{b[0].c.c[0], b[1].c.c[0], b[2].c.c[0], b[3].c.c[0]}});
        const V x = cst(X1);
        V t = mul_add(b[3], x, b[2]);
        t = mul_add(t, x, b[1]);
        return mul_add(t, x, b[0]);
    }
    std::array<V, 4> unpack(const V& v) {  // the four base coordinates of an extension value
        std::array<V, 4> out;
        if (v.k) {
            for (int k = 0; k < 4; k++) out[k] = cst_base(v.c.c[k]);
            return out;
        }
        auto it = unpacked_.find(v.wire);
        if (it != unpacked_.end()) return it->second;
        for (int k = 0; k < 4; k++) {
            Src s;
            s.kind = S_HINT_COORD, s.a = v.wire, s.b = (uint32_t)k;
            out[k] = in_base(s);
        }
        assert_eq(pack(out), v);
        unpacked_.emplace(v.wire, out);
        return out;
    }
    V lane_base(const Lane& l) {
        if (l.coord < 0) return l.v;
        return unpack(l.v)[l.coord];
    }
    V pack_lanes(const Lane* l) {
        bool same = l[0].coord == 0 && !l[0].v.k;
        for (int k = 1; k < 4 && same; k++) same = l[k].coord == k && !l[k].v.k && l[k].v.wire == l[0].v.wire;
        if (same) return l[0].v;
        std::array<V, 4> b;
        for (int k = 0; k < 4; k++) {
            if (l[k].v.k) b[k] = cst_base(l[k].v.c.c[l[k].coord < 0 ? 0 : l[k].coord]);
            else b[k] = lane_base(l[k]);
        }
        return pack(b);
    }
    static void ext_lanes(const V& v, std::vector<Lane>* out) {
        for (int k = 0; k < 4; k++) out->push_back(Lane{v, k});
    }

    // ---- sponge (p2_hash_slice) and compression (p2_compress) ----
    std::array<V, 2> sponge(const std::vector<Lane>& in) {
        Lane st[16];
        for (auto& l : st) l = Lane{cst(ext_zero()), 0};
        size_t i = 0;
        while (i < in.size()) {
            const size_t n = std::min<size_t>(8, in.size() - i);
            for (size_t k = 0; k < n; k++) st[k] = in[i + k];
            i += n;
            std::array<V, 4> w;
            for (int j = 0; j < 4; j++) w[j] = pack_lanes(st + 4 * j);
            const std::array<V, 4> o = permute(w);
            for (int k = 0; k < 16; k++) st[k] = Lane{o[k / 4], k % 4};
        }
        return {pack_lanes(st), pack_lanes(st + 4)};
    }
    std::array<V, 2> compress(const std::array<V, 2>& l, const std::array<V, 2>& r) {
        const std::array<V, 4> o = permute({l[0], l[1], r[0], r[1]});
        return {o[0], o[1]};
    }

    // bits of the canonical representative of a base value, least significant first; `n_used` low bits are returned
    std::vector<V> canonical_bits(const V& x, unsigned n_used) {
        std::vector<V> bits(31);
        if (x.k) {
            const uint32_t v = from_monty(x.c.c[0]);
            for (unsigned i = 0; i < 31; i++) bits[i] = cst_u((v >> i) & 1);
            bits.resize(n_used);
            return bits;
        }
        for (unsigned i = 0; i < 31; i++) {
            Src s;
            s.kind = S_HINT_BIT, s.a = x.wire, s.b = i;
            bits[i] = in_base(s);
            assert_bool(bits[i]);
        }
        // sum_i 2^i b_i = x, two bits per gate
        V acc = lin3(bits[0], ONE, bits[1], to_monty(2), V{}, 0);
        for (unsigned i = 2; i < 31; i += 2) {
            if (i + 1 < 31) acc = lin3(bits[i], to_monty(1u << i), bits[i + 1], to_monty(1u << (i + 1)), acc, ONE);
            else acc = lin3(bits[i], to_monty(1u << i), V{}, 0, acc, ONE);
        }
        assert_eq(acc, x);
        // canonical: p = 2^31 - 2^27 + 1, so bits 27..30 all set forces bits 0..26 to zero
        V hi = mul(mul(bits[30], bits[29]), mul(bits[28], bits[27]));
        V low = lin3(bits[0], ONE, bits[1], ONE, bits[2], ONE);
        for (unsigned i = 3; i < 27; i += 2) low = lin3(bits[i], ONE, bits[i + 1], ONE, low, ONE);
        assert_product_zero(hi, low);
        bits.resize(n_used);
        return bits;
    }

  private:
    std::map<std::array<uint32_t, 4>, uint32_t> consts_;
    std::map<uint32_t, std::array<V, 4>> unpacked_;
    size_t in_gate_[2] = {SIZE_MAX, SIZE_MAX};
    unsigned in_fill_[2] = {0, 0};
};

// ---- the transcript (verifier.hip HostChallenger), on lanes ----
struct SymChallenger {
    Builder& b;
    Lane state[16];
    std::vector<Lane> in_buf;
    Lane out_buf[8];
    unsigned n_out = 0;
    explicit SymChallenger(Builder& bb) : b(bb) {
        for (auto& l : state) l = Lane{cst(ext_zero()), 0};
    }
    void duplex() {
        for (size_t i = 0; i < in_buf.size(); i++) state[i] = in_buf[i];
        in_buf.clear();
        std::array<V, 4> w;
        for (int j = 0; j < 4; j++) w[j] = b.pack_lanes(state + 4 * j);
        const std::array<V, 4> o = b.permute(w);
        for (int k = 0; k < 16; k++) state[k] = Lane{o[k / 4], k % 4};
        for (int k = 0; k < 8; k++) out_buf[k] = state[k];
        n_out = 8;
    }
    void observe(const Lane& l) {
        n_out = 0;
        in_buf.push_back(l);
        if (in_buf.size() == 8) duplex();
    }
    void observe_base(const V& v) { observe(Lane{v, -1}); }
    void observe_const(uint32_t canon) { observe(Lane{cst_u(canon), -1}); }
    void observe_ext(const V& v) {
        for (int k = 0; k < 4; k++) observe(Lane{v, k});
    }
    Lane sample() {
        if (!in_buf.empty() || n_out == 0) duplex();
        return out_buf[--n_out];
    }
    V sample_ext() {
        Lane l[4];
        for (int i = 0; i < 4; i++) l[i] = sample();
        return b.pack_lanes(l);
    }
    std::vector<V> sample_bits(unsigned bits) { return b.canonical_bits(b.lane_base(sample()), bits); }
    void check_witness(unsigned bits, const V& w) {
        observe_base(w);
        const std::vector<V> bs = sample_bits(bits);
        if (bs.empty()) return;
        V acc = bs[0];
        for (size_t i = 1; i < bs.size(); i += 2) acc = i + 1 < bs.size() ? b.lin3(bs[i], b.ONE, bs[i + 1], b.ONE, acc, b.ONE) : b.add(acc, bs[i]);
        b.assert_zero(acc);
    }
};

// where the chained state of a child lives in its public values
struct StmtSpec {
    std::vector<std::pair<uint32_t, uint32_t>> start, end;  // (AIR, public-value index), K entries each
    bool child_is_node = false;
};

struct ChildVk {
    zkhip_params prm;
    std::vector<std::vector<uint32_t>> programs;
    std::vector<AirProgram> pg;
    std::vector<unsigned> log_heights;
    std::vector<size_t> widths, n_pvs;
    std::vector<std::array<uint32_t, 8>> prep_commit;  // canonical
    std::vector<char> has_prep;
    uint32_t digest[8];  // Montgomery: sponge of the constant part of the transcript preamble
    size_t proof_words = 0;
    uint32_t header[4] = {0, 0, 0, 0};  // the proof's shape words (magic + flags, AIR count, tallest LDE, FRI layers)
};

struct ChildValues {  // what one verified child hands to the statement logic
    std::vector<std::vector<V>> pvs;
};

// recompute the root implied by an opening of a mixed-height commitment (verifier.hip verify_opening) and require it
static void verify_opening_sym(Builder& b, const std::array<V, 2>& root, const std::vector<unsigned>& lhs, const std::vector<size_t>& ws,
                               const std::vector<V>& index_bits, const std::vector<Lane>& cells, const std::vector<std::array<V, 2>>& path) {
    unsigned lh = 0;
    for (unsigned h : lhs) lh = std::max(lh, h);
    std::array<V, 2> cur;
    for (unsigned level = lh;; level--) {
        std::vector<Lane> lanes;
        size_t off = 0;
        bool any = false;
        for (size_t m = 0; m < lhs.size(); m++) {
            if (lhs[m] == level) {
                for (size_t k = 0; k < ws[m]; k++) lanes.push_back(cells[off + k]);
                any = true;
            }
            off += ws[m];
        }
        if (level == lh) {
            cur = b.sponge(lanes);
        } else {
            const unsigned l = lh - level - 1;
            std::array<V, 2> left, right;
            for (int k = 0; k < 2; k++) b.swap_if(index_bits[l], cur[k], path[l][k], &left[k], &right[k]);
            cur = b.compress(left, right);
            if (any) cur = b.compress(cur, b.sponge(lanes));
        }
        if (level == 0) break;
    }
    b.assert_eq(cur[0], root[0]);
    b.assert_eq(cur[1], root[1]);
}

// prod_j (bit_j ? root_j : 1) over base constants root_j
static V bit_product(Builder& b, const std::vector<V>& bits, const std::vector<uint32_t>& roots_monty, const V& start) {
    V acc = start;
    for (size_t j = 0; j < bits.size(); j++) {
        const V f = b.lin(0, bits[j], V{}, msub(roots_monty[j], MONTY_ONE), 0, V{}, 0, ext_one());  // 1 + bit (root - 1)
        acc = b.mul(acc, f);
    }
    return acc;
}

// The symbolic twin of zkhip_verify (verifier.hip:218-576) for child `ci`.
// `prep_sym` (uniform node): the child's preprocessed commitments are VALUES of the circuit (8 base words per AIR, indexed by AIR)
// instead of constants of the child verifying key -- one circuit then verifies proofs of any key with this AIR set and these heights.
static ChildValues verify_child_sym(Builder& b, const ChildVk& vk, uint32_t ci, const std::vector<std::array<V, 8>>* prep_sym = nullptr) {
    const zkhip_params& prm = vk.prm;
    const unsigned bl = prm.log_blowup, lfp = prm.log_final_poly_len;
    const size_t n_airs = vk.pg.size(), n_fin = (size_t)1 << lfp;
    const std::vector<AirProgram>& pg = vk.pg;
    unsigned hmax = 0;
    size_t n_lu = 0, n_prep = 0, n_cached = 0;
    for (size_t a = 0; a < n_airs; a++) {
        hmax = std::max(hmax, vk.log_heights[a] + bl);
        if (!pg[a].ints.empty()) n_lu++;
        if (pg[a].prep_width) n_prep++;
        if (pg[a].cached_width) n_cached++;
    }
    struct CMat {
        unsigned lh, h;
        size_t width;
        unsigned n_pts;
        size_t open_off;
    };
    std::vector<CMat> cm;
    for (size_t a = 0; a < n_airs; a++) cm.push_back({vk.log_heights[a], vk.log_heights[a] + bl, vk.widths[a] - pg[a].cached_width, 2, 0});
    for (size_t a = 0; a < n_airs; a++)
        if (pg[a].cached_width) cm.push_back({vk.log_heights[a], vk.log_heights[a] + bl, pg[a].cached_width, 2, 0});
    for (size_t a = 0; a < n_airs; a++)
        if (pg[a].prep_width) cm.push_back({vk.log_heights[a], vk.log_heights[a] + bl, pg[a].prep_width, 2, 0});
    for (size_t a = 0; a < n_airs; a++)
        if (!pg[a].ints.empty()) cm.push_back({vk.log_heights[a], vk.log_heights[a] + bl, pg[a].perm_width(), 2, 0});
    for (size_t a = 0; a < n_airs; a++)
        for (unsigned j = 0; j < pg[a].qd(); j++) cm.push_back({vk.log_heights[a], vk.log_heights[a] + bl, 4, 1, 0});
    const size_t cm_cached0 = n_airs, cm_prep0 = n_airs + n_cached, cm_perm0 = cm_prep0 + n_prep, cm_quot0 = cm_perm0 + n_lu;
    std::vector<size_t> qoff(n_airs + 1, 0);
    for (size_t a = 0; a < n_airs; a++) qoff[a + 1] = qoff[a] + pg[a].qd();
    size_t n_open = 0;
    for (auto& m : cm) {
        m.open_off = n_open;
        n_open += m.width * m.n_pts;
    }
    const unsigned n_layers = hmax - bl - lfp;

    // ---- the proof's fixed part as inputs (word offsets as in verifier.hip) ----
    b.phase = "proof inputs";
    size_t r = 4;
    auto in_digest = [&](size_t off) -> std::array<V, 2> {
        Src s;
        s.kind = S_PROOF_EXT, s.child = ci;
        std::array<V, 2> d;
        s.a = (uint32_t)off, d[0] = b.in_ext(s);
        s.a = (uint32_t)off + 4, d[1] = b.in_ext(s);
        return d;
    };
    auto in_ext_at = [&](size_t off) {
        Src s;
        s.kind = S_PROOF_EXT, s.child = ci, s.a = (uint32_t)off;
        return b.in_ext(s);
    };
    auto in_base_at = [&](size_t off) {
        Src s;
        s.kind = S_PROOF_BASE, s.child = ci, s.a = (uint32_t)off;
        return b.in_base(s);
    };
    const std::array<V, 2> root_main = in_digest(r);
    r += 8;
    std::vector<std::array<V, 2>> roots_cached;
    for (size_t k = 0; k < n_cached; k++) roots_cached.push_back(in_digest(r)), r += 8;
    std::array<V, 2> root_perm{};
    std::vector<std::array<V, 4>> exposed_coords;  // per AIR with interactions: the 4 base coordinates
    std::vector<V> exposed;
    if (n_lu) {
        root_perm = in_digest(r);
        r += 8;
        for (size_t k = 0; k < n_lu; k++) {
            std::array<V, 4> co;
            for (int q = 0; q < 4; q++) co[q] = in_base_at(r + q);
            exposed_coords.push_back(co);
            exposed.push_back(b.pack(co));
            r += 4;
        }
    }
    const std::array<V, 2> root_quot = in_digest(r);
    r += 8;
    std::vector<V> opened(n_open);
    for (size_t i = 0; i < n_open; i++) opened[i] = in_ext_at(r + 4 * i);
    r += 4 * n_open;
    std::vector<std::array<V, 2>> fri_roots(n_layers);
    std::vector<V> fri_pow(n_layers);
    for (unsigned l = 0; l < n_layers; l++) {
        fri_roots[l] = in_digest(r + 9 * (size_t)l);
        fri_pow[l] = in_base_at(r + 9 * (size_t)l + 8);
    }
    r += 9 * (size_t)n_layers;
    std::vector<V> fin(n_fin);
    for (size_t j = 0; j < n_fin; j++) fin[j] = in_ext_at(r + 4 * j);
    r += 4 * n_fin;
    const V qpow = in_base_at(r++);
    ChildValues out;
    out.pvs.resize(n_airs);
    for (size_t a = 0; a < n_airs; a++)
        for (size_t i = 0; i < vk.n_pvs[a]; i++) {
            Src s;
            s.kind = S_PV, s.child = ci, s.a = (uint32_t)a, s.b = (uint32_t)i;
            out.pvs[a].push_back(b.in_base(s));
        }

    // ---- transcript ----
    b.phase = "transcript";
    SymChallenger ch(b);
    {
        const uint32_t hdr[7] = {PROTO_TAG, (uint32_t)n_airs, prm.log_blowup, prm.log_final_poly_len, prm.num_queries, prm.commit_pow_bits, prm.query_pow_bits};
        for (uint32_t w : hdr) ch.observe_const(w);
        for (size_t a = 0; a < n_airs; a++) {
            std::vector<uint32_t> pm(vk.programs[a].size());
            for (size_t i = 0; i < pm.size(); i++) pm[i] = to_monty(vk.programs[a][i]);
            uint32_t dg[8];
            p2_hash_slice(pm.data(), pm.size(), dg);
            ch.observe_const(vk.log_heights[a]), ch.observe_const((uint32_t)vk.widths[a]), ch.observe_const((uint32_t)vk.n_pvs[a]);
            for (int i = 0; i < 8; i++) ch.observe(Lane{cst_base(dg[i]), -1});
            if (pg[a].prep_width)
                for (int i = 0; i < 8; i++) {
                    if (prep_sym) ch.observe_base((*prep_sym)[a][i]);
                    else ch.observe_const(vk.prep_commit[a][i]);
                }
            for (size_t i = 0; i < vk.n_pvs[a]; i++) ch.observe_base(out.pvs[a][i]);
        }
    }
    for (size_t k = 0; k < n_cached; k++) ch.observe_ext(roots_cached[k][0]), ch.observe_ext(roots_cached[k][1]);
    ch.observe_ext(root_main[0]), ch.observe_ext(root_main[1]);
    std::vector<V> chal(N_CHAL);  // base coordinates, materialised on demand
    std::vector<char> chal_have(N_CHAL, 0);
    std::vector<V> chal_ext;  // gamma, beta^1 ..
    if (n_lu) {
        const V gamma = ch.sample_ext(), beta = ch.sample_ext();
        unsigned max_f = 0;
        for (size_t a = 0; a < n_airs; a++)
            for (uint32_t i = 0; i < pg[a].n_nodes; i++)
                if (pg[a].nodes[3 * i] == A_CHAL) max_f = std::max(max_f, pg[a].nodes[3 * i + 1] / 4);
        chal_ext.push_back(gamma);
        V cur = beta;
        for (unsigned i = 1; i <= max_f; i++) {
            chal_ext.push_back(cur);
            if (i < max_f) cur = b.mul(cur, beta);
        }
        ch.observe_ext(root_perm[0]), ch.observe_ext(root_perm[1]);
        for (size_t k = 0; k < n_lu; k++)
            for (int q = 0; q < 4; q++) ch.observe_base(exposed_coords[k][q]);
        V tot = exposed[0];
        for (size_t k = 1; k < n_lu; k++) tot = b.add(tot, exposed[k]);
        b.assert_zero(tot);
    }
    auto chal_at = [&](uint32_t x) -> V {
        if (!chal_have[x]) {
            if (x / 4 >= chal_ext.size()) throw BuildError{"challenge index out of range"};
            const std::array<V, 4> co = b.unpack(chal_ext[x / 4]);
            for (int q = 0; q < 4; q++) chal[4 * (x / 4) + q] = co[q], chal_have[4 * (x / 4) + q] = 1;
        }
        return chal[x];
    };
    const V alpha = ch.sample_ext();
    ch.observe_ext(root_quot[0]), ch.observe_ext(root_quot[1]);
    const V zeta = ch.sample_ext();
    for (size_t i = 0; i < n_open; i++) ch.observe_ext(opened[i]);
    const V alpha_f = ch.sample_ext();
    const uint32_t gen = to_monty(FIELD_GEN_CANON);

    // ---- constraints at zeta ----
    b.phase = "constraints at zeta";
    size_t k_lu = 0, k_prep = 0, k_cached = 0;
    std::map<unsigned, V> zn_of;  // zeta^(2^lh)
    for (size_t a = 0; a < n_airs; a++) {
        const unsigned lh = vk.log_heights[a], h = lh + bl;
        const size_t W = vk.widths[a], CW = pg[a].cached_width;
        if (!zn_of.count(lh)) {
            V zn = zeta;
            for (unsigned k = 0; k < lh; k++) zn = b.mul(zn, zn);
            zn_of[lh] = zn;
        }
        const V zn = zn_of[lh];
        const V zh = b.add_const(zn, ext_from_base(mneg(MONTY_ONE)));
        const V d1 = b.add_const(zeta, ext_from_base(mneg(MONTY_ONE)));
        const V is_first = b.div(zh, d1);
        const V is_trans = b.add_const(zeta, ext_from_base(mneg(minv(two_adic_generator(lh)))));
        const V is_last = b.div(zh, is_trans);
        const V inv_zh = b.inv(zh);
        std::vector<V> mrow(2 * W);
        if (CW) {
            const V* co = &opened[cm[cm_cached0 + k_cached++].open_off];
            for (size_t k = 0; k < CW; k++) mrow[k] = co[k], mrow[W + k] = co[CW + k];
        }
        for (size_t k = 0; k < W - CW; k++) mrow[CW + k] = opened[cm[a].open_off + k], mrow[W + CW + k] = opened[cm[a].open_off + (W - CW) + k];
        const V *plocal = nullptr, *pnext = nullptr, *qlocal = nullptr, *qnext = nullptr;
        const std::array<V, 4>* expo = nullptr;
        if (pg[a].prep_width) {
            qlocal = &opened[cm[cm_prep0 + k_prep].open_off];
            qnext = qlocal + pg[a].prep_width;
            k_prep++;
        }
        if (!pg[a].ints.empty()) {
            plocal = &opened[cm[cm_perm0 + k_lu].open_off];
            pnext = plocal + pg[a].perm_width();
            expo = &exposed_coords[k_lu];
            k_lu++;
        }
        std::vector<V> vals(pg[a].n_nodes);
        for (uint32_t i = 0; i < pg[a].n_nodes; i++) {
            const uint32_t op = pg[a].nodes[3 * i], x = pg[a].nodes[3 * i + 1], y = pg[a].nodes[3 * i + 2];
            switch (op) {
                case A_VAR: vals[i] = y ? mrow[W + x] : mrow[x]; break;
                case A_PUB: vals[i] = out.pvs[a][x]; break;
                case A_CONST: vals[i] = cst_u(x); break;
                case A_FIRST: vals[i] = is_first; break;
                case A_LAST: vals[i] = is_last; break;
                case A_TRANS: vals[i] = is_trans; break;
                case A_ADD: vals[i] = b.add(vals[x], vals[y]); break;
                case A_SUB: vals[i] = b.sub(vals[x], vals[y]); break;
                case A_MUL: vals[i] = b.mul(vals[x], vals[y]); break;
                case A_NEG: vals[i] = b.scale(vals[x], b.NEG1); break;
                case A_PERM: vals[i] = y ? pnext[x] : plocal[x]; break;
                case A_CHAL: vals[i] = chal_at(x); break;
                case A_PREP: vals[i] = y ? qnext[x] : qlocal[x]; break;
                default: vals[i] = (*expo)[x]; break;
            }
        }
        V acc = cst(ext_zero());
        for (uint32_t k = 0; k < pg[a].n_cons; k++) acc = b.mul_add(acc, alpha, vals[pg[a].cons[k]]);
        const V lhs = b.mul(acc, inv_zh);
        const uint32_t wM = two_adic_generator(h);
        V rhs = cst(ext_zero());
        const unsigned nch = pg[a].qd();
        for (unsigned j = 0; j < nch; j++) {
            const uint32_t sj = mmul(gen, mpow(wM, bitrev32(j, bl)));
            V zps = cst(ext_one());
            for (unsigned k = 0; k < nch; k++) {
                if (k == j) continue;
                const uint32_t sk = mmul(gen, mpow(wM, bitrev32(k, bl)));
                // ((zeta / sk)^N - 1) / ((sj / sk)^N - 1)
                const uint32_t skN_inv = mpow(minv(sk), (uint64_t)1 << lh);
                const uint32_t den = msub(mpow(mmul(sj, minv(sk)), (uint64_t)1 << lh), MONTY_ONE);
                const V t = b.lin(0, zn, V{}, mmul(skN_inv, minv(den)), 0, V{}, 0, ext_from_base(mneg(minv(den))));
                zps = b.mul(zps, t);
            }
            const V* chunk = &opened[cm[cm_quot0 + qoff[a] + j].open_off];
            const V x = cst(b.X1);
            V v = b.mul_add(chunk[3], x, chunk[2]);
            v = b.mul_add(v, x, chunk[1]);
            v = b.mul_add(v, x, chunk[0]);
            rhs = b.mul_add(v, zps, rhs);
        }
        b.assert_eq(lhs, rhs);
    }

    // ---- FRI transcript ----
    b.phase = "transcript";
    std::vector<V> betas(n_layers), betas_sq(n_layers);
    for (unsigned l = 0; l < n_layers; l++) {
        ch.observe_ext(fri_roots[l][0]), ch.observe_ext(fri_roots[l][1]);
        ch.check_witness(prm.commit_pow_bits, fri_pow[l]);
        betas[l] = ch.sample_ext();
    }
    for (size_t j = 0; j < n_fin; j++) ch.observe_ext(fin[j]);
    ch.check_witness(prm.query_pow_bits, qpow);

    struct Batch {
        size_t first, n;
        std::array<V, 2> root;
        std::vector<unsigned> lhs;
        std::vector<size_t> ws;
        size_t tw = 0;
        unsigned bh = 0;
    };
    std::vector<Batch> batches;
    auto add_batch = [&](size_t first, size_t n, const std::array<V, 2>& root) {
        Batch bt;
        bt.first = first, bt.n = n, bt.root = root;
        for (size_t m = first; m < first + n; m++) {
            bt.lhs.push_back(cm[m].h), bt.ws.push_back(cm[m].width);
            bt.tw += cm[m].width;
            bt.bh = std::max(bt.bh, cm[m].h);
        }
        batches.push_back(bt);
    };
    add_batch(0, n_airs, root_main);
    for (size_t k = 0; k < n_cached; k++) add_batch(cm_cached0 + k, 1, roots_cached[k]);
    {
        size_t k = 0;
        for (size_t a = 0; a < n_airs; a++)
            if (pg[a].prep_width) {
                const auto& pc = vk.prep_commit[a];
                std::array<V, 2> root{cst(Ext{{to_monty(pc[0]), to_monty(pc[1]), to_monty(pc[2]), to_monty(pc[3])}}),
                                      cst(Ext{{to_monty(pc[4]), to_monty(pc[5]), to_monty(pc[6]), to_monty(pc[7])}})};
                if (prep_sym) {
                    const std::array<V, 8>& ps = (*prep_sym)[a];
                    root = {b.pack({ps[0], ps[1], ps[2], ps[3]}), b.pack({ps[4], ps[5], ps[6], ps[7]})};
                }
                add_batch(cm_prep0 + k++, 1, root);
            }
    }
    if (n_lu) add_batch(cm_perm0, n_lu, root_perm);
    add_batch(cm_quot0, qoff[n_airs], root_quot);

    b.phase = "reduced openings, per proof";
    // ---- per-proof parts of the reduced openings: alpha_f powers, sum_k alpha_f^k p_k(z) per (matrix, point), alpha_f^offset ----
    size_t max_w = 1;
    for (const auto& m : cm) max_w = std::max(max_w, m.width);
    std::vector<V> apow(max_w + 1);
    apow[0] = cst(ext_one());
    for (size_t k = 1; k <= max_w; k++) apow[k] = k == 1 ? alpha_f : b.mul(apow[k - 1], alpha_f);
    struct RoTerm {
        V ry[2], off_pow[2];
    };
    std::vector<RoTerm> ro_terms(cm.size());
    std::vector<char> has(hmax + 1, 0);
    {
        std::vector<V> run(hmax + 1, cst(ext_one()));
        size_t oi = 0;
        for (const auto& bt : batches)
            for (size_t mi = 0; mi < bt.n; mi++) {
                const size_t m = bt.first + mi;
                const CMat& M = cm[m];
                has[M.h] = 1;
                for (unsigned pt = 0; pt < M.n_pts; pt++) {
                    V ry = cst(ext_zero());
                    for (size_t k = M.width; k-- > 0;) ry = b.mul_add(ry, alpha_f, opened[oi + k]);  // Horner: sum_k alpha_f^k p_k
                    ro_terms[m].ry[pt] = ry;
                    ro_terms[m].off_pow[pt] = run[M.h];
                    run[M.h] = b.mul(run[M.h], apow[M.width]);
                    oi += M.width;
                }
            }
    }
    for (unsigned l = 0; l < n_layers; l++)
        if (has[hmax - l - 1]) betas_sq[l] = b.mul(betas[l], betas[l]);
    // z - x denominators: z per (height, point) is a per-proof value
    std::map<std::pair<unsigned, unsigned>, V> z_of;
    for (const auto& M : cm)
        for (unsigned pt = 0; pt < M.n_pts; pt++)
            if (!z_of.count({M.h, pt})) z_of[{M.h, pt}] = pt == 0 ? zeta : b.scale(zeta, two_adic_generator(M.lh));

    // ---- queries ----
    b.phase = "query indices";
    const uint32_t half = minv(to_monty(2));
    // the query indices first (the challenger is a chain: nothing else touches it from here on), then the queries, each a part of the
    // circuit of its own
    std::vector<std::vector<V>> idx_of(prm.num_queries);
    for (unsigned qn = 0; qn < prm.num_queries; qn++) idx_of[qn] = ch.sample_bits(hmax);  // least significant first
    b.c.sub.emplace_back();
    for (unsigned qn = 0; qn < prm.num_queries; qn++) {
        b.close_input_rows();
        b.c.sub.back().push_back(b.c.order.size());
        const std::vector<V>& idx = idx_of[qn];
        std::vector<std::vector<Lane>> rows_of(batches.size());
        for (size_t bi = 0; bi < batches.size(); bi++) {
            b.phase = "query: opened rows (inputs, sponge, paths)";
            const Batch& bt = batches[bi];
            // The opened row of the batch comes in PACKED, four cells to a value, in the order the level's sponge takes them (per level: its
            // matrices one after the other): the sponge then absorbs the values as they are -- no row that packs four base wires --, and the
            // reduced opening below walks the coordinates of the same values (Builder::hstep).
            rows_of[bi].resize(bt.tw);
            {
                std::vector<size_t> moff(bt.n + 1, 0);
                for (size_t mi = 0; mi < bt.n; mi++) moff[mi + 1] = moff[mi] + bt.ws[mi];
                std::vector<unsigned> levels(bt.lhs);
                std::sort(levels.begin(), levels.end(), std::greater<unsigned>());
                levels.erase(std::unique(levels.begin(), levels.end()), levels.end());
                for (unsigned lv : levels) {
                    std::vector<size_t> seq;
                    for (size_t mi = 0; mi < bt.n; mi++)
                        if (bt.lhs[mi] == lv)
                            for (size_t k = 0; k < bt.ws[mi]; k++) seq.push_back(moff[mi] + k);
                    for (size_t j = 0; j < seq.size(); j += 4) {
                        std::array<uint32_t, 4> offs;
                        for (size_t q = 0; q < 4; q++) offs[q] = j + q < seq.size() ? (uint32_t)(r + seq[j + q]) : ~0u;
                        const V w = b.in_packed(ci, offs);
                        for (size_t q = 0; q < 4 && j + q < seq.size(); q++) rows_of[bi][seq[j + q]] = Lane{w, (int)q};
                    }
                }
            }
            r += bt.tw;
            std::vector<std::array<V, 2>> path(bt.bh);
            for (unsigned l = 0; l < bt.bh; l++) path[l] = in_digest(r + 8 * (size_t)l);
            r += 8 * (size_t)bt.bh;
            const std::vector<V> ibits(idx.begin() + (hmax - bt.bh), idx.end());
            verify_opening_sym(b, bt.root, bt.lhs, bt.ws, ibits, rows_of[bi], path);
        }
        // evaluation points of this query per LDE height, and 1 / (z - x)
        b.phase = "query: reduced openings";
        std::map<unsigned, V> x_of;
        std::map<std::pair<unsigned, unsigned>, V> invd;
        std::vector<V> roq(hmax + 1, cst(ext_zero()));
        for (size_t bi = 0; bi < batches.size(); bi++) {
            size_t off = 0;
            for (size_t mi = 0; mi < batches[bi].n; mi++) {
                const size_t m = batches[bi].first + mi;
                const CMat& M = cm[m];
                const unsigned h = M.h;
                if (!x_of.count(h)) {
                    // x = gen * w_h^bitrev(ih, h), ih = idx >> (hmax - h): bit j of ih contributes w_h^(2^(h-1-j))
                    std::vector<V> bits(idx.begin() + (hmax - h), idx.end());
                    std::vector<uint32_t> roots(h);
                    for (unsigned j = 0; j < h; j++) roots[j] = mpow(two_adic_generator(h), (uint64_t)1 << (h - 1 - j));
                    x_of[h] = bit_product(b, bits, roots, cst_base(gen));
                }
                // sum_k alpha_f^k p_k(x), last cell first: the cells of this matrix that share a packed value take ONE row
                V rrow = cst(ext_zero());
                for (size_t k = M.width; k > 0;) {
                    const Lane& top = rows_of[bi][off + k - 1];
                    unsigned mask = 0;
                    while (k > 0 && rows_of[bi][off + k - 1].v.wire == top.v.wire) mask |= 1u << rows_of[bi][off + k - 1].coord, k--;
                    rrow = b.hstep(rrow, top.v, mask, alpha_f);
                }
                for (unsigned pt = 0; pt < M.n_pts; pt++) {
                    if (!invd.count({h, pt})) invd[{h, pt}] = b.inv(b.sub(z_of[{h, pt}], x_of[h]));
                    const V t = b.mul(b.sub(ro_terms[m].ry[pt], rrow), invd[{h, pt}]);
                    roq[h] = b.mul_add(t, ro_terms[m].off_pow[pt], roq[h]);
                }
                off += M.width;
            }
        }
        b.phase = "query: fold layers";
        V eval = roq[hmax];
        V xi;  // 1 / x of the current layer's pair
        for (unsigned l = 0; l < n_layers; l++) {
            const unsigned log_len = hmax - l;
            const V sib = in_ext_at(r);
            r += 4;
            std::vector<std::array<V, 2>> path(log_len - 1);
            for (unsigned q = 0; q + 1 < log_len; q++) path[q] = in_digest(r + 8 * (size_t)q);
            r += 8 * (size_t)(log_len - 1);
            V lo, hi;
            b.swap_if(idx[l], eval, sib, &lo, &hi);
            const std::vector<V> kbits(idx.begin() + l + 1, idx.end());  // pair index il >> 1
            {
                // leaf = sponge of the 8 words (lo, hi): one permutation of (lo, hi, 0, 0)
                std::vector<Lane> lanes;
                Builder::ext_lanes(lo, &lanes), Builder::ext_lanes(hi, &lanes);
                std::array<V, 2> cur = b.sponge(lanes);
                for (unsigned q = 0; q + 1 < log_len; q++) {
                    std::array<V, 2> left, right;
                    for (int k = 0; k < 2; k++) b.swap_if(kbits[q], cur[k], path[q][k], &left[k], &right[k]);
                    cur = b.compress(left, right);
                }
                b.assert_eq(cur[0], fri_roots[l][0]);
                b.assert_eq(cur[1], fri_roots[l][1]);
            }
            // 1 / x, x = w_{log_len}^bitrev(k, log_len - 1): from the bits for the first layer, then x' = x^2 (1 - 2 k_0)
            if (l == 0) {
                const unsigned L = log_len - 1;
                std::vector<uint32_t> roots(L);
                for (unsigned j = 0; j < L; j++) roots[j] = minv(mpow(two_adic_generator(log_len), (uint64_t)1 << (L - 1 - j)));
                xi = bit_product(b, kbits, roots, cst(ext_one()));
            } else {
                const V sgn = b.lin(0, idx[l], V{}, mneg(to_monty(2)), 0, V{}, 0, ext_one());
                xi = b.mul(b.mul(xi, xi), sgn);
            }
            // fold_row: e0 + (beta - x)(e1 - e0) / (-2x) = e0 + d / 2 - (beta / x) d / 2
            const V d = b.sub(hi, lo);
            const V t = b.mul(b.mul(betas[l], xi), d);
            eval = b.lin3(lo, b.ONE, d, half, t, mneg(half));
            if (has[log_len - 1]) eval = b.mul_add(betas_sq[l], roq[log_len - 1], eval);
        }
        V want = fin[n_fin - 1];
        if (lfp) {
            const unsigned L = bl + lfp;
            std::vector<V> bits(idx.begin() + n_layers, idx.end());
            std::vector<uint32_t> roots(L);
            for (unsigned j = 0; j < L; j++) roots[j] = mpow(two_adic_generator(L), (uint64_t)1 << (L - 1 - j));
            const V xf = bit_product(b, bits, roots, cst(ext_one()));
            for (size_t j = n_fin - 1; j-- > 0;) want = b.mul_add(want, xf, fin[j]);
        }
        b.assert_eq(eval, want);
    }
    b.close_input_rows();
    b.c.sub.back().push_back(b.c.order.size());
    b.phase = "statement";
    if (r != vk.proof_words) throw BuildError{"internal: the circuit's walk over the proof does not end at its last word"};
    return out;
}

}  // namespace rec
}  // namespace zk

using namespace zk;
using namespace zk::rec;

// ---- the node circuit: C ABI object -------------------------------------------------------------------------------------
// What is fixed when the circuit is built (shared by every fork: zkhip_recursion_fork) ...
struct RecCore {
    ChildVk vk;
    std::unique_ptr<ChildVk> vk_b;   // join: the key of slot 1 (the deferral node's)
    const ChildVk& vk_of(size_t slot) const { return mode == 4 && slot == 1 ? *vk_b : vk; }
    StmtSpec spec;
    // 0 = leaf (children: proofs of a fixed key), 1 = node of the level below (per-depth keys), 2 = uniform node (one key),
    // 3 = deferral node (children: root proofs under a fixed aggregation key), 4 = join (a root + a deferral node proof)
    int mode = 0;
    bool uniform = false;  // public values end with [leaf commitment (8) | internal commitment (8)]
    size_t n_leaf_shapes = 1;          // uniform node: how many leaf circuits (one per set of chips a segment may carry) its leaf children come from
    uint32_t region_index = 0;         // deferral node over JOIN proofs: the node of a child's memory tree above its deferral region (0: plain roots)
    bool has_app_id = false;           // leaf: the statement's first 8 words are this constant instead of the child key's digest
    uint32_t app_id[8] = {};           // (Montgomery)
    size_t max_children = 0, n_state = 0;
    Circuit c;
    size_t child_proof_words = 0, n_aux = 0;
    // the three chips
    std::vector<uint32_t> prog[3], prep[3];
    unsigned log_height[3] = {0, 0, 0};
    size_t width[3] = {GATE_WIDTH, ZKHIP_POSEIDON2_AIR_WIDTH, 1}, prep_width[3] = {GATE_PREP, P2W_PREP, 0};
    // device copies of the wiring (ids per slot) for the trace generators, one pair per device, shared by the forks
    std::mutex dev_mu;
    std::map<int, std::pair<uint32_t*, uint32_t*>> dev_ids;
    ~RecCore() {
        for (auto& kv : dev_ids) {
            if (kv.second.first) (void)hipFree(kv.second.first);
            if (kv.second.second) (void)hipFree(kv.second.second);
        }
    }
};

// The wire values of one run: a mapping of its own (never the shared heap), so that page-locking it for the copy to the device
// (hipHostRegister, once, in zkhip_recursion_tracegen) locks exactly these pages and nothing the allocator may hand out or give back
// while they are registered.  (Round 3 registered the storage of a std::vector in place)
struct WireBuf {
    Ext* p = nullptr;
    size_t n = 0, bytes = 0;
    bool registered = false;
    void reset(size_t n_values) {
        release();
        const size_t page = 1 << 21;
        bytes = (n_values * sizeof(Ext) + page - 1) / page * page;
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        p = (Ext*)m, n = n_values;   // (anonymous pages are zero)
    }
    void release() {
        if (!p) return;
        if (registered) (void)hipHostUnregister(p), registered = false;
        munmap(p, bytes);
        p = nullptr, n = 0, bytes = 0;
    }
    size_t size() const { return n; }
    Ext& operator[](size_t i) { return p[i]; }
    const Ext& operator[](size_t i) const { return p[i]; }
    ~WireBuf() { release(); }
    WireBuf() = default;
    WireBuf(const WireBuf&) = delete;
    WireBuf& operator=(const WireBuf&) = delete;
};

// ... and the state of one user of it: the witness of the last zkhip_recursion_witness call and its device buffers
struct zkhip_recursion {
    std::shared_ptr<RecCore> k;
    WireBuf vals;
    std::vector<uint32_t> node_pvs;
    std::string error;
    int dev_ready_device = -1;
    uint32_t *d_wires = nullptr, *d_p2_inputs = nullptr;
};

namespace {

thread_local std::string g_build_error;

int upload(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

// the chips' programs and preprocessed traces; `min_log` pads the gate / Poseidon2 chip (leaf and internal circuits of ONE key share heights)
void build_programs(RecCore& K, const unsigned min_log[2]) {
    using namespace zkhip::air;
    Circuit& c = K.c;
    // gate chip
    {
        AirBuilder b(GATE_WIDTH, 0, GATE_PREP);
        auto slot = [&](int s) { return std::array<Expr, 4>{b.var(4 * s), b.var(4 * s + 1), b.var(4 * s + 2), b.var(4 * s + 3)}; };
        const auto a = slot(0), bb = slot(1), cc = slot(2), d = slot(3);
        const Expr qM = b.prep(8), qA = b.prep(9), qB = b.prep(10), qC = b.prep(11), qD = b.prep(12), qbase = b.prep(17);
        const Expr ab[4] = {a[0] * bb[0] + (a[1] * bb[3] + a[2] * bb[2] + a[3] * bb[1]) * 11,
                            a[0] * bb[1] + a[1] * bb[0] + (a[2] * bb[3] + a[3] * bb[2]) * 11,
                            a[0] * bb[2] + a[1] * bb[1] + a[2] * bb[0] + a[3] * bb[3] * 11,
                            a[0] * bb[3] + a[1] * bb[2] + a[2] * bb[1] + a[3] * bb[0]};
        for (int k = 0; k < 4; k++) b.assert_zero(qM * ab[k] + qA * a[k] + qB * bb[k] + qC * cc[k] + qD * d[k] + b.prep(13 + k));
        for (int s = 0; s < 4; s++)
            for (int k = 1; k < 4; k++) b.assert_zero(qbase * b.var(4 * s + k));
        // the Horner row (K_HSTEP): t0 = a; t_{i+1} = taken(3 - i) ? t_i d + b[3 - i] : t_i; c = t_4 -- the three values between in slots 4 .. 6.
        // h_j = qH * [coordinate j taken] is ONE preprocessed column, so the products stay of degree 3.  (On other rows qH = h_j = 0.)
        {
            const Expr qH = b.prep(18);
            std::array<Expr, 4> t = a;
            for (int i = 0; i < 4; i++) {
                const int j = 3 - i;
                const Expr hj = b.prep(19 + j);
                const std::array<Expr, 4> nx = i < 3 ? slot(4 + i) : cc;
                const Expr td[4] = {t[0] * d[0] + (t[1] * d[3] + t[2] * d[2] + t[3] * d[1]) * 11,
                                    t[0] * d[1] + t[1] * d[0] + (t[2] * d[3] + t[3] * d[2]) * 11,
                                    t[0] * d[2] + t[1] * d[1] + t[2] * d[0] + t[3] * d[3] * 11,
                                    t[0] * d[3] + t[1] * d[2] + t[2] * d[1] + t[3] * d[0]};
                for (int k = 0; k < 4; k++) {
                    Expr step = td[k] - t[k];
                    if (k == 0) step = step + bb[j];
                    b.assert_zero(qH * (nx[k] - t[k]) - hj * step);
                }
                t = nx;
            }
        }
        for (int s = 0; s < 4; s++)
            b.push_interaction(WIRE_BUS, {b.prep(s), b.var(4 * s), b.var(4 * s + 1), b.var(4 * s + 2), b.var(4 * s + 3)}, b.prep(4 + s), Kind::Send);
        K.prog[0] = b.program();
    }
    // Poseidon2 chip with wire interactions
    {
        AirBuilder b(POSEIDON2_AIR_WIDTH, 0, P2W_PREP);
        poseidon2_air(b);
        const size_t o0 = POSEIDON2_AIR_WIDTH - 16;
        for (int j = 0; j < 4; j++)
            b.push_interaction(WIRE_BUS, {b.prep(j), b.var(4 * j), b.var(4 * j + 1), b.var(4 * j + 2), b.var(4 * j + 3)}, b.prep(12), Kind::Receive);
        for (int j = 0; j < 4; j++)
            b.push_interaction(WIRE_BUS, {b.prep(4 + j), b.var(o0 + 4 * j), b.var(o0 + 4 * j + 1), b.var(o0 + 4 * j + 2), b.var(o0 + 4 * j + 3)},
                               b.prep(8 + j), Kind::Send);
        K.prog[1] = b.program();
    }
    // public-value chip: one row, one (constant) wire id per group of four public values
    {
        const size_t groups = c.pv_wires.size();
        AirBuilder b(1, c.n_pvs, groups);
        K.prep_width[2] = groups;
        for (size_t g = 0; g < groups; g++) {
            std::vector<Expr> f{b.prep(g)};
            for (size_t k = 0; k < 4; k++) f.push_back(4 * g + k < c.n_pvs ? b.pub(4 * g + k) : b.constant(0));
            b.push_interaction(WIRE_BUS, f, b.constant(1), Kind::Receive);
        }
        K.prog[2] = b.program();
    }
    // preprocessed traces (canonical, column-major)
    auto ceil_log = [](size_t n) {
        unsigned l = 0;
        while (((size_t)1 << l) < n) l++;
        return l;
    };
    K.log_height[0] = std::max(ceil_log(std::max<size_t>(c.gates.size(), 2)), min_log ? min_log[0] : 0u);
    K.log_height[1] = std::max(ceil_log(std::max<size_t>(c.perms.size(), 2)), min_log ? min_log[1] : 0u);
    K.log_height[2] = 0;
    {
        const size_t N = (size_t)1 << K.log_height[0];
        auto& pr = K.prep[0];
        pr.assign(GATE_PREP * N, 0);
        // (column-major: 18 scattered stores per gate, 6.7 M gates for a 51-chip child -- ranges of rows side by side)
        const size_t n_gates = c.gates.size(), n_th = n_gates >= ((size_t)1 << 18) ? 4 : 1;
        auto fill = [&](size_t g0, size_t g1) {
            for (size_t g = g0; g < g1; g++) {
                const Gate& G = c.gates[g];
                for (int s = 0; s < 4; s++) {
                    pr[(size_t)s * N + g] = G.w[s];
                    pr[(size_t)(4 + s) * N + g] = G.role[s] == 1 ? zk::P - 1 : (G.role[s] == 2 ? c.fanout[G.w[s]] % zk::P : 0);
                }
                pr[8 * N + g] = from_monty(G.qM), pr[9 * N + g] = from_monty(G.qA), pr[10 * N + g] = from_monty(G.qB);
                pr[11 * N + g] = from_monty(G.qC), pr[12 * N + g] = from_monty(G.qD);
                for (int k = 0; k < 4; k++) pr[(size_t)(13 + k) * N + g] = from_monty(G.qK.c[k]);
                pr[17 * N + g] = G.base;
                if (G.kind == K_HSTEP) {
                    pr[18 * N + g] = 1;
                    for (int j = 0; j < 4; j++) pr[(size_t)(19 + j) * N + g] = G.hmask >> j & 1u;
                }
            }
        };
        std::vector<std::thread> th;
        for (size_t t = 1; t < n_th; t++) th.emplace_back(fill, n_gates * t / n_th, n_gates * (t + 1) / n_th);
        fill(0, n_gates / n_th);
        for (auto& t : th) t.join();
    }
    {
        const size_t N = (size_t)1 << K.log_height[1];
        auto& pr = K.prep[1];
        pr.assign(P2W_PREP * N, 0);
        for (size_t i = 0; i < c.perms.size(); i++) {
            for (int j = 0; j < 4; j++) {
                pr[(size_t)j * N + i] = c.perms[i].in[j];
                pr[(size_t)(4 + j) * N + i] = c.perms[i].out[j];
                pr[(size_t)(8 + j) * N + i] = c.fanout[c.perms[i].out[j]] % zk::P;
            }
            pr[12 * N + i] = 1;
        }
    }
    K.prep[2].assign(c.pv_wires.begin(), c.pv_wires.end());
}

}  // namespace

// reads one child verifying key: programs, heights, commitments, the digest of its transcript preamble, the shape of its proofs
static int load_child_vk(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, bool commitments_are_values, ChildVk& vk) {
    if (!prm || !airs || n_airs == 0) return ZKHIP_ERR_INVALID;
    if (prm->log_final_poly_len > ZKHIP_MAX_LOG_FINAL_POLY || prm->log_blowup < 1 || prm->log_blowup > 4 || prm->num_queries == 0 ||
        prm->commit_pow_bits > 30 || prm->query_pow_bits > 30)
        return ZKHIP_ERR_INVALID;
    vk.prm = *prm;
    vk.programs.resize(n_airs), vk.pg.resize(n_airs), vk.prep_commit.resize(n_airs), vk.has_prep.assign(n_airs, 0);
    for (size_t a = 0; a < n_airs; a++) {
        if (!airs[a].program || airs[a].width == 0) return ZKHIP_ERR_INVALID;
        vk.programs[a].assign(airs[a].program, airs[a].program + airs[a].program_len);
        for (uint32_t w : vk.programs[a])
            if (w >= P) return ZKHIP_ERR_INVALID;
        if (parse_air(vk.programs[a].data(), vk.programs[a].size(), airs[a].width, &vk.pg[a], nullptr) != 0) return ZKHIP_ERR_INVALID;
        if (vk.pg[a].n_pvs != airs[a].n_pvs || airs[a].log_height + prm->log_blowup > 27 || airs[a].log_height < prm->log_final_poly_len) return ZKHIP_ERR_INVALID;
        if (vk.pg[a].log_qd() > prm->log_blowup) return ZKHIP_ERR_CONSTRAINT;
        vk.log_heights.push_back(airs[a].log_height), vk.widths.push_back(airs[a].width), vk.n_pvs.push_back(airs[a].n_pvs);
        if (vk.pg[a].prep_width && commitments_are_values) {
            vk.has_prep[a] = 1;   // (the commitments are values of the circuit, not part of the key it is built for)
        } else if (vk.pg[a].prep_width) {
            if (!airs[a].prep_commit) return ZKHIP_ERR_INVALID;
            for (int k = 0; k < 8; k++) {
                if (airs[a].prep_commit[k] >= P) return ZKHIP_ERR_INVALID;
                vk.prep_commit[a][k] = airs[a].prep_commit[k];
            }
            vk.has_prep[a] = 1;
        }
    }
    if (!logup_bus_counts_bounded(vk.pg.data(), vk.log_heights.data(), n_airs)) return ZKHIP_ERR_INVALID;
    // digest of the child verifying key: the constant part of the transcript preamble
    {
        std::vector<uint32_t> pre{PROTO_TAG, (uint32_t)n_airs, prm->log_blowup, prm->log_final_poly_len, prm->num_queries, prm->commit_pow_bits, prm->query_pow_bits};
        for (size_t a = 0; a < n_airs; a++) {
            std::vector<uint32_t> pm(vk.programs[a].size());
            for (size_t i = 0; i < pm.size(); i++) pm[i] = to_monty(vk.programs[a][i]);
            uint32_t dg[8];
            p2_hash_slice(pm.data(), pm.size(), dg);
            pre.push_back(vk.log_heights[a]), pre.push_back((uint32_t)vk.widths[a]), pre.push_back((uint32_t)vk.n_pvs[a]);
            for (int i = 0; i < 8; i++) pre.push_back(from_monty(dg[i]));
            if (vk.has_prep[a] && !commitments_are_values)
                for (int i = 0; i < 8; i++) pre.push_back(vk.prep_commit[a][i]);
        }
        for (auto& w : pre) w = to_monty(w);
        p2_hash_slice(pre.data(), pre.size(), vk.digest);
    }
    std::vector<zkhip_air> za(airs, airs + n_airs);
    zkhip_proof_layout lay;
    if (zkhip_proof_layout_of(prm, za.data(), n_airs, &lay) != ZKHIP_OK) return ZKHIP_ERR_INVALID;
    vk.proof_words = lay.n_words;
    unsigned hmax = 0;
    bool lu = false, prep = false, cached = false;
    for (size_t a = 0; a < n_airs; a++) {
        hmax = std::max(hmax, vk.log_heights[a] + prm->log_blowup);
        lu |= !vk.pg[a].ints.empty(), prep |= vk.pg[a].prep_width != 0, cached |= vk.pg[a].cached_width != 0;
    }
    vk.header[0] = PROOF_MAGIC + (lu ? 1u : 0u) + (prep ? 2u : 0u) + (cached ? 4u : 0u);
    vk.header[1] = (uint32_t)n_airs, vk.header[2] = hmax, vk.header[3] = hmax - prm->log_blowup - prm->log_final_poly_len;
    return ZKHIP_OK;
}

// zkhip_recursion_key_commit of a fixed key, as a constant of the circuit (two packed wires)
static std::array<V, 2> key_commit_const(const ChildVk& vk) {
    std::vector<uint32_t> w;
    for (size_t a = 0; a < vk.pg.size(); a++)
        if (vk.has_prep[a])
            for (int k = 0; k < 8; k++) w.push_back(to_monty(vk.prep_commit[a][k]));
    uint32_t dg[8];
    p2_hash_slice(w.data(), w.size(), dg);
    return {cst(Ext{{dg[0], dg[1], dg[2], dg[3]}}), cst(Ext{{dg[4], dg[5], dg[6], dg[7]}})};
}

// deferral node: auxiliary words per child -- the opening of its public values (16 cells + 27 siblings), and for a JOIN child the opening
// of its deferral region (4096 cells + 19 siblings; include/zkhip_vm_flow.hpp DEFERRAL_REGION_BYTES) and the claim-count flags
constexpr size_t DEFERRAL_PV_AUX = 16 + 27 * 8, DEFERRAL_REGION_CELLS = 4096, DEFERRAL_REGION_SIBS = 19, DEFERRAL_MAX_CLAIMS = 63;

// The statement logic behind the verified children, then the public-value binding and the parallelism check.
static int finish_build(std::unique_ptr<zkhip_recursion>& R, const unsigned* min_log_height, zkhip_recursion** out) {
    RecCore& K = *R->k;
    const size_t max_children = K.max_children;
    const StmtSpec& sp = K.spec;
    const bool timing = getenv("ZKHIP_RECURSION_TIMING") != nullptr;   // stage times of the circuit builder on stderr
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (timing) std::fprintf(stderr, "[recursion build] %-28s %.3f s\n", what, std::chrono::duration<double>(now - tprev).count());
        tprev = now;
    };
    try {
        Builder b;
        const size_t NS = K.n_state;
        std::vector<ChildValues> kids;
        std::vector<std::vector<std::array<V, 8>>> prep_in(max_children);   // uniform node: the children's preprocessed commitments
        for (size_t ci = 0; ci < max_children; ci++) {
            const ChildVk& vk = K.vk_of(ci);
            const size_t n_airs = vk.pg.size();
            b.close_input_rows();
            b.c.sections.push_back(b.c.order.size());
            if (K.mode == 2) {
                prep_in[ci].resize(n_airs);
                for (size_t a = 0; a < n_airs; a++)
                    if (vk.has_prep[a])
                        for (uint32_t k = 0; k < 8; k++) {
                            Src sr;
                            sr.kind = S_PREP, sr.child = (uint32_t)ci, sr.a = (uint32_t)a, sr.b = k;
                            prep_in[ci][a][k] = b.in_base(sr);
                        }
            }
            kids.push_back(verify_child_sym(b, vk, (uint32_t)ci, K.mode == 2 ? &prep_in[ci] : nullptr));
        }
        b.close_input_rows();
        b.c.sections.push_back(b.c.order.size());
        lap("children verified");
        if (timing) {
            std::map<std::string, size_t> rows, perms;   // (one literal may live at several addresses)
            for (const auto& kv : b.phase_gates) rows[kv.first] += kv.second;
            for (const auto& kv : b.phase_perms) perms[kv.first] += kv.second;
            for (const auto& kv : rows) std::fprintf(stderr, "[recursion build]   rows %-48s %10zu\n", kv.first.c_str(), kv.second);
            for (const auto& kv : perms) std::fprintf(stderr, "[recursion build]   perm %-48s %10zu\n", kv.first.c_str(), kv.second);
        }
        const ChildVk& vk = K.vk;
        const size_t n_airs = vk.pg.size();
        std::vector<Lane> pvl;   // the node's public values
        auto pack8 = [&](const V* v) { return std::array<V, 2>{b.pack({v[0], v[1], v[2], v[3]}), b.pack({v[4], v[5], v[6], v[7]})}; };
        if (K.mode == 4) {
            // JOIN (a batch-like proof = the guest's root + the proof of what it deferred): slot 0 is a root under aggregation key A
            // (its internal commitment must be A's own), slot 1 a deferral node whose chain starts at zero.  Public values: the root's
            // statement, then the deferral accumulator -- the verifier opens the guest's claims in the final memory root and hashes them.
            const std::vector<V>& rp = kids[0].pvs[K.vk.pg.size() - 1];
            const std::vector<V>& dp = kids[1].pvs[K.vk_b->pg.size() - 1];
            const std::array<V, 2> ic = key_commit_const(K.vk);
            const std::array<V, 2> got = pack8(&rp[rp.size() - 8]);
            for (int k = 0; k < 2; k++) b.assert_eq(got[k], ic[k]);
            // (slot 1 may also be a FOLD of deferral nodes -- a node over their proofs with the chain as its chained state,
            // [key digest (8) | chain before (8) | chain after (8) | accumulator (8)]: a task with more children than one deferral node takes)
            const size_t o = dp.size() == 32 ? 8 : 0;
            for (int k = 0; k < 8; k++) b.assert_zero(dp[o + k]);
            for (const V& v : rp) pvl.push_back(Lane{v, -1});
            for (int k = 0; k < 8; k++) pvl.push_back(Lane{dp[o + 8 + k], -1});
        } else {
        // presence flags: child 0 is present, present children form a prefix
        std::vector<V> flag(max_children);
        flag[0] = cst(ext_one());
        for (size_t ci = 1; ci < max_children; ci++) {
            Src s;
            s.kind = S_FLAG, s.child = (uint32_t)ci;
            flag[ci] = b.in_base(s);
            b.assert_bool(flag[ci]);
            if (ci > 1) b.assert_product_zero(flag[ci], b.lin(0, flag[ci - 1], V{}, b.NEG1, 0, V{}, 0, ext_one()));
        }
        if (K.mode == 3) {
            // DEFERRAL NODE (crates/prover/src/prover/mod.rs:200-282 VerifyProver; crates/types/circuit/src/lib.rs:137-154 verify_stark):
            // every child is a ROOT proof under the fixed aggregation key of the child app.  Per child the circuit derives the claim a
            // parent guest makes about it -- input commitment (sponge of the root's statement), exe commitment (entry pc, initial memory
            // root), vm commitment (app-vk digest, leaf commitment), the 32 public-value bytes opened in the final memory root (their 16
            // cells and the 27 siblings above the block pair are auxiliary inputs) -- requires exit code 0 and the key's own internal
            // commitment, and chains acc <- compress(acc, chunk) over the claim's five 8-element chunks.
            const size_t pa = n_airs - 1, NSc = 9;
            const std::array<V, 2> ic = key_commit_const(vk);
            std::array<V, 8> acc0;
            for (uint32_t k = 0; k < 8; k++) {
                Src sr;
                sr.kind = S_COMMIT, sr.a = 2, sr.b = k;
                acc0[k] = b.in_base(sr);
            }
            std::array<V, 2> acc = pack8(acc0.data());
            for (size_t ci = 0; ci < max_children; ci++) {
                const std::vector<V>& pv = kids[ci].pvs[pa];
                const V* app = &pv[0];
                const V pc_start = pv[8], pc_end = pv[8 + NSc];
                const V *root0 = &pv[9], *root1 = &pv[9 + NSc], *lc = &pv[16 + 2 * NSc], *icv = &pv[24 + 2 * NSc];
                if (!K.region_index) {   // (a join pins its root's internal commitment itself)
                    const std::array<V, 2> got_ic = pack8(icv);
                    for (int k = 0; k < 2; k++) b.assert_eq(got_ic[k], ic[k]);
                }
                b.assert_zero(pc_end);
                std::vector<V> aux(K.n_aux);
                for (uint32_t i = 0; i < K.n_aux; i++) {
                    Src sr;
                    sr.kind = S_AUX, sr.child = (uint32_t)ci, sr.a = i;
                    aux[i] = b.in_base(sr);
                }
                const V zero = cst(ext_zero());
                const std::array<V, 2> z2{zero, zero};
                const std::array<V, 2> cells0 = pack8(&aux[0]), cells1 = pack8(&aux[8]);
                std::array<V, 2> cur = b.compress(b.compress(cells0, z2), b.compress(cells1, z2));   // the two public-value blocks are siblings
                uint32_t idx = (3u << 26) >> 1;
                for (unsigned l = 27; l > 0; l--, idx >>= 1) {
                    const std::array<V, 2> sib = pack8(&aux[16 + 8 * (27 - l)]);
                    cur = (idx & 1u) ? b.compress(sib, cur) : b.compress(cur, sib);
                }
                const std::array<V, 2> r1 = pack8(root1);
                for (int k = 0; k < 2; k++) b.assert_eq(cur[k], r1[k]);
                if (K.region_index) {
                    // The child is a JOIN proof (a batch under a bundle: a proof whose own guest deferred verification).  Its statement ends
                    // with the chain its deferral node verified; what a verifier of the join does on the host -- open the guest's deferral
                    // region (4096 cells = 512 blocks, one subtree of the memory tree) in the final memory root, read the n claims, chain
                    // them, compare -- happens here: cells, the 20 siblings above the subtree and n one-hot-prefix flags are auxiliary inputs.
                    // (A commitment word is read as the field element lo + 2^16 hi: a word >= p, which the host verifier refuses, is its
                    // residue here.)
                    const V* rc = &aux[DEFERRAL_PV_AUX];
                    const V* rs = rc + DEFERRAL_REGION_CELLS;
                    const V* fl = rs + 8 * DEFERRAL_REGION_SIBS;
                    std::vector<std::array<V, 2>> level(DEFERRAL_REGION_CELLS / 8);
                    for (size_t blk = 0; blk < level.size(); blk++) level[blk] = b.compress(pack8(rc + 8 * blk), z2);
                    while (level.size() > 1) {
                        std::vector<std::array<V, 2>> up(level.size() / 2);
                        for (size_t i = 0; i < up.size(); i++) up[i] = b.compress(level[2 * i], level[2 * i + 1]);
                        level = std::move(up);
                    }
                    std::array<V, 2> node = level[0];
                    uint32_t ridx = K.region_index;
                    for (size_t l = 0; l < DEFERRAL_REGION_SIBS; l++, ridx >>= 1) {
                        const std::array<V, 2> sib = pack8(rs + 8 * l);
                        node = (ridx & 1u) ? b.compress(sib, node) : b.compress(node, sib);
                    }
                    for (int k = 0; k < 2; k++) b.assert_eq(node[k], r1[k]);
                    // word 0 of the region = n <= 63: flags f_0 >= f_1 >= ... with sum n; at least one claim (a join without claims is refused)
                    b.assert_zero(rc[1]);
                    V count = cst(ext_zero());
                    for (size_t k = 0; k < DEFERRAL_MAX_CLAIMS; k++) {
                        b.assert_bool(fl[k]);
                        if (k > 0) b.assert_product_zero(fl[k], b.lin(0, fl[k - 1], V{}, b.NEG1, 0, V{}, 0, ext_one()));
                        count = b.add(count, fl[k]);
                    }
                    b.assert_eq(count, rc[0]);
                    b.assert_eq(fl[0], cst(ext_one()));
                    std::array<V, 2> chain{zero, zero};
                    const uint32_t two16 = to_monty(65536);
                    for (size_t k = 0; k < DEFERRAL_MAX_CLAIMS; k++) {
                        const V* cw = rc + 2 * (32 + 32 * k);   // the claim's 32 words as 64 cells
                        std::array<V, 2> nx = chain;
                        for (int c3 = 0; c3 < 3; c3++) {
                            V e[8];
                            for (int j = 0; j < 8; j++) e[j] = b.lin3(cw[2 * (8 * c3 + j)], b.ONE, cw[2 * (8 * c3 + j) + 1], two16, V{}, 0);
                            nx = b.compress(nx, pack8(e));
                        }
                        for (int h = 0; h < 2; h++) nx = b.compress(nx, pack8(cw + 48 + 8 * h));
                        for (int q = 0; q < 2; q++) chain[q] = b.select(fl[k], nx[q], chain[q]);
                    }
                    const std::array<V, 2> stated = pack8(&pv[50]);
                    for (int k = 0; k < 2; k++) b.assert_eq(chain[k], stated[k]);
                }
                std::vector<Lane> stmt_lanes;
                for (const V& v : pv) stmt_lanes.push_back(Lane{v, -1});
                const std::array<V, 2> input_commit = b.sponge(stmt_lanes);
                const std::array<V, 2> exe_commit = b.compress(pack8(root0), {b.pack({pc_start, zero, zero, zero}), zero});
                const std::array<V, 2> vm_commit = b.compress(pack8(app), pack8(lc));
                std::array<V, 2> nx = acc;
                for (const std::array<V, 2>& chunk : {input_commit, exe_commit, vm_commit, cells0, cells1}) nx = b.compress(nx, chunk);
                for (int k = 0; k < 2; k++) acc[k] = b.select(flag[ci], nx[k], acc[k]);
            }
            for (int k = 0; k < 8; k++) pvl.push_back(Lane{acc0[k], -1});
            Builder::ext_lanes(acc[0], &pvl), Builder::ext_lanes(acc[1], &pvl);
        } else {
        auto pv_of = [&](size_t ci, const std::pair<uint32_t, uint32_t>& at) { return kids[ci].pvs[at.first][at.second]; };
        // app vk digest: the constant digest of the child vk (leaf level) or what the children carry (all equal)
        std::array<V, 8> vkd;
        if (!sp.child_is_node) {
            // (several leaf circuits of one app -- one per set of chips a segment may carry -- state ONE app id: the digest of the full set)
            for (int k = 0; k < 8; k++) vkd[k] = cst_base(K.has_app_id ? K.app_id[k] : vk.digest[k]);
        } else {
            const size_t pa = n_airs - 1;
            for (int k = 0; k < 8; k++) vkd[k] = kids[0].pvs[pa][k];
            for (size_t ci = 1; ci < max_children; ci++)
                for (int k = 0; k < 8; k++) b.assert_product_zero(flag[ci], b.sub(kids[ci].pvs[pa][k], vkd[k]));
        }
        // one key: every child is a proof of the leaf circuit or of the internal circuit -- the digest of ITS preprocessed commitments
        // equals the leaf / internal commitment this node states, and an internal child states the same pair (by induction every
        // node below is under one of the two; the verifier holds the pair)
        std::array<V, 8> leaf_commit, internal_commit;
        for (int k = 0; k < 8; k++) leaf_commit[k] = internal_commit[k] = cst(ext_zero());
        if (K.mode == 2) {
            // the leaf circuits' commitments (one per shape) and the internal circuit's: values of this node; what it STATES as its leaf
            // commitment is the one commitment (one shape) or the sponge of the list
            const size_t S = K.n_leaf_shapes;
            std::vector<std::array<V, 8>> leaf_list(S);
            for (uint32_t j = 0; j < S; j++)
                for (uint32_t k = 0; k < 8; k++) {
                    Src sr;
                    sr.kind = S_COMMIT, sr.a = 0, sr.b = 8 * j + k;
                    leaf_list[j][k] = b.in_base(sr);
                }
            for (uint32_t k = 0; k < 8; k++) {
                Src sr;
                sr.kind = S_COMMIT, sr.a = 1, sr.b = k;
                internal_commit[k] = b.in_base(sr);
            }
            std::vector<std::array<V, 2>> lcs(S);
            for (size_t j = 0; j < S; j++) lcs[j] = {b.pack({leaf_list[j][0], leaf_list[j][1], leaf_list[j][2], leaf_list[j][3]}),
                                                   b.pack({leaf_list[j][4], leaf_list[j][5], leaf_list[j][6], leaf_list[j][7]})};
            if (S == 1) {
                leaf_commit = leaf_list[0];
            } else {
                std::vector<Lane> lanes;
                for (size_t j = 0; j < S; j++)
                    for (int k = 0; k < 8; k++) lanes.push_back(Lane{leaf_list[j][k], -1});
                const std::array<V, 2> h = b.sponge(lanes);
                const std::array<V, 4> h0 = b.unpack(h[0]), h1 = b.unpack(h[1]);
                for (int k = 0; k < 4; k++) leaf_commit[k] = h0[k], leaf_commit[4 + k] = h1[k];
            }
            const std::array<V, 2> ic{b.pack({internal_commit[0], internal_commit[1], internal_commit[2], internal_commit[3]}),
                                      b.pack({internal_commit[4], internal_commit[5], internal_commit[6], internal_commit[7]})};
            const size_t pa = n_airs - 1, o = 16 + 2 * NS;
            for (size_t ci = 0; ci < max_children; ci++) {   // (an absent slot repeats child 0, commitments and kind included)
                Src sr;
                sr.kind = S_KIND, sr.child = (uint32_t)ci, sr.a = 0;
                const V is_leaf = b.in_base(sr);
                b.assert_bool(is_leaf);
                // which leaf circuit: a one-hot selector over the shapes (one shape: the selector is is_leaf itself)
                std::vector<V> sel(S, is_leaf);
                if (S > 1) {
                    V sum = cst(ext_zero());
                    for (uint32_t j = 0; j < S; j++) {
                        sr.a = j + 1;
                        sel[j] = b.in_base(sr);
                        b.assert_bool(sel[j]);
                        sum = b.add(sum, sel[j]);
                    }
                    b.assert_eq(sum, is_leaf);
                }
                std::vector<Lane> lanes;
                for (size_t a = 0; a < n_airs; a++)
                    if (vk.has_prep[a])
                        for (int k = 0; k < 8; k++) lanes.push_back(Lane{prep_in[ci][a][k], -1});
                const std::array<V, 2> d = b.sponge(lanes);
                for (int k = 0; k < 2; k++) {
                    V target = ic[k];   // ic + sum_j sel_j (lc_j - ic)
                    for (size_t j = 0; j < S; j++) target = b.mul_add(sel[j], b.sub(lcs[j][k], ic[k]), target);
                    b.assert_eq(d[k], target);
                }
                const V is_node = b.lin(0, is_leaf, V{}, b.NEG1, 0, V{}, 0, ext_one());
                for (int k = 0; k < 8; k++) {
                    b.assert_product_zero(is_node, b.sub(kids[ci].pvs[pa][o + k], leaf_commit[k]));
                    b.assert_product_zero(is_node, b.sub(kids[ci].pvs[pa][o + 8 + k], internal_commit[k]));
                }
            }
        }
        // chained state
        std::vector<V> start(NS), end(NS);
        for (size_t k = 0; k < NS; k++) start[k] = pv_of(0, sp.start[k]), end[k] = pv_of(0, sp.end[k]);
        for (size_t ci = 1; ci < max_children; ci++)
            for (size_t k = 0; k < NS; k++) {
                b.assert_product_zero(flag[ci], b.sub(pv_of(ci, sp.start[k]), end[k]));
                end[k] = b.select(flag[ci], pv_of(ci, sp.end[k]), end[k]);
            }
        // accumulator over the children's payloads
        std::array<V, 2> acc{cst(ext_zero()), cst(ext_zero())};
        for (size_t ci = 0; ci < max_children; ci++) {
            std::array<V, 2> h;
            if (!sp.child_is_node) {
                std::vector<Lane> lanes;
                for (size_t a = 0; a < n_airs; a++)
                    for (const V& v : kids[ci].pvs[a]) lanes.push_back(Lane{v, -1});
                h = b.sponge(lanes);
            } else {
                const size_t pa = n_airs - 1, o = 8 + 2 * NS;
                Lane l[8];
                for (int k = 0; k < 8; k++) l[k] = Lane{kids[ci].pvs[pa][o + k], -1};
                h = {b.pack_lanes(l), b.pack_lanes(l + 4)};
            }
            const std::array<V, 2> nx = b.compress(acc, h);
            for (int k = 0; k < 2; k++) acc[k] = b.select(flag[ci], nx[k], acc[k]);
        }
        // public values of the node: [vk(8) start(K) end(K) acc(8)] (+ [leaf commitment (8) | internal commitment (8)] under one key: zero
        // at a leaf, whose proofs must have the public-value layout of the internal circuit's), bound four at a time
        for (int k = 0; k < 8; k++) pvl.push_back(Lane{vkd[k], -1});
        for (size_t k = 0; k < NS; k++) pvl.push_back(Lane{start[k], -1});
        for (size_t k = 0; k < NS; k++) pvl.push_back(Lane{end[k], -1});
        Builder::ext_lanes(acc[0], &pvl), Builder::ext_lanes(acc[1], &pvl);
        if (K.uniform) {
            for (int k = 0; k < 8; k++) pvl.push_back(Lane{leaf_commit[k], -1});
            for (int k = 0; k < 8; k++) pvl.push_back(Lane{internal_commit[k], -1});
        }
        }
        }
        b.c.n_pvs = pvl.size();
        while (pvl.size() % 4) pvl.push_back(Lane{cst(ext_zero()), -1});
        for (size_t g = 0; g < pvl.size(); g += 4) b.c.pv_wires.push_back(b.use(b.pack_lanes(&pvl[g])));
        lap("statement");
        // May the queries of a child run side by side?  Every wire a query part reads must be written by a constant row, by the child's
        // part before its queries, or by the same query part (roles: 2 = written here, 1 = read here; a permutation reads in, writes out).
        {
            Circuit& c = b.c;
            const size_t n_children = c.sections.size() - 1;
            bool ok = c.sub.size() == n_children;
            auto is_const_row = [](const Gate& G) { return G.kind == K_LIN && !G.role[0] && !G.role[1] && !G.role[3]; };
            std::vector<uint32_t> part_of_wire(c.n_wires + 1, 0);   // 0 = constant, else 1 + part id
            // part ids: child i: pre = 3 i (n_q + 2) ... keep it simple: id = (i, k) -> i * stride + k, k = 0 pre, 1 .. n_q queries, n_q + 1 post
            size_t stride = 0;
            for (const auto& sb : c.sub) stride = std::max(stride, sb.size() + 2);
            auto part_at = [&](size_t oi) -> uint32_t {
                size_t i = std::upper_bound(c.sections.begin(), c.sections.end(), oi) - c.sections.begin() - 1;
                if (i >= n_children) return (uint32_t)(n_children * stride + 1);   // the statement logic behind the children
                const auto& sb = c.sub[i];
                const size_t k = std::upper_bound(sb.begin(), sb.end(), oi) - sb.begin();   // 0 = before the queries
                return (uint32_t)(i * stride + k + 1);
            };
            for (size_t pass = 0; pass < 2 && ok; pass++) {
                size_t cur_i = 0, cur_k = 0;   // part_at(oi) for rising oi without its two binary searches (11 M rows for a 51-chip child)
                for (size_t oi = 0; oi < c.order.size() && ok; oi++) {
                    const Op& op = c.order[oi];
                    uint32_t part;
                    if (oi < c.sections[0]) {
                        part = part_at(oi);
                    } else {
                        while (cur_i + 1 < c.sections.size() && c.sections[cur_i + 1] <= oi) cur_i++, cur_k = 0;
                        if (cur_i >= n_children) {
                            part = (uint32_t)(n_children * stride + 1);
                        } else {
                            const auto& sb = c.sub[cur_i];
                            while (cur_k < sb.size() && sb[cur_k] <= oi) cur_k++;
                            part = (uint32_t)(cur_i * stride + cur_k + 1);
                        }
                    }
                    const size_t i = (part - 1) / stride, k = (part - 1) % stride;
                    const bool is_query = i < n_children && k >= 1 && k + 1 < c.sub[i].size() + 1 && k <= c.sub[i].size() - 1;
                    auto wr = [&](uint32_t w, bool constant) {
                        if (pass == 0) part_of_wire[w] = constant ? 0 : part;
                    };
                    auto rd = [&](uint32_t w) {
                        if (pass == 0 || !is_query || !w) return;
                        const uint32_t src = part_of_wire[w];
                        if (src != 0 && src != part && src != (uint32_t)(i * stride + 1)) ok = false;   // not constant, not own, not the child's part before the queries
                    };
                    if (op.is_perm) {
                        for (int j = 0; j < 4; j++) rd(c.perms[op.idx].in[j]), wr(c.perms[op.idx].out[j], false);
                    } else {
                        const Gate& G = c.gates[op.idx];
                        for (int sl = 0; sl < 4; sl++) {
                            if (G.role[sl] == 2) wr(G.w[sl], is_const_row(G));
                            else if (G.role[sl] == 1) rd(G.w[sl]);
                            if (G.kind == K_INPUT && G.role[sl] == 2 && (G.src[sl].kind == S_HINT_BIT || G.src[sl].kind == S_HINT_COORD)) rd(G.src[sl].a);
                        }
                    }
                }
            }
            c.query_parallel = ok && process_config().parallel_queries;
        }
        lap("query independence");
        K.c = std::move(b.c);
    } catch (const BuildError& e) {
        g_build_error = e.msg;
        return ZKHIP_ERR_INVALID;
    }
    build_programs(K, min_log_height);
    lap("chip programs");
    *out = R.release();
    return ZKHIP_OK;
}

extern "C" {

int zkhip_recursion_build(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, size_t max_children, const zkhip_recursion_stmt* stmt,
                          zkhip_recursion** out) {
    if (!prm || !airs || !out || n_airs == 0 || max_children == 0 || max_children > 8 || !stmt) return ZKHIP_ERR_INVALID;
    *out = nullptr;
    if (stmt->child_is_node < 0 || stmt->child_is_node > 3) return ZKHIP_ERR_INVALID;
    std::unique_ptr<zkhip_recursion> R(new zkhip_recursion());
    R->k.reset(new RecCore());
    RecCore& K = *R->k;
    K.mode = stmt->child_is_node;
    K.uniform = K.mode == 2 || ((K.mode == 0 || K.mode == 1) && stmt->uniform);
    if (stmt->n_leaf_shapes > 8) return ZKHIP_ERR_INVALID;
    K.n_leaf_shapes = K.mode == 2 ? std::max<size_t>(1, stmt->n_leaf_shapes) : 1;
    if (K.mode == 0 && stmt->app_id) {
        K.has_app_id = true;
        for (int k = 0; k < 8; k++) {
            if (stmt->app_id[k] >= P) return ZKHIP_ERR_INVALID;
            K.app_id[k] = to_monty(stmt->app_id[k]);
        }
    }
    ChildVk& vk = K.vk;
    {
        const int rc = load_child_vk(prm, airs, n_airs, K.mode == 2, vk);
        if (rc != ZKHIP_OK) return rc;
    }
    // statement layout
    StmtSpec& sp = K.spec;
    sp.child_is_node = K.mode == 1 || K.mode == 2;
    if (K.mode == 3) {
        // the children are ROOT proofs of a guest flow under one aggregation key: [app (8) | pc, memory root | pc, memory root | acc (8) | leaf (8) | internal (8)]
        // (region_index != 0: JOIN proofs -- the same 50 words followed by the chain (8) the join's deferral node verified)
        K.region_index = stmt->region_index;
        if (n_airs != 3 || vk.n_pvs[2] != 8 + 9 + 9 + 8 + 16 + (K.region_index ? 8u : 0u) || !vk.has_prep[0] || !vk.has_prep[1] || !vk.has_prep[2]) return ZKHIP_ERR_INVALID;
        if (K.region_index >> (DEFERRAL_REGION_SIBS + 1)) return ZKHIP_ERR_INVALID;   // (a node of the level 19 below the root)
        K.n_state = 8;
        K.n_aux = DEFERRAL_PV_AUX + (K.region_index ? DEFERRAL_REGION_CELLS + 8 * DEFERRAL_REGION_SIBS + DEFERRAL_MAX_CLAIMS : 0);
    } else if (sp.child_is_node) {
        // the child is a node circuit: its last AIR is the public-value chip with [vk(8) start(K) end(K) acc(8)] (+ [leaf commitment (8) |
        // internal commitment (8)] under one key)
        // (child_is_node = 1 with `uniform`: a WRAPPER -- it verifies proofs of ONE fixed node key whose public values have the one-key
        // layout and states them in the same layout, so that a leaf circuit too large for the tree's common heights enters the tree
        // through a circuit of the common size)
        const size_t np = vk.n_pvs[n_airs - 1], fixed = K.uniform ? 32 : 16;
        if (np < fixed || (np - fixed) % 2) return ZKHIP_ERR_INVALID;
        if (K.mode == 2 && n_airs != 3) return ZKHIP_ERR_INVALID;
        K.n_state = (np - fixed) / 2;
        for (size_t k = 0; k < K.n_state; k++) {
            sp.start.push_back({(uint32_t)n_airs - 1, (uint32_t)(8 + k)});
            sp.end.push_back({(uint32_t)n_airs - 1, (uint32_t)(8 + K.n_state + k)});
        }
    } else {
        if (stmt->n_state > 16 || (stmt->n_state && (!stmt->start_air || !stmt->start_idx || !stmt->end_air || !stmt->end_idx))) return ZKHIP_ERR_INVALID;
        K.n_state = stmt->n_state;
        for (size_t k = 0; k < stmt->n_state; k++) {
            if (stmt->start_air[k] >= n_airs || stmt->start_idx[k] >= vk.n_pvs[stmt->start_air[k]] || stmt->end_air[k] >= n_airs ||
                stmt->end_idx[k] >= vk.n_pvs[stmt->end_air[k]])
                return ZKHIP_ERR_INVALID;
            sp.start.push_back({stmt->start_air[k], stmt->start_idx[k]});
            sp.end.push_back({stmt->end_air[k], stmt->end_idx[k]});
        }
    }
    K.max_children = max_children;
    K.child_proof_words = vk.proof_words;
    return finish_build(R, stmt->min_log_height, out);
}

// The JOIN circuit of a guest that defers verification (crates/prover/src/prover/mod.rs:200-282 `enable_deferral`): child 0 = the guest's
// root proof under aggregation key A (params_a / airs_a: the internal circuit's key, commitments included), child 1 = the proof of the
// deferral node (params_b / airs_b) that verified the child proofs the guest makes claims about.  Public values: the root's statement
// followed by the deferral accumulator (8).
int zkhip_recursion_build_join(const zkhip_params* params_a, const zkhip_air* airs_a, size_t n_airs_a, const zkhip_params* params_b,
                               const zkhip_air* airs_b, size_t n_airs_b, zkhip_recursion** out) {
    if (!out) return ZKHIP_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<zkhip_recursion> R(new zkhip_recursion());
    R->k.reset(new RecCore());
    RecCore& K = *R->k;
    K.mode = 4;
    K.vk_b.reset(new ChildVk());
    int rc = load_child_vk(params_a, airs_a, n_airs_a, false, K.vk);
    if (rc == ZKHIP_OK) rc = load_child_vk(params_b, airs_b, n_airs_b, false, *K.vk_b);
    if (rc != ZKHIP_OK) return rc;
    if (n_airs_a != 3 || K.vk.n_pvs[2] < 32 || !K.vk.has_prep[0] || !K.vk.has_prep[1] || !K.vk.has_prep[2] ||
        (K.vk_b->n_pvs[n_airs_b - 1] != 16 && K.vk_b->n_pvs[n_airs_b - 1] != 32))
        return ZKHIP_ERR_INVALID;
    K.max_children = 2;
    K.child_proof_words = K.vk.proof_words;
    const unsigned none[2] = {0, 0};
    return finish_build(R, none, out);
}

void zkhip_recursion_destroy(zkhip_recursion* R) {
    if (!R) return;
    for (uint32_t* p : {R->d_wires, R->d_p2_inputs})
        if (p) (void)hipFree(p);
    delete R;   // (the wire buffer unregisters and unmaps itself; the shared part goes with its last user)
}

// A second user of the same circuit: the wiring, programs and preprocessed traces are shared (read-only), the witness and its device
// buffers are the fork's own -- the levels of an aggregation tree above the leaves all run ONE internal circuit, side by side.
int zkhip_recursion_fork(const zkhip_recursion* R, zkhip_recursion** out) {
    if (!R || !out) return ZKHIP_ERR_INVALID;
    zkhip_recursion* F = new zkhip_recursion();
    F->k = R->k;
    *out = F;
    return ZKHIP_OK;
}

// Lays the preprocessed traces out again at heights of at least 2^log_height[0] (gate chip) / 2^log_height[1] (Poseidon2 chip): the
// leaf and the internal circuit of one aggregation key share one height set.  Not while a fork is in use.
int zkhip_recursion_pad(zkhip_recursion* R, const unsigned log_height[2]) {
    if (!R || !log_height || log_height[0] > 27 || log_height[1] > 27) return ZKHIP_ERR_INVALID;
    if (R->k.use_count() != 1) return ZKHIP_ERR_INVALID;
    RecCore& K = *R->k;
    if (log_height[0] < K.log_height[0] || log_height[1] < K.log_height[1]) return ZKHIP_ERR_INVALID;
    build_programs(K, log_height);
    return ZKHIP_OK;
}

size_t zkhip_recursion_n_airs(const zkhip_recursion* R) { return R ? 3 : 0; }
size_t zkhip_recursion_n_pvs(const zkhip_recursion* R) { return R ? R->k->c.n_pvs : 0; }
size_t zkhip_recursion_n_state(const zkhip_recursion* R) { return R ? R->k->n_state : 0; }
size_t zkhip_recursion_max_children(const zkhip_recursion* R) { return R ? R->k->max_children : 0; }
size_t zkhip_recursion_child_proof_bytes(const zkhip_recursion* R) { return R ? 4 * R->k->child_proof_words : 0; }

int zkhip_recursion_stats(const zkhip_recursion* R, size_t out[4]) {
    if (!R || !out) return ZKHIP_ERR_INVALID;
    const RecCore& K = *R->k;
    out[0] = K.c.n_wires, out[1] = K.c.gates.size(), out[2] = K.c.perms.size(), out[3] = K.c.n_pvs;
    return ZKHIP_OK;
}

// digest of a node key's preprocessed commitments (n_commits x 8 canonical words): what a uniform node states as its leaf / internal
// circuit commitment -- the sponge the circuit computes over its child's commitments
int zkhip_recursion_key_commit(const uint32_t* prep_commits, size_t n_commits, uint32_t out[8]) {
    if (!prep_commits || !out || n_commits == 0 || n_commits > 64) return ZKHIP_ERR_INVALID;
    std::vector<uint32_t> w(8 * n_commits);
    for (size_t i = 0; i < w.size(); i++) {
        if (prep_commits[i] >= P) return ZKHIP_ERR_INVALID;
        w[i] = to_monty(prep_commits[i]);
    }
    uint32_t dg[8];
    p2_hash_slice(w.data(), w.size(), dg);
    for (int k = 0; k < 8; k++) out[k] = from_monty(dg[k]);
    return ZKHIP_OK;
}

// digest of a verifying key's transcript preamble (parameters, AIR programs, heights, preprocessed commitments): what a leaf circuit built
// for this key states as the app's digest -- without building the circuit
int zkhip_recursion_vk_digest(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, uint32_t out[8]) {
    if (!out) return ZKHIP_ERR_INVALID;
    ChildVk vk;
    const int rc = load_child_vk(prm, airs, n_airs, false, vk);
    if (rc != ZKHIP_OK) return rc;
    for (int k = 0; k < 8; k++) out[k] = from_monty(vk.digest[k]);
    return ZKHIP_OK;
}

int zkhip_recursion_air(const zkhip_recursion* R, size_t i, zkhip_air* out) {
    if (!R || !out || i >= 3) return ZKHIP_ERR_INVALID;
    const RecCore& K = *R->k;
    out->program = K.prog[i].data(), out->program_len = K.prog[i].size();
    out->log_height = K.log_height[i], out->width = K.width[i];
    out->n_pvs = i == 2 ? K.c.n_pvs : 0;
    out->prep_trace = K.prep[i].data();
    out->prep_commit = nullptr;
    return ZKHIP_OK;
}

int zkhip_recursion_child_vk_digest(const zkhip_recursion* R, uint32_t out[8]) {
    if (!R || !out) return ZKHIP_ERR_INVALID;
    const RecCore& K = *R->k;
    for (int k = 0; k < 8; k++) out[k] = from_monty(K.vk.digest[k]);
    return ZKHIP_OK;
}

const char* zkhip_recursion_last_error(const zkhip_recursion* R) { return R ? R->error.c_str() : g_build_error.c_str(); }

// Runs the circuit on `n_present` child proofs (1 <= n_present <= max_children; the absent slots are filled with copies of
// child 0, whose verification is repeated but does not enter the statement).  Returns ZKHIP_ERR_VERIFY when an assertion of
// the circuit fails -- the wire values are kept either way, so that a test can show that a proof of the resulting traces
// does not verify.
}  // extern "C"

namespace {
// what a uniform node is told about its children beside their proofs
struct UniformIn {
    const uint32_t* prep_commits;   // [child][AIR with a preprocessed trace][8], canonical
    const int* is_leaf;             // [child]
    const uint32_t *leaf_commit, *internal_commit;
};
// ... and a deferral node about its
struct DeferralIn {
    const uint32_t* aux;         // [child][16 public-value cells | 27 x 8 sibling digests above the block pair, bottom-up], canonical
    const uint32_t* acc_start;   // the chain's value before this node's children
};
int recursion_witness(zkhip_recursion* R, const uint8_t* const* proofs, const size_t* proof_lens, const uint32_t* const* const* child_pvs,
                      size_t n_present, uint32_t* node_pvs_out, const UniformIn* uni, const DeferralIn* def = nullptr) {
    if (!R || !proofs || !proof_lens || n_present == 0 || n_present > R->k->max_children) return ZKHIP_ERR_INVALID;
    const RecCore& K = *R->k;
    if ((K.mode == 2) != (uni != nullptr) || (K.mode == 3) != (def != nullptr)) {
        R->error = K.mode == 2   ? "a uniform node circuit takes its children's commitments and kinds (zkhip_recursion_witness_uniform)"
                   : K.mode == 3 ? "a deferral node circuit takes the openings of its children's public values (zkhip_recursion_witness_deferral)"
                                 : "this circuit is built for one child verifying key (zkhip_recursion_witness)";
        return ZKHIP_ERR_INVALID;
    }
    if (K.mode == 4 && n_present != 2) {
        R->error = "a join takes the root proof and the deferral node's proof";
        return ZKHIP_ERR_INVALID;
    }
    size_t n_prep_airs = 0;
    std::vector<size_t> prep_slot(K.vk.pg.size(), 0);
    for (size_t a = 0; a < K.vk.pg.size(); a++)
        if (K.vk.has_prep[a]) prep_slot[a] = n_prep_airs++;
    if (uni) {
        if (!uni->prep_commits || !uni->is_leaf || !uni->leaf_commit || !uni->internal_commit) return ZKHIP_ERR_INVALID;
        for (size_t i = 0; i < 8 * n_prep_airs * n_present; i++)
            if (uni->prep_commits[i] >= P) return ZKHIP_ERR_INVALID;
        for (size_t k = 0; k < 8 * K.n_leaf_shapes; k++)
            if (uni->leaf_commit[k] >= P) return ZKHIP_ERR_INVALID;
        for (int k = 0; k < 8; k++)
            if (uni->internal_commit[k] >= P) return ZKHIP_ERR_INVALID;
        for (size_t c = 0; c < n_present; c++)
            if (uni->is_leaf[c] < 0 || (size_t)uni->is_leaf[c] > K.n_leaf_shapes) return ZKHIP_ERR_INVALID;
    }
    if (def) {
        if (!def->aux || !def->acc_start) return ZKHIP_ERR_INVALID;
        for (size_t i = 0; i < K.n_aux * n_present; i++)
            if (def->aux[i] >= P) return ZKHIP_ERR_INVALID;
        for (int k = 0; k < 8; k++)
            if (def->acc_start[k] >= P) return ZKHIP_ERR_INVALID;
    }
    const Circuit& c = K.c;
    std::vector<const uint32_t*> pw(K.max_children);
    for (size_t ci = 0; ci < K.max_children; ci++) {
        const size_t src = ci < n_present ? ci : 0;
        const ChildVk& cvk = K.vk_of(ci);
        if (!proofs[src] || proof_lens[src] != 4 * cvk.proof_words) {
            R->error = "child proof " + std::to_string(src) + " has the wrong size for this circuit's child verifying key";
            return ZKHIP_ERR_INVALID;
        }
        pw[ci] = reinterpret_cast<const uint32_t*>(proofs[src]);
        for (size_t a = 0; a < cvk.pg.size(); a++)
            if (cvk.n_pvs[a] && (!child_pvs || !child_pvs[src] || !child_pvs[src][a])) return ZKHIP_ERR_INVALID;
    }
    for (size_t ci = 0; ci < n_present; ci++) {
        const ChildVk& cvk = K.vk_of(ci);
        // the shape words are not part of the transcript: they are fixed by the child verifying key and checked here
        if (memcmp(pw[ci], cvk.header, 16) != 0) {
            R->error = "child proof " + std::to_string(ci) + " does not have the shape of this circuit's child verifying key";
            return ZKHIP_ERR_VERIFY;
        }
        for (size_t i = 0; i < cvk.proof_words; i++)
            if (pw[ci][i] >= P) {
                R->error = "child proof word not canonical";
                return ZKHIP_ERR_VERIFY;
            }
        for (size_t a = 0; a < cvk.pg.size(); a++)
            for (size_t i = 0; i < cvk.n_pvs[a]; i++)
                if (child_pvs[ci][a][i] >= P) return ZKHIP_ERR_INVALID;
    }
    WireBuf& vals = R->vals;
    if (vals.size() != (size_t)c.n_wires + 1) {   // (every wire has exactly one defining row: nothing stale survives a run)
        try {
            vals.reset((size_t)c.n_wires + 1);
        } catch (const std::bad_alloc&) {
            R->error = "no memory for the wire values";
            return ZKHIP_ERR_INVALID;
        }
    }
    auto lin_value = [&](const Gate& G) {
        Ext r = G.qK;
        const Ext &a = vals[G.w[0]], &bq = vals[G.w[1]], &d = vals[G.w[3]];
        if (G.qM) r = ext_add(r, ext_mul_base(ext_mul(a, bq), G.qM));
        if (G.qA) r = ext_add(r, ext_mul_base(a, G.qA));
        if (G.qB) r = ext_add(r, ext_mul_base(bq, G.qB));
        if (G.qD) r = ext_add(r, ext_mul_base(d, G.qD));
        return r;
    };
    auto is_const_row = [](const Gate& G) { return G.kind == K_LIN && !G.role[0] && !G.role[1] && !G.role[3]; };
    // evaluates order[lo, hi); returns the first failing gate or -1
    auto run = [&](size_t lo, size_t hi, bool skip_const) -> long {
        long first_bad = -1;
        for (size_t oi = lo; oi < hi; oi++) {
            const Op& op = c.order[oi];
            if (op.is_perm) {
                const Perm& p = c.perms[op.idx];
                uint32_t s[16];
                for (int j = 0; j < 4; j++)
                    for (int k = 0; k < 4; k++) s[4 * j + k] = vals[p.in[j]].c[k];
                poseidon2_permute_host(s);
                for (int j = 0; j < 4; j++) vals[p.out[j]] = Ext{{s[4 * j], s[4 * j + 1], s[4 * j + 2], s[4 * j + 3]}};
                continue;
            }
            const Gate& G = c.gates[op.idx];
            switch (G.kind) {
                case K_INPUT:
                    for (int s = 0; s < 4; s++) {
                        if (G.role[s] != 2) continue;
                        const Src& sr = G.src[s];
                        Ext v = ext_zero();
                        switch (sr.kind) {
                            case S_PROOF_BASE: v.c[0] = to_monty(pw[sr.child][sr.a]); break;
                            case S_PROOF_EXT:
                                for (int k = 0; k < 4; k++) v.c[k] = to_monty(pw[sr.child][sr.a + k]);
                                break;
                            case S_PROOF_PACK: {
                                const std::array<uint32_t, 4>& o = c.packs[sr.a];
                                for (int k = 0; k < 4; k++) v.c[k] = o[k] == ~0u ? 0u : to_monty(pw[sr.child][o[k]]);
                                break;
                            }
                            case S_PV: v.c[0] = to_monty(child_pvs[sr.child < n_present ? sr.child : 0][sr.a][sr.b]); break;
                            case S_FLAG: v.c[0] = sr.child < n_present ? MONTY_ONE : 0; break;
                            case S_PREP: v.c[0] = to_monty(uni->prep_commits[((sr.child < n_present ? sr.child : 0) * n_prep_airs + prep_slot[sr.a]) * 8 + sr.b]); break;
                            case S_KIND: {   // a = 0: is the child a leaf proof; a = j + 1: is it a proof of leaf circuit j
                                const int kind = uni->is_leaf[sr.child < n_present ? sr.child : 0];
                                v.c[0] = (sr.a == 0 ? kind != 0 : kind == (int)sr.a) ? MONTY_ONE : 0;
                                break;
                            }
                            case S_COMMIT: v.c[0] = to_monty((sr.a == 2 ? def->acc_start : sr.a ? uni->internal_commit : uni->leaf_commit)[sr.b]); break;
                            case S_AUX: v.c[0] = to_monty(def->aux[(sr.child < n_present ? sr.child : 0) * K.n_aux + sr.a]); break;
                            case S_HINT_BIT: v.c[0] = ((from_monty(vals[sr.a].c[0]) >> sr.b) & 1) ? MONTY_ONE : 0; break;
                            case S_HINT_COORD: v.c[0] = vals[sr.a].c[sr.b]; break;
                            default: break;
                        }
                        vals[G.w[s]] = v;
                    }
                    break;
                case K_LIN:
                    if (skip_const && is_const_row(G)) break;
                    vals[G.w[2]] = lin_value(G);
                    break;
                case K_HSTEP: {
                    Ext t = vals[G.w[0]];
                    const Ext &wv = vals[G.w[1]], &al = vals[G.w[3]];
                    for (int j = 3; j >= 0; j--)
                        if (G.hmask >> j & 1u) {
                            t = ext_mul(t, al);
                            t.c[0] = madd(t.c[0], wv.c[j]);
                        }
                    vals[G.w[2]] = t;
                    break;
                }
                case K_INV: {
                    const Ext& a = vals[G.w[0]];
                    if (is_zero(a)) {
                        if (first_bad < 0) first_bad = (long)op.idx;
                        vals[G.w[1]] = ext_zero();
                    } else {
                        vals[G.w[1]] = ext_inv(a);
                    }
                    break;
                }
                case K_DIV: {
                    const Ext& d = vals[G.w[1]];
                    const Ext n = G.qD ? vals[G.w[3]] : ext_neg(G.qK);
                    if (is_zero(d)) {
                        if (first_bad < 0 && !is_zero(n)) first_bad = (long)op.idx;
                        vals[G.w[0]] = ext_zero();
                    } else {
                        vals[G.w[0]] = ext_mul(n, ext_inv(d));
                    }
                    break;
                }
                default:
                    if (!is_zero(lin_value(G)) && first_bad < 0) first_bad = (long)op.idx;
                    break;
            }
        }
        return first_bad;
    };
    // SIXTEEN query parts side by side: each part is advanced to its next permutation (gates are evaluated on the way), the sixteen
    // pending states are permuted in ONE call -- a state word of all sixteen per 512-bit register (poseidon2_permute16_host) -- and
    // scattered back.  The parts are independent (query_parallel) and built by the same code, so they stay in step; one that ends
    // early rides along with a zero state.  Most of a node's ~170 k permutations are in the query parts.
    auto run16 = [&](const size_t (*ranges)[2], size_t n_parts, long* bad_out) {
        size_t pos[16];
        for (size_t k = 0; k < 16; k++) pos[k] = k < n_parts ? ranges[k][0] : 0;
        alignas(64) uint32_t t[256];
        const Perm* pend[16];
        for (;;) {
            size_t n_pending = 0;
            for (size_t k = 0; k < 16; k++) {
                pend[k] = nullptr;
                if (k >= n_parts) continue;
                // gates up to the part's next permutation
                size_t q = pos[k];
                const size_t hi = ranges[k][1];
                while (q < hi && !c.order[q].is_perm) q++;
                if (q > pos[k]) {
                    const long b = run(pos[k], q, true);
                    if (b >= 0 && (bad_out[k] < 0 || b < bad_out[k])) bad_out[k] = b;
                }
                pos[k] = q;
                if (q < hi) pend[k] = &c.perms[c.order[q].idx], n_pending++;
            }
            if (!n_pending) return;
            for (size_t k = 0; k < 16; k++)
                for (int j = 0; j < 4; j++)
                    for (int w = 0; w < 4; w++) t[16 * (4 * j + w) + k] = pend[k] ? vals[pend[k]->in[j]].c[w] : 0u;
            poseidon2_permute16_host(t);
            for (size_t k = 0; k < 16; k++) {
                if (!pend[k]) continue;
                for (int j = 0; j < 4; j++) vals[pend[k]->out[j]] = Ext{{t[16 * (4 * j) + k], t[16 * (4 * j + 1) + k], t[16 * (4 * j + 2) + k], t[16 * (4 * j + 3) + k]}};
                pos[k]++;
            }
        }
    };
    // constant rows first (a constant's row sits in whichever section used it first), then one thread per child section,
    // then the statement logic behind them
    for (const Gate& G : c.gates)
        if (is_const_row(G)) vals[G.w[2]] = G.qK;
    long first_bad = -1;
    {
        const size_t n_sec = c.sections.size() - 1;
        std::vector<long> bad(n_sec, -1);
        auto keep_first = [](long& acc, long b) {
            if (b >= 0 && (acc < 0 || b < acc)) acc = b;
        };
        // (a thread the system refuses -- a process at its thread limit -- is not fatal: the caller does that share itself)
        auto spawn = [](std::vector<std::thread>& th, auto&& fn) {
            try {
                th.emplace_back(fn);
                return true;
            } catch (const std::system_error&) {
                return false;
            }
        };
        if (!c.query_parallel) {
            std::vector<std::thread> th;
            for (size_t i = 1; i < n_sec; i++) {
                auto part = [&, i]() { bad[i] = run(c.sections[i], c.sections[i + 1], true); };
                if (!spawn(th, part)) part();
            }
            bad[0] = run(c.sections[0], c.sections[1], true);
            for (auto& t : th) t.join();
        } else {
            // the children's parts before their queries side by side, then every (child, query) part from a shared counter, then what
            // follows the queries
            auto each_child = [&](auto&& f) {
                std::vector<std::thread> th;
                for (size_t i = 1; i < n_sec; i++) {
                    auto part = [&, i]() { f(i); };
                    if (!spawn(th, part)) part();
                }
                f(0);
                for (auto& t : th) t.join();
            };
            each_child([&](size_t i) { keep_first(bad[i], run(c.sections[i], c.sub[i].front(), true)); });
            std::vector<std::pair<size_t, size_t>> tasks;
            for (size_t i = 0; i < n_sec; i++)
                for (size_t q = 0; q + 1 < c.sub[i].size(); q++) tasks.push_back({i, q});
            // (all cores the process may use -- zkhip_host_cpus: the cgroup's quota, not the 256 host threads a container sees;
            // measured in the one-flow pipeline with 4 / 8 / 16 threads on 16 cores -- the segment phase loses ~40 ms to the
            // contention, the tree's tail gains ~100 ms; ZKHIP_WITNESS_THREADS overrides)
            const unsigned cap = process_config().witness_threads ? process_config().witness_threads : zkhip_host_cpus();
            const size_t n_threads = std::min<size_t>((tasks.size() + 15) / 16, std::max(1u, cap));
            std::atomic<size_t> next{0};
            std::vector<std::vector<long>> tb(n_threads, std::vector<long>(n_sec, -1));
            std::vector<std::thread> th;
            auto worker = [&](size_t t) {
                // bundles of sixteen query parts from the shared counter, advanced side by side
                for (size_t k0; (k0 = next.fetch_add(16)) < tasks.size();) {
                    const size_t n = std::min<size_t>(16, tasks.size() - k0);
                    size_t ranges[16][2];
                    long bad16[16];
                    for (size_t k = 0; k < n; k++) ranges[k][0] = c.sub[tasks[k0 + k].first][tasks[k0 + k].second], ranges[k][1] = c.sub[tasks[k0 + k].first][tasks[k0 + k].second + 1], bad16[k] = -1;
                    run16(ranges, n, bad16);
                    for (size_t k = 0; k < n; k++) keep_first(tb[t][tasks[k0 + k].first], bad16[k]);
                }
            };
            for (size_t t = 1; t < n_threads; t++)
                if (!spawn(th, [&, t]() { worker(t); })) break;
            if (n_threads) worker(0);   // this thread takes tasks from the same counter
            for (auto& t : th) t.join();
            for (size_t t = 0; t < n_threads; t++)
                for (size_t i = 0; i < n_sec; i++) keep_first(bad[i], tb[t][i]);
            each_child([&](size_t i) { keep_first(bad[i], run(c.sub[i].back(), c.sections[i + 1], true)); });
        }
        for (long b : bad)
            if (b >= 0 && (first_bad < 0 || b < first_bad)) first_bad = b;
        const long tail = run(c.sections[n_sec], c.order.size(), true);
        if (first_bad < 0) first_bad = tail;
    }
    R->node_pvs.assign(c.n_pvs, 0);
    for (size_t i = 0; i < c.n_pvs; i++) R->node_pvs[i] = from_monty(vals[c.pv_wires[i / 4]].c[i % 4]);
    if (node_pvs_out) memcpy(node_pvs_out, R->node_pvs.data(), 4 * c.n_pvs);
    if (first_bad >= 0) {
        R->error = "the circuit is not satisfied: gate " + std::to_string(first_bad) + " fails (a child proof does not verify)";
        return ZKHIP_ERR_VERIFY;
    }
    R->error.clear();
    return ZKHIP_OK;
}
}  // namespace

extern "C" {

int zkhip_recursion_witness(zkhip_recursion* R, const uint8_t* const* proofs, const size_t* proof_lens, const uint32_t* const* const* child_pvs,
                            size_t n_present, uint32_t* node_pvs_out) {
    return recursion_witness(R, proofs, proof_lens, child_pvs, n_present, node_pvs_out, nullptr);
}

// The witness of a UNIFORM node (zkhip_recursion_stmt.child_is_node = 2): child c is a proof under the key whose preprocessed commitments
// are child_prep_commits[c] (8 canonical words per AIR with a preprocessed trace, in AIR order), of the leaf circuit (child_is_leaf[c] != 0)
// or of the internal circuit; leaf_commit / internal_commit = zkhip_recursion_key_commit of the two circuits' keys, which the node
// states in its public values and requires of every node below it.
int zkhip_recursion_witness_uniform(zkhip_recursion* R, const uint8_t* const* proofs, const size_t* proof_lens, const uint32_t* const* const* child_pvs,
                                    const uint32_t* child_prep_commits, const int* child_is_leaf, const uint32_t leaf_commit[8],
                                    const uint32_t internal_commit[8], size_t n_present, uint32_t* node_pvs_out) {
    const UniformIn uni{child_prep_commits, child_is_leaf, leaf_commit, internal_commit};
    return recursion_witness(R, proofs, proof_lens, child_pvs, n_present, node_pvs_out, &uni);
}

// The witness of a DEFERRAL node (zkhip_recursion_stmt.child_is_node = 3): child c is a guest flow's root proof; child_aux[c] = the 16 cells of
// its two public-value blocks (bytes 2 j, 2 j + 1 of the 32 public-value bytes = cell j) and the 27 sibling digests above the block pair
// in its final memory root, bottom-up (232 canonical words); acc_start = the claim chain before this node (zero for the first node).
int zkhip_recursion_witness_deferral(zkhip_recursion* R, const uint8_t* const* proofs, const size_t* proof_lens, const uint32_t* const* const* child_pvs,
                                     const uint32_t* child_aux, const uint32_t acc_start[8], size_t n_present, uint32_t* node_pvs_out) {
    const DeferralIn def{child_aux, acc_start};
    return recursion_witness(R, proofs, proof_lens, child_pvs, n_present, node_pvs_out, nullptr, &def);
}
size_t zkhip_recursion_n_aux(const zkhip_recursion* R) { return R ? R->k->n_aux : 0; }

// wire values of the last witness call, canonical: [n_wires + 1][4] (wire 0 unused).  Host-side twin data for the oracle's
// trace generators (tests) -- the product path keeps them inside and goes through zkhip_recursion_tracegen.
int zkhip_recursion_wires(const zkhip_recursion* R, uint32_t* out, size_t cap_words, size_t* n_words) {
    if (!R) return ZKHIP_ERR_INVALID;
    const size_t n = 4 * R->vals.size();
    if (n_words) *n_words = n;
    if (!out) return ZKHIP_OK;
    if (cap_words < n) return ZKHIP_ERR_SMALL_BUFFER;
    for (size_t i = 0; i < R->vals.size(); i++)
        for (int k = 0; k < 4; k++) out[4 * i + k] = from_monty(R->vals[i].c[k]);
    return ZKHIP_OK;
}

}  // extern "C"

// ---- device trace generation: gather the wire values into the chips' traces ----
namespace {

// gate chip: trace[4 s + k][row] = wires[ids[s][row]][k]; on a Horner row (ids[4][row] = its coordinate mask, else 0) the three values between
// its steps are computed here into slots 4 .. 6 (they are no wires)
__global__ __launch_bounds__(256) void k_gate_trace(const uint32_t* __restrict__ ids, const uint4* __restrict__ wires, size_t n_gates, size_t N,
                                                    uint32_t* __restrict__ trace) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    uint4 v[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        v[s] = make_uint4(0, 0, 0, 0);
        if (row < n_gates) v[s] = wires[ids[(size_t)s * n_gates + row]];
        trace[(size_t)(4 * s + 0) * N + row] = v[s].x, trace[(size_t)(4 * s + 1) * N + row] = v[s].y;
        trace[(size_t)(4 * s + 2) * N + row] = v[s].z, trace[(size_t)(4 * s + 3) * N + row] = v[s].w;
    }
    const uint32_t mask = row < n_gates ? ids[(size_t)4 * n_gates + row] : 0u;
    Ext t = Ext{{v[0].x, v[0].y, v[0].z, v[0].w}};
    const Ext al = Ext{{v[3].x, v[3].y, v[3].z, v[3].w}};
    const uint32_t wv[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int j = 3 - i;
        if (mask >> j & 1u) {
            t = ext_mul(t, al);
            t.c[0] = madd(t.c[0], wv[j]);
        }
        const Ext o = mask ? t : ext_zero();
#pragma unroll
        for (int k = 0; k < 4; k++) trace[(size_t)(16 + 4 * i + k) * N + row] = o.c[k];
    }
}
// Poseidon2 chip: inputs[row][4 j + k] = wires[in_ids[j][row]][k]
__global__ __launch_bounds__(256) void k_p2w_inputs(const uint32_t* __restrict__ ids, const uint4* __restrict__ wires, size_t n_perms,
                                                    uint4* __restrict__ inputs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 4 * n_perms) return;
    const size_t row = i / 4, j = i % 4;
    inputs[i] = wires[ids[j * n_perms + row]];
}

}  // namespace

extern "C" int zkhip_recursion_tracegen(zkhip_ctx* ctx, zkhip_recursion* R, uint32_t* d_gate_trace, uint32_t* d_p2_trace, uint32_t* d_pv_trace) {
    ZK_BIND_DEVICE(ctx);
    if (!ctx || !R || !d_gate_trace || !d_p2_trace || !d_pv_trace) return ZKHIP_ERR_INVALID;
    RecCore& K = *R->k;
    const Circuit& c = K.c;
    if (R->vals.size() != (size_t)c.n_wires + 1) return set_error(ctx, ZKHIP_ERR_INVALID, "recursion_tracegen: no witness (call zkhip_recursion_witness first)");
    const size_t ng = c.gates.size(), np = c.perms.size();
    uint32_t *d_gate_ids = nullptr, *d_perm_ids = nullptr;
    {
        // the wiring ids of this device: uploaded by the first user, shared by the forks
        std::lock_guard<std::mutex> lk(K.dev_mu);
        auto it = K.dev_ids.find(ctx->device);
        if (it == K.dev_ids.end()) {
            ZK_HIP_CHECK(ctx, hipMalloc(&d_gate_ids, std::max<size_t>(1, 5 * ng) * 4));
            ZK_HIP_CHECK(ctx, hipMalloc(&d_perm_ids, std::max<size_t>(1, 4 * np) * 4));
            std::vector<uint32_t> ids(5 * ng), pids(4 * np);
            for (size_t g = 0; g < ng; g++) {
                for (int s = 0; s < 4; s++) ids[(size_t)s * ng + g] = c.gates[g].w[s];
                ids[(size_t)4 * ng + g] = c.gates[g].kind == K_HSTEP ? c.gates[g].hmask : 0u;
            }
            for (size_t i = 0; i < np; i++)
                for (int j = 0; j < 4; j++) pids[(size_t)j * np + i] = c.perms[i].in[j];
            ZK_TRY(upload(ctx, d_gate_ids, ids.data(), ids.size() * 4));
            ZK_TRY(upload(ctx, d_perm_ids, pids.data(), pids.size() * 4));
            it = K.dev_ids.emplace(ctx->device, std::make_pair(d_gate_ids, d_perm_ids)).first;
        }
        d_gate_ids = it->second.first, d_perm_ids = it->second.second;
    }
    if (R->dev_ready_device != ctx->device) {
        for (uint32_t** p : {&R->d_wires, &R->d_p2_inputs})
            if (*p) (void)hipFree(*p), *p = nullptr;
        ZK_HIP_CHECK(ctx, hipMalloc(&R->d_wires, ((size_t)c.n_wires + 1) * 16));
        ZK_HIP_CHECK(ctx, hipMalloc(&R->d_p2_inputs, std::max<size_t>(1, np) * 64));
        R->dev_ready_device = ctx->device;
    }
    // The value array is a mapping of its own (WireBuf): page-locked once, for the life of the circuit's user, so that the 40 MB of a
    // node's wire values cross PCIe without a staging copy.  zkhip_config.pin_witness = 0 leaves it pageable.
    if (!R->vals.registered && ctx->cfg.pin_witness) {
        R->vals.registered = hipHostRegister(R->vals.p, R->vals.bytes, hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
    }
    ZK_TRY(upload(ctx, R->d_wires, R->vals.p, R->vals.size() * 16));
    const size_t N0 = (size_t)1 << K.log_height[0];
    {
        KernelScope ks(ctx, "recursion_gate_trace");
        hipLaunchKernelGGL(k_gate_trace, dim3((unsigned)((N0 + 255) / 256)), dim3(256), 0, ctx->stream, d_gate_ids, (const uint4*)R->d_wires, ng, N0,
                           d_gate_trace);
    }
    if (np) {
        KernelScope ks(ctx, "recursion_p2_inputs");
        hipLaunchKernelGGL(k_p2w_inputs, dim3((unsigned)((4 * np + 255) / 256)), dim3(256), 0, ctx->stream, d_perm_ids, (const uint4*)R->d_wires, np,
                           (uint4*)R->d_p2_inputs);
    }
    ZK_HIP_CHECK(ctx, hipGetLastError());
    ZK_TRY(poseidon2_air_tracegen(ctx, R->d_p2_inputs, np, K.log_height[1], d_p2_trace));
    ZK_HIP_CHECK(ctx, hipMemsetAsync(d_pv_trace, 0, 4, ctx->stream));
    return ZKHIP_OK;
}
