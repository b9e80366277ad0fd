// zkhip_air.hpp -- C++ host-side AIR builder for libzkhip: symbolic expressions -> the constraint bytecode that
// zkhip_keygen / zkhip_verify consume (format: DESIGN.md section 4, "AIR bytecode").
//
// Mirrors, for the compiled side of the reference, the builder interface its chips are written against:
//   * p3-air 0.4.3 `AirBuilder` (main-trace window with rotation 0/1, public values, is_first_row / is_last_row /
//     is_transition, assert_zero, when_*), `PairBuilder::preprocessed`;
//   * openvm-stark-backend `InteractionBuilder::push_interaction(bus, fields, count, kind)` -- the bus messages every
//     OpenVM chip of the chunk circuit declares (Cargo.lock pins);
// and lowers interactions to the LogUp constraints exactly like zkvm-prover_amd/air.py (word-for-word the same
// programs: tests/test_air_builder_cpp.py), so a Rust/C++ integrator needs no Python.  Header-only, no device code.
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <optional>
#include <stdexcept>
#include <tuple>
#include <vector>

namespace zkhip {
namespace air {

constexpr uint32_t P = 2013265921u;
constexpr uint32_t AIR_MAGIC = 0x31414B5Au, PREP_MAGIC = 0x50504B5Au, LOGUP_MAGIC = 0x554C4B5Au, CACHED_MAGIC = 0x43414B5Au;
enum Op : uint32_t { VAR, PUB, CONST, FIRST, LAST, TRANS, ADD, SUB, MUL, NEG, PERM, CHAL, EXPOSED, PREP };
constexpr unsigned LOGUP_MAX_FIELDS = 32;  // challenge vector = gamma, beta^1 .. beta^32
constexpr uint32_t EXT_W = 11;             // x^4 = 11
enum class Kind { Send, Receive };

class AirBuilder;

struct Expr {
    AirBuilder* b = nullptr;
    uint32_t idx = 0;
    unsigned deg = 0;
};

class AirBuilder {
  public:
    AirBuilder(size_t width, size_t n_pvs = 0, size_t prep_width = 0) : width_(width), n_pvs_(n_pvs), prep_width_(prep_width) {}
    // cached main partition (OpenVM-v1 cached main): the first `cw` main columns are committed in a tree of their own
    void set_cached_width(size_t cw) {
        if (cw >= width_) throw std::out_of_range("cached_width must leave a common part");
        cached_width_ = cw;
    }
    size_t cached_width() const { return cached_width_; }
    AirBuilder(const AirBuilder&) = delete;  // expressions point at their builder
    AirBuilder& operator=(const AirBuilder&) = delete;

    size_t width() const { return width_; }
    size_t n_pvs() const { return n_pvs_; }
    size_t prep_width() const { return prep_width_; }
    size_t n_nodes() const { return nodes_.size(); }
    size_t n_constraints() const { return cons_.size(); }
    unsigned max_constraint_degree = 3;  // degree budget of the LogUp grouping (2^log_blowup + 1)

    Expr var(size_t col, unsigned rot = 0) {
        if (col >= width_ || rot > 1) throw std::out_of_range("var");
        return node(VAR, (uint32_t)col, rot, 1);
    }
    Expr next(size_t col) { return var(col, 1); }
    Expr prep(size_t col, unsigned rot = 0) {  // cell of the preprocessed trace (p3 PairBuilder::preprocessed)
        if (col >= prep_width_ || rot > 1) throw std::out_of_range("prep");
        return node(PREP, (uint32_t)col, rot, 1);
    }
    Expr pub(size_t i) {
        if (i >= n_pvs_) throw std::out_of_range("pub");
        return node(PUB, (uint32_t)i, 0, 0);
    }
    Expr constant(int64_t v) { return node(CONST, (uint32_t)(((v % (int64_t)P) + (int64_t)P) % (int64_t)P), 0, 0); }
    Expr is_first_row() { return node(FIRST, 0, 0, 1); }
    Expr is_last_row() { return node(LAST, 0, 0, 1); }
    Expr is_transition() { return node(TRANS, 0, 0, 0); }
    void assert_zero(Expr e) { cons_.push_back(e.idx); }
    inline void when_first_row(Expr e);
    inline void when_last_row(Expr e);
    inline void when_transition(Expr e);

    // leaves of the after-challenge (LogUp) phase; chips do not use them directly
    Expr perm(size_t col, unsigned rot = 0) { return node(PERM, (uint32_t)col, rot, 1); }
    Expr chal(size_t i) { return node(CHAL, (uint32_t)i, 0, 0); }
    Expr exposed(size_t i) { return node(EXPOSED, (uint32_t)i, 0, 0); }

    // Bus message: `fields` and `count` are expressions of the CURRENT row (main / preprocessed cells with rotation 0,
    // public values, constants, + - * neg).  Send adds count/denominator to the bus, Receive subtracts it.
    void push_interaction(uint32_t bus, const std::vector<Expr>& fields, Expr count, Kind kind) {
        if (logup_done_) throw std::logic_error("push_interaction after program()");
        if (bus >= (1u << 20) || fields.empty() || fields.size() > LOGUP_MAX_FIELDS) throw std::invalid_argument("interaction");
        for (const Expr& f : fields)
            if (!row_local(f.idx)) throw std::invalid_argument("interaction operands must be expressions of the current row");
        if (!row_local(count.idx)) throw std::invalid_argument("interaction operands must be expressions of the current row");
        ints_.push_back({bus, kind == Kind::Send ? 0u : 1u, count, fields, 0u});
    }

    unsigned max_degree() const {
        std::vector<unsigned> deg(nodes_.size());
        for (size_t i = 0; i < nodes_.size(); i++) {
            const auto& [op, a, b] = nodes_[i];
            switch (op) {
                case VAR: case PERM: case PREP: case FIRST: case LAST: deg[i] = 1; break;
                case PUB: case CONST: case TRANS: case CHAL: case EXPOSED: deg[i] = 0; break;
                case ADD: case SUB: deg[i] = std::max(deg[a], deg[b]); break;
                case MUL: deg[i] = deg[a] + deg[b]; break;
                default: deg[i] = deg[a];
            }
        }
        unsigned m = 0;
        for (uint32_t c : cons_) m = std::max(m, deg[c]);
        return m;
    }

    // Serialises the AIR (appending the LogUp constraints of the interactions pushed so far, once).
    inline std::vector<uint32_t> program();
    const std::vector<uint32_t>& interaction_groups() const { return groups_; }

    Expr node(uint32_t op, uint32_t a, uint32_t b, unsigned deg) {
        const auto key = std::make_tuple(op, a, b);
        auto it = cache_.find(key);
        if (it != cache_.end()) return Expr{this, it->second.first, it->second.second};
        nodes_.push_back(key);
        cache_.emplace(key, std::make_pair((uint32_t)nodes_.size() - 1, deg));
        return Expr{this, (uint32_t)nodes_.size() - 1, deg};
    }

  private:
    struct Interaction {
        uint32_t bus, sign;
        Expr count;
        std::vector<Expr> fields;
        uint32_t group;
    };
    using Ext = std::array<std::optional<Expr>, 4>;  // extension-field expression, nullopt = zero coordinate

    bool row_local(uint32_t idx) const {
        std::vector<uint32_t> stack{idx};
        std::vector<uint8_t> seen(nodes_.size(), 0);
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (seen[i]) continue;
            seen[i] = 1;
            const auto& [op, a, b] = nodes_[i];
            if (op == VAR || op == PREP) {
                if (b != 0) return false;
            } else if (op == PUB || op == CONST) {
            } else if (op == ADD || op == SUB || op == MUL) {
                stack.push_back(a);
                stack.push_back(b);
            } else if (op == NEG) {
                stack.push_back(a);
            } else {
                return false;
            }
        }
        return true;
    }
    inline Ext ext_mul(const Ext& a, const Ext& b);
    inline Ext ext_add(const Ext& a, const Ext& b);
    inline void finalize_interactions();

    size_t width_, n_pvs_, prep_width_, cached_width_ = 0;
    std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> nodes_;
    std::vector<uint32_t> cons_;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::pair<uint32_t, unsigned>> cache_;
    std::vector<Interaction> ints_;
    std::vector<uint32_t> groups_;
    bool logup_done_ = false;
};

// ---- expression operators (an integer operand becomes a constant node, reduced mod p) ----
inline Expr operator+(Expr x, Expr y) { return x.b->node(ADD, x.idx, y.idx, std::max(x.deg, y.deg)); }
inline Expr operator-(Expr x, Expr y) { return x.b->node(SUB, x.idx, y.idx, std::max(x.deg, y.deg)); }
inline Expr operator*(Expr x, Expr y) { return x.b->node(MUL, x.idx, y.idx, x.deg + y.deg); }
inline Expr operator-(Expr x) { return x.b->node(NEG, x.idx, 0, x.deg); }
inline Expr operator+(Expr x, int64_t c) { return x + x.b->constant(c); }
inline Expr operator+(int64_t c, Expr x) { return x + x.b->constant(c); }
inline Expr operator-(Expr x, int64_t c) { return x - x.b->constant(c); }
inline Expr operator-(int64_t c, Expr x) { return x.b->constant(c) - x; }
inline Expr operator*(Expr x, int64_t c) { return x * x.b->constant(c); }
inline Expr operator*(int64_t c, Expr x) { return x * x.b->constant(c); }

inline void AirBuilder::when_first_row(Expr e) { assert_zero(is_first_row() * e); }
inline void AirBuilder::when_last_row(Expr e) { assert_zero(is_last_row() * e); }
inline void AirBuilder::when_transition(Expr e) { assert_zero(is_transition() * e); }

inline AirBuilder::Ext AirBuilder::ext_mul(const Ext& a, const Ext& b) {
    Ext out;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            if (!a[i] || !b[j]) continue;
            Expr t = *a[i] * *b[j];
            if (i + j >= 4) t = t * (int64_t)EXT_W;
            const int m = (i + j) % 4;
            out[m] = out[m] ? *out[m] + t : t;
        }
    return out;
}
inline AirBuilder::Ext AirBuilder::ext_add(const Ext& a, const Ext& b) {
    Ext out;
    for (int k = 0; k < 4; k++) out[k] = !b[k] ? a[k] : (!a[k] ? b[k] : std::optional<Expr>(*a[k] + *b[k]));
    return out;
}

// LogUp constraints of the pushed interactions: interactions are packed greedily into
// column groups while the group constraint phi_g * prod den_j = sum_j (+-count_j) prod_{k != j} den_k stays within the
// degree budget; the running sum of all phi_g owns the last four permutation columns.
inline void AirBuilder::finalize_interactions() {
    if (ints_.empty() || logup_done_) return;
    logup_done_ = true;
    const unsigned budget = max_constraint_degree;
    std::vector<Ext> dens;
    std::vector<unsigned> dds;
    for (const Interaction& it : ints_) {
        Ext d;
        for (int k = 0; k < 4; k++) {
            Expr e = chal(k);
            if (k == 0) e = e + (int64_t)(it.bus + 1);
            for (size_t i = 0; i < it.fields.size(); i++) e = e + chal(4 * (i + 1) + k) * it.fields[i];
            d[k] = e;
        }
        dens.push_back(d);
        unsigned dd = 0;
        for (const Expr& f : it.fields) dd = std::max(dd, f.deg);
        dds.push_back(dd);
    }
    std::vector<std::vector<size_t>> groups;
    for (size_t j = 0; j < ints_.size(); j++) {
        bool placed = false;
        if (!groups.empty() && groups.back().size() < 4) {
            std::vector<size_t> g = groups.back();
            g.push_back(j);
            unsigned dsum = 0;
            for (size_t k : g) dsum += dds[k];
            bool ok = 1 + dsum <= budget;
            for (size_t k : g) ok = ok && ints_[k].count.deg + dsum - dds[k] <= budget;
            if (ok) {
                groups.back() = g;
                placed = true;
            }
        }
        if (!placed) groups.push_back({j});
    }
    groups_.assign(ints_.size(), 0);
    for (size_t gi = 0; gi < groups.size(); gi++)
        for (size_t j : groups[gi]) groups_[j] = ints_[j].group = (uint32_t)gi;
    const size_t n_grp = groups.size();
    for (size_t gi = 0; gi < n_grp; gi++) {
        const auto& g = groups[gi];
        Ext lhs;
        for (int k = 0; k < 4; k++) lhs[k] = perm(4 * gi + k);
        for (size_t j : g) lhs = ext_mul(lhs, dens[j]);
        Ext rhs;
        for (size_t j : g) {
            Ext term;
            term[0] = ints_[j].sign == 0 ? ints_[j].count : -ints_[j].count;
            for (size_t k : g)
                if (k != j) term = ext_mul(term, dens[k]);
            rhs = ext_add(rhs, term);
        }
        for (int m = 0; m < 4; m++) assert_zero(rhs[m] ? *lhs[m] - *rhs[m] : *lhs[m]);
    }
    for (int k = 0; k < 4; k++) {
        // node creation order follows air.py: s_loc[0..3], s_nxt[0..3] first
        (void)perm(4 * n_grp + k);
    }
    for (int k = 0; k < 4; k++) (void)perm(4 * n_grp + k, 1);
    for (int k = 0; k < 4; k++) {
        const Expr s_loc = perm(4 * n_grp + k), s_nxt = perm(4 * n_grp + k, 1);
        std::optional<Expr> row_sum, nxt_sum;
        for (size_t gi = 0; gi < n_grp; gi++) {
            row_sum = row_sum ? *row_sum + perm(4 * gi + k) : perm(4 * gi + k);
            nxt_sum = nxt_sum ? *nxt_sum + perm(4 * gi + k, 1) : perm(4 * gi + k, 1);
        }
        when_first_row(s_loc - *row_sum);
        when_transition(s_nxt - s_loc - *nxt_sum);
        when_last_row(s_loc - exposed(k));
    }
}

inline std::vector<uint32_t> AirBuilder::program() {
    finalize_interactions();
    std::vector<uint32_t> w{AIR_MAGIC, (uint32_t)nodes_.size(), (uint32_t)cons_.size(), (uint32_t)n_pvs_};
    for (const auto& [op, a, b] : nodes_) {
        w.push_back(op);
        w.push_back(a);
        w.push_back(b);
    }
    w.insert(w.end(), cons_.begin(), cons_.end());
    if (prep_width_) {
        w.push_back(PREP_MAGIC);
        w.push_back((uint32_t)prep_width_);
    }
    if (cached_width_) {
        w.push_back(CACHED_MAGIC);
        w.push_back((uint32_t)cached_width_);
    }
    if (!ints_.empty()) {
        w.push_back(LOGUP_MAGIC);
        w.push_back((uint32_t)ints_.size());
        for (const Interaction& it : ints_) {
            w.push_back(it.bus);
            w.push_back(it.sign);
            w.push_back(it.count.idx);
            w.push_back((uint32_t)it.fields.size());
            for (const Expr& f : it.fields) w.push_back(f.idx);
            w.push_back(it.group);
        }
    }
    return w;
}

// ---- the Poseidon2 AIR whose trace zkhip_poseidon2_air_tracegen generates (include/zkhip.h) --------------------------
// One permutation per row, the structure of p3-poseidon2-air 0.4.3 with one committed register x^3 per S-box:
//   inputs[16] | 4 x { sbox[16], post[16] } | 13 x { sbox, post_sbox } | 4 x { sbox[16], post[16] }   (298 columns)
// 282 constraints of degree 3.  With `bus` >= 0 the chip serves compression requests like OpenVM's Poseidon2 periphery
// chip: column 298 = multiplicity, receive(bus, inputs[0..16] ++ outputs[0..8], mult).
constexpr size_t POSEIDON2_AIR_WIDTH = 16 + 4 * 32 + 13 * 2 + 4 * 32;

// 141 canonical round constants: Poseidon Grain LFSR for (prime field, x^7, n = 31, t = 16, R_F = 8, R_P = 13)
inline std::vector<uint32_t> poseidon2_round_constants() {
    std::vector<uint8_t> bits;
    auto put = [&](unsigned v, int n) { for (int i = 0; i < n; i++) bits.push_back((v >> (n - 1 - i)) & 1); };
    put(1, 2), put(0, 4), put(31, 12), put(16, 12), put(8, 10), put(13, 10);
    while (bits.size() < 80) bits.push_back(1);
    size_t head = 0;  // bits[head .. head+80) is the register
    auto step = [&]() {
        const uint8_t nb = bits[head + 62] ^ bits[head + 51] ^ bits[head + 38] ^ bits[head + 23] ^ bits[head + 13] ^ bits[head];
        bits.push_back(nb);
        head++;
        return nb;
    };
    auto next_bit = [&]() {
        for (;;) {
            const uint8_t a = step(), c = step();
            if (a) return c;
        }
    };
    for (int i = 0; i < 160; i++) step();
    std::vector<uint32_t> out;
    while (out.size() < 141) {
        uint32_t v = 0;
        for (int i = 0; i < 31; i++) v = (v << 1) | next_bit();
        if (v < P) out.push_back(v);
    }
    return out;
}

inline std::array<uint32_t, 16> poseidon2_internal_diag() {
    auto powm = [](uint64_t b, uint64_t e) {
        uint64_t r = 1;
        for (b %= P; e; e >>= 1, b = b * b % P)
            if (e & 1) r = r * b % P;
        return (uint32_t)r;
    };
    const uint32_t i2 = powm(2, P - 2);
    auto neg = [](uint32_t x) { return x ? P - x : 0u; };
    return {P - 2, 1, 2, i2, 3, 4, neg(i2), P - 3, P - 4, powm(i2, 8), powm(i2, 2), powm(i2, 3), powm(i2, 27),
            neg(powm(i2, 8)), neg(powm(i2, 4)), neg(powm(i2, 27))};
}

namespace detail {
// circ(2 M4, M4, M4, M4), M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]], on 16 expressions
inline std::vector<Expr> p2_external(const std::vector<Expr>& s) {
    std::vector<Expr> out(16);
    for (int blk = 0; blk < 16; blk += 4) {
        const Expr x0 = s[blk], x1 = s[blk + 1], x2 = s[blk + 2], x3 = s[blk + 3];
        const Expr t01 = x0 + x1;
        const Expr t23 = x2 + x3;
        const Expr t0123 = t01 + t23;
        const Expr t01123 = t0123 + x1;
        const Expr t01233 = t0123 + x3;
        out[blk + 3] = t01233 + x0 * 2;
        out[blk + 1] = t01123 + x2 * 2;
        out[blk + 0] = t01123 + t01;
        out[blk + 2] = t01233 + t23;
    }
    std::vector<Expr> sums;
    for (int k = 0; k < 4; k++) {
        const Expr lo = out[k] + out[4 + k];
        const Expr hi = out[8 + k] + out[12 + k];
        sums.push_back(lo + hi);
    }
    std::vector<Expr> r;
    for (int i = 0; i < 16; i++) r.push_back(out[i] + sums[i % 4]);
    return r;
}
}  // namespace detail

// Fills `b` (constructed with width POSEIDON2_AIR_WIDTH, or +1 with a bus) with the constraints of the Poseidon2 AIR.
inline void poseidon2_air(AirBuilder& b, int bus = -1, int out_lanes = 8) {   // out_lanes: 8 serve compressions, 16 a sponge (chips::duplex_air)
    if (b.width() != POSEIDON2_AIR_WIDTH + (bus >= 0 ? 1 : 0)) throw std::invalid_argument("poseidon2_air: builder width");
    const std::vector<uint32_t> rc = poseidon2_round_constants();
    const std::array<uint32_t, 16> diag = poseidon2_internal_diag();
    size_t col = 16;
    std::vector<Expr> in;
    for (int i = 0; i < 16; i++) in.push_back(b.var(i));
    std::vector<Expr> state = detail::p2_external(in);
    auto full_round = [&](const uint32_t* rcs) {
        std::vector<Expr> outs;
        for (int i = 0; i < 16; i++) {
            const Expr y = state[i] + (int64_t)rcs[i];
            const Expr reg = b.var(col + i);
            const Expr y2 = y * y;
            b.assert_zero(reg - y2 * y);
            const Expr r2 = reg * reg;
            outs.push_back(r2 * y);
        }
        const std::vector<Expr> lin = detail::p2_external(outs);
        std::vector<Expr> post;
        for (int i = 0; i < 16; i++) post.push_back(b.var(col + 16 + i));
        for (int i = 0; i < 16; i++) b.assert_zero(post[i] - lin[i]);
        state = post;
        col += 32;
    };
    for (int r = 0; r < 4; r++) full_round(&rc[16 * r]);
    for (int r = 0; r < 13; r++) {
        const Expr y = state[0] + (int64_t)rc[64 + r];
        const Expr reg = b.var(col), post = b.var(col + 1);
        const Expr y2 = y * y;
        b.assert_zero(reg - y2 * y);
        const Expr r2 = reg * reg;
        b.assert_zero(post - r2 * y);
        col += 2;
        state[0] = post;
        Expr total = state[0];
        for (int i = 1; i < 16; i++) total = total + state[i];
        for (int i = 0; i < 16; i++) state[i] = state[i] * (int64_t)diag[i] + total;
    }
    for (int r = 0; r < 4; r++) full_round(&rc[77 + 16 * r]);
    if (bus >= 0) {
        std::vector<Expr> msg = in;
        msg.insert(msg.end(), state.begin(), state.begin() + out_lanes);
        b.push_interaction((uint32_t)bus, msg, b.var(POSEIDON2_AIR_WIDTH), Kind::Receive);
    }
}

}  // namespace air
}  // namespace zkhip
