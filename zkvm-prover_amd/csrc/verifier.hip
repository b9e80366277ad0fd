// verifier.hip -- host-side verifier of libzkhip proofs (no device code; compiled with hipcc only
// for uniformity).  Replaces, for this backend's proofs, what the reference reaches through
// `Sdk::verify_proof` (crates/verifier/src/verifier.rs:82) and runs as the prover's self-check
// (crates/prover/src/prover/mod.rs:407-411).  Written against babybear.hpp / poseidon2.hpp
// (Montgomery arithmetic); it shares no code with oracle/.
#include <string.h>

#include <vector>

#include "air_compile.hpp"
#include "poseidon2.hpp"
#include "../../include/zkhip.h"

namespace zk {
namespace {

struct HostChallenger {
    uint32_t state[16];
    uint32_t in_buf[8], out_buf[8];
    unsigned n_in = 0, n_out = 0;
    HostChallenger() { memset(state, 0, sizeof state); }
    void duplex() {
        for (unsigned i = 0; i < n_in; i++) state[i] = in_buf[i];
        n_in = 0;
        poseidon2_permute_host(state);
        memcpy(out_buf, state, sizeof out_buf);
        n_out = 8;
    }
    void observe(uint32_t v_monty) {
        n_out = 0;
        in_buf[n_in++] = v_monty;
        if (n_in == 8) duplex();
    }
    void observe_canon(const uint32_t* v, size_t n) {
        for (size_t i = 0; i < n; i++) observe(to_monty(v[i]));
    }
    uint32_t sample() {
        if (n_in != 0 || n_out == 0) duplex();
        return out_buf[--n_out];
    }
    Ext sample_ext() {
        Ext e;
        for (int i = 0; i < 4; i++) e.c[i] = sample();
        return e;
    }
    uint32_t sample_bits(unsigned bits) { return from_monty(sample()) & (uint32_t)(((uint64_t)1 << bits) - 1); }
    bool check_witness(unsigned bits, uint32_t w_canon) {
        observe(to_monty(w_canon));
        return sample_bits(bits) == 0;
    }
};

Ext ext_from_canon(const uint32_t* p) { return Ext{{to_monty(p[0]), to_monty(p[1]), to_monty(p[2]), to_monty(p[3])}}; }

// recompute the root implied by an opening of a mixed-height commitment
bool verify_opening(const uint32_t root_m[8], const std::vector<unsigned>& lhs, const std::vector<size_t>& ws,
                    size_t index, const uint32_t* opening_canon) {
    unsigned lh = 0;
    size_t total = 0;
    for (size_t m = 0; m < lhs.size(); m++) {
        if (lhs[m] > lh) lh = lhs[m];
        total += ws[m];
    }
    std::vector<uint32_t> tmp(total + 1);
    const uint32_t* path = opening_canon + total;
    uint32_t cur[8];
    for (unsigned level = lh;; level--) {
        size_t len = 0, off = 0;
        bool any = false;
        for (size_t m = 0; m < lhs.size(); m++) {
            if (lhs[m] == level) {
                for (size_t k = 0; k < ws[m]; k++) tmp[len++] = to_monty(opening_canon[off + k]);
                any = true;
            }
            off += ws[m];
        }
        if (level == lh) {
            p2_hash_slice(tmp.data(), len, cur);
        } else {
            unsigned l = lh - level - 1;
            uint32_t sib[8];
            for (int k = 0; k < 8; k++) sib[k] = to_monty(path[8 * l + k]);
            if (((index >> l) & 1) == 0) p2_compress(cur, sib, cur);
            else p2_compress(sib, cur, cur);
            if (any) {
                uint32_t hsh[8];
                p2_hash_slice(tmp.data(), len, hsh);
                p2_compress(cur, hsh, cur);
            }
        }
        if (level == 0) break;
    }
    return memcmp(cur, root_m, 32) == 0;
}

// p3-fri `fold_row` for arity 2: pair k of a layer of 2^(log_n_out+1) values in bit-reversed order sits on the points
// +-x, x = g^bitrev(k) with g the generator of that size; the line through (x, e0), (-x, e1) is evaluated at beta.
Ext fold_row(size_t k, unsigned log_n_out, const Ext& beta, const Ext& e0, const Ext& e1) {
    const uint32_t xx = mpow(two_adic_generator(log_n_out + 1), bitrev32((uint32_t)k, log_n_out));
    const uint32_t c = mneg(mmul(minv(xx), minv(to_monty(2))));
    Ext bx = beta;
    bx.c[0] = msub(bx.c[0], xx);
    return ext_add(e0, ext_mul(bx, ext_mul_base(ext_sub(e1, e0), c)));
}

}  // namespace
}  // namespace zk

using namespace zk;

// ---- the verifier's primitives as host entry points (no device): what p3's `Mmcs::verify_batch`, `fold_row` and the
//      permutation are to the reference's verifier (crates/verifier/src/verifier.rs:82 -> Sdk::verify_proof).  They let a
//      consumer -- and tests/test_ref_vectors_cpu.py, against the reference's own stored proofs -- check openings and
//      fold steps of ANY BabyBear-Poseidon2 v1 proof, not only the ones this backend writes. ----
extern "C" int zkhip_poseidon2_permute_host(uint32_t state[16]) {
    if (!state) return ZKHIP_ERR_INVALID;
    uint32_t s[16];
    for (int i = 0; i < 16; i++) {
        if (state[i] >= P) return ZKHIP_ERR_INVALID;
        s[i] = to_monty(state[i]);
    }
    poseidon2_permute(s);
    for (int i = 0; i < 16; i++) state[i] = from_monty(s[i]);
    return ZKHIP_OK;
}

extern "C" int zkhip_poseidon2_permute_host_avx512(uint32_t state[16]) {
    if (!state) return ZKHIP_ERR_INVALID;
#ifndef ZK_HAVE_HOST_AVX512
    return 1;
#else
    if (!__builtin_cpu_supports("avx512f") || !__builtin_cpu_supports("avx512dq")) return 1;
    uint32_t s[16];
    for (int i = 0; i < 16; i++) {
        if (state[i] >= P) return ZKHIP_ERR_INVALID;
        s[i] = to_monty(state[i]);
    }
    zk::poseidon2_permute_avx512(s);
    for (int i = 0; i < 16; i++) state[i] = from_monty(s[i]);
    return ZKHIP_OK;
#endif
}

// sixteen independent permutations at once (the aggregation witness generator's form): states[16 k + w] = word w of state k, canonical
extern "C" int zkhip_poseidon2_permute16_host(uint32_t states[256]) {
    if (!states) return ZKHIP_ERR_INVALID;
    uint32_t t[256];
    for (int k = 0; k < 16; k++)
        for (int w = 0; w < 16; w++) {
            if (states[16 * k + w] >= P) return ZKHIP_ERR_INVALID;
            t[16 * w + k] = to_monty(states[16 * k + w]);
        }
    poseidon2_permute16_host(t);
    for (int k = 0; k < 16; k++)
        for (int w = 0; w < 16; w++) states[16 * k + w] = from_monty(t[16 * w + k]);
    return ZKHIP_OK;
}

extern "C" int zkhip_mmcs_verify(const uint32_t root[8], const unsigned* log_heights, const size_t* widths, size_t n_mats,
                                 uint64_t index, const uint32_t* opening) {
    if (!root || !log_heights || !widths || !opening || n_mats == 0) return ZKHIP_ERR_INVALID;
    std::vector<unsigned> lhs(log_heights, log_heights + n_mats);
    std::vector<size_t> ws(widths, widths + n_mats);
    unsigned lh = 0;
    size_t total = 0;
    for (size_t m = 0; m < n_mats; m++) {
        if (lhs[m] > 27) return ZKHIP_ERR_INVALID;
        lh = std::max(lh, lhs[m]);
        total += ws[m];
    }
    if (index >> lh) return ZKHIP_ERR_INVALID;
    for (size_t i = 0; i < total + 8 * (size_t)lh; i++)
        if (opening[i] >= P) return ZKHIP_ERR_VERIFY;
    uint32_t root_m[8];
    for (int k = 0; k < 8; k++) {
        if (root[k] >= P) return ZKHIP_ERR_VERIFY;
        root_m[k] = to_monty(root[k]);
    }
    return verify_opening(root_m, lhs, ws, (size_t)index, opening) ? ZKHIP_OK : ZKHIP_ERR_VERIFY;
}

extern "C" int zkhip_fri_fold_row(uint64_t index, unsigned log_height, const uint32_t beta[4], const uint32_t e0[4],
                                  const uint32_t e1[4], uint32_t out[4]) {
    if (!beta || !e0 || !e1 || !out || log_height > 26 || (index >> log_height)) return ZKHIP_ERR_INVALID;
    for (int k = 0; k < 4; k++)
        if (beta[k] >= P || e0[k] >= P || e1[k] >= P) return ZKHIP_ERR_INVALID;
    const Ext r = fold_row((size_t)index, log_height, ext_from_canon(beta), ext_from_canon(e0), ext_from_canon(e1));
    for (int k = 0; k < 4; k++) out[k] = from_monty(r.c[k]);
    return ZKHIP_OK;
}

// The bus check of the verifier on its own: the exposed cumulative sums of the AIRs with interactions cancel (crates/verifier/src/
// verifier.rs:82 -> Sdk::verify_proof does this inside the engine; the `exposed_values_after_challenge` of the reference's stored
// proofs are the vectors, tests/test_ref_vectors_cpu.py).
extern "C" int zkhip_logup_exposed_check(const uint32_t* exposed, size_t n) {
    if (n && !exposed) return ZKHIP_ERR_INVALID;
    Ext tot = ext_zero();
    for (size_t k = 0; k < n; k++) {
        for (int q = 0; q < 4; q++)
            if (exposed[4 * k + q] >= P) return ZKHIP_ERR_INVALID;
        tot = ext_add(tot, ext_from_canon(exposed + 4 * k));
    }
    return ext_eq(tot, ext_zero()) ? ZKHIP_OK : ZKHIP_ERR_VERIFY;
}

// Word offsets of the fields of a proof: what Proof::<SC>::decode_from_bytes gives the reference's
// verifier (crates/verifier/src/verifier.rs:62) -- here the layout is static, so "decoding" is a table of offsets.
extern "C" int zkhip_proof_layout_of(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, zkhip_proof_layout* out) {
    if (!prm || !airs || !out || n_airs == 0) return ZKHIP_ERR_INVALID;
    const unsigned b = prm->log_blowup;
    const unsigned lfp = prm->log_final_poly_len;
    if (lfp > ZKHIP_MAX_LOG_FINAL_POLY || b < 1 || b > 4) return ZKHIP_ERR_INVALID;
    unsigned hmax = 0;
    size_t n_lu = 0, n_open = 0, n_cached = 0;
    size_t main_w = 0, perm_w = 0, quot_w = 0, prep_words = 0, cached_words = 0;   // opened row words per query, per input commitment
    unsigned main_h = 0, perm_h = 0;
    for (size_t a = 0; a < n_airs; a++) {
        AirProgram pg;
        if (parse_air(airs[a].program, airs[a].program_len, airs[a].width, &pg, nullptr) != 0) return ZKHIP_ERR_INVALID;
        if (airs[a].log_height + b > 27 || airs[a].log_height < lfp) return ZKHIP_ERR_INVALID;
        const unsigned h = airs[a].log_height + b;
        hmax = std::max(hmax, h);
        main_h = std::max(main_h, h);
        main_w += airs[a].width - pg.cached_width;
        if (pg.cached_width) n_cached++, cached_words += pg.cached_width + 8 * (size_t)h;
        if (pg.log_qd() > b) return ZKHIP_ERR_CONSTRAINT;
        const unsigned nch = pg.qd();
        quot_w += 4 * (size_t)nch;
        n_open += 2 * airs[a].width + 4 * (size_t)nch;
        if (pg.prep_width) {
            prep_words += pg.prep_width + 8 * (size_t)h;
            n_open += 2 * pg.prep_width;
        }
        if (!pg.ints.empty()) {
            n_lu++;
            perm_w += pg.perm_width();
            perm_h = std::max(perm_h, h);
            n_open += 2 * pg.perm_width();
        }
    }
    const unsigned n_layers = hmax - b - lfp;
    memset(out, 0, sizeof *out);
    size_t r = 4;
    out->root_main = r, r += 8;
    out->roots_cached = n_cached ? r : 0, out->n_cached = n_cached, r += 8 * n_cached;
    if (n_lu) {
        out->root_perm = r, r += 8;
        out->exposed = r, out->n_exposed = n_lu, r += 4 * n_lu;
    }
    out->root_quot = r, r += 8;
    out->opened = r, out->n_opened = n_open, r += 4 * n_open;
    out->fri_layers = r, out->n_fri_layers = n_layers, r += 9 * (size_t)n_layers;
    out->final_poly = r, out->n_final_poly = (size_t)1 << lfp, r += (size_t)4 << lfp;
    out->query_pow = r, r += 1;
    out->queries = r;
    size_t qw = main_w + 8 * (size_t)main_h + cached_words + prep_words + (n_lu ? perm_w + 8 * (size_t)perm_h : 0) + quot_w + 8 * (size_t)main_h;
    for (unsigned l = 0; l < n_layers; l++) qw += 4 + 8 * (size_t)(hmax - l - 1);
    out->query_words = qw;
    out->n_queries = prm->num_queries;
    out->n_words = r + qw * prm->num_queries;
    return ZKHIP_OK;
}

// a rejection notes the line of this file that refused (zkhip_verify_where: which check a proof fails, for diagnosis)
static inline int rejected(int* where, int line) {
    if (where) *where = line;
    return ZKHIP_ERR_VERIFY;
}
static int verify_where(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, const uint32_t* const* pvs,
                        const uint8_t* proof_bytes, size_t len, int* where) {
    if (!prm || !airs || !proof_bytes || n_airs == 0 || (len & 3)) return ZKHIP_ERR_INVALID;
    const unsigned b = prm->log_blowup;
    const unsigned lfp = prm->log_final_poly_len;  // the fold loop stops at 2^(b+lfp) values: a polynomial of degree < 2^lfp
    if (lfp > ZKHIP_MAX_LOG_FINAL_POLY || b < 1 || b > 4) return ZKHIP_ERR_INVALID;
    if (prm->num_queries == 0 || prm->commit_pow_bits > 30 || prm->query_pow_bits > 30) return ZKHIP_ERR_INVALID;  // as zkhip_keygen
    const size_t n_fin = (size_t)1 << lfp;
    const size_t n_words = len / 4;
    if (n_words < 4) return rejected(where, __LINE__);
    std::vector<uint32_t> pw(n_words);
    memcpy(pw.data(), proof_bytes, len);
    const uint32_t* proof = pw.data();
    for (size_t i = 0; i < n_words; i++)
        if (proof[i] >= P) return rejected(where, __LINE__);
    std::vector<AirProgram> pg(n_airs);
    unsigned hmax = 0;
    size_t n_lu = 0, n_prep = 0, n_cached = 0;
    for (size_t a = 0; a < n_airs; a++) {
        if (parse_air(airs[a].program, airs[a].program_len, airs[a].width, &pg[a], nullptr) != 0) return ZKHIP_ERR_INVALID;
        if (pg[a].cached_width) n_cached++;
        if (pg[a].n_pvs != airs[a].n_pvs || airs[a].log_height + b > 27 || airs[a].log_height < lfp || airs[a].width == 0) return ZKHIP_ERR_INVALID;
        if (pg[a].log_qd() > b) return ZKHIP_ERR_CONSTRAINT;
        if (airs[a].n_pvs && (!pvs || !pvs[a])) return ZKHIP_ERR_INVALID;
        hmax = std::max(hmax, airs[a].log_height + b);
        if (!pg[a].ints.empty()) n_lu++;
        if (pg[a].prep_width) {
            if (!airs[a].prep_commit) return ZKHIP_ERR_INVALID;  // the verifying key must hold the commitment
            for (int k = 0; k < 8; k++)
                if (airs[a].prep_commit[k] >= P) return ZKHIP_ERR_INVALID;
            n_prep++;
        }
    }
    // committed matrices in opening order: main (every AIR), permutation (AIRs with interactions), quotient chunks
    struct CMat {
        unsigned lh, h;
        size_t width;
        unsigned n_pts;
        size_t open_off;
    };
    std::vector<CMat> cm;
    // a cached main partition (leading columns of an AIR's main trace) is a matrix of its own, in a tree of its own
    for (size_t a = 0; a < n_airs; a++) cm.push_back({airs[a].log_height, airs[a].log_height + b, airs[a].width - pg[a].cached_width, 2, 0});
    for (size_t a = 0; a < n_airs; a++)
        if (pg[a].cached_width) cm.push_back({airs[a].log_height, airs[a].log_height + b, pg[a].cached_width, 2, 0});
    for (size_t a = 0; a < n_airs; a++)
        if (pg[a].prep_width) cm.push_back({airs[a].log_height, airs[a].log_height + b, pg[a].prep_width, 2, 0});
    for (size_t a = 0; a < n_airs; a++)
        if (!pg[a].ints.empty()) cm.push_back({airs[a].log_height, airs[a].log_height + b, pg[a].perm_width(), 2, 0});
    for (size_t a = 0; a < n_airs; a++)
        for (unsigned j = 0; j < pg[a].qd(); j++) cm.push_back({airs[a].log_height, airs[a].log_height + b, 4, 1, 0});
    const size_t cm_cached0 = n_airs, cm_prep0 = n_airs + n_cached, cm_perm0 = cm_prep0 + n_prep, cm_quot0 = cm_perm0 + n_lu;
    std::vector<size_t> qoff(n_airs + 1, 0);  // AIR a's chunk j is quotient matrix qoff[a] + j
    for (size_t a = 0; a < n_airs; a++) qoff[a + 1] = qoff[a] + pg[a].qd();
    size_t n_open = 0;
    for (auto& m : cm) {
        m.open_off = n_open;
        n_open += m.width * m.n_pts;
    }
    {
        std::vector<unsigned> lhs(n_airs);
        for (size_t a = 0; a < n_airs; a++) lhs[a] = airs[a].log_height;
        if (!logup_bus_counts_bounded(pg.data(), lhs.data(), n_airs)) return ZKHIP_ERR_INVALID;
    }
    const unsigned n_layers = hmax - b - lfp;
    size_t r = 0;
    const size_t lu_words = n_lu ? 8 + 4 * n_lu : 0;
    if (n_words < 4 + 16 + 8 * n_cached + lu_words + 4 * n_open + 9 * (size_t)n_layers + 4 * n_fin + 1) return rejected(where, __LINE__);
    if (proof[0] != PROOF_MAGIC + (n_lu ? 1u : 0u) + (n_prep ? 2u : 0u) + (n_cached ? 4u : 0u) || proof[1] != n_airs || proof[2] != hmax ||
        proof[3] != n_layers)
        return rejected(where, __LINE__);
    r = 4;
    const uint32_t* root_main = proof + r;
    r += 8;
    const uint32_t* roots_cached = proof + r;  // one per AIR with a cached partition, AIR order
    r += 8 * n_cached;
    const uint32_t *root_perm = nullptr, *exposed_c = nullptr;
    if (n_lu) {
        root_perm = proof + r;
        exposed_c = proof + r + 8;
        r += lu_words;
    }
    const uint32_t* root_quot = proof + r;
    r += 8;
    const uint32_t* opened_c = proof + r;
    r += 4 * n_open;
    const uint32_t* fri_hdr = proof + r;
    r += 9 * (size_t)n_layers;
    const uint32_t* fin_c = proof + r;  // 2^lfp coefficients of the final polynomial
    r += 4 * n_fin;
    const uint32_t qpow = proof[r++];
    std::vector<Ext> opened(n_open);
    for (size_t i = 0; i < n_open; i++) opened[i] = ext_from_canon(opened_c + 4 * i);

    HostChallenger ch;
    {
        uint32_t hdr[7] = {PROTO_TAG, (uint32_t)n_airs, prm->log_blowup, prm->log_final_poly_len, prm->num_queries,
                           prm->commit_pow_bits, prm->query_pow_bits};
        ch.observe_canon(hdr, 7);
        for (size_t a = 0; a < n_airs; a++) {
            std::vector<uint32_t> pm(airs[a].program_len);
            for (size_t i = 0; i < pm.size(); i++) {
                if (airs[a].program[i] >= P) return ZKHIP_ERR_INVALID;
                pm[i] = to_monty(airs[a].program[i]);
            }
            uint32_t dg[8];
            p2_hash_slice(pm.data(), pm.size(), dg);
            uint32_t meta[3] = {airs[a].log_height, (uint32_t)airs[a].width, (uint32_t)airs[a].n_pvs};
            ch.observe_canon(meta, 3);
            for (int i = 0; i < 8; i++) ch.observe(dg[i]);
            if (pg[a].prep_width) ch.observe_canon(airs[a].prep_commit, 8);
            for (size_t i = 0; i < airs[a].n_pvs; i++) {
                if (pvs[a][i] >= P) return ZKHIP_ERR_INVALID;
                ch.observe(to_monty(pvs[a][i]));
            }
        }
    }
    ch.observe_canon(roots_cached, 8 * n_cached);  // main-trace commitments in the reference's order: cached..., common
    ch.observe_canon(root_main, 8);
    // LogUp phase: interaction challenges, permutation commitment, exposed sums (must cancel over all AIRs)
    uint32_t chal[N_CHAL] = {};
    if (n_lu) {
        const Ext gamma = ch.sample_ext(), beta = ch.sample_ext();
        Ext cur = beta;
        for (int k = 0; k < 4; k++) chal[k] = gamma.c[k];
        for (unsigned i = 1; i <= LOGUP_MAX_FIELDS; i++) {
            for (int k = 0; k < 4; k++) chal[4 * i + k] = cur.c[k];
            cur = ext_mul(cur, beta);
        }
        ch.observe_canon(root_perm, 8);
        ch.observe_canon(exposed_c, 4 * n_lu);
        if (zkhip_logup_exposed_check(exposed_c, n_lu) != ZKHIP_OK) return rejected(where, __LINE__);
    }
    const Ext alpha = ch.sample_ext();
    ch.observe_canon(root_quot, 8);
    const Ext zeta = ch.sample_ext();
    ch.observe_canon(opened_c, 4 * n_open);
    const Ext alpha_f = ch.sample_ext();
    const uint32_t gen = to_monty(FIELD_GEN_CANON);

    // ---- constraints at zeta ----
    size_t k_lu = 0, k_prep = 0, k_cached = 0;
    for (size_t a = 0; a < n_airs; a++) {
        const unsigned lh = airs[a].log_height, h = lh + b;
        const size_t W = airs[a].width, CW = pg[a].cached_width;
        // selectors of H at zeta
        Ext zn = zeta;
        for (unsigned k = 0; k < lh; k++) zn = ext_mul(zn, zn);
        Ext zh = zn;
        zh.c[0] = msub(zh.c[0], MONTY_ONE);
        Ext d1 = zeta;
        d1.c[0] = msub(d1.c[0], MONTY_ONE);
        const Ext is_first = ext_mul(zh, ext_inv(d1));
        Ext is_trans = zeta;
        is_trans.c[0] = msub(is_trans.c[0], minv(two_adic_generator(lh)));
        const Ext is_last = ext_mul(zh, ext_inv(is_trans));
        const Ext inv_zh = ext_inv(zh);
        std::vector<Ext> vals(pg[a].n_nodes);
        // the AIR's main row = the cached partition's columns, then the common columns
        std::vector<Ext> mrow(2 * W);
        if (CW) {
            const Ext* co = &opened[cm[cm_cached0 + k_cached++].open_off];
            for (size_t k = 0; k < CW; k++) mrow[k] = co[k], mrow[W + k] = co[CW + k];
        }
        for (size_t k = 0; k < W - CW; k++) mrow[CW + k] = opened[cm[a].open_off + k], mrow[W + CW + k] = opened[cm[a].open_off + (W - CW) + k];
        const Ext* local = mrow.data();
        const Ext* next = local + W;
        const Ext *plocal = nullptr, *pnext = nullptr;
        const uint32_t* expo = nullptr;
        const Ext *qlocal = nullptr, *qnext = nullptr;
        if (pg[a].prep_width) {
            qlocal = &opened[cm[cm_prep0 + k_prep].open_off];
            qnext = qlocal + pg[a].prep_width;
            k_prep++;
        }
        if (!pg[a].ints.empty()) {
            plocal = &opened[cm[cm_perm0 + k_lu].open_off];
            pnext = plocal + pg[a].perm_width();
            expo = exposed_c + 4 * k_lu;
            k_lu++;
        }
        for (uint32_t i = 0; i < pg[a].n_nodes; i++) {
            uint32_t op = pg[a].nodes[3 * i], x = pg[a].nodes[3 * i + 1], y = pg[a].nodes[3 * i + 2];
            switch (op) {
                case A_VAR: vals[i] = y ? next[x] : local[x]; break;
                case A_PUB: vals[i] = ext_from_base(to_monty(pvs[a][x])); break;
                case A_CONST: vals[i] = ext_from_base(to_monty(x)); break;
                case A_FIRST: vals[i] = is_first; break;
                case A_LAST: vals[i] = is_last; break;
                case A_TRANS: vals[i] = is_trans; break;
                case A_ADD: vals[i] = ext_add(vals[x], vals[y]); break;
                case A_SUB: vals[i] = ext_sub(vals[x], vals[y]); break;
                case A_MUL: vals[i] = ext_mul(vals[x], vals[y]); break;
                case A_NEG: vals[i] = ext_neg(vals[x]); break;
                case A_PERM: vals[i] = y ? pnext[x] : plocal[x]; break;
                case A_CHAL: vals[i] = ext_from_base(chal[x]); break;
                case A_PREP: vals[i] = y ? qnext[x] : qlocal[x]; break;
                default: vals[i] = ext_from_base(to_monty(expo[x])); break;
            }
        }
        Ext acc = ext_zero();
        for (uint32_t k = 0; k < pg[a].n_cons; k++) acc = ext_add(ext_mul(acc, alpha), vals[pg[a].cons[k]]);
        const Ext lhs = ext_mul(acc, inv_zh);
        // quotient(zeta) from its chunks
        const uint32_t wM = two_adic_generator(h);
        Ext rhs = ext_zero();
        const unsigned nch = pg[a].qd();
        for (unsigned j = 0; j < nch; j++) {
            const uint32_t sj = mmul(gen, mpow(wM, bitrev32(j, b)));
            Ext zps = ext_one();
            for (unsigned k = 0; k < nch; k++) {
                if (k == j) continue;
                const uint32_t sk = mmul(gen, mpow(wM, bitrev32(k, b)));
                Ext t = ext_mul_base(zeta, minv(sk));
                for (unsigned q = 0; q < lh; q++) t = ext_mul(t, t);
                t.c[0] = msub(t.c[0], MONTY_ONE);
                uint32_t den = msub(mpow(mmul(sj, minv(sk)), (uint64_t)1 << lh), MONTY_ONE);
                zps = ext_mul(zps, ext_mul_base(t, minv(den)));
            }
            const Ext* chunk = &opened[cm[cm_quot0 + qoff[a] + j].open_off];
            Ext v = ext_zero();
            for (int k = 0; k < 4; k++) {
                Ext e = ext_zero();
                e.c[k] = MONTY_ONE;
                v = ext_add(v, ext_mul(e, chunk[k]));
            }
            rhs = ext_add(rhs, ext_mul(v, zps));
        }
        if (!ext_eq(lhs, rhs)) return rejected(where, __LINE__);
    }

    // ---- FRI transcript ----
    std::vector<Ext> betas(n_layers);
    std::vector<uint32_t> froots_m(8 * (size_t)n_layers);
    for (unsigned l = 0; l < n_layers; l++) {
        ch.observe_canon(fri_hdr + 9 * l, 8);
        for (int k = 0; k < 8; k++) froots_m[8 * l + k] = to_monty(fri_hdr[9 * l + k]);
        if (!ch.check_witness(prm->commit_pow_bits, fri_hdr[9 * l + 8])) return rejected(where, __LINE__);
        betas[l] = ch.sample_ext();
    }
    ch.observe_canon(fin_c, 4 * n_fin);
    if (!ch.check_witness(prm->query_pow_bits, qpow)) return rejected(where, __LINE__);
    std::vector<Ext> fin(n_fin);
    for (size_t j = 0; j < n_fin; j++) fin[j] = ext_from_canon(fin_c + 4 * j);

    // the input batches (one commitment each) as ranges of `cm`
    struct Batch {
        size_t first, n;
        uint32_t root_m[8];
        std::vector<unsigned> lhs;
        std::vector<size_t> ws;
        size_t tw = 0;
        unsigned bh = 0;
    };
    std::vector<Batch> batches;
    auto add_batch = [&](size_t first, size_t n, const uint32_t* root_c) {
        Batch bt;
        bt.first = first, bt.n = n;
        for (int k = 0; k < 8; k++) bt.root_m[k] = to_monty(root_c[k]);
        for (size_t m = first; m < first + n; m++) {
            bt.lhs.push_back(cm[m].h);
            bt.ws.push_back(cm[m].width);
            bt.tw += cm[m].width;
            bt.bh = std::max(bt.bh, cm[m].h);
        }
        batches.push_back(bt);
    };
    add_batch(0, n_airs, root_main);
    {
        size_t k = 0;
        for (size_t a = 0; a < n_airs; a++)
            if (pg[a].cached_width) add_batch(cm_cached0 + k, 1, roots_cached + 8 * k), k++;
    }
    {
        size_t k = 0;
        for (size_t a = 0; a < n_airs; a++)
            if (pg[a].prep_width) add_batch(cm_prep0 + k++, 1, airs[a].prep_commit);  // each preprocessed trace has its own tree
    }
    if (n_lu) add_batch(cm_perm0, n_lu, root_perm);
    add_batch(cm_quot0, qoff[n_airs], root_quot);
    std::vector<Ext> roq(hmax + 1);
    std::vector<char> has(hmax + 1);
    std::vector<uint64_t> num_reduced(hmax + 1);
    std::vector<const uint32_t*> rows_of(batches.size());
    for (unsigned qn = 0; qn < prm->num_queries; qn++) {
        const size_t idx = ch.sample_bits(hmax);
        for (size_t bi = 0; bi < batches.size(); bi++) {
            const Batch& bt = batches[bi];
            const size_t n_op = bt.tw + 8 * (size_t)bt.bh;
            if (r + n_op > n_words) return rejected(where, __LINE__);
            rows_of[bi] = proof + r;
            r += n_op;
            if (!verify_opening(bt.root_m, bt.lhs, bt.ws, idx >> (hmax - bt.bh), rows_of[bi])) return rejected(where, __LINE__);
        }
        std::fill(roq.begin(), roq.end(), ext_zero());
        std::fill(has.begin(), has.end(), 0);
        std::fill(num_reduced.begin(), num_reduced.end(), 0);
        size_t oi = 0;
        for (size_t bi = 0; bi < batches.size(); bi++) {
            const uint32_t* rows = rows_of[bi];
            for (size_t mi = 0; mi < batches[bi].n; mi++) {
                const CMat& M = cm[batches[bi].first + mi];
                const unsigned h = M.h;
                const size_t W = M.width;
                const size_t ih = idx >> (hmax - h);
                const uint32_t x = mmul(gen, mpow(two_adic_generator(h), bitrev32((uint32_t)ih, h)));
                has[h] = 1;
                std::vector<Ext> apow(W);
                Ext rrow = ext_zero(), cur = ext_one();
                for (size_t k = 0; k < W; k++) {
                    apow[k] = cur;
                    rrow = ext_add(rrow, ext_mul_base(cur, to_monty(rows[k])));
                    cur = ext_mul(cur, alpha_f);
                }
                for (unsigned pt = 0; pt < M.n_pts; pt++) {
                    Ext z = pt == 0 ? zeta : ext_mul_base(zeta, two_adic_generator(M.lh));
                    Ext ry = ext_zero();
                    for (size_t k = 0; k < W; k++) ry = ext_add(ry, ext_mul(apow[k], opened[oi + k]));
                    Ext d = z;
                    d.c[0] = msub(d.c[0], x);
                    Ext u = ext_mul(ext_mul(ext_sub(ry, rrow), ext_inv(d)), ext_pow(alpha_f, num_reduced[h]));
                    roq[h] = ext_add(roq[h], u);
                    num_reduced[h] += W;
                    oi += W;
                }
                rows += W;
            }
        }
        Ext eval = roq[hmax];
        for (unsigned l = 0; l < n_layers; l++) {
            const unsigned log_len = hmax - l;
            const size_t il = idx >> l, n_path = 8 * (size_t)(log_len - 1);
            if (r + 4 + n_path > n_words) return rejected(where, __LINE__);
            const uint32_t *sib = proof + r, *path = proof + r + 4;
            r += 4 + n_path;
            std::vector<uint32_t> opening(8 + n_path);
            for (int k = 0; k < 4; k++) {
                opening[4 * (il & 1) + k] = from_monty(eval.c[k]);
                opening[4 * ((il & 1) ^ 1) + k] = sib[k];
            }
            memcpy(opening.data() + 8, path, n_path * 4);
            if (!verify_opening(&froots_m[8 * l], {log_len - 1}, {8}, il >> 1, opening.data())) return rejected(where, __LINE__);
            const Ext e0 = ext_from_canon(opening.data()), e1 = ext_from_canon(opening.data() + 4);
            eval = fold_row(il >> 1, log_len - 1, betas[l], e0, e1);
            if (has[log_len - 1]) eval = ext_add(eval, ext_mul(ext_mul(betas[l], betas[l]), roq[log_len - 1]));
        }
        // the folded value must be the final polynomial at this query's point of the last domain (Horner)
        Ext want = fin[n_fin - 1];
        if (lfp) {
            const uint32_t xf = mpow(two_adic_generator(b + lfp), bitrev32((uint32_t)(idx >> n_layers), b + lfp));
            for (size_t j = n_fin - 1; j-- > 0;) want = ext_add(ext_mul_base(want, xf), fin[j]);
        }
        if (!ext_eq(eval, want)) return rejected(where, __LINE__);
    }
    if (r != n_words) return rejected(where, __LINE__);
    return ZKHIP_OK;
}

extern "C" int zkhip_verify(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, const uint32_t* const* pvs, const uint8_t* proof_bytes,
                            size_t len) {
    return verify_where(prm, airs, n_airs, pvs, proof_bytes, len, nullptr);
}
extern "C" int zkhip_verify_where(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, const uint32_t* const* pvs,
                                  const uint8_t* proof_bytes, size_t len, int* where) {
    if (where) *where = 0;
    return verify_where(prm, airs, n_airs, pvs, proof_bytes, len, where);
}
