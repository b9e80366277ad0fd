// codec.hip -- C ABI over include/zkhip_codec.hpp (host only): reads / writes the reference's stored-proof container
// (OpenVM-v1 `Proof<SC>`, bincode) and converts between it and this backend's static proof layout.
//
// Replaces, for v1-format proofs, what `Proof::<SC>::decode_from_bytes` (crates/verifier/src/verifier.rs:62) and
// `encode_to_vec` (crates/prover/src/prover/mod.rs:375-378) are to the reference.  Field order follows serde's
// derive order of openvm-stark-backend 1.x `Proof` (pinned by the eight stored proofs, which decode to the last byte
// and re-encode identically: tests/test_codec_v1_cpu.py).
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_codec.hpp"
#include "air_compile.hpp"

using namespace zk;
namespace cd = zkhip_codec;

namespace {

unsigned log2_u64(uint64_t v) {
    unsigned l = 0;
    while (v >>= 1) l++;
    return l;
}

void summarize(const std::vector<cd::ProofV1>& ps, zkhip_v1_summary* s) {
    memset(s, 0, sizeof *s);
    s->n_proofs = ps.size();
    if (ps.empty()) return;
    const cd::ProofV1& p = ps[0];
    s->n_airs = p.per_air.size();
    s->n_queries = p.fri.query_proofs.size();
    s->n_fri_layers = p.fri.commit_phase_commits.size();
    s->n_final_poly = p.fri.final_poly.size();
    s->n_main_commits = p.main_trace.size();
    s->n_after_challenge_commits = p.after_challenge.size();
    s->n_preprocessed = p.values.preprocessed.size();
    s->has_logup_pow = p.has_logup_pow ? 1 : 0;
    unsigned maxdeg = 0;
    for (size_t a = 0; a < p.per_air.size(); a++) {
        const unsigned l = log2_u64(p.per_air[a].degree);
        if (a < ZKHIP_V1_MAX_AIRS) s->log_degree[a] = l;
        maxdeg = std::max(maxdeg, l);
    }
    if (!p.fri.query_proofs.empty()) {
        const auto& q = p.fri.query_proofs[0];
        s->n_input_batches = q.input_proof.size();
        size_t h = 0;
        for (const auto& b : q.input_proof) h = std::max(h, b.opening_proof.size());
        s->log_max_height = (unsigned)h;
        s->log_blowup = h >= maxdeg ? (unsigned)(h - maxdeg) : 0;
    }
}

// the committed matrices of a zkhip proof, in ITS opening order: main (every AIR), preprocessed, permutation, quotient chunks
struct Shape {
    std::vector<AirProgram> pg;
    unsigned b = 0, lfp = 0, hmax = 0, n_layers = 0;  // AIR a has pg[a].qd() quotient chunks
    size_t n_airs = 0, n_lu = 0, n_prep = 0, n_cached = 0, n_open = 0, n_fin = 0;
    unsigned main_h = 0, perm_h = 0;
};

int shape_of(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, Shape* s) {
    if (!prm || !airs || n_airs == 0) return ZKHIP_ERR_INVALID;
    s->b = prm->log_blowup, s->lfp = prm->log_final_poly_len;
    if (s->lfp > ZKHIP_MAX_LOG_FINAL_POLY || s->b < 1 || s->b > 4) return ZKHIP_ERR_INVALID;
    s->n_airs = n_airs;
    s->n_fin = (size_t)1 << s->lfp;
    s->pg.resize(n_airs);
    for (size_t a = 0; a < n_airs; a++) {
        if (parse_air(airs[a].program, airs[a].program_len, airs[a].width, &s->pg[a], nullptr) != 0) return ZKHIP_ERR_INVALID;
        if (airs[a].log_height + s->b > 27 || airs[a].log_height < s->lfp) return ZKHIP_ERR_INVALID;
        const unsigned h = airs[a].log_height + s->b;
        s->hmax = std::max(s->hmax, h);
        if (s->pg[a].log_qd() > s->b) return ZKHIP_ERR_CONSTRAINT;
        s->n_open += 2 * airs[a].width + 4 * (size_t)s->pg[a].qd();
        if (s->pg[a].prep_width) s->n_prep++, s->n_open += 2 * s->pg[a].prep_width;
        if (s->pg[a].cached_width) s->n_cached++;
        if (!s->pg[a].ints.empty()) s->n_lu++, s->perm_h = std::max(s->perm_h, h), s->n_open += 2 * s->pg[a].perm_width();
    }
    s->main_h = s->hmax;
    s->n_layers = s->hmax - s->b - s->lfp;
    return ZKHIP_OK;
}

cd::Digest digest_m(const uint32_t* canon) {
    cd::Digest d;
    for (int k = 0; k < 8; k++) d[k] = to_monty(canon[k]);
    return d;
}
cd::ExtWords ext_m(const uint32_t* canon) {
    cd::ExtWords e;
    for (int k = 0; k < 4; k++) e[k] = to_monty(canon[k]);
    return e;
}
void put_canon(std::vector<uint32_t>& out, const uint32_t* monty, size_t n) {
    for (size_t i = 0; i < n; i++) out.push_back(from_monty(monty[i]));
}

int emit(const std::vector<uint8_t>& enc, uint8_t* out, size_t cap, size_t* out_len) {
    if (out_len) *out_len = enc.size();
    if (!out || cap < enc.size()) return ZKHIP_ERR_SMALL_BUFFER;
    if (!enc.empty()) memcpy(out, enc.data(), enc.size());
    return ZKHIP_OK;
}

}  // namespace

extern "C" int zkhip_proof_decode_v1(const uint8_t* bytes, size_t len, int kind, zkhip_v1_summary* out) {
    if (!bytes || !out) return ZKHIP_ERR_INVALID;
    try {
        std::vector<cd::ProofV1> ps;
        if (kind == ZKHIP_V1_VEC) ps = cd::decode_proofs(bytes, len);
        else if (kind == ZKHIP_V1_SINGLE) ps.push_back(cd::decode_proof(bytes, len));
        else return ZKHIP_ERR_INVALID;
        for (const auto& p : ps)
            if (!cd::well_formed(p, P)) return ZKHIP_ERR_VERIFY;
        summarize(ps, out);
        return ZKHIP_OK;
    } catch (const cd::DecodeError&) {
        return ZKHIP_ERR_VERIFY;
    } catch (const std::bad_alloc&) {
        return ZKHIP_ERR_NOMEM;
    }
}

extern "C" int zkhip_proof_reencode_v1(const uint8_t* bytes, size_t len, int kind, uint8_t* out, size_t cap, size_t* out_len) {
    if (!bytes) return ZKHIP_ERR_INVALID;
    try {
        if (kind == ZKHIP_V1_VEC) return emit(cd::encode_proofs(cd::decode_proofs(bytes, len)), out, cap, out_len);
        if (kind == ZKHIP_V1_SINGLE) return emit(cd::encode_proof(cd::decode_proof(bytes, len)), out, cap, out_len);
        return ZKHIP_ERR_INVALID;
    } catch (const cd::DecodeError&) {
        return ZKHIP_ERR_VERIFY;
    } catch (const std::bad_alloc&) {
        return ZKHIP_ERR_NOMEM;
    }
}

// zkhip proof words (canonical) -> bincode(Proof<SC>) (Montgomery words).  The v1 container has no field for per-layer
// commit-phase proof-of-work witnesses (a v2-era parameter, openvm.toml:5): they must be absent (commit_pow_bits == 0).
extern "C" int zkhip_proof_to_v1(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, const uint32_t* const* pvs,
                                 const uint8_t* proof_bytes, size_t len, uint8_t* out, size_t cap, size_t* out_len) {
    Shape s;
    int rc = shape_of(prm, airs, n_airs, &s);
    if (rc != ZKHIP_OK) return rc;
    if (prm->commit_pow_bits != 0 || !proof_bytes || (len & 3)) return ZKHIP_ERR_INVALID;
    zkhip_proof_layout lay;
    if ((rc = zkhip_proof_layout_of(prm, airs, n_airs, &lay)) != ZKHIP_OK) return rc;
    if (len != 4 * lay.n_words) return ZKHIP_ERR_INVALID;
    std::vector<uint32_t> w(lay.n_words);
    memcpy(w.data(), proof_bytes, len);
    for (uint32_t v : w)
        if (v >= P) return ZKHIP_ERR_VERIFY;
    if (w[0] != PROOF_MAGIC + (s.n_lu ? 1u : 0u) + (s.n_prep ? 2u : 0u) + (s.n_cached ? 4u : 0u) || w[1] != n_airs || w[2] != s.hmax || w[3] != s.n_layers)
        return ZKHIP_ERR_VERIFY;
    try {
        cd::ProofV1 p;
        // main-trace commitments in the reference's order: cached partitions (AIR order), then the common main
        for (size_t k = 0; k < s.n_cached; k++) p.main_trace.push_back(digest_m(&w[lay.roots_cached + 8 * k]));
        p.main_trace.push_back(digest_m(&w[lay.root_main]));
        if (s.n_lu) p.after_challenge.push_back(digest_m(&w[lay.root_perm]));
        p.quotient = digest_m(&w[lay.root_quot]);
        // opened values: zkhip order = main (all AIRs), preprocessed, permutation, quotient chunks
        size_t o = lay.opened;
        auto take = [&](size_t n) {
            std::vector<cd::ExtWords> v(n);
            for (size_t i = 0; i < n; i++, o += 4) v[i] = ext_m(&w[o]);
            return v;
        };
        p.values.main.resize(s.n_cached + 1);
        for (size_t a = 0; a < n_airs; a++) {  // zkhip order: the common parts of every AIR first ...
            cd::AdjacentOpenedValues adj;
            const size_t wc = airs[a].width - s.pg[a].cached_width;
            adj.local = take(wc), adj.next = take(wc);
            p.values.main[s.n_cached].push_back(std::move(adj));
        }
        {
            size_t k = 0;  // ... then the cached partitions, one commitment each
            for (size_t a = 0; a < n_airs; a++)
                if (s.pg[a].cached_width) {
                    cd::AdjacentOpenedValues adj;
                    adj.local = take(s.pg[a].cached_width), adj.next = take(s.pg[a].cached_width);
                    p.values.main[k++].push_back(std::move(adj));
                }
        }
        for (size_t a = 0; a < n_airs; a++)
            if (s.pg[a].prep_width) {
                cd::AdjacentOpenedValues adj;
                adj.local = take(s.pg[a].prep_width), adj.next = take(s.pg[a].prep_width);
                p.values.preprocessed.push_back(std::move(adj));
            }
        if (s.n_lu) p.values.after_challenge.resize(1);
        for (size_t a = 0; a < n_airs; a++)
            if (!s.pg[a].ints.empty()) {
                cd::AdjacentOpenedValues adj;
                adj.local = take(s.pg[a].perm_width()), adj.next = take(s.pg[a].perm_width());
                p.values.after_challenge[0].push_back(std::move(adj));
            }
        p.values.quotient.resize(n_airs);
        for (size_t a = 0; a < n_airs; a++)
            for (unsigned j = 0; j < s.pg[a].qd(); j++) p.values.quotient[a].push_back(take(4));
        // FRI
        for (size_t l = 0; l < s.n_layers; l++) {
            p.fri.commit_phase_commits.push_back(digest_m(&w[lay.fri_layers + 9 * l]));
            if (w[lay.fri_layers + 9 * l + 8] != 0) return ZKHIP_ERR_INVALID;  // a commit-phase witness cannot be carried
        }
        for (size_t j = 0; j < s.n_fin; j++) p.fri.final_poly.push_back(ext_m(&w[lay.final_poly + 4 * j]));
        p.fri.pow_witness = to_monty(w[lay.query_pow]);
        p.fri.query_proofs.resize(lay.n_queries);
        for (size_t qi = 0; qi < lay.n_queries; qi++) {
            size_t r = lay.queries + qi * lay.query_words;
            auto rows = [&](size_t width) {
                std::vector<uint32_t> v(width);
                for (size_t k = 0; k < width; k++) v[k] = to_monty(w[r++]);
                return v;
            };
            auto path = [&](unsigned h) {
                std::vector<cd::Digest> v(h);
                for (unsigned k = 0; k < h; k++, r += 8) v[k] = digest_m(&w[r]);
                return v;
            };
            cd::BatchOpening bmain, bperm, bquot;
            std::vector<cd::BatchOpening> bprep, bcached;
            for (size_t a = 0; a < n_airs; a++) bmain.opened_values.push_back(rows(airs[a].width - s.pg[a].cached_width));
            bmain.opening_proof = path(s.main_h);
            for (size_t a = 0; a < n_airs; a++)
                if (s.pg[a].cached_width) {
                    cd::BatchOpening bc;
                    bc.opened_values.push_back(rows(s.pg[a].cached_width));
                    bc.opening_proof = path(airs[a].log_height + s.b);
                    bcached.push_back(std::move(bc));
                }
            for (size_t a = 0; a < n_airs; a++)
                if (s.pg[a].prep_width) {
                    cd::BatchOpening bp;
                    bp.opened_values.push_back(rows(s.pg[a].prep_width));
                    bp.opening_proof = path(airs[a].log_height + s.b);
                    bprep.push_back(std::move(bp));
                }
            if (s.n_lu) {
                for (size_t a = 0; a < n_airs; a++)
                    if (!s.pg[a].ints.empty()) bperm.opened_values.push_back(rows(s.pg[a].perm_width()));
                bperm.opening_proof = path(s.perm_h);
            }
            for (size_t a = 0; a < n_airs; a++)
                for (unsigned j = 0; j < s.pg[a].qd(); j++) bquot.opened_values.push_back(rows(4));
            bquot.opening_proof = path(s.main_h);
            auto& q = p.fri.query_proofs[qi];
            for (auto& bp : bprep) q.input_proof.push_back(std::move(bp));  // v1 order: preprocessed, main (cached..., common), after-challenge, quotient
            for (auto& bc : bcached) q.input_proof.push_back(std::move(bc));
            q.input_proof.push_back(std::move(bmain));
            if (s.n_lu) q.input_proof.push_back(std::move(bperm));
            q.input_proof.push_back(std::move(bquot));
            for (unsigned l = 0; l < s.n_layers; l++) {
                cd::CommitPhaseStep st;
                st.sibling_value = ext_m(&w[r]);
                r += 4;
                st.opening_proof = path(s.hmax - l - 1);
                q.commit_phase_openings.push_back(std::move(st));
            }
            if (r != lay.queries + (qi + 1) * lay.query_words) return ZKHIP_ERR_VERIFY;
        }
        size_t k_lu = 0;
        for (size_t a = 0; a < n_airs; a++) {
            cd::AirProofData d;
            d.air_id = a;
            d.degree = (uint64_t)1 << airs[a].log_height;
            if (!s.pg[a].ints.empty()) d.exposed_values_after_challenge.push_back({ext_m(&w[lay.exposed + 4 * k_lu++])});
            for (size_t i = 0; i < airs[a].n_pvs; i++) {
                if (!pvs || !pvs[a] || pvs[a][i] >= P) return ZKHIP_ERR_INVALID;
                d.public_values.push_back(to_monty(pvs[a][i]));
            }
            p.per_air.push_back(std::move(d));
        }
        p.has_logup_pow = s.n_lu != 0;  // Some(witness) exactly when there is an after-challenge phase; no LogUp grinding here
        p.logup_pow_witness = 0;
        return emit(cd::encode_proof(p), out, cap, out_len);
    } catch (const std::bad_alloc&) {
        return ZKHIP_ERR_NOMEM;
    }
}

// bincode(Proof<SC>) -> zkhip proof words; the shapes must be those of (params, airs).  pvs_out (optional): per AIR a
// buffer of n_pvs canonical words.
extern "C" int zkhip_proof_from_v1(const zkhip_params* prm, const zkhip_air* airs, size_t n_airs, const uint8_t* v1, size_t v1_len,
                                   uint8_t* out, size_t cap, size_t* out_len, uint32_t* const* pvs_out) {
    Shape s;
    int rc = shape_of(prm, airs, n_airs, &s);
    if (rc != ZKHIP_OK) return rc;
    if (!v1) return ZKHIP_ERR_INVALID;
    zkhip_proof_layout lay;
    if ((rc = zkhip_proof_layout_of(prm, airs, n_airs, &lay)) != ZKHIP_OK) return rc;
    try {
        const cd::ProofV1 p = cd::decode_proof(v1, v1_len);
        if (!cd::well_formed(p, P)) return ZKHIP_ERR_VERIFY;
        for (size_t a = 0; a < p.per_air.size(); a++)
            if (p.per_air[a].air_id != a) return ZKHIP_ERR_VERIFY;   // this backend's proofs carry every AIR of the key, in key order
        // shape checks against the key
        if (p.main_trace.size() != s.n_cached + 1 || p.after_challenge.size() != (s.n_lu ? 1u : 0u) || p.per_air.size() != n_airs ||
            p.fri.commit_phase_commits.size() != s.n_layers || p.fri.final_poly.size() != s.n_fin ||
            p.fri.query_proofs.size() != lay.n_queries || p.values.main.size() != s.n_cached + 1 || p.values.main[s.n_cached].size() != n_airs ||
            p.values.preprocessed.size() != s.n_prep || p.values.after_challenge.size() != (s.n_lu ? 1u : 0u) ||
            p.values.quotient.size() != n_airs)
            return ZKHIP_ERR_VERIFY;
        std::vector<uint32_t> w;
        w.reserve(lay.n_words);
        w.push_back(PROOF_MAGIC + (s.n_lu ? 1u : 0u) + (s.n_prep ? 2u : 0u) + (s.n_cached ? 4u : 0u));
        w.push_back((uint32_t)n_airs), w.push_back(s.hmax), w.push_back(s.n_layers);
        put_canon(w, p.main_trace[s.n_cached].data(), 8);
        for (size_t k = 0; k < s.n_cached; k++) put_canon(w, p.main_trace[k].data(), 8);
        if (s.n_lu) {
            put_canon(w, p.after_challenge[0].data(), 8);
            for (size_t a = 0; a < n_airs; a++)
                if (!s.pg[a].ints.empty()) {
                    const auto& ex = p.per_air[a].exposed_values_after_challenge;
                    if (ex.size() != 1 || ex[0].size() != 1) return ZKHIP_ERR_VERIFY;
                    put_canon(w, ex[0][0].data(), 4);
                }
        }
        put_canon(w, p.quotient.data(), 8);
        auto put_adj = [&](const cd::AdjacentOpenedValues& adj, size_t width) {
            if (adj.local.size() != width || adj.next.size() != width) return false;
            for (const auto& e : adj.local) put_canon(w, e.data(), 4);
            for (const auto& e : adj.next) put_canon(w, e.data(), 4);
            return true;
        };
        for (size_t a = 0; a < n_airs; a++)
            if (!put_adj(p.values.main[s.n_cached][a], airs[a].width - s.pg[a].cached_width)) return ZKHIP_ERR_VERIFY;
        {
            size_t k = 0;
            for (size_t a = 0; a < n_airs; a++)
                if (s.pg[a].cached_width) {
                    if (p.values.main[k].size() != 1 || !put_adj(p.values.main[k][0], s.pg[a].cached_width)) return ZKHIP_ERR_VERIFY;
                    k++;
                }
            k = 0;
            for (size_t a = 0; a < n_airs; a++)
                if (s.pg[a].prep_width && !put_adj(p.values.preprocessed[k++], s.pg[a].prep_width)) return ZKHIP_ERR_VERIFY;
            k = 0;
            if (s.n_lu && p.values.after_challenge[0].size() != s.n_lu) return ZKHIP_ERR_VERIFY;
            for (size_t a = 0; a < n_airs; a++)
                if (!s.pg[a].ints.empty() && !put_adj(p.values.after_challenge[0][k++], s.pg[a].perm_width())) return ZKHIP_ERR_VERIFY;
        }
        for (size_t a = 0; a < n_airs; a++) {
            if (p.values.quotient[a].size() != s.pg[a].qd()) return ZKHIP_ERR_VERIFY;
            for (const auto& c : p.values.quotient[a]) {
                if (c.size() != 4) return ZKHIP_ERR_VERIFY;
                for (const auto& e : c) put_canon(w, e.data(), 4);
            }
        }
        for (size_t l = 0; l < s.n_layers; l++) {
            put_canon(w, p.fri.commit_phase_commits[l].data(), 8);
            w.push_back(0);  // no commit-phase witness in the v1 container
        }
        for (const auto& e : p.fri.final_poly) put_canon(w, e.data(), 4);
        w.push_back(from_monty(p.fri.pow_witness));
        const size_t n_batches = s.n_prep + s.n_cached + 2 + (s.n_lu ? 1 : 0);
        for (const auto& q : p.fri.query_proofs) {
            if (q.input_proof.size() != n_batches || q.commit_phase_openings.size() != s.n_layers) return ZKHIP_ERR_VERIFY;
            auto put_batch = [&](const cd::BatchOpening& b, const std::vector<size_t>& widths, unsigned h) {
                if (b.opened_values.size() != widths.size() || b.opening_proof.size() != h) return false;
                for (size_t m = 0; m < widths.size(); m++) {
                    if (b.opened_values[m].size() != widths[m]) return false;
                    put_canon(w, b.opened_values[m].data(), widths[m]);
                }
                for (const auto& d : b.opening_proof) put_canon(w, d.data(), 8);
                return true;
            };
            std::vector<size_t> wmain, wperm, wquot;
            for (size_t a = 0; a < n_airs; a++) {
                wmain.push_back(airs[a].width - s.pg[a].cached_width);
                if (!s.pg[a].ints.empty()) wperm.push_back(s.pg[a].perm_width());
                for (unsigned j = 0; j < s.pg[a].qd(); j++) wquot.push_back(4);
            }
            if (!put_batch(q.input_proof[s.n_prep + s.n_cached], wmain, s.main_h)) return ZKHIP_ERR_VERIFY;
            size_t k = 0;
            for (size_t a = 0; a < n_airs; a++)
                if (s.pg[a].cached_width && !put_batch(q.input_proof[s.n_prep + k++], {s.pg[a].cached_width}, airs[a].log_height + s.b)) return ZKHIP_ERR_VERIFY;
            k = 0;
            for (size_t a = 0; a < n_airs; a++)
                if (s.pg[a].prep_width && !put_batch(q.input_proof[k++], {s.pg[a].prep_width}, airs[a].log_height + s.b)) return ZKHIP_ERR_VERIFY;
            if (s.n_lu && !put_batch(q.input_proof[s.n_prep + s.n_cached + 1], wperm, s.perm_h)) return ZKHIP_ERR_VERIFY;
            if (!put_batch(q.input_proof[n_batches - 1], wquot, s.main_h)) return ZKHIP_ERR_VERIFY;
            for (unsigned l = 0; l < s.n_layers; l++) {
                const auto& st = q.commit_phase_openings[l];
                if (st.opening_proof.size() != s.hmax - l - 1) return ZKHIP_ERR_VERIFY;
                put_canon(w, st.sibling_value.data(), 4);
                for (const auto& d : st.opening_proof) put_canon(w, d.data(), 8);
            }
        }
        if (w.size() != lay.n_words) return ZKHIP_ERR_VERIFY;
        for (size_t a = 0; a < n_airs; a++) {
            if (p.per_air[a].public_values.size() != airs[a].n_pvs || p.per_air[a].degree != ((uint64_t)1 << airs[a].log_height)) return ZKHIP_ERR_VERIFY;
            if (pvs_out && pvs_out[a])
                for (size_t i = 0; i < airs[a].n_pvs; i++) pvs_out[a][i] = from_monty(p.per_air[a].public_values[i]);
        }
        if (out_len) *out_len = 4 * w.size();
        if (!out || cap < 4 * w.size()) return ZKHIP_ERR_SMALL_BUFFER;
        memcpy(out, w.data(), 4 * w.size());
        return ZKHIP_OK;
    } catch (const cd::DecodeError&) {
        return ZKHIP_ERR_VERIFY;
    } catch (const std::bad_alloc&) {
        return ZKHIP_ERR_NOMEM;
    }
}
