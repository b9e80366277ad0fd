// zkhip_codec.hpp -- the reference's stored-proof container (OpenVM-v1 `Proof<SC>`, bincode) in C++.
//
// The proofs the reference keeps under crates/verifier/testdata/proofs/ and crates/prover/testdata/ are
// `VmInternalStarkProof { proofs: Vec<Proof<SC>>, public_values: Vec<BabyBear> }` (crates/types/src/proof.rs:69-74),
// each field base64(bincode-v1) (crates/types/src/utils.rs:20-39).  `Proof<SC>` is the proof of the quotient + FRI
// pipeline (openvm-stark-backend 1.x `proof.rs`, SC = BabyBearPoseidon2Config): the same objects zkhip_prove
// produces, in serde field order.  This header reads and writes that container byte-exactly
// (tests/test_codec_v1_cpu.py: decode -> encode of all eight reference files reproduces the input); the C ABI entry
// points built on it are zkhip_proof_decode_v1 / zkhip_proof_reencode_v1 / zkhip_proof_to_v1 / zkhip_proof_from_v1
// (include/zkhip.h).  It replaces what `Proof::<SC>::decode_from_bytes` / `encode_to_vec` are to the reference
// (crates/verifier/src/verifier.rs:62, crates/prover/src/prover/mod.rs:375-378) for v1-format proofs; the v2
// (`Encode`) format needs the un-vendored openvm-stark-backend 2.0 sources.
//
// bincode v1, default options: integers little-endian fixed width, `Vec<T>` = u64 length + items, arrays and structs
// inline, `Option<T>` = u8 tag (0/1) + T.  A field element is the u32 p3's MontyField31 keeps in memory
// (Montgomery form, R = 2^32): words are carried as stored; canonical(v) = v * 2^-32 mod p.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace zkhip_codec {

using Digest = std::array<uint32_t, 8>;
using ExtWords = std::array<uint32_t, 4>;

struct BatchOpening {
    std::vector<std::vector<uint32_t>> opened_values;  // one row per matrix of the commitment
    std::vector<Digest> opening_proof;                 // sibling digests, bottom-up
};
struct CommitPhaseStep {
    ExtWords sibling_value;
    std::vector<Digest> opening_proof;
};
struct QueryProof {
    std::vector<BatchOpening> input_proof;  // preprocessed trees, main commitments, after-challenge, quotient
    std::vector<CommitPhaseStep> commit_phase_openings;
};
struct FriProof {
    std::vector<Digest> commit_phase_commits;
    std::vector<QueryProof> query_proofs;
    std::vector<ExtWords> final_poly;
    uint32_t pow_witness = 0;
};
struct AdjacentOpenedValues {
    std::vector<ExtWords> local, next;
};
struct OpenedValues {
    std::vector<AdjacentOpenedValues> preprocessed;               // one per AIR that has a preprocessed trace
    std::vector<std::vector<AdjacentOpenedValues>> main;          // per main commitment, per matrix
    std::vector<std::vector<AdjacentOpenedValues>> after_challenge;  // per phase, per matrix
    std::vector<std::vector<std::vector<ExtWords>>> quotient;     // per AIR, per chunk, 4 values
};
struct AirProofData {
    uint64_t air_id = 0, degree = 0;
    std::vector<std::vector<ExtWords>> exposed_values_after_challenge;  // per phase
    std::vector<uint32_t> public_values;
};
struct ProofV1 {
    std::vector<Digest> main_trace, after_challenge;
    Digest quotient{};
    FriProof fri;
    OpenedValues values;
    std::vector<AirProofData> per_air;
    bool has_logup_pow = false;  // rap_phase_seq_proof: Option<FriLogUpPartialProof { logup_pow_witness }>
    uint32_t logup_pow_witness = 0;
};

struct DecodeError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Reader {
public:
    Reader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    size_t remaining() const { return n_ - o_; }
    size_t offset() const { return o_; }
    uint8_t u8() {
        need(1);
        return p_[o_++];
    }
    uint32_t u32() {
        need(4);
        uint32_t v;
        memcpy(&v, p_ + o_, 4);
        o_ += 4;
        return v;
    }
    uint64_t u64() {
        need(8);
        uint64_t v;
        memcpy(&v, p_ + o_, 8);
        o_ += 8;
        return v;
    }
    template <size_t N>
    std::array<uint32_t, N> arr() {
        need(4 * N);
        std::array<uint32_t, N> a;
        memcpy(a.data(), p_ + o_, 4 * N);
        o_ += 4 * N;
        return a;
    }
    // length prefix of a Vec whose items take at least `min_item_bytes` each: rejects lengths the input cannot hold
    size_t len(size_t min_item_bytes) {
        uint64_t n = u64();
        if (min_item_bytes && n > remaining() / min_item_bytes) throw DecodeError("vector length exceeds the input");
        return (size_t)n;
    }
    std::vector<uint32_t> words() {
        size_t n = len(4);
        std::vector<uint32_t> v(n);
        if (n) memcpy(v.data(), p_ + o_, 4 * n);
        o_ += 4 * n;
        return v;
    }
    template <size_t N>
    std::vector<std::array<uint32_t, N>> arrs() {
        size_t n = len(4 * N);
        std::vector<std::array<uint32_t, N>> v(n);
        if (n) memcpy(v.data(), p_ + o_, 4 * N * n);
        o_ += 4 * N * n;
        return v;
    }

private:
    void need(size_t k) const {
        if (k > n_ - o_) throw DecodeError("truncated input");
    }
    const uint8_t* p_;
    size_t n_, o_ = 0;
};

class Writer {
public:
    std::vector<uint8_t> out;
    void u8(uint8_t v) { out.push_back(v); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    template <size_t N>
    void arr(const std::array<uint32_t, N>& a) {
        raw(a.data(), 4 * N);
    }
    void words(const std::vector<uint32_t>& v) {
        u64(v.size());
        if (!v.empty()) raw(v.data(), 4 * v.size());
    }
    template <size_t N>
    void arrs(const std::vector<std::array<uint32_t, N>>& v) {
        u64(v.size());
        if (!v.empty()) raw(v.data(), 4 * N * v.size());
    }

private:
    void raw(const void* p, size_t n) {
        const uint8_t* b = static_cast<const uint8_t*>(p);
        out.insert(out.end(), b, b + n);
    }
};

inline AdjacentOpenedValues read_adj(Reader& r) {
    AdjacentOpenedValues a;
    a.local = r.arrs<4>();
    a.next = r.arrs<4>();
    return a;
}
inline void write_adj(Writer& w, const AdjacentOpenedValues& a) {
    w.arrs<4>(a.local);
    w.arrs<4>(a.next);
}

inline ProofV1 read_proof(Reader& r) {
    ProofV1 p;
    p.main_trace = r.arrs<8>();
    p.after_challenge = r.arrs<8>();
    p.quotient = r.arr<8>();
    p.fri.commit_phase_commits = r.arrs<8>();
    p.fri.query_proofs.resize(r.len(16));
    for (auto& q : p.fri.query_proofs) {
        q.input_proof.resize(r.len(16));
        for (auto& b : q.input_proof) {
            b.opened_values.resize(r.len(8));
            for (auto& row : b.opened_values) row = r.words();
            b.opening_proof = r.arrs<8>();
        }
        q.commit_phase_openings.resize(r.len(24));
        for (auto& s : q.commit_phase_openings) {
            s.sibling_value = r.arr<4>();
            s.opening_proof = r.arrs<8>();
        }
    }
    p.fri.final_poly = r.arrs<4>();
    p.fri.pow_witness = r.u32();
    p.values.preprocessed.resize(r.len(16));
    for (auto& a : p.values.preprocessed) a = read_adj(r);
    p.values.main.resize(r.len(8));
    for (auto& c : p.values.main) {
        c.resize(r.len(16));
        for (auto& a : c) a = read_adj(r);
    }
    p.values.after_challenge.resize(r.len(8));
    for (auto& c : p.values.after_challenge) {
        c.resize(r.len(16));
        for (auto& a : c) a = read_adj(r);
    }
    p.values.quotient.resize(r.len(8));
    for (auto& a : p.values.quotient) {
        a.resize(r.len(8));
        for (auto& c : a) c = r.arrs<4>();
    }
    p.per_air.resize(r.len(32));
    for (auto& a : p.per_air) {
        a.air_id = r.u64();
        a.degree = r.u64();
        a.exposed_values_after_challenge.resize(r.len(8));
        for (auto& ph : a.exposed_values_after_challenge) ph = r.arrs<4>();
        a.public_values = r.words();
    }
    const uint8_t tag = r.u8();
    if (tag > 1) throw DecodeError("invalid Option tag");
    p.has_logup_pow = tag == 1;
    if (p.has_logup_pow) p.logup_pow_witness = r.u32();
    return p;
}

inline void write_proof(Writer& w, const ProofV1& p) {
    w.arrs<8>(p.main_trace);
    w.arrs<8>(p.after_challenge);
    w.arr<8>(p.quotient);
    w.arrs<8>(p.fri.commit_phase_commits);
    w.u64(p.fri.query_proofs.size());
    for (const auto& q : p.fri.query_proofs) {
        w.u64(q.input_proof.size());
        for (const auto& b : q.input_proof) {
            w.u64(b.opened_values.size());
            for (const auto& row : b.opened_values) w.words(row);
            w.arrs<8>(b.opening_proof);
        }
        w.u64(q.commit_phase_openings.size());
        for (const auto& s : q.commit_phase_openings) {
            w.arr<4>(s.sibling_value);
            w.arrs<8>(s.opening_proof);
        }
    }
    w.arrs<4>(p.fri.final_poly);
    w.u32(p.fri.pow_witness);
    w.u64(p.values.preprocessed.size());
    for (const auto& a : p.values.preprocessed) write_adj(w, a);
    w.u64(p.values.main.size());
    for (const auto& c : p.values.main) {
        w.u64(c.size());
        for (const auto& a : c) write_adj(w, a);
    }
    w.u64(p.values.after_challenge.size());
    for (const auto& c : p.values.after_challenge) {
        w.u64(c.size());
        for (const auto& a : c) write_adj(w, a);
    }
    w.u64(p.values.quotient.size());
    for (const auto& a : p.values.quotient) {
        w.u64(a.size());
        for (const auto& c : a) w.arrs<4>(c);
    }
    w.u64(p.per_air.size());
    for (const auto& a : p.per_air) {
        w.u64(a.air_id);
        w.u64(a.degree);
        w.u64(a.exposed_values_after_challenge.size());
        for (const auto& ph : a.exposed_values_after_challenge) w.arrs<4>(ph);
        w.words(a.public_values);
    }
    w.u8(p.has_logup_pow ? 1 : 0);
    if (p.has_logup_pow) w.u32(p.logup_pow_witness);
}

// bincode(Proof<SC>)
inline ProofV1 decode_proof(const uint8_t* bytes, size_t len) {
    Reader r(bytes, len);
    ProofV1 p = read_proof(r);
    if (r.remaining()) throw DecodeError("trailing bytes after the proof");
    return p;
}
inline std::vector<uint8_t> encode_proof(const ProofV1& p) {
    Writer w;
    write_proof(w, p);
    return std::move(w.out);
}
// bincode(Vec<Proof<SC>>): the `proofs` field of VmInternalStarkProof
inline std::vector<ProofV1> decode_proofs(const uint8_t* bytes, size_t len) {
    Reader r(bytes, len);
    std::vector<ProofV1> v(r.len(64));
    for (auto& p : v) p = read_proof(r);
    if (r.remaining()) throw DecodeError("trailing bytes after the proofs");
    return v;
}
inline std::vector<uint8_t> encode_proofs(const std::vector<ProofV1>& v) {
    Writer w;
    w.u64(v.size());
    for (const auto& p : v) write_proof(w, p);
    return std::move(w.out);
}

// structural checks a decoder can make without the verifying key: every field element below p (as Montgomery words are
// too), degrees powers of two, every query shaped like the first
inline bool well_formed(const ProofV1& p, uint32_t modulus = 2013265921u) {
    auto okw = [&](uint32_t v) { return v < modulus; };
    auto okd = [&](const Digest& d) {
        for (uint32_t v : d)
            if (!okw(v)) return false;
        return true;
    };
    auto oke = [&](const ExtWords& e) {
        for (uint32_t v : e)
            if (!okw(v)) return false;
        return true;
    };
    for (const auto& d : p.main_trace)
        if (!okd(d)) return false;
    for (const auto& d : p.after_challenge)
        if (!okd(d)) return false;
    if (!okd(p.quotient)) return false;
    for (const auto& d : p.fri.commit_phase_commits)
        if (!okd(d)) return false;
    for (const auto& e : p.fri.final_poly)
        if (!oke(e)) return false;
    for (const auto& a : p.per_air) {
        if (a.degree == 0 || (a.degree & (a.degree - 1))) return false;
        for (uint32_t v : a.public_values)
            if (!okw(v)) return false;
        for (const auto& ph : a.exposed_values_after_challenge)
            for (const auto& e : ph)
                if (!oke(e)) return false;
    }
    // every opened value, the proof-of-work witnesses: a word w + p would decode to the same field element as w
    auto oka = [&](const AdjacentOpenedValues& v) {
        for (const auto& e : v.local)
            if (!oke(e)) return false;
        for (const auto& e : v.next)
            if (!oke(e)) return false;
        return true;
    };
    for (const auto& v : p.values.preprocessed)
        if (!oka(v)) return false;
    for (const auto& c : p.values.main)
        for (const auto& v : c)
            if (!oka(v)) return false;
    for (const auto& c : p.values.after_challenge)
        for (const auto& v : c)
            if (!oka(v)) return false;
    for (const auto& a : p.values.quotient)
        for (const auto& ch : a)
            for (const auto& e : ch)
                if (!oke(e)) return false;
    if (!okw(p.fri.pow_witness) || (p.has_logup_pow && !okw(p.logup_pow_witness))) return false;
    const QueryProof* q0 = p.fri.query_proofs.empty() ? nullptr : &p.fri.query_proofs[0];
    for (const auto& q : p.fri.query_proofs) {
        if (q.input_proof.size() != q0->input_proof.size() || q.commit_phase_openings.size() != p.fri.commit_phase_commits.size())
            return false;
        for (size_t b = 0; b < q.input_proof.size(); b++) {
            const auto &x = q.input_proof[b], &y = q0->input_proof[b];
            if (x.opened_values.size() != y.opened_values.size() || x.opening_proof.size() != y.opening_proof.size()) return false;
            for (size_t m = 0; m < x.opened_values.size(); m++) {
                if (x.opened_values[m].size() != y.opened_values[m].size()) return false;
                for (uint32_t v : x.opened_values[m])
                    if (!okw(v)) return false;
            }
            for (const auto& d : x.opening_proof)
                if (!okd(d)) return false;
        }
        for (size_t l = 0; l < q.commit_phase_openings.size(); l++) {
            if (q.commit_phase_openings[l].opening_proof.size() != q0->commit_phase_openings[l].opening_proof.size()) return false;
            if (!oke(q.commit_phase_openings[l].sibling_value)) return false;
            for (const auto& d : q.commit_phase_openings[l].opening_proof)
                if (!okd(d)) return false;
        }
    }
    return true;
}

}  // namespace zkhip_codec
