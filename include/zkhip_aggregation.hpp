// zkhip_aggregation.hpp -- the aggregation layer over the verifier circuit of the C ABI (zkhip_recursion_*): what the
// reference does with its leaf / internal aggregation provers.
//
//   crates/prover/src/prover/mod.rs:47-60    leaf / internal parameter sets, DEFAULT_AGG_TREE_CONFIG (arity 4 / 3)
//   crates/prover/src/prover/mod.rs:200-282  `commit_child_vk` -> VerifyProver: a node proves "my children verify under the child vk"
//   crates/integration/src/lib.rs:461-514    the data handed from one level to the next
//
// ONE AGGREGATION KEY (crates/prover/src/prover/mod.rs:147-170 caches one `agg_vk`; crates/verifier/src/verifier.rs:96-111 verifies
// every root under it): the LEAF circuit is built for the app's verifying key (zkhip_recursion_build: its wiring is preprocessed, so
// its commitments pin the app) and verifies <= 4 segment proofs; the INTERNAL circuit verifies <= 3 proofs of the leaf circuit OR OF
// ITSELF -- both circuits share the AIR programs and, padded, the heights; the child's preprocessed commitments are values of the
// internal circuit, and every node states the pair (leaf commitment, internal commitment) it requires below it.  The root of a tree
// of any depth is a proof of the internal circuit: the aggregation key = (internal verifying key, leaf commitment, app-vk digest).
// A node's public values are [app-vk digest (8) | start state (K) | end state (K) | accumulator (8) | leaf commitment (8) | internal
// commitment (8)]; the children's states are chained in-circuit.  Every node proof is self-verified (mod.rs:407-411 does that for
// every proof it returns).  AggregationTreeConfig::one_key = false keeps round 3's per-depth keys (level l hard-wires level l - 1).
#pragma once
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <deque>
#include <memory>
#include <map>
#include <mutex>
#include <thread>

#include "zkhip_prover.hpp"

namespace scroll_zkvm_hip {

struct StatementSpec {
    std::vector<std::pair<uint32_t, uint32_t>> start, end;  // (AIR, public-value index) of the state words of a segment proof
};

// a verified unit entering a node: proof bytes + the public values of every AIR of its verifying key
struct ChildProof {
    std::vector<uint8_t> proof;
    std::vector<std::vector<uint32_t>> pvs;
};

// verifying key of a node circuit (or of the app): everything zkhip_verify needs
struct VerifyingKey {
    zkhip_params params{};
    std::vector<AirDesc> airs;      // program, width, n_pvs, prep_commit (no tables)
    std::vector<unsigned> heights;  // fixed trace heights
    // an AGGREGATION key also pins what the root must state beneath it: the leaf circuit's commitment (-> the app) and the app-vk digest
    std::vector<uint32_t> leaf_commit, app_digest;   // 8 words each, empty on any other key
    // a JOIN key (deferral: the root's statement is followed by the deferral accumulator; the internal commitment of the tree beneath is
    // pinned inside the join circuit, not by this key's own commitments)
    bool join = false;

    std::vector<zkhip_air> as_airs() const {
        std::vector<zkhip_air> za(airs.size());
        for (size_t a = 0; a < airs.size(); a++) {
            za[a] = zkhip_air{airs[a].program.data(), airs[a].program.size(), heights[a], airs[a].width, airs[a].n_pvs, nullptr, nullptr};
            if (airs[a].has_prep) za[a].prep_commit = airs[a].prep_commit.size() == 8 ? airs[a].prep_commit.data() : nullptr;
        }
        return za;
    }
    bool verify(const ChildProof& p) const {
        if (p.pvs.size() != airs.size()) return false;
        std::vector<zkhip_air> za = as_airs();
        std::vector<const uint32_t*> pv(airs.size());
        for (size_t a = 0; a < airs.size(); a++) {
            if (p.pvs[a].size() != airs[a].n_pvs) return false;
            pv[a] = p.pvs[a].data();
        }
        return zkhip_verify(&params, za.data(), za.size(), pv.data(), p.proof.data(), p.proof.size()) == ZKHIP_OK;
    }
    // for failure messages: 0 when the proof verifies, else the line of csrc/verifier.hip that refuses it (-1: malformed for this key)
    int refused_at(const ChildProof& p) const {
        if (p.pvs.size() != airs.size()) return -1;
        std::vector<zkhip_air> za = as_airs();
        std::vector<const uint32_t*> pv(airs.size());
        for (size_t a = 0; a < airs.size(); a++) {
            if (p.pvs[a].size() != airs[a].n_pvs) return -1;
            pv[a] = p.pvs[a].data();
        }
        int where = 0;
        const int rc = zkhip_verify_where(&params, za.data(), za.size(), pv.data(), p.proof.data(), p.proof.size(), &where);
        return rc == ZKHIP_OK ? 0 : (where ? where : -1);
    }
    // the app-file form (encode_app_exe, tables omitted): UniversalVerifier::setup reads it back.  An aggregation key appends
    // [AGGKEY_MAGIC | leaf commitment (8) | app-vk digest (8)] (readers of the app-file form stop before it).
    static constexpr uint32_t AGGKEY_MAGIC = 0x4B474741u, JOINKEY_MAGIC = 0x4B4E4F4Au;
    std::vector<uint8_t> to_app_exe() const {
        std::vector<AirDesc> a = airs;
        for (size_t i = 0; i < a.size(); i++) {
            a[i].prep.clear();
            if (a[i].has_prep) a[i].prep_log_height = heights[i];
        }
        std::vector<uint8_t> out = encode_app_exe(a);
        if (leaf_commit.size() == 8 && app_digest.size() == 8) {
            std::vector<uint32_t> t{join ? JOINKEY_MAGIC : AGGKEY_MAGIC};
            t.insert(t.end(), leaf_commit.begin(), leaf_commit.end()), t.insert(t.end(), app_digest.begin(), app_digest.end());
            const uint8_t* b = reinterpret_cast<const uint8_t*>(t.data());
            out.insert(out.end(), b, b + 4 * t.size());
        }
        return out;
    }
    static VerifyingKey read(const std::string& path, const zkhip_params& params) {
        VerifyingKey vk;
        vk.params = params;
        vk.airs = read_app_exe(path);
        for (const auto& a : vk.airs) vk.heights.push_back(a.prep_log_height);
        std::ifstream f(path, std::ios::binary);
        const std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const size_t used = encode_app_exe(vk.airs).size();
        if (raw.size() == used + 4 * 17) {
            uint32_t t[17];
            memcpy(t, raw.data() + used, sizeof t);
            if (t[0] == AGGKEY_MAGIC || t[0] == JOINKEY_MAGIC) vk.leaf_commit.assign(t + 1, t + 9), vk.app_digest.assign(t + 9, t + 17), vk.join = t[0] == JOINKEY_MAGIC;
        }
        return vk;
    }
    bool is_aggregation_key() const { return leaf_commit.size() == 8 && app_digest.size() == 8; }
    // digest of this key's preprocessed commitments: what a node of the tree states as its internal (or leaf) commitment
    std::vector<uint32_t> key_commit() const {
        std::vector<uint32_t> pc;
        for (const auto& a : airs)
            if (a.has_prep) {
                if (a.prep_commit.size() != 8) throw Error(Error::Setup, "a verifying key without its preprocessed commitments");
                pc.insert(pc.end(), a.prep_commit.begin(), a.prep_commit.end());
            }
        std::vector<uint32_t> out(8);
        if (pc.empty() || zkhip_recursion_key_commit(pc.data(), pc.size() / 8, out.data()) != ZKHIP_OK) throw Error(Error::Setup, "zkhip_recursion_key_commit");
        return out;
    }
    // What a ROOT under an aggregation key must state beside verifying: the internal commitment is this key's own, the leaf commitment
    // and the app-vk digest are the key's (root public values: [app digest (8) | start (K) | end (K) | accumulator (8) | leaf (8) | internal (8)]).
    bool root_statement_matches(const std::vector<uint32_t>& root_pvs, std::string* why = nullptr) const {
        auto fail = [&](const char* m) {
            if (why) *why = m;
            return false;
        };
        if (!is_aggregation_key()) return fail("not an aggregation key (no leaf commitment / app digest)");
        if (root_pvs.size() < 32 + (join ? 8u : 0u)) return fail("the root statement is too short for an aggregation key");
        const size_t n = root_pvs.size() - (join ? 8 : 0);   // (a join's statement ends with the deferral accumulator)
        if (!std::equal(app_digest.begin(), app_digest.end(), root_pvs.begin())) return fail("the root is not about this app (app verifying-key digest)");
        if (!std::equal(leaf_commit.begin(), leaf_commit.end(), root_pvs.begin() + (n - 16))) return fail("the tree's leaves are not proofs of this key's leaf circuit");
        if (join) return true;   // (the join circuit requires the internal commitment of the key it was built for)
        const std::vector<uint32_t> ic = key_commit();
        if (!std::equal(ic.begin(), ic.end(), root_pvs.begin() + (n - 8))) return fail("the tree's internal nodes are not proofs of this key's internal circuit");
        return true;
    }
};

// FRI parameters of the node proofs: `leaf` for the nodes over segment proofs, `internal` for every level above
// (crates/prover/src/prover/mod.rs:47-52 `default_agg_params`: leaf_params_with_100_bits_security / internal_params_with_100_bits_security
// of the un-vendored SDK).  A node circuit is built for its CHILD's parameters and proven under its own, and a level's verifying key
// carries its parameters (VerifyingKey::params).  By default every level uses the app's parameters: the file form of a verifying key
// (`root.vk`) does not carry parameters and the command-line verifier reads them from the app's openvm.toml.  with_100_bits_security()
// is the pair that follows the rule the reference's own parameters obey (openvm.toml: blow-up 2 with 100 queries; its stored proofs:
// blow-up 4 with 44 queries; 16 + 16 proof-of-work bits): an internal proof costs more to make and less than half to verify in the
// next circuit.
struct AggregationSystemParams {
    zkhip_params leaf{}, internal{};
    static AggregationSystemParams defaults_for(const zkhip_params& app) {
        AggregationSystemParams a;
        a.leaf = a.internal = app;
        return a;
    }
    static AggregationSystemParams with_100_bits_security() {
        AggregationSystemParams a;
        a.leaf = zkhip_params{1, 0, 100, 16, 16};
        a.internal = zkhip_params{2, 0, 44, 16, 16};
        return a;
    }
    // under ONE key leaf and internal proofs are verified by the same circuit: one parameter set for every node (the segments keep the app's)
    static AggregationSystemParams nodes_100_bits_security() {
        AggregationSystemParams a;
        a.leaf = a.internal = zkhip_params{2, 0, 44, 16, 16};
        return a;
    }
    bool same() const { return memcmp(&leaf, &internal, sizeof(zkhip_params)) == 0; }
};

using Digest8 = std::array<uint32_t, 8>;
inline Digest8 p2_compress8(const Digest8& l, const Digest8& r) {
    uint32_t st[16];
    for (int i = 0; i < 8; i++) st[i] = l[i], st[8 + i] = r[i];
    zkhip_poseidon2_permute_host(st);
    Digest8 d;
    std::copy(st, st + 8, d.begin());
    return d;
}
inline Digest8 p2_sponge8(const uint32_t* v, size_t n) {   // PaddingFreeSponge<16, 8, 8>: overwrite, permute, no padding
    uint32_t st[16] = {};
    for (size_t i = 0; i < n; i += 8) {
        for (size_t k = 0; k < 8 && i + k < n; k++) st[k] = v[i + k];
        zkhip_poseidon2_permute_host(st);
    }
    Digest8 d;
    std::copy(st, st + 8, d.begin());
    return d;
}
class AggregationProver {
public:
    struct Stats {
        size_t nodes = 0;
        std::vector<size_t> nodes_per_slot;   // how the tree's nodes spread over the device slots
        double witness_seconds = 0, tracegen_prove_seconds = 0, verify_seconds = 0, keygen_seconds = 0, build_seconds = 0;
        size_t leafs_at_setup = 0, leafs_on_demand = 0;   // leaf circuits built with the key / when a shape's first segment proof arrived
    };

    // PER-PROOF CHIP PRESENCE (the reference proves only the chips a segment used: AGENTS.md:183-185): an app may have several SHAPES -- sets of
    // chips a segment carries, each with a segment key of its own; the LAST one is the full set -- and the tree one leaf circuit per shape
    // (one_key only).  A leaf node takes proofs of ONE shape; the internal circuit takes leaf proofs of any listed shape.
    static AggregationProver setup_shapes(const std::vector<VerifyingKey>& shapes, const StatementSpec& spec, AggregationTreeConfig cfg = {}, int device = 0,
                                          const AggregationSystemParams* agg_params = nullptr) {
        if (shapes.empty() || shapes.size() > 8) throw Error(Error::Setup, "aggregation: 1..8 shapes");
        if (shapes.size() > 1 && !cfg.one_key) throw Error(Error::Setup, "aggregation: several shapes need the one-key tree");
        AggregationProver p = setup(shapes.back(), spec, cfg, device, agg_params);
        p.apps_ = shapes;
        for (const auto& a : shapes)
            if (a.airs.size() != a.heights.size() || a.airs.empty()) throw Error(Error::Setup, "aggregation: the app verifying key needs one height per AIR");
        return p;
    }
    static AggregationProver setup(const VerifyingKey& app, const StatementSpec& spec, AggregationTreeConfig cfg = {}, int device = 0,
                                   const AggregationSystemParams* agg_params = nullptr) {
        AggregationProver p;
        p.app_ = app, p.apps_ = {app}, p.spec_ = spec, p.cfg_ = cfg, p.device_ = device;
        p.agg_params_ = agg_params ? *agg_params : AggregationSystemParams::defaults_for(app.params);
        p.levels_mu_.reset(new std::mutex), p.build_mu_.reset(new std::mutex);
        p.devices_ = {device};
        if (app.airs.size() != app.heights.size() || app.airs.empty()) throw Error(Error::Setup, "aggregation: the app verifying key needs one height per AIR");
        if (spec.start.size() != spec.end.size()) throw Error(Error::Setup, "aggregation: start and end state must have the same length");
        if (cfg.one_key && !p.agg_params_.same())
            throw Error(Error::Setup, "aggregation: under one key the leaf and the internal proofs are verified by one circuit and need one parameter set "
                                      "(AggregationTreeConfig::one_key = false keeps a key per depth)");
        return p;
    }
    // The GPUs the tree's nodes run on ("Aggregation-tree proving of the gathered proofs ... sharded one node per GPU per tree
    // level"): every device gets its own copy of the node proving keys and its own circuit forks; a node goes to whichever device's
    // pipeline is free first.  A device may be listed more than once (several pipelines on one GPU).  Before the first node only.
    void set_devices(const std::vector<int>& devices) {
        if (devices.empty()) throw Error(Error::Setup, "aggregation: an empty device list");
        if (!slots_.empty()) throw Error(Error::Setup, "aggregation: the device list is fixed once keys exist");
        devices_ = devices;
    }
    size_t n_slots() const { return devices_.size(); }
    // (a node proof gives way to a key generation that waits for the slot: a leaf circuit built on demand would otherwise queue behind
    // every node proof the slot's workers re-lock for -- std::mutex is not fair; measured 2.5 s of waiting in a guest of two wide shapes)
    void wait_turn(size_t slot) {
        const std::atomic<int>& u = *slots_.at(slot).urgent;
        while (u.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    std::mutex& slot_mutex(size_t slot) {
        wait_turn(slot);
        return *slots_.at(slot).mu;
    }
    AggregationProver(AggregationProver&& o) noexcept { *this = std::move(o); }
    AggregationProver& operator=(AggregationProver&& o) noexcept {
        reset();
        app_ = std::move(o.app_), apps_ = std::move(o.apps_), leafs_ = std::move(o.leafs_), bigleafs_ = std::move(o.bigleafs_), policy_ = std::move(o.policy_), spec_ = std::move(o.spec_), cfg_ = o.cfg_, agg_params_ = o.agg_params_, device_ = o.device_, levels_ = std::move(o.levels_), stats = o.stats;
        leaf_commit_ = std::move(o.leaf_commit_), internal_commit_ = std::move(o.internal_commit_), leaf_list_ = std::move(o.leaf_list_);
        used_ = std::move(o.used_), used_on_disk_ = std::move(o.used_on_disk_), nat_ = std::move(o.nat_), cache_path_ = std::move(o.cache_path_), cache_stored_ = o.cache_stored_;
        lazy_ = std::move(o.lazy_), prefetching_ = std::move(o.prefetching_), prefetch_threads_ = std::move(o.prefetch_threads_), common_h_[0] = o.common_h_[0], common_h_[1] = o.common_h_[1];
        levels_mu_ = std::move(o.levels_mu_), build_mu_ = std::move(o.build_mu_), shape_mu_ = std::move(o.shape_mu_);
        devices_ = std::move(o.devices_), slots_ = std::move(o.slots_);
        o.slots_.clear();
        return *this;
    }
    ~AggregationProver() { reset(); }
    // A segment of `shape` has been EXECUTED (its proof is still being made): if the shape's leaf circuit was left out at setup, start
    // building it now, beside the segment proving -- by the time the tree needs it, it is there (or ensure_leaf waits for the rest).
    void prefetch_leaf(size_t shape) {
        {
            std::lock_guard<std::mutex> lk(*levels_mu_);
            if (shape >= lazy_.size() || !lazy_[shape]) return;
            if (prefetching_.size() < lazy_.size()) prefetching_.resize(lazy_.size(), 0);
            if (prefetching_[shape]) return;
            prefetching_[shape] = 1;
        }
        prefetch_threads_.emplace_back([this, shape] {
            try {
                ensure_leaf(shape);
            } catch (...) {   // (the node that needs the shape runs ensure_leaf itself and reports)
            }
        });
    }
    void reset() {
        for (auto& t : prefetch_threads_)
            if (t.joinable()) t.join();
        prefetch_threads_.clear();
        remember_used_shapes();
        for (auto* set : {&levels_, &leafs_, &bigleafs_})
            for (auto& L : *set) free_level(L);
        levels_.clear(), leafs_.clear(), bigleafs_.clear();
        for (auto& sl : slots_)
            if (sl.ctx) zkhip_ctx_destroy(sl.ctx);
        slots_.clear();
    }

    // the cache file learns which shapes this guest's flows use (see AggCache); called when a flow ends and at reset()
    void remember_used_shapes() {
        if (!levels_mu_) return;   // (moved from)
        std::lock_guard<std::mutex> lk(*levels_mu_);
        if (!cache_stored_ || cache_path_.empty() || used_.size() != used_on_disk_.size() || nat_.size() != used_.size()) return;
        bool grew = false;
        for (size_t sh = 0; sh < used_.size(); sh++)
            if (used_[sh] && !used_on_disk_[sh]) used_on_disk_[sh] = 1, grew = true;
        if (grew) store_agg_cache(cache_path_, common_h_, nat_, leaf_list_, used_on_disk_);
    }
    void mark_used(size_t shape) {
        std::lock_guard<std::mutex> lk(*levels_mu_);
        if (shape < used_.size()) used_[shape] = 1;
    }

    Stats stats;

    // verifying key of the node circuit of `level` (0 = leaf; under one key every level above is THE internal circuit)
    const VerifyingKey& node_vk(size_t level, size_t shape = 0) {
        ensure_level(level);
        if (level == 0 && cfg_.one_key) ensure_leaf(shape);
        return lv(level, shape).vk;
    }
    size_t n_shapes() const { return apps_.size(); }
    const VerifyingKey& app_vk(size_t shape = 0) const { return apps_.at(shape); }
    // the key a root of a tree with `n_levels` levels verifies under: one key = the aggregation key whatever the depth
    const VerifyingKey& root_vk(size_t n_levels = 2) {
        if (!cfg_.one_key) return node_vk(n_levels - 1);
        ensure_level(1);
        return lv(1).vk;
    }
    bool one_key() const { return cfg_.one_key; }
    const AggregationTreeConfig& tree_config() const { return cfg_; }
    size_t first_root_layer() const { return cfg_.one_key ? 2 : 1; }
    size_t arity(size_t level, size_t shape = 0) const {
        if (level > 0) return cfg_.num_children_internal;
        return shape < policy_.size() && policy_[shape].arity ? policy_[shape].arity : cfg_.num_children_leaf;
    }
    // How a shape's segment proofs enter the tree.  A leaf circuit's size grows with the total width of the chips it verifies; the tree's
    // circuits share ONE height set, so a shape much wider than the base set would make every node of every tree as large as ITS leaf
    // circuit (measured: the reference's chunk-circuit configuration, 51 chips, 2^25 gate rows instead of 2^22).  `arity`: how many
    // segment proofs of the shape a leaf node takes (0 = the tree's leaf arity).  `wrapped`: the shape's leaf circuit keeps its natural
    // (large) heights and its proofs enter the tree through a WRAPPER -- a circuit of the common size that verifies one proof of that
    // leaf circuit and restates its public values; the wrapper's commitment is the shape's entry in the leaf-commitment list.
    struct ShapePolicy {
        unsigned arity = 0;     // (a wrapped shape: proofs of its leaf circuit -- one segment proof each -- per wrapper)
        bool wrapped = false;
    };
    void set_shape_policies(const std::vector<ShapePolicy>& p) {
        if (!slots_.empty()) throw Error(Error::Setup, "aggregation: shape policies are fixed once keys exist");
        if (p.size() != apps_.size()) throw Error(Error::Setup, "aggregation: one policy per shape");
        for (const auto& q : p)
            if (q.arity > 8) throw Error(Error::Setup, "aggregation: a leaf node takes at most 8 proofs");
        policy_ = p;
    }
    bool wrapped(size_t shape) const { return shape < policy_.size() && policy_[shape].wrapped; }
    // the shapes an earlier process of this guest put into the tree, as the key cache remembers them (empty: no cache, or no key yet) -- what the
    // flow's other lanes build their segment keys for at setup instead of inside the first segment proof of that shape
    std::vector<char> shapes_used_before() const { return used_on_disk_; }

    // proves one node: level 0 verifies up to 4 proofs of the app, level l > 0 up to 3 proofs of level l - 1
    ChildProof prove_node(size_t level, const std::vector<const ChildProof*>& kids, size_t shape = 0, const std::vector<size_t>* kid_shapes = nullptr, size_t slot = 0) {
        using clk = std::chrono::steady_clock;
        auto t0 = clk::now();
        std::vector<uint32_t> npv = witness_node(level, kids, shape, kid_shapes, slot);
        auto t1 = clk::now();
        ChildProof out = prove_witnessed(level, std::move(npv), shape, slot);
        auto t2 = clk::now();
        if (!lv(level, shape).vk.verify(out)) throw Error(Error::VerifyProof, "aggregation: the node proof does not verify");
        stats.nodes++;
        stats.witness_seconds += std::chrono::duration<double>(t1 - t0).count();
        stats.tracegen_prove_seconds += std::chrono::duration<double>(t2 - t1).count();
        stats.verify_seconds += std::chrono::duration<double>(clk::now() - t2).count();
        return out;
    }

    // "execution" of a node: runs the verifier circuit on the children (host; one thread per child inside the library) and
    // returns the node's public values.  Throws if a child does not verify or the states do not chain.
    // `shape` (level 0): which shape's segment proofs the node takes; `kid_shapes` (level 1): the shape of each leaf child (default 0)
    // `kinds` (one key, level > 0; optional): per child 0 = a proof of the internal circuit, j + 1 = a proof of leaf circuit j -- the
    // internal circuit takes any mix, which is what lets the tree take ANY shape (TreeStream's greedy fold)
    std::vector<uint32_t> witness_node(size_t level, const std::vector<const ChildProof*>& kids, size_t shape = 0, const std::vector<size_t>* kid_shapes = nullptr,
                                       size_t slot = 0, const std::vector<int>* kinds_in = nullptr) {
        ensure_level(level);
        if (level == 0 && cfg_.one_key) ensure_leaf(shape), mark_used(shape);
        Replica& L = lv(level, shape).rep.at(slot);
        if (kids.empty() || kids.size() > arity(level, shape)) throw Error(Error::GenProof, "aggregation: a node of level " + std::to_string(level) + " takes 1.." + std::to_string(arity(level, shape)) + " children");
        const VerifyingKey& cvk = level == 0 ? apps_.at(shape) : lv(level - 1).vk;   // (one key: the leaf's and the internal vk differ in the commitments only)
        std::vector<const uint8_t*> proofs;
        std::vector<size_t> lens;
        std::vector<std::vector<const uint32_t*>> pv_rows(kids.size());
        std::vector<const uint32_t* const*> pv_ptrs;
        for (size_t c = 0; c < kids.size(); c++) {
            if (kids[c]->pvs.size() != cvk.airs.size()) throw Error(Error::GenProof, "aggregation: child public values do not match the child verifying key");
            proofs.push_back(kids[c]->proof.data()), lens.push_back(kids[c]->proof.size());
            for (size_t a = 0; a < cvk.airs.size(); a++) {
                if (kids[c]->pvs[a].size() != cvk.airs[a].n_pvs) throw Error(Error::GenProof, "aggregation: child public values do not match the child verifying key");
                pv_rows[c].push_back(kids[c]->pvs[a].data());
            }
            pv_ptrs.push_back(pv_rows[c].data());
        }
        std::vector<uint32_t> npv(zkhip_recursion_n_pvs(L.circ));
        int rc;
        if (cfg_.one_key && level > 0) {
            // children of level 1 are proofs of the leaf circuit, above of the internal circuit itself
            std::vector<uint32_t> pc;
            std::vector<int> kinds(kids.size(), 0);
            for (size_t c = 0; c < kids.size(); c++) {
                const size_t ks = level == 1 && kid_shapes && !kinds_in ? kid_shapes->at(c) : 0;
                if (kinds_in) kinds[c] = kinds_in->at(c);
                else if (level == 1) kinds[c] = (int)ks + 1;   // (0 = a proof of the internal circuit, j + 1 = of leaf circuit j)
                if (kinds[c] < 0 || (size_t)kinds[c] > leafs_.size()) throw Error(Error::GenProof, "aggregation: unknown kind of child");
                const VerifyingKey& kvk = kinds[c] ? lv(0, (size_t)kinds[c] - 1).vk : lv(1).vk;
                for (const auto& a : kvk.airs) pc.insert(pc.end(), a.prep_commit.begin(), a.prep_commit.end());
            }
            rc = zkhip_recursion_witness_uniform(L.circ, proofs.data(), lens.data(), pv_ptrs.data(), pc.data(), kinds.data(), leaf_list_.data(),
                                                 internal_commit_.data(), kids.size(), npv.data());
        } else {
            rc = zkhip_recursion_witness(L.circ, proofs.data(), lens.data(), pv_ptrs.data(), kids.size(), npv.data());
        }
        if (rc != ZKHIP_OK) {
            // which child, and whose fault: the same proofs through the host verifier (a child it accepts and the circuit refuses points
            // at the witness generator; one it refuses too at the prover that made it)
            std::string detail = zkhip_recursion_last_error(L.circ);
            if (rc == ZKHIP_ERR_VERIFY) {
                for (size_t c = 0; c < kids.size(); c++) {
                    const VerifyingKey* kvk = &cvk;
                    if (cfg_.one_key && level > 0) {
                        const int kind = kinds_in ? kinds_in->at(c) : level == 1 ? (int)(kid_shapes ? kid_shapes->at(c) : 0) + 1 : 0;
                        kvk = kind ? &lv(0, (size_t)kind - 1).vk : &lv(1).vk;
                    }
                    int at = -1;
                    try {
                        at = kvk->refused_at(*kids[c]);
                    } catch (...) {
                    }
                    detail += std::string("; child ") + std::to_string(c) +
                              (at == 0 ? " verifies on the host" : " does NOT verify on the host (verifier.hip:" + std::to_string(at) + ")");
                }
            }
            throw Error(Error::GenProof, "aggregation: " + detail);
        }
        return npv;
    }
    // A leaf node of a WRAPPED shape: the shape's (large) leaf circuit over its segment proofs, then the wrapper over that proof -- what
    // enters the tree is the wrapper's proof.  Host witness and device work of both, one after the other, on one slot.
    ChildProof prove_wrapped(const std::vector<const ChildProof*>& kids, size_t shape, size_t slot = 0) {
        ensure_level(0);
        if (cfg_.one_key) ensure_leaf(shape), mark_used(shape);
        if (!wrapped(shape)) throw Error(Error::GenProof, "aggregation: not a wrapped shape");
        Level& B = bigleafs_.at(shape);
        Level& W = lv(0, shape);
        auto run = [&](Level& L, const VerifyingKey& cvk, const std::vector<const ChildProof*>& ks) {
            Replica& r = L.rep.at(slot);
            std::vector<const uint8_t*> proofs;
            std::vector<size_t> lens;
            std::vector<std::vector<const uint32_t*>> rows(ks.size());
            std::vector<const uint32_t* const*> pv;
            for (size_t c = 0; c < ks.size(); c++) {
                if (ks[c]->pvs.size() != cvk.airs.size()) throw Error(Error::GenProof, "aggregation: child public values do not match the child verifying key");
                proofs.push_back(ks[c]->proof.data()), lens.push_back(ks[c]->proof.size());
                for (const auto& v : ks[c]->pvs) rows[c].push_back(v.data());
                pv.push_back(rows[c].data());
            }
            std::vector<uint32_t> npv(zkhip_recursion_n_pvs(r.circ));
            if (zkhip_recursion_witness(r.circ, proofs.data(), lens.data(), pv.data(), ks.size(), npv.data()) != ZKHIP_OK)
                throw Error(Error::GenProof, std::string("aggregation: ") + zkhip_recursion_last_error(r.circ));
            wait_turn(slot);
            std::lock_guard<std::mutex> dev(*slots_.at(slot).mu);
            zkhip_ctx* c = slots_[slot].ctx;
            check(zkhip_recursion_tracegen(c, r.circ, (uint32_t*)r.d_traces[0], (uint32_t*)r.d_traces[1], (uint32_t*)r.d_traces[2]), slot);
            ChildProof out;
            out.pvs.resize(3);
            out.pvs[2] = std::move(npv);
            out.proof.resize(zkhip_proof_size(r.pk));
            const uint32_t* dt[3] = {(const uint32_t*)r.d_traces[0], (const uint32_t*)r.d_traces[1], (const uint32_t*)r.d_traces[2]};
            const uint32_t* pvp[3] = {nullptr, nullptr, out.pvs[2].data()};
            size_t len = 0;
            check(zkhip_prove(c, r.pk, dt, pvp, out.proof.data(), out.proof.size(), &len), slot);
            out.proof.resize(len);
            return out;
        };
        if (kids.empty() || kids.size() > arity(0, shape)) throw Error(Error::GenProof, "aggregation: too many segment proofs for a leaf node of this shape");
        // (second session of round 5) the large leaf circuit takes ONE segment proof; the wrapper takes up to arity(0, shape) of its proofs -- two for the
        // chunk circuit's full shape: a wrapper per PAIR of wide segments instead of one each
        std::vector<ChildProof> bigs;
        for (const ChildProof* k : kids) bigs.push_back(run(B, apps_.at(shape), {k}));
        std::vector<const ChildProof*> bp;
        for (const ChildProof& b : bigs) bp.push_back(&b);
        return run(W, B.vk, bp);
    }
    // device trace generation + proof of the node whose witness was computed last on this level
    ChildProof prove_witnessed(size_t level, std::vector<uint32_t> node_pvs, size_t shape = 0, size_t slot = 0) {
        wait_turn(slot);
        std::lock_guard<std::mutex> dev(*slots_.at(slot).mu);
        upload_witness(level, shape, slot);
        return prove_uploaded(level, std::move(node_pvs), shape, slot);
    }

    // The aggregation tree as a STREAM: segment proofs are pushed as they complete (any order, any thread) and every node starts as
    // soon as its children exist -- leaf nodes while later segments are still being executed and proven, internal nodes while other
    // nodes of the level below are still in flight.  The grouping is AggregationPlan's (node k of a level = children [a k, a k + a) of
    // the level below), so full groups are known before the number of segments is; finish(n) fixes the partial ones and returns the
    // root.  Per level two host threads: one runs the verifier circuit of the next node (witness, host cores), one generates its
    // traces on the device and proves it (the levels share the prover's context: device sections are serialised, while the segment
    // lanes keep proving on contexts of their own); every node proof is self-verified on a thread of its own.
    // (a template over the prover, so that the scheduling -- threads, queues, the greedy fold -- runs under ThreadSanitizer against a prover
    // that makes stub proofs: tests/tree_stream_tsan.cpp)
    template <class Agg>
    class TreeStreamT {
    public:
        // `greedy` (one key only): above the leaf nodes the tree has no fixed shape -- whenever `arity` ADJACENT node proofs exist (leaf or
        // internal, any mix) they are folded, leftmost first.  Every fold but the last takes a full set of children, so the number of
        // internal nodes is the balanced tree's; but when the nodes keep up with the segments the result is a comb -- root = (everything
        // before, the last two leaf nodes) -- and what is left once the last segment proof exists is ONE leaf node and the root instead of
        // one node per level; when they do not keep up, waiting proofs fold among themselves and the shape drifts to the balanced one.
        explicit TreeStreamT(Agg& agg, bool greedy = false) : agg_(agg), greedy_(greedy && agg.one_key()), fold_(agg.arity(1)) {
            layers_.emplace_back();
            start_level(0);
            if (greedy_) {
                std::lock_guard<std::mutex> lk(mu_);
                while (layers_.size() <= 3) layers_.emplace_back();
                layers_[1].started = true;
                for (size_t level : {1, 2})   // two internal-circuit workers per device slot (the forks the levels of a fixed tree use)
                    for (size_t slot = 0; slot < agg_.n_slots(); slot++) start_pair_locked(level, slot);
            }
        }
        TreeStreamT(const TreeStreamT&) = delete;
        TreeStreamT& operator=(const TreeStreamT&) = delete;
        ~TreeStreamT() {
            {
                std::lock_guard<std::mutex> lk(mu_);
                stop_ = true;
            }
            cv_.notify_all();
            join_all();
        }
        bool trace = false;   // a line per event on stderr: milliseconds since the stream was made (FlowOptions::trace_tree)
        void note(const char* what, size_t level, size_t k, size_t slot, size_t n_kids = 0) const {
            if (!trace) return;
            std::fprintf(stderr, "[tree %8.2f ms] %-14s level %zu node %zu slot %zu kids %zu\n",
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_made_).count(), what, level, k, slot, n_kids);
        }
        void push(size_t index, ChildProof seg, size_t shape = 0) {
            note("segment", 0, index, 0);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (seg_shape_.size() <= index) seg_shape_.resize(index + 1, 0);
                seg_shape_[index] = shape;
            }
            put(0, index, std::move(seg));
        }
        // all `n_segments` proofs have been (or will be) pushed: waits for the root.  `all` (optional) receives every level's proofs.
        ChildProof finish(size_t n_segments, std::vector<std::vector<ChildProof>>* all = nullptr) {
            if (n_segments == 0) throw Error(Error::GenProof, "aggregation: no segment proofs");
            std::unique_lock<std::mutex> lk(mu_);
            layers_[0].total = n_segments;
            cv_.notify_all();
            size_t root_layer = 0;
            if (greedy_) greedy_schedule_locked();
            cv_.wait(lk, [&] {
                if (!err_.empty()) return true;
                if (greedy_) return groot_ != nullptr;
                for (size_t i = agg_.first_root_layer(); i < layers_.size(); i++)
                    if (layers_[i].total == 1 && !layers_[i].items.empty() && layers_[i].items[0]) {
                        root_layer = i;
                        return true;
                    }
                return false;
            });
            if (!err_.empty()) {
                const std::string e = err_;
                stop_ = true;
                lk.unlock();
                cv_.notify_all();
                join_all();
                throw Error(e.find("does not verify") != std::string::npos && e.find("node ") != std::string::npos ? Error::VerifyProof : Error::GenProof, e);
            }
            stop_ = true;
            lk.unlock();
            cv_.notify_all();
            join_all();   // the self-verification of the last nodes
            if (!err_.empty()) throw Error(Error::VerifyProof, err_);
            if (greedy_) {   // all[0]: the leaf nodes in order; all[1]: the internal nodes in the order they were formed, the root last
                if (all) {
                    all->assign(2, {});
                    for (const ChildProof* q : gleaves_) (*all)[0].push_back(*q);
                    for (const ChildProof* q : ginternal_) (*all)[1].push_back(*q);
                }
                return *groot_;
            }
            if (all)
                for (size_t i = 1; i <= root_layer; i++) {
                    all->emplace_back();
                    for (auto& it : layers_[i].items) all->back().push_back(*it);
                }
            return *layers_[root_layer].items[0];
        }
        // node proofs formed and waiting for a free pipeline (every level).  The flow's lanes look at it before they start a segment proof: a tree that
        // lags behind the segment proofs is finished alone, three streams on an under-filled GPU (mixed guest at frames of 2^20: a 700 ms tail).
        // (counted in SEGMENT PROOFS handed in whose leaf node's proof does not exist yet: queued node proofs are no measure -- a pipeline's witness
        // thread takes a node off the queue as soon as its buffer is free)
        size_t backlog() {
            std::lock_guard<std::mutex> lk(mu_);
            return segs_pushed_ - std::min(segs_pushed_, segs_in_leaf_proofs_);
        }
        size_t levels() const {
            if (greedy_) return 1 + fold_.root_depth();
            size_t n = 0;
            for (size_t i = 1; i < layers_.size(); i++) n += !layers_[i].items.empty();
            return n;
        }

    private:
        struct Task {   // one node of a level, formed by the level's grouper
            size_t k = 0, shape = 0;
            std::vector<const ChildProof*> kids;
            std::vector<size_t> kid_shapes;
            std::vector<int> kinds;         // greedy fold: per child 0 internal, j + 1 leaf circuit j
            FoldLine::Fold fold;            // greedy fold: what FoldLine handed out
        };
        struct Layer {
            std::deque<std::unique_ptr<ChildProof>> items;   // layer 0 = segment proofs, layer l + 1 = node proofs of level l
            size_t total = SIZE_MAX;                          // number of items once known
            bool started = false;                             // the threads of the level that consumes this layer exist
            std::deque<Task> tasks;                           // nodes of the level that consumes this layer, waiting for a free slot
            bool tasks_closed = false;
        };
        Agg& agg_;
        const std::chrono::steady_clock::time_point t_made_ = std::chrono::steady_clock::now();
        std::mutex mu_;
        std::condition_variable cv_;
        std::deque<Layer> layers_;
        size_t segs_pushed_ = 0, segs_in_leaf_proofs_ = 0;   // backlog(): segment proofs handed in / covered by finished leaf nodes
        std::deque<size_t> leaf_n_kids_;                     // segment proofs under leaf node k (written when the node is formed)
        std::vector<size_t> seg_shape_;     // shape of segment i (layer 0)
        std::deque<size_t> leaf_shape_;     // shape of leaf proof k (layer 1), written when the node is formed
        std::vector<std::thread> threads_, verifiers_;
        std::string err_;
        bool stop_ = false;
        // greedy fold (see the constructor; the decisions are FoldLine's, include/zkhip_prover.hpp)
        const bool greedy_;
        FoldLine fold_;
        std::deque<Task> gq_;                                // folds waiting for a free internal worker
        std::deque<std::unique_ptr<ChildProof>> gstore_;     // the internal nodes' proofs (stable addresses)
        std::vector<const ChildProof*> gleaves_, ginternal_;
        size_t gnext_id_ = 0;
        bool gq_closed_ = false;
        const ChildProof* groot_ = nullptr;

        // mu_ held: queues every fold that can start now; notices the root
        void greedy_schedule_locked() {
            if (stop_ || !err_.empty() || groot_) return;
            if (layers_[1].total != SIZE_MAX) fold_.set_total(layers_[1].total);   // the number of leaf nodes once known
            FoldLine::Fold f;
            while (fold_.next(&f)) {
                Task t;
                t.k = gnext_id_++, t.kinds = f.kinds;
                for (const void* k : f.kids) t.kids.push_back((const ChildProof*)k);
                t.fold = std::move(f);
                gq_.push_back(std::move(t));
            }
            if (fold_.root()) groot_ = (const ChildProof*)fold_.root(), gq_closed_ = true;
            cv_.notify_all();
        }
        // a fold's proof exists
        const ChildProof* greedy_done(const Task& t, ChildProof p) {
            std::lock_guard<std::mutex> lk(mu_);
            gstore_.emplace_back(new ChildProof(std::move(p)));
            const ChildProof* made = gstore_.back().get();
            ginternal_.push_back(made);
            fold_.done(t.fold, made);
            greedy_schedule_locked();
            return made;
        }

        void join_all() {
            for (;;) {   // threads may start further threads until stop_ is seen
                std::vector<std::thread> t;
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    t.swap(threads_);
                    for (auto& v : verifiers_) t.push_back(std::move(v));
                    verifiers_.clear();
                }
                if (t.empty()) return;
                for (auto& th : t)
                    if (th.joinable()) th.join();
            }
        }
        void fail(const std::string& e) {
            std::lock_guard<std::mutex> lk(mu_);
            if (err_.empty()) err_ = e;
            cv_.notify_all();
        }
        void put(size_t layer, size_t index, ChildProof p) {
            std::lock_guard<std::mutex> lk(mu_);
            while (layers_.size() <= layer) layers_.emplace_back();
            Layer& L = layers_[layer];
            if (L.items.size() <= index) L.items.resize(index + 1);
            L.items[index].reset(new ChildProof(std::move(p)));
            if (layer == 0) segs_pushed_++;
            if (layer == 1) segs_in_leaf_proofs_ += index < leaf_n_kids_.size() ? leaf_n_kids_[index] : 1;
            if (greedy_ && layer == 1) {   // a leaf node's proof: a piece of the line
                const ChildProof* made = L.items[index].get();
                fold_.add(index, index + 1, (int)(index < leaf_shape_.size() ? leaf_shape_[index] : 0) + 1, 0, made);
                if (gleaves_.size() <= index) gleaves_.resize(index + 1, nullptr);
                gleaves_[index] = made;
                greedy_schedule_locked();
            } else if (layer > 0 && !L.started && !stop_) {
                start_level_locked(layer);
            }
            cv_.notify_all();
        }
        void start_level(size_t level) {
            std::lock_guard<std::mutex> lk(mu_);
            start_level_locked(level);
        }
        // level `level` consumes layer `level` and produces layer `level + 1`.  One GROUPER thread forms the level's nodes in order (above
        // the leaves node k takes children [a k, a k + a) of the layer below; a LEAF node takes the next run of segment proofs of one shape,
        // at most a) and queues them; every device slot has a pair of threads that takes nodes from the queue as it gets free: one runs
        // the verifier circuit (witness, host cores) into the slot's fork of the circuit, one generates the traces on the slot's device and
        // proves the node -- witness (node n + 1) beside device (node n) per slot, the slots beside each other.
        void start_level_locked(size_t level) {
            while (layers_.size() <= level + 1) layers_.emplace_back();
            layers_[level].started = true;
            threads_.emplace_back([this, level] {
                try {
                    size_t next = 0;   // first child of the next node
                    for (size_t k = 0;; k++) {
                        std::unique_lock<std::mutex> lk(mu_);
                        bool end = false;
                        size_t hi = next;
                        cv_.wait(lk, [&] {
                            // (a leaf node takes as many proofs as its shape's policy says)
                            const size_t a = agg_.arity(level, level == 0 && seg_shape_.size() > next ? seg_shape_[next] : 0);
                            if (stop_ || !err_.empty()) return true;
                            const Layer& in = layers_[level];
                            if (in.total != SIZE_MAX && (next >= in.total || (level >= agg_.first_root_layer() && in.total == 1))) {
                                end = true;   // no further node on this level (a single item of a root layer is the root)
                                return true;
                            }
                            // the node's children: up to `a` present items from `next`; a leaf node stops where the shape changes
                            for (hi = next; hi < next + a; hi++) {
                                if (in.total != SIZE_MAX && hi >= in.total) break;
                                if (in.items.size() <= hi || !in.items[hi]) return false;
                                if (level == 0 && hi > next && seg_shape_[hi] != seg_shape_[next]) break;
                            }
                            return true;
                        });
                        if (end) layers_[level + 1].total = (level >= agg_.first_root_layer() && layers_[level].total == 1) ? 0 : k;
                        if (end || stop_ || !err_.empty()) {
                            layers_[level].tasks_closed = true;
                            if (greedy_ && level == 0 && end) greedy_schedule_locked();   // (the number of leaf nodes is known now)
                            cv_.notify_all();
                            break;
                        }
                        Task t;
                        t.k = k;
                        const Layer& in = layers_[level];
                        for (size_t c = next; c < hi; c++) {
                            t.kids.push_back(in.items[c].get());
                            if (level == 1) t.kid_shapes.push_back(c < leaf_shape_.size() ? leaf_shape_[c] : 0);
                        }
                        if (level == 0) {
                            t.shape = seg_shape_.size() > next ? seg_shape_[next] : 0;
                            if (leaf_shape_.size() <= k) leaf_shape_.resize(k + 1, 0);
                            leaf_shape_[k] = t.shape;
                            if (leaf_n_kids_.size() <= k) leaf_n_kids_.resize(k + 1, 0);
                            leaf_n_kids_[k] = hi - next;
                        }
                        next = hi;
                        layers_[level].tasks.push_back(std::move(t));
                        cv_.notify_all();
                    }
                } catch (const std::exception& e) {
                    fail(e.what());
                    std::lock_guard<std::mutex> lk(mu_);
                    layers_[level].tasks_closed = true;
                    cv_.notify_all();
                }
            });
            for (size_t slot = 0; slot < agg_.n_slots(); slot++) start_pair_locked(level, slot);
        }
        void start_pair_locked(size_t level, size_t slot) {
            struct Shared {
                std::mutex m;
                std::condition_variable c;
                bool witness_ready = false, buffer_free = true, done = false;
                size_t node = 0, shape = 0;
                std::vector<uint32_t> npv;
                Task task;
            };
            auto sh = std::make_shared<Shared>();
            // witness thread of the slot: the next queued node, once the slot's witness buffer of that circuit is free
            threads_.emplace_back([this, level, slot, sh] {
                try {
                    for (;;) {
                        Task t;
                        {
                            std::unique_lock<std::mutex> lk(mu_);
                            const bool fold = greedy_ && level >= 1;
                            std::deque<Task>& q = fold ? gq_ : layers_[level].tasks;
                            cv_.wait(lk, [&] { return stop_ || !err_.empty() || !q.empty() || (fold ? gq_closed_ : layers_[level].tasks_closed); });
                            if (stop_ || !err_.empty() || q.empty()) break;
                            t = std::move(q.front());
                            q.pop_front();
                        }
                        {
                            std::unique_lock<std::mutex> lk(sh->m);
                            sh->c.wait(lk, [&] { return sh->buffer_free; });
                        }
                        if (level == 0 && agg_.wrapped(t.shape)) {
                            // the shape's own leaf circuit, then its wrapper: both here, the wrapper's proof enters layer 1
                            const auto tw = std::chrono::steady_clock::now();
                            ChildProof out = agg_.prove_wrapped(t.kids, t.shape, slot);
                            const double dw = std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
                            {
                                std::lock_guard<std::mutex> lk(mu_);
                                agg_.stats.tracegen_prove_seconds += dw, agg_.stats.nodes += 2;
                            }
                            put(level + 1, t.k, std::move(out));
                            continue;
                        }
                        const auto t0 = std::chrono::steady_clock::now();
                        note("witness starts", level, t.k, slot, t.kids.size());
                        std::vector<uint32_t> npv = agg_.witness_node(level, t.kids, t.shape, &t.kid_shapes, slot, t.kinds.empty() ? nullptr : &t.kinds);
                        note("witness done", level, t.k, slot, t.kids.size());
                        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        {
                            std::lock_guard<std::mutex> lk(mu_);
                            agg_.stats.witness_seconds += dt;
                        }
                        std::lock_guard<std::mutex> lk(sh->m);
                        sh->npv = std::move(npv), sh->node = t.k, sh->shape = t.shape, sh->task = std::move(t), sh->witness_ready = true, sh->buffer_free = false;
                        sh->c.notify_all();
                    }
                } catch (const std::exception& e) {
                    fail(e.what());
                }
                std::lock_guard<std::mutex> lk(sh->m);
                sh->done = true;
                sh->c.notify_all();
            });
            // device thread of the slot: trace generation (frees the witness buffer) and proof of the node, then its self-verification elsewhere
            threads_.emplace_back([this, level, slot, sh] {
                try {
                    for (;;) {
                        std::vector<uint32_t> npv;
                        size_t k, shape;
                        Task task;
                        {
                            std::unique_lock<std::mutex> lk(sh->m);
                            sh->c.wait(lk, [&] { return sh->witness_ready || sh->done; });
                            if (!sh->witness_ready) return;
                            npv = std::move(sh->npv), k = sh->node, shape = sh->shape, task = std::move(sh->task), sh->witness_ready = false;
                        }
                        const auto t = std::chrono::steady_clock::now();
                        ChildProof out;
                        note("device waits", level, k, slot);
                        {
                            std::lock_guard<std::mutex> dev(agg_.slot_mutex(slot));
                            note("device starts", level, k, slot);
                            agg_.upload_witness(level, shape, slot);
                            note("traces done", level, k, slot);
                            {
                                std::lock_guard<std::mutex> lk(sh->m);
                                sh->buffer_free = true;
                                sh->c.notify_all();
                            }
                            out = agg_.prove_uploaded(level, std::move(npv), shape, slot);
                        }
                        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
                        {
                            std::lock_guard<std::mutex> lk(mu_);
                            agg_.stats.tracegen_prove_seconds += dt, agg_.stats.nodes++;
                            if (agg_.stats.nodes_per_slot.size() <= slot) agg_.stats.nodes_per_slot.resize(slot + 1, 0);
                            agg_.stats.nodes_per_slot[slot]++;
                        }
                        note("proof done", level, k, slot);
                        const ChildProof* folded = nullptr;
                        if (greedy_ && level >= 1) folded = greedy_done(task, std::move(out));
                        else put(level + 1, k, std::move(out));
                        std::lock_guard<std::mutex> lk(mu_);
                        const ChildProof* made = folded ? folded : layers_[level + 1].items[k].get();
                        verifiers_.emplace_back([this, level, k, made, shape] {
                            const auto tv = std::chrono::steady_clock::now();
                            const bool ok = agg_.level_vk(level, shape).verify(*made);
                            const double dv = std::chrono::duration<double>(std::chrono::steady_clock::now() - tv).count();
                            std::lock_guard<std::mutex> lk(mu_);
                            agg_.stats.verify_seconds += dv;
                            if (!ok && err_.empty()) {
                                err_ = "aggregation: the proof of node " + std::to_string(k) + " of level " + std::to_string(level) + " does not verify";
                                cv_.notify_all();
                            }
                        });
                    }
                } catch (const std::exception& e) {
                    fail(e.what());
                    std::lock_guard<std::mutex> lk(sh->m);
                    sh->buffer_free = true;
                    sh->c.notify_all();
                }
            });
        }
    };
    using TreeStream = TreeStreamT<AggregationProver>;

    // The whole tree of `plan` over the segment proofs; returns the root, `all` (optional) receives every level.
    // (`greedy`: TreeStream's fold without a fixed shape above the leaf nodes -- one key only; `all` then holds two layers)
    ChildProof prove_tree(const AggregationPlan& plan, const std::vector<ChildProof>& segments, std::vector<std::vector<ChildProof>>* all = nullptr,
                          const std::vector<size_t>* shapes = nullptr, bool greedy = false) {
        if (segments.size() != plan.n_segments) throw Error(Error::GenProof, "aggregation: segment proof count does not match the plan");
        TreeStream ts(*this, greedy);
        for (size_t i = 0; i < segments.size(); i++) ts.push(i, segments[i], shapes ? shapes->at(i) : 0);
        return ts.finish(segments.size(), all);
    }

    // ---- StarkProof container <-> ChildProof ----
    static ChildProof from_stark_proof(const StarkProof& sp, const VerifyingKey& vk) {
        ChildProof c;
        c.proof = sp.proof;
        size_t off = 0;
        for (const auto& a : vk.airs) {
            if (off + 4 * a.n_pvs > sp.user_pvs_proof.size()) throw Error(Error::GenProof, "aggregation: short public values in a child proof");
            std::vector<uint32_t> v(a.n_pvs);
            if (a.n_pvs) memcpy(v.data(), sp.user_pvs_proof.data() + off, 4 * a.n_pvs);
            off += 4 * a.n_pvs;
            c.pvs.push_back(std::move(v));
        }
        if (sp.baseline.size() != vk.heights.size()) throw Error(Error::GenProof, "aggregation: a child proof's heights do not match the fixed heights of the child verifying key");
        for (size_t a = 0; a < vk.heights.size(); a++)
            if (sp.baseline[a] != vk.heights[a]) throw Error(Error::GenProof, "aggregation: a child proof's heights do not match the fixed heights of the child verifying key");
        return c;
    }
    static StarkProof to_stark_proof(const ChildProof& c, const VerifyingKey& vk) {
        StarkProof sp;
        sp.proof = c.proof;
        for (const auto& p : c.pvs) {
            const uint8_t* b = reinterpret_cast<const uint8_t*>(p.data());
            sp.user_pvs_proof.insert(sp.user_pvs_proof.end(), b, b + 4 * p.size());
        }
        for (unsigned h : vk.heights) sp.baseline.push_back((uint8_t)h);
        return sp;
    }

private:
    AggregationProver() : levels_mu_(new std::mutex), build_mu_(new std::mutex) {}
    // a level's copy on one device slot: the circuit user (slot 0: the level's own circuit; others: forks), the proving key on that device
    // (one key: the levels above the first internal one share its key), the trace buffers
    struct Replica {
        zkhip_recursion* circ = nullptr;
        bool owns_circ = false;
        zkhip_pk* pk = nullptr;
        bool owns_pk = true;
        std::vector<void*> d_traces;
    };
    struct Level {
        zkhip_recursion* circ = nullptr;   // the level's circuit (owned)
        VerifyingKey vk;
        std::vector<Replica> rep;          // per device slot
    };
    struct Slot {
        int device = 0;
        zkhip_ctx* ctx = nullptr;
        std::unique_ptr<std::mutex> mu;    // any call on ctx
        std::unique_ptr<std::atomic<int>> urgent = std::unique_ptr<std::atomic<int>>(new std::atomic<int>(0));   // key generations waiting for mu
    };
    std::vector<int> devices_;
    std::deque<Slot> slots_;
    void free_level(Level& L) {
        for (size_t sl = 0; sl < L.rep.size(); sl++) {
            Replica& r = L.rep[sl];
            zkhip_ctx* c = slots_.at(sl).ctx;
            for (void* d : r.d_traces)
                if (d) zkhip_free(c, d);
            if (r.pk && r.owns_pk) zkhip_pk_destroy(c, r.pk);
            if (r.circ && r.owns_circ) zkhip_recursion_destroy(r.circ);
        }
        L.rep.clear();
        if (L.circ) zkhip_recursion_destroy(L.circ), L.circ = nullptr;
    }
    // one key: zkhip_recursion_key_commit of the circuits' keys: leaf_list_ = the S leaf commitments (8 S words); leaf_commit_ = what a node
    // STATES as its leaf commitment: the commitment itself (one shape) or the sponge of the list
    std::vector<uint32_t> leaf_list_, leaf_commit_, internal_commit_;
    // LAZY leaf circuits (round 5): a shape whose commitment and natural heights are known from the on-disk cache of this key
    // (agg_cache_path) is not built at setup -- the wide shapes of the chunk configuration cost 3 - 4 s of circuit building and key
    // generation per process, used or not -- but when its first segment proof arrives (ensure_leaf), and what is built then must commit
    // to exactly what the cache said (the aggregation key already states it).
    std::vector<char> lazy_, prefetching_;
    std::vector<char> used_, used_on_disk_;            // shapes whose proofs entered the tree in this process / as the cache file knows them
    std::vector<std::array<unsigned, 2>> nat_;         // (what the cache file holds beside leaf_list_: rewritten when `used` grows)
    std::string cache_path_;
    bool cache_stored_ = false;
    std::vector<std::thread> prefetch_threads_;   // (joined in reset(): they call ensure_leaf on this object)
    unsigned common_h_[2] = {0, 0};
    std::vector<VerifyingKey> apps_;   // the shapes' segment keys (apps_.back() = app_ = the full set)
    std::deque<Level> leafs_;          // one key: the leaf circuit of every shape (a wrapped shape: its wrapper); levels_[0] is then unused
    std::deque<Level> bigleafs_;       // ... and, for a wrapped shape, its own (large) leaf circuit
    std::vector<ShapePolicy> policy_;
    VerifyingKey app_;
    StatementSpec spec_;
    AggregationTreeConfig cfg_;
    AggregationSystemParams agg_params_;
    int device_ = 0;
    std::deque<Level> levels_;                       // (stable references: levels are added while others are in use)
    // levels_mu_: the container (brief); build_mu_: one level is built at a time; a slot's mu: any call on its context.  Order: build ->
    // slot; levels_mu_ is never held while another one is taken.
    std::unique_ptr<std::mutex> levels_mu_, build_mu_;
    std::vector<std::unique_ptr<std::mutex>> shape_mu_;   // ensure_leaf: one per shape.  Order: shape -> slot; shape -> levels_mu_ (brief)
    Level& lv(size_t level, size_t shape = 0) {
        std::lock_guard<std::mutex> lk(*levels_mu_);
        if (level == 0 && cfg_.one_key) return leafs_.at(shape);
        return levels_.at(level);
    }
    size_t n_levels() {
        std::lock_guard<std::mutex> lk(*levels_mu_);
        return levels_.size();
    }
    const VerifyingKey& level_vk(size_t level, size_t shape = 0) { return lv(level, shape).vk; }

    void check(int rc, size_t slot = 0) const {
        if (rc != ZKHIP_OK) throw Error(Error::GenProof, std::string("aggregation: ") + zkhip_last_error(slots_.at(slot).ctx));
    }
    void upload_witness(size_t level, size_t shape = 0, size_t slot = 0) {
        Replica& L = lv(level, shape).rep.at(slot);
        check(zkhip_recursion_tracegen(slots_[slot].ctx, L.circ, (uint32_t*)L.d_traces[0], (uint32_t*)L.d_traces[1], (uint32_t*)L.d_traces[2]), slot);
    }
    ChildProof prove_uploaded(size_t level, std::vector<uint32_t> node_pvs, size_t shape = 0, size_t slot = 0) {
        Replica& L = lv(level, shape).rep.at(slot);
        zkhip_ctx* ctx_ = slots_[slot].ctx;
        ChildProof out;
        out.pvs.resize(3);
        out.pvs[2] = std::move(node_pvs);
        out.proof.resize(zkhip_proof_size(L.pk));
        const uint32_t* dt[3] = {(const uint32_t*)L.d_traces[0], (const uint32_t*)L.d_traces[1], (const uint32_t*)L.d_traces[2]};
        const uint32_t* pv[3] = {nullptr, nullptr, out.pvs[2].data()};
        size_t len = 0;
        check(zkhip_prove(ctx_, L.pk, dt, pv, out.proof.data(), out.proof.size(), &len), slot);
        out.proof.resize(len);
        return out;
    }
    // a circuit's three chips as the AIRs of its verifying key (no commitments yet) + the zkhip_air view the key generator takes
    static void circuit_airs(zkhip_recursion* circ, VerifyingKey* vk, std::vector<zkhip_air>* na) {
        na->assign(3, zkhip_air{});
        vk->airs.clear(), vk->heights.clear();
        for (size_t i = 0; i < 3; i++) {
            if (zkhip_recursion_air(circ, i, &(*na)[i]) != ZKHIP_OK) throw Error(Error::Setup, "aggregation: zkhip_recursion_air");
            AirDesc d;
            d.width = (*na)[i].width, d.n_pvs = (*na)[i].n_pvs, d.program.assign((*na)[i].program, (*na)[i].program + (*na)[i].program_len);
            d.has_prep = true, d.prep_log_height = (*na)[i].log_height;
            vk->airs.push_back(std::move(d));
            vk->heights.push_back((*na)[i].log_height);
        }
    }
    // proving keys (one per device slot) + preprocessed commitments + trace buffers + circuit forks of a built circuit
    void keygen_level(Level& L, std::vector<zkhip_air>& na) {
        L.rep.resize(slots_.size());
        for (size_t sl = 1; sl < slots_.size(); sl++) {
            if (zkhip_recursion_fork(L.circ, &L.rep[sl].circ) != ZKHIP_OK) throw Error(Error::Setup, "aggregation: zkhip_recursion_fork");
            L.rep[sl].owns_circ = true;
        }
        L.rep[0].circ = L.circ, L.rep[0].owns_circ = false;
        // one key per device slot (a key holds the scratch of ITS proofs), generated side by side: every slot has a context of its own
        std::vector<std::array<uint32_t, 24>> commits(slots_.size());
        std::vector<std::string> errs(slots_.size());
        auto one = [&](size_t sl) {
            try {
                Replica& r = L.rep[sl];
                slots_[sl].urgent->fetch_add(1, std::memory_order_acq_rel);
                std::unique_lock<std::mutex> dev(*slots_[sl].mu);
                slots_[sl].urgent->fetch_sub(1, std::memory_order_acq_rel);
                zkhip_ctx* c = slots_[sl].ctx;
                int rc = zkhip_keygen(c, &L.vk.params, na.data(), 3, &r.pk);
                if (rc != ZKHIP_OK) throw Error(Error::Keygen, std::string("failed to generate STARK proving key: ") + zkhip_last_error(c));
                for (size_t i = 0; i < 3; i++) check(zkhip_pk_prep_commitment(c, r.pk, i, &commits[sl][8 * i]), sl);
                alloc_traces(L, sl);
            } catch (const std::exception& e) {
                errs[sl] = e.what();
            }
        };
        std::vector<std::thread> th;
        for (size_t sl = 1; sl < slots_.size(); sl++) th.emplace_back(one, sl);
        one(0);
        for (auto& t : th) t.join();
        for (const std::string& e : errs)
            if (!e.empty()) throw Error(Error::Keygen, e);
        for (size_t i = 0; i < 3; i++) L.vk.airs[i].prep_commit.assign(&commits[0][8 * i], &commits[0][8 * i] + 8);
        for (size_t sl = 1; sl < slots_.size(); sl++)
            if (commits[sl] != commits[0]) throw Error(Error::Keygen, "aggregation: two devices disagree on a preprocessed commitment");
    }
    void alloc_traces(Level& L, size_t sl) {
        for (size_t i = 0; i < 3; i++) {
            void* d = nullptr;
            check(zkhip_malloc(slots_[sl].ctx, (L.vk.airs[i].width << L.vk.heights[i]) * 4, &d), sl);
            L.rep[sl].d_traces.push_back(d);
        }
    }
    void ensure_ctx() {
        if (!slots_.empty()) return;
        for (int d : devices_) {
            Slot sl;
            sl.device = d, sl.mu.reset(new std::mutex);
            int rc = zkhip_ctx_create(d, &sl.ctx);
            if (rc != ZKHIP_OK) {
                for (auto& x : slots_) zkhip_ctx_destroy(x.ctx);
                slots_.clear();
                throw Error(Error::Keygen, "no gfx950 device " + std::to_string(d) + " for the HIP backend (zkhip_ctx_create returned " + std::to_string(rc) + ")");
            }
            slots_.push_back(std::move(sl));
        }
    }
    zkhip_recursion_stmt leaf_stmt(std::vector<uint32_t> (&cols)[4]) const {
        zkhip_recursion_stmt st{};
        for (auto& p : spec_.start) cols[0].push_back(p.first), cols[1].push_back(p.second);
        for (auto& p : spec_.end) cols[2].push_back(p.first), cols[3].push_back(p.second);
        st.n_state = cols[0].size(), st.start_air = cols[0].data(), st.start_idx = cols[1].data(), st.end_air = cols[2].data(), st.end_idx = cols[3].data();
        return st;
    }
    // ---- the pieces of build_one_key, per shape (also what ensure_leaf runs for a shape that was left out at setup) ----
    void build_leaf_circuit(size_t sh, const std::vector<uint32_t>& app_id, Level* leaf) {
        std::vector<uint32_t> cols[4];
        zkhip_recursion_stmt st = leaf_stmt(cols);
        st.uniform = 1;
        st.app_id = apps_.size() > 1 ? app_id.data() : nullptr;
        std::vector<zkhip_air> za = apps_[sh].as_airs();
        const int rc = zkhip_recursion_build(&apps_[sh].params, za.data(), za.size(), wrapped(sh) ? 1 : arity(0, sh), &st, &leaf->circ);
        if (rc != ZKHIP_OK) throw Error(Error::Setup, std::string("aggregation: cannot build the leaf verifier circuit of shape ") + std::to_string(sh) + ": " + zkhip_recursion_last_error(nullptr));
    }
    // leaf (built, natural heights) -> big = that circuit with its keys, leaf = the wrapper over it
    void wrap_leaf(size_t sh, Level* leaf, Level* big) {
        *big = std::move(*leaf);
        *leaf = Level{};
        big->vk.params = agg_params_.leaf;
        std::vector<zkhip_air> ba;
        circuit_airs(big->circ, &big->vk, &ba);
        keygen_level(*big, ba);
        std::vector<zkhip_air> za = big->vk.as_airs();
        zkhip_recursion_stmt st{};
        st.child_is_node = 1, st.uniform = 1;
        const int rc = zkhip_recursion_build(&big->vk.params, za.data(), za.size(), arity(0, sh), &st, &leaf->circ);
        if (rc != ZKHIP_OK) throw Error(Error::Setup, std::string("aggregation: cannot build the wrapper of shape ") + std::to_string(sh) + ": " + zkhip_recursion_last_error(nullptr));
    }
    void pad_leaf(Level& leaf, const Level& internal, std::vector<zkhip_air>* la) {
        if (zkhip_recursion_pad(leaf.circ, common_h_) != ZKHIP_OK) throw Error(Error::Setup, "aggregation: zkhip_recursion_pad");
        leaf.vk.params = agg_params_.leaf;
        circuit_airs(leaf.circ, &leaf.vk, la);
        for (size_t i = 0; i < 3; i++)
            if (leaf.vk.airs[i].program != internal.vk.airs[i].program || leaf.vk.heights[i] != internal.vk.heights[i])
                throw Error(Error::Setup, "aggregation: the leaf and the internal circuit do not share one AIR set");
    }
    // The on-disk cache of ONE aggregation key: file name = a digest of everything the key depends on (every shape's segment-key digest,
    // the node parameters, arities, shape policies, the statement layout, and the size and time stamp of libzkhip.so: another build of the
    // circuit builder is another key).  Content: [magic, S, H0, H1, then per shape: natural heights (2), commitment (8), used (1)].
    // Directory: ZKHIP_AGG_CACHE_DIR, else the library's jit_cache_dir; none = no cache.
    // `used`: the key depends on the guest (the program chip's commitment is part of every shape's key), so the file is per guest -- and
    // it remembers which shapes this guest's flows put into the tree.  A shape the guest used before is built at setup again (its circuit
    // is needed, and building it BESIDE the segment proving costs more than it saves: measured 1.5 - 2.5 M instr/s against 2.9 M for the
    // mixed guest on the 16-CPU boxes); a shape no flow of this guest ever used stays out until a segment proof of it arrives.
    struct AggCache {
        bool hit = false;
        std::string path;
        unsigned h[2] = {0, 0};
        std::vector<std::array<unsigned, 2>> nat;
        std::vector<std::array<uint32_t, 8>> commit;
        std::vector<char> used;   // a flow of an earlier process put segment proofs of this shape into the tree
    };
    std::string agg_cache_path(const std::vector<uint32_t>& app_id) const {
        std::string dir;
        if (const char* e = getenv("ZKHIP_AGG_CACHE_DIR")) {
            dir = e;
        } else {
            zkhip_config c;
            zkhip_config_default(&c);
            dir = c.jit_cache_dir;
        }
        if (dir.empty()) return "";
        std::vector<uint32_t> words{0x41474B43u /* "AGKC" */, (uint32_t)apps_.size(), (uint32_t)cfg_.num_children_leaf, (uint32_t)cfg_.num_children_internal, (uint32_t)spec_.start.size()};
        words.insert(words.end(), app_id.begin(), app_id.end());
        for (const VerifyingKey& a : apps_) {
            uint32_t d[8] = {};
            std::vector<zkhip_air> za = a.as_airs();
            if (zkhip_recursion_vk_digest(&a.params, za.data(), za.size(), d) != ZKHIP_OK) return "";
            words.insert(words.end(), d, d + 8);
        }
        for (const zkhip_params* q : {&agg_params_.leaf, &agg_params_.internal})
            for (uint32_t v : {q->log_blowup, q->log_final_poly_len, q->num_queries, q->commit_pow_bits, q->query_pow_bits}) words.push_back(v);
        for (const ShapePolicy& q : policy_) words.push_back(q.arity), words.push_back(q.wrapped ? 1u : 0u);
        for (const auto& q : spec_.start) words.push_back(q.first), words.push_back(q.second);
        for (const auto& q : spec_.end) words.push_back(q.first), words.push_back(q.second);
        {   // the builder's identity: the CONTENT of the loaded library (size + a 64-bit hash of its bytes, once per process) and its version --
            // not its modification time: reproducible builds and images fix the mtime, and a cache directory may travel between boxes
            // 
            static const std::array<uint32_t, 4> id = [] {
                std::array<uint32_t, 4> out{0, 0, 0, (uint32_t)zkhip_version()};
                Dl_info info;
                if (!dladdr((const void*)&zkhip_recursion_build, &info) || !info.dli_fname) return out;
                FILE* f = std::fopen(info.dli_fname, "rb");
                if (!f) return out;
                uint64_t h = 1469598103934665603ull, n = 0;   // FNV-1a over 8-byte words (the tail byte-wise)
                std::vector<unsigned char> buf(1 << 20);
                for (size_t got; (got = std::fread(buf.data(), 1, buf.size(), f)) > 0; n += got)
                    for (size_t i = 0; i < got; i++) h = (h ^ buf[i]) * 1099511628211ull;
                std::fclose(f);
                out[0] = (uint32_t)n, out[1] = (uint32_t)h, out[2] = (uint32_t)(h >> 32);
                return out;
            }();
            words.insert(words.end(), id.begin(), id.end());
        }
        for (uint32_t& w : words) w %= 0x78000001u;
        const Digest8 h = p2_sponge8(words.data(), words.size());
        char name[96];
        std::snprintf(name, sizeof name, "/agg_%08x%08x%08x%08x.key", h[0], h[1], h[2], h[3]);
        return dir + name;
    }
    static constexpr uint32_t AGG_CACHE_MAGIC = 0x41474B45u;
    static Digest8 agg_cache_digest(const uint32_t* w, size_t n) {
        std::vector<uint32_t> r(w, w + n);
        for (uint32_t& x : r) x %= 0x78000001u;
        return p2_sponge8(r.data(), r.size());
    }
    AggCache load_agg_cache(const std::vector<uint32_t>& app_id) const {
        AggCache c;
        c.path = agg_cache_path(app_id);
        if (c.path.empty()) return c;
        FILE* f = std::fopen(c.path.c_str(), "rb");
        if (!f) return c;
        const size_t S = apps_.size();
        // [magic, S, H0, H1, per shape: natural heights (2), commitment (8), used (1)] + the Poseidon2 sponge of all that (8 words): a file of
        // another length, with another magic or a body that does not hash to its last eight words is no cache at all (the setup builds
        // every shape and rewrites it)
        std::vector<uint32_t> w(4 + 11 * S + 8 + 1);
        const size_t got = std::fread(w.data(), 4, w.size(), f);
        std::fclose(f);
        bool ok = got == 4 + 11 * S + 8 && w[0] == AGG_CACHE_MAGIC && w[1] == S;
        if (ok) {
            const Digest8 d = agg_cache_digest(w.data(), 4 + 11 * S);
            ok = std::equal(d.begin(), d.end(), w.begin() + 4 + 11 * S);
        }
        if (!ok) {
            std::fprintf(stderr, "[zkhip aggregation] %s is not a key cache of this build (length, magic or digest): ignored and rewritten\n", c.path.c_str());
            return c;
        }
        c.h[0] = w[2], c.h[1] = w[3];
        for (size_t sh = 0; sh < S; sh++) {
            c.nat.push_back({w[4 + 11 * sh], w[5 + 11 * sh]});
            std::array<uint32_t, 8> k;
            std::copy(w.begin() + 6 + 11 * sh, w.begin() + 14 + 11 * sh, k.begin());
            c.commit.push_back(k);
            c.used.push_back(w[14 + 11 * sh] ? 1 : 0);
        }
        c.hit = true;
        return c;
    }
    void store_agg_cache(const std::string& path, const unsigned H[2], const std::vector<std::array<unsigned, 2>>& nat, const std::vector<uint32_t>& leaf_list,
                         const std::vector<char>& used) const {
        if (path.empty()) return;
        const size_t S = apps_.size();
        std::vector<uint32_t> w{AGG_CACHE_MAGIC, (uint32_t)S, H[0], H[1]};
        for (size_t sh = 0; sh < S; sh++) {
            w.push_back(nat[sh][0]), w.push_back(nat[sh][1]);
            w.insert(w.end(), leaf_list.begin() + 8 * sh, leaf_list.begin() + 8 * sh + 8);
            w.push_back(sh < used.size() && used[sh] ? 1u : 0u);
        }
        const Digest8 d = agg_cache_digest(w.data(), w.size());
        w.insert(w.end(), d.begin(), d.end());
        const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)getpid());
        if (FILE* f = std::fopen(tmp.c_str(), "wb")) {
            const bool ok = std::fwrite(w.data(), 4, w.size(), f) == w.size();
            std::fclose(f);
            if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
        }
    }
public:
    // Every shape whose commitment came from the key cache WITHOUT being recomputed in this process is built now and must commit to what the
    // aggregation key states (ensure_leaf throws otherwise).  What a process does before it hands the key to someone ELSE -- a verifier, a
    // parent guest's program commitment (UniversalProver::get_agg_vk / program_commitment): the flow's own root proof is verified under the
    // key either way, but a stale cache would make that key differ from the one a cache-less process derives.
    void verify_lazy_shapes() {
        (void)root_vk();
        size_t n;
        {
            std::lock_guard<std::mutex> lk(*levels_mu_);
            n = lazy_.size();
        }
        for (size_t sh = 0; sh < n; sh++) ensure_leaf(sh);
    }

private:
    // a shape that was left out at setup: built now (its first segment proof has arrived), padded to the common heights, given its keys --
    // and it must commit to what the aggregation key already states
    void ensure_leaf(size_t sh) {
        {
            std::lock_guard<std::mutex> lk(*levels_mu_);
            if (sh >= lazy_.size() || !lazy_[sh]) return;
        }
        // one shape is built once; DIFFERENT shapes side by side (the circuit builder is host code, seconds for a 50-chip key; the key
        // generation takes the device slots' own locks) -- a guest whose segments fall into two wide shapes waits for the longer build, not
        // for their sum
        std::mutex* shape_mu;
        {
            std::lock_guard<std::mutex> lk(*levels_mu_);
            shape_mu = shape_mu_.at(sh).get();
        }
        std::lock_guard<std::mutex> build_lock(*shape_mu);
        {
            std::lock_guard<std::mutex> lk(*levels_mu_);
            if (!lazy_[sh]) return;
        }
        using clk = std::chrono::steady_clock;
        const auto t0 = clk::now();
        std::vector<uint32_t> app_id(8);
        {
            std::vector<zkhip_air> za = app_.as_airs();
            if (zkhip_recursion_vk_digest(&app_.params, za.data(), za.size(), app_id.data()) != ZKHIP_OK) throw Error(Error::Setup, "aggregation: zkhip_recursion_vk_digest");
        }
        Level leaf, big;
        std::vector<zkhip_air> la;
        try {
            build_leaf_circuit(sh, app_id, &leaf);
            if (policy_[sh].wrapped) wrap_leaf(sh, &leaf, &big);
            pad_leaf(leaf, lv(1), &la);
            keygen_level(leaf, la);
            const std::vector<uint32_t> c = leaf.vk.key_commit();
            if (!std::equal(c.begin(), c.end(), leaf_list_.begin() + 8 * sh))
                throw Error(Error::Setup, "aggregation: shape " + std::to_string(sh) + " commits to another key than the cached one the aggregation key states (a stale key cache: delete it)");
        } catch (...) {
            free_level(leaf), free_level(big);
            throw;
        }
        std::lock_guard<std::mutex> lk(*levels_mu_);
        stats.build_seconds += std::chrono::duration<double>(clk::now() - t0).count();
        stats.leafs_on_demand++;
        leafs_[sh] = std::move(leaf);
        bigleafs_[sh] = std::move(big);
        lazy_[sh] = 0;
    }
    // ONE KEY: the leaf circuit (for the app's key) and the internal circuit (for the node AIR set at the common heights) are built
    // together -- the internal circuit verifies proofs of its own height, so the common height set is a fixed point: start from the leaf
    // circuit's natural heights, build the internal circuit for children of that height, grow until it fits; then pad both.
    void build_one_key() {
        using clk = std::chrono::steady_clock;
        ensure_ctx();
        auto t0 = clk::now();
        const size_t S = apps_.size();
        if (policy_.size() != S) policy_.assign(S, ShapePolicy{});
        std::deque<Level> leafs(S), bigs(S);
        Level internal;
        std::vector<std::vector<zkhip_air>> la(S);
        std::vector<zkhip_air> ia;
        internal.vk.params = agg_params_.internal;
        // several shapes state ONE app id: the digest of the full set's key (one shape: the leaf circuit's own child digest, as before)
        std::vector<uint32_t> app_id(8);
        {
            std::vector<zkhip_air> za = app_.as_airs();
            if (zkhip_recursion_vk_digest(&app_.params, za.data(), za.size(), app_id.data()) != ZKHIP_OK) throw Error(Error::Setup, "aggregation: zkhip_recursion_vk_digest");
        }
        try {
            // what an earlier process of this very key left on disk: per shape the natural heights of its leaf circuit (or wrapper) and the
            // commitment of its padded key.  Shape 0 (the smallest: every run uses it, and its chips' programs are the node AIR set) is
            // always built; the others only if the cache does not know them.
            AggCache cache = load_agg_cache(app_id);
            lazy_.assign(S, 0);
            shape_mu_.clear();
            for (size_t sh = 0; sh < S; sh++) shape_mu_.emplace_back(new std::mutex);
            for (size_t sh = 1; sh < S; sh++) lazy_[sh] = cache.hit && !cache.used.at(sh) && !getenv("ZKHIP_AGG_NO_LAZY") ? 1 : 0;
            cache_path_ = cache.path, nat_.assign(S, {0, 0});
            used_.assign(S, 0), used_on_disk_.assign(S, 0);
            if (cache.hit) used_on_disk_ = cache.used;
            // the leaf circuits side by side (host only: seconds each for a 50-chip key)
            std::vector<std::string> errs(S);
            std::vector<std::thread> th;
            for (size_t sh = 0; sh < S; sh++)
                if (!lazy_[sh])
                    th.emplace_back([&, sh] {
                        try {
                            build_leaf_circuit(sh, app_id, &leafs[sh]);
                        } catch (const std::exception& e) {
                            errs[sh] = e.what();
                        }
                    });
            for (auto& t : th) t.join();
            for (const auto& e : errs)
                if (!e.empty()) throw Error(Error::Setup, e);
            // a WRAPPED shape: its leaf circuit keeps its natural heights and gets its keys now (the wrapper is a circuit FOR that key); what
            // takes the shape's place among the tree's leaf circuits is the wrapper (one child, the one-key public-value layout restated)
            for (size_t sh = 0; sh < S; sh++)
                if (policy_[sh].wrapped && !lazy_[sh]) wrap_leaf(sh, &leafs[sh], &bigs[sh]);
            unsigned H[2] = {0, 0};
            std::vector<std::array<unsigned, 2>> nat(S);
            for (size_t sh = 0; sh < S; sh++) {
                if (lazy_[sh]) {
                    nat[sh] = cache.nat.at(sh);
                } else {
                    leafs[sh].vk.params = agg_params_.leaf;
                    circuit_airs(leafs[sh].circ, &leafs[sh].vk, &la[sh]);
                    nat[sh] = {la[sh][0].log_height, la[sh][1].log_height};
                }
                H[0] = std::max(H[0], nat[sh][0]), H[1] = std::max(H[1], nat[sh][1]);
            }
            for (int round = 0;; round++) {
                if (round > 8) throw Error(Error::Setup, "aggregation: the common height of the leaf and internal circuits does not settle");
                std::vector<zkhip_air> child = la[0];   // the node AIR set: programs of the leaf circuit's chips, heights H
                child[0].log_height = H[0], child[1].log_height = H[1];
                for (auto& c : child) c.prep_trace = nullptr, c.prep_commit = nullptr;
                zkhip_recursion_stmt is{};
                is.child_is_node = 2, is.min_log_height[0] = H[0], is.min_log_height[1] = H[1], is.n_leaf_shapes = S;
                if (internal.circ) zkhip_recursion_destroy(internal.circ), internal.circ = nullptr;
                const int rc = zkhip_recursion_build(&agg_params_.leaf, child.data(), 3, arity(1), &is, &internal.circ);
                if (rc != ZKHIP_OK) throw Error(Error::Setup, std::string("aggregation: cannot build the internal verifier circuit: ") + zkhip_recursion_last_error(nullptr));
                circuit_airs(internal.circ, &internal.vk, &ia);
                if (ia[0].log_height == H[0] && ia[1].log_height == H[1]) break;
                H[0] = ia[0].log_height, H[1] = ia[1].log_height;
            }
            common_h_[0] = H[0], common_h_[1] = H[1];
            for (size_t sh = 0; sh < S; sh++) stats.leafs_at_setup += lazy_[sh] ? 0 : 1;
            if (cache.hit && (cache.h[0] != H[0] || cache.h[1] != H[1])) throw Error(Error::Setup, "aggregation: the key cache " + cache.path + " does not belong to this build (common heights differ): delete it");
            for (size_t sh = 0; sh < S; sh++) {
                if (lazy_[sh]) continue;
                pad_leaf(leafs[sh], internal, &la[sh]);
            }
            auto t1 = clk::now();
            leaf_list_.assign(8 * S, 0);
            for (size_t sh = 0; sh < S; sh++) {
                if (lazy_[sh]) {
                    std::copy(cache.commit.at(sh).begin(), cache.commit.at(sh).end(), leaf_list_.begin() + 8 * sh);
                    continue;
                }
                keygen_level(leafs[sh], la[sh]);
                const std::vector<uint32_t> c = leafs[sh].vk.key_commit();
                std::copy(c.begin(), c.end(), leaf_list_.begin() + 8 * sh);
                if (cache.hit && !std::equal(c.begin(), c.end(), cache.commit.at(sh).begin()))
                    throw Error(Error::Setup, "aggregation: the key cache " + cache.path + " does not belong to this build (shape " + std::to_string(sh) + " commits differently): delete it");
            }
            nat_ = nat;
            if (!cache.hit) {
                bool all = true;
                for (size_t sh = 0; sh < S; sh++) all = all && !lazy_[sh];
                if (all) store_agg_cache(cache_path_, H, nat, leaf_list_, used_on_disk_), cache_stored_ = true;
            } else {
                cache_stored_ = true;
            }
            keygen_level(internal, ia);
            internal_commit_ = internal.vk.key_commit();
            if (S == 1) {
                leaf_commit_ = leaf_list_;
            } else {
                const Digest8 h = p2_sponge8(leaf_list_.data(), leaf_list_.size());
                leaf_commit_.assign(h.begin(), h.end());
            }
            // the aggregation key: the internal verifying key + what a root must state beneath it
            internal.vk.leaf_commit = leaf_commit_;
            internal.vk.app_digest = app_id;   // (one shape: what its leaf circuit states anyway -- the digest of the key it is built for)
            stats.build_seconds += std::chrono::duration<double>(t1 - t0).count();
            stats.keygen_seconds += std::chrono::duration<double>(clk::now() - t1).count();
        } catch (...) {
            free_level(internal);
            for (auto& L : leafs) free_level(L);
            for (auto& L : bigs) free_level(L);
            throw;
        }
        std::lock_guard<std::mutex> lk(*levels_mu_);
        leafs_ = std::move(leafs);
        bigleafs_ = std::move(bigs);
        levels_.emplace_back();   // (level 0 lives in leafs_)
        levels_.push_back(std::move(internal));
    }
    void ensure_level(size_t level) {
        using clk = std::chrono::steady_clock;
        while (n_levels() <= level) {
            std::lock_guard<std::mutex> build_lock(*build_mu_);   // per level: a thread that needs level 1 gets in between levels 1 and 2 of a warm-up
            const size_t l = n_levels();
            if (l > level) break;
            if (cfg_.one_key) {
                if (l == 0) {
                    build_one_key();
                    continue;
                }
                // a further level of the tree: forks of THE internal circuit (own witness, own trace buffers), the same proving keys
                Level& I = lv(1);
                Level L;
                if (zkhip_recursion_fork(I.circ, &L.circ) != ZKHIP_OK) throw Error(Error::Setup, "aggregation: zkhip_recursion_fork");
                L.vk = I.vk;
                L.rep.resize(slots_.size());
                for (size_t sl = 0; sl < slots_.size(); sl++) {
                    Replica& r = L.rep[sl];
                    if (sl == 0) r.circ = L.circ, r.owns_circ = false;
                    else if (zkhip_recursion_fork(I.circ, &r.circ) != ZKHIP_OK) throw Error(Error::Setup, "aggregation: zkhip_recursion_fork");
                    else r.owns_circ = true;
                    r.pk = I.rep[sl].pk, r.owns_pk = false;
                    std::lock_guard<std::mutex> dev(*slots_[sl].mu);
                    alloc_traces(L, sl);
                }
                std::lock_guard<std::mutex> lk(*levels_mu_);
                levels_.push_back(std::move(L));
                continue;
            }
            ensure_ctx();
            const VerifyingKey& cvk = l == 0 ? app_ : lv(l - 1).vk;
            std::vector<zkhip_air> za = cvk.as_airs();
            std::vector<uint32_t> cols[4];
            zkhip_recursion_stmt st{};
            if (l == 0) st = leaf_stmt(cols);
            else st.child_is_node = 1;
            Level L;
            auto t0 = clk::now();
            int rc = zkhip_recursion_build(&cvk.params, za.data(), za.size(), arity(l), &st, &L.circ);
            if (rc != ZKHIP_OK) throw Error(Error::Setup, std::string("aggregation: cannot build the verifier circuit of level ") + std::to_string(l) + ": " + zkhip_recursion_last_error(nullptr));
            auto t1 = clk::now();
            L.vk.params = l == 0 ? agg_params_.leaf : agg_params_.internal;
            std::vector<zkhip_air> na;
            try {
                circuit_airs(L.circ, &L.vk, &na);
                keygen_level(L, na);
            } catch (...) {
                free_level(L);
                throw;
            }
            stats.build_seconds += std::chrono::duration<double>(t1 - t0).count();
            stats.keygen_seconds += std::chrono::duration<double>(clk::now() - t1).count();
            std::lock_guard<std::mutex> lk(*levels_mu_);
            levels_.push_back(std::move(L));
        }
    }
};


// ---- deferral (crates/prover/src/prover/mod.rs:200-282 `enable_deferral`; crates/integration/src/lib.rs:461-514 `compute_deferral_data`;
//      guest side crates/types/circuit/src/lib.rs:137-154 `verify_stark::<0>(input_commit, &expected)`) ---------------------------------------
// A parent guest (batch over chunks, bundle over batches) does not verify its children's proofs itself: it STATES claims -- "the proof
// with this input commitment verifies, for the program with this exe / vm commitment, with these public values" -- and the proof system
// backs every claim with a verified child proof.  Here a claim is 32 words the guest stores in its deferral region (see
// zkhip_vm::deferral_base): [input commitment (8) | exe commitment (8) | vm commitment (8) | the child's 32 public-value bytes (8 words)];
// the DEFERRAL NODE (zkhip_recursion_stmt.child_is_node = 3) verifies the child root proofs under the child app's aggregation key, derives
// the same 32 words from each proof and chains them; the JOIN (zkhip_recursion_build_join) verifies the guest's own root and the deferral
// node's proof and states [root statement | chain]; the verifier opens the region in the guest's final memory root and hashes it.
// the program commitments a parent guest holds about a child app (crates/types/circuit/src/lib.rs `ProgramCommitment { exe, vm }`; the
// reference generates them into crates/circuits/*-circuit/*_commit.rs): exe = compress(initial memory root, entry pc), vm =
// compress(app-vk digest, leaf-circuit commitment of the app's aggregation key)
struct ProgramCommitment {
    Digest8 exe{}, vm{};
    static ProgramCommitment of(const VerifyingKey& agg_key, uint32_t entry_pc, const Digest8& image_root) {
        if (!agg_key.is_aggregation_key()) throw Error(Error::Setup, "program commitment: not an aggregation key");
        ProgramCommitment c;
        c.exe = p2_compress8(image_root, Digest8{entry_pc, 0, 0, 0, 0, 0, 0, 0});
        Digest8 app, lc;
        std::copy(agg_key.app_digest.begin(), agg_key.app_digest.end(), app.begin());
        std::copy(agg_key.leaf_commit.begin(), agg_key.leaf_commit.end(), lc.begin());
        c.vm = p2_compress8(app, lc);
        return c;
    }
};
// one deferred child: its root proof, the root's statement (50 words), the 32 public-value bytes and the sibling digests above the
// public-value block pair in its final memory root (27 x 8, bottom-up)
inline bool chain_claim(Digest8& acc, const uint32_t claim[32]);
struct DeferralInput {
    ChildProof root;
    std::vector<uint8_t> public_values;
    std::vector<uint32_t> siblings;
    // a JOIN child (a proof of a guest that itself deferred: a batch under a bundle): the opening of ITS deferral region in its final
    // memory root -- 4096 cells, then the 19 sibling digests above the region's subtree (StarkProof::deferral_merkle_proofs)
    std::vector<uint32_t> region;
    static constexpr size_t REGION_CELLS = 4096, REGION_SIBS = 19, MAX_CLAIMS = 63;
    // from the StarkProof a guest flow returns: user_pvs_proof = [statement (50 words; a join's: 58) | 32 bytes | openings of both blocks (2 x 28 x 8)]
    static DeferralInput from_stark_proof(const StarkProof& sp, bool join = false) {
        const size_t n_stmt = join ? 58 : 50, n_open = 2 * 8 * 28;
        if (sp.user_pvs_proof.size() != 4 * n_stmt + 32 + 4 * n_open) throw Error(Error::GenProof, "deferral: a child proof is not a guest flow's root (user_pvs_proof size)");
        DeferralInput d;
        if (join) {
            if (sp.deferral_merkle_proofs.size() != 4 * (REGION_CELLS + 8 * REGION_SIBS)) throw Error(Error::GenProof, "deferral: a child proof under a join key carries no opening of its deferral region");
            d.region.resize(REGION_CELLS + 8 * REGION_SIBS);
            memcpy(d.region.data(), sp.deferral_merkle_proofs.data(), sp.deferral_merkle_proofs.size());
        }
        d.root.proof = sp.proof;
        d.root.pvs.resize(3);
        d.root.pvs[2].resize(n_stmt);
        memcpy(d.root.pvs[2].data(), sp.user_pvs_proof.data(), 4 * n_stmt);
        d.public_values.assign(sp.user_pvs_proof.begin() + 4 * n_stmt, sp.user_pvs_proof.begin() + 4 * n_stmt + 32);
        // block 0's opening, bottom-up: its first sibling is block 1's leaf digest (recomputed in-circuit); the 27 above are the pair's
        d.siblings.resize(27 * 8);
        memcpy(d.siblings.data(), sp.user_pvs_proof.data() + 4 * n_stmt + 32 + 4 * 8, 4 * 27 * 8);
        return d;
    }
    std::array<uint32_t, 16> cells() const {
        std::array<uint32_t, 16> c;
        for (int j = 0; j < 16; j++) c[j] = public_values[2 * j] | ((uint32_t)public_values[2 * j + 1] << 8);
        return c;
    }
    std::vector<uint32_t> aux() const {
        const std::array<uint32_t, 16> c = cells();
        std::vector<uint32_t> a(c.begin(), c.end());
        a.insert(a.end(), siblings.begin(), siblings.end());
        if (!region.empty()) {   // + the region's opening and the flags [k < n]
            a.insert(a.end(), region.begin(), region.end());
            const uint32_t n = region[0] | (region[1] << 16);
            for (uint32_t k = 0; k < MAX_CLAIMS; k++) a.push_back(k < n ? 1u : 0u);
        }
        return a;
    }
    // a JOIN child on the host (what the deferral node does in the circuit): the region opens in the child's final memory root as node
    // `region_index` of its level, and its claims chain to the value the child's statement ends with
    bool region_backs_the_chain(uint32_t region_index, std::string* why = nullptr) const {
        auto fail = [&](const char* m) {
            if (why) *why = m;
            return false;
        };
        const std::vector<uint32_t>& s = root.pvs[2];
        if (s.size() != 58 || region.size() != REGION_CELLS + 8 * REGION_SIBS) return fail("not a join child");
        for (size_t i = 0; i < REGION_CELLS; i++)
            if (region[i] > 0xffffu) return fail("a cell of the deferral region is not 16 bits");
        std::vector<Digest8> level(REGION_CELLS / 8);
        for (size_t b = 0; b < level.size(); b++) {
            Digest8 cells;
            std::copy(region.begin() + 8 * b, region.begin() + 8 * b + 8, cells.begin());
            level[b] = p2_compress8(cells, Digest8{});
        }
        while (level.size() > 1) {
            std::vector<Digest8> up(level.size() / 2);
            for (size_t i = 0; i < up.size(); i++) up[i] = p2_compress8(level[2 * i], level[2 * i + 1]);
            level = std::move(up);
        }
        Digest8 cur = level[0];
        uint32_t idx = region_index;
        for (size_t l = 0; l < REGION_SIBS; l++, idx >>= 1) {
            Digest8 sib;
            std::copy(region.begin() + REGION_CELLS + 8 * l, region.begin() + REGION_CELLS + 8 * l + 8, sib.begin());
            cur = (idx & 1u) ? p2_compress8(sib, cur) : p2_compress8(cur, sib);
        }
        if (!std::equal(cur.begin(), cur.end(), s.begin() + 18)) return fail("the deferral region does not open in the child's final memory root");
        auto word = [&](size_t w) { return region[2 * w] | (region[2 * w + 1] << 16); };
        const uint32_t n = word(0);
        if (n == 0 || n > MAX_CLAIMS) return fail("the child states no claims, or more than the region holds");
        Digest8 acc{};
        for (uint32_t k = 0; k < n; k++) {
            uint32_t c[32];
            for (size_t j = 0; j < 32; j++) c[j] = word(32 + 32 * k + j);
            if (!chain_claim(acc, c)) return fail("a claim's commitment is not a field element");
        }
        if (!std::equal(acc.begin(), acc.end(), s.begin() + 50)) return fail("the child's claims are not the ones its deferral node verified");
        return true;
    }
    // the 32 words of the claim a parent guest states about this child
    Digest8 input_commit() const { return p2_sponge8(root.pvs[2].data(), root.pvs[2].size()); }
    std::array<uint32_t, 32> claim_words() const {
        const std::vector<uint32_t>& s = root.pvs[2];
        Digest8 app, lc, r0;
        std::copy(s.begin(), s.begin() + 8, app.begin()), std::copy(s.begin() + 34, s.begin() + 42, lc.begin()), std::copy(s.begin() + 9, s.begin() + 17, r0.begin());
        const Digest8 ic = input_commit(), exe = p2_compress8(r0, Digest8{s[8], 0, 0, 0, 0, 0, 0, 0}), vm = p2_compress8(app, lc);
        std::array<uint32_t, 32> w;
        std::copy(ic.begin(), ic.end(), w.begin()), std::copy(exe.begin(), exe.end(), w.begin() + 8), std::copy(vm.begin(), vm.end(), w.begin() + 16);
        for (int k = 0; k < 8; k++) memcpy(&w[24 + k], public_values.data() + 4 * k, 4);
        return w;
    }
};
// acc <- compress(acc, chunk) over the five chunks of a claim: the three commitments as field elements, the public-value words as
// their 16-bit cells (a word need not be a field element)
inline bool chain_claim(Digest8& acc, const uint32_t claim[32]) {
    for (int k = 0; k < 24; k++)
        if (claim[k] >= 2013265921u) return false;   // a commitment word that is no field element backs nothing
    for (int c = 0; c < 3; c++) {
        Digest8 chunk;
        std::copy(claim + 8 * c, claim + 8 * c + 8, chunk.begin());
        acc = p2_compress8(acc, chunk);
    }
    for (int h = 0; h < 2; h++) {
        Digest8 chunk;
        for (int j = 0; j < 4; j++) chunk[2 * j] = claim[24 + 4 * h + j] & 0xffffu, chunk[2 * j + 1] = claim[24 + 4 * h + j] >> 16;
        acc = p2_compress8(acc, chunk);
    }
    return true;
}

class DeferralProver {
public:
    // `child_key`: the CHILD app's aggregation key (crates/prover/src/prover/mod.rs:213 `child_prover.load_agg_vk()`); node_params: the
    // parameters the deferral node and the join are proven under (mod.rs:239 `internal_params_with_100_bits_security`)
    // `child_region_index`: for a child app that itself defers (its key is a JOIN key: a bundle over batches) -- where the CHILD guest's
    // deferral region sits in its memory tree, ((2 << 26) | deferral_base(child exe) / 16) >> 9 (zkhip_vm::deferral_region_index)
    // `max_nodes` > 1: a task may have more children than one deferral node takes (a batch holds up to 45 chunks: crates/types/batch/src/
    // payload/v6.rs:10) -- up to max_nodes deferral nodes, each continuing the chain of the one before, and a FOLD over their proofs (a node
    // circuit with the chain as its chained state); the join then verifies the fold, for EVERY task of this prover (one key per app)
    static std::unique_ptr<DeferralProver> setup(const VerifyingKey& child_key, const zkhip_params& node_params, int device = 0, size_t max_children = 4,
                                                 uint32_t child_region_index = 0, size_t max_nodes = 1) {
        if (!child_key.is_aggregation_key() || child_key.airs.size() != 3 || child_key.airs[2].n_pvs != (child_key.join ? 58u : 50u))
            throw Error(Error::Setup, "deferral: the child key is not the aggregation key of a guest flow");
        if (child_key.join != (child_region_index != 0))
            throw Error(Error::Setup, child_key.join ? "deferral: the child key is a join key (the child app defers): its deferral region's place in its memory is needed"
                                                     : "deferral: a region index for a child app that does not defer");
        std::unique_ptr<DeferralProver> p(new DeferralProver());
        if (max_nodes == 0 || max_nodes > 8) throw Error(Error::Setup, "deferral: 1..8 deferral nodes per task");
        p->child_key_ = child_key, p->params_ = node_params, p->device_ = device, p->max_children_ = max_children, p->region_index_ = child_region_index, p->max_nodes_ = max_nodes;
        int rc = zkhip_ctx_create(device, &p->ctx_);
        if (rc != ZKHIP_OK) throw Error(Error::Keygen, "no gfx950 device for the HIP backend (zkhip_ctx_create returned " + std::to_string(rc) + ")");
        std::vector<zkhip_air> za = child_key.as_airs();
        zkhip_recursion_stmt st{};
        st.child_is_node = 3;
        st.region_index = child_region_index;
        rc = zkhip_recursion_build(&child_key.params, za.data(), za.size(), max_children, &st, &p->def_.circ);
        if (rc != ZKHIP_OK) throw Error(Error::Setup, std::string("deferral: cannot build the deferral node: ") + zkhip_recursion_last_error(nullptr));
        p->keygen(p->def_);
        if (max_nodes > 1) {   // the fold: children = deferral-node proofs, chained state = (chain before, chain after)
            std::vector<zkhip_air> da = p->def_.vk.as_airs();
            uint32_t air2[8], lo[8], hi[8];
            for (uint32_t k = 0; k < 8; k++) air2[k] = 2, lo[k] = k, hi[k] = 8 + k;
            zkhip_recursion_stmt fs{};
            fs.n_state = 8, fs.start_air = air2, fs.start_idx = lo, fs.end_air = air2, fs.end_idx = hi, fs.child_is_node = 0;
            rc = zkhip_recursion_build(&p->def_.vk.params, da.data(), da.size(), max_nodes, &fs, &p->fold_.circ);
            if (rc != ZKHIP_OK) throw Error(Error::Setup, std::string("deferral: cannot build the fold of deferral nodes: ") + zkhip_recursion_last_error(nullptr));
            p->keygen(p->fold_);
        }
        return p;
    }
    ~DeferralProver() {
        for (Node* n : {&def_, &fold_, &join_}) {
            for (void* d : n->d_traces)
                if (d) zkhip_free(ctx_, d);
            if (n->pk) zkhip_pk_destroy(ctx_, n->pk);
            if (n->circ) zkhip_recursion_destroy(n->circ);
        }
        if (ctx_) zkhip_ctx_destroy(ctx_);
    }
    DeferralProver(const DeferralProver&) = delete;
    DeferralProver& operator=(const DeferralProver&) = delete;
    size_t max_children() const { return max_children_ * max_nodes_; }   // per task
    size_t max_nodes() const { return max_nodes_; }
    uint32_t child_region_index() const { return region_index_; }
    const VerifyingKey& join_child_vk() const { return max_nodes_ > 1 ? fold_.vk : def_.vk; }   // what the join verifies beside the root
    const VerifyingKey& child_key() const { return child_key_; }
    const VerifyingKey& deferral_vk() const { return def_.vk; }
    // crates/integration/src/lib.rs:461-514: what the task and the prover need from the child proofs -- the input commitments (they go
    // into the guest's input stream), the deferral inputs, the chain's final value.  Every child is checked under the child key first.
    struct Data {
        std::vector<std::array<uint8_t, 32>> input_commits;
        std::vector<DeferralInput> inputs;
        Digest8 state{};
    };
    Data compute_deferral_data(const std::vector<const StarkProof*>& proofs) const {
        if (proofs.empty()) throw Error(Error::GenProof, "no child proofs to compute deferral data");
        if (proofs.size() > max_children()) throw Error(Error::GenProof, "deferral: " + std::to_string(proofs.size()) + " child proofs, this prover's deferral nodes take " + std::to_string(max_children()));
        Data d;
        for (size_t i = 0; i < proofs.size(); i++) {
            DeferralInput in = DeferralInput::from_stark_proof(*proofs[i], child_key_.join);
            std::string why;
            if (!child_key_.verify(in.root)) throw Error(Error::GenProof, "deferral: child proof " + std::to_string(i) + " does not verify under the child aggregation key");
            if (!child_key_.root_statement_matches(in.root.pvs[2], &why)) throw Error(Error::GenProof, "deferral: child proof " + std::to_string(i) + ": " + why);
            if (child_key_.join && !in.region_backs_the_chain(region_index_, &why)) throw Error(Error::GenProof, "deferral: child proof " + std::to_string(i) + ": " + why);
            const Digest8 ic = in.input_commit();
            std::array<uint8_t, 32> b;
            memcpy(b.data(), ic.data(), 32);
            d.input_commits.push_back(b);
            const std::array<uint32_t, 32> w = in.claim_words();
            if (!chain_claim(d.state, w.data())) throw Error(Error::GenProof, "deferral: a commitment word is not a field element");
            d.inputs.push_back(std::move(in));
        }
        return d;
    }
    // the deferral node(s) over the child roots: witness (host) + traces and proof (device); self-verified.  With max_nodes > 1: the
    // children in groups of max_children per node, every node's chain starting where the one before ended, then the fold over the nodes
    ChildProof prove_deferral(const std::vector<DeferralInput>& inputs) {
        if (inputs.empty() || inputs.size() > max_children()) throw Error(Error::GenProof, "deferral: 1.." + std::to_string(max_children()) + " child proofs");
        std::vector<ChildProof> nodes;
        uint32_t acc[8] = {};
        for (size_t g = 0; g < inputs.size(); g += max_children_) {
            const size_t n = std::min(max_children_, inputs.size() - g);
            std::vector<const uint8_t*> proofs;
            std::vector<size_t> lens;
            std::vector<std::vector<const uint32_t*>> rows(n);
            std::vector<const uint32_t* const*> pv_ptrs;
            std::vector<uint32_t> aux;
            for (size_t c = 0; c < n; c++) {
                const DeferralInput& in = inputs[g + c];
                proofs.push_back(in.root.proof.data()), lens.push_back(in.root.proof.size());
                for (const auto& v : in.root.pvs) rows[c].push_back(v.data());
                pv_ptrs.push_back(rows[c].data());
                const std::vector<uint32_t> a = in.aux();
                aux.insert(aux.end(), a.begin(), a.end());
            }
            std::vector<uint32_t> npv(zkhip_recursion_n_pvs(def_.circ));
            int rc = zkhip_recursion_witness_deferral(def_.circ, proofs.data(), lens.data(), pv_ptrs.data(), aux.data(), acc, n, npv.data());
            if (rc != ZKHIP_OK) throw Error(Error::GenProof, std::string("deferral: ") + zkhip_recursion_last_error(def_.circ));
            std::copy(npv.begin() + 8, npv.begin() + 16, acc);   // the next node continues here
            nodes.push_back(prove(def_, std::move(npv)));
        }
        if (max_nodes_ == 1) return std::move(nodes[0]);
        std::vector<const uint8_t*> proofs;
        std::vector<size_t> lens;
        std::vector<std::vector<const uint32_t*>> rows(nodes.size());
        std::vector<const uint32_t* const*> pv_ptrs;
        for (size_t c = 0; c < nodes.size(); c++) {
            proofs.push_back(nodes[c].proof.data()), lens.push_back(nodes[c].proof.size());
            for (const auto& v : nodes[c].pvs) rows[c].push_back(v.data());
            pv_ptrs.push_back(rows[c].data());
        }
        std::vector<uint32_t> fpv(zkhip_recursion_n_pvs(fold_.circ));
        int rc = zkhip_recursion_witness(fold_.circ, proofs.data(), lens.data(), pv_ptrs.data(), nodes.size(), fpv.data());
        if (rc != ZKHIP_OK) throw Error(Error::GenProof, std::string("deferral fold: ") + zkhip_recursion_last_error(fold_.circ));
        last_nodes_ = std::move(nodes);
        return prove(fold_, std::move(fpv));
    }
    const std::vector<ChildProof>& last_node_proofs() const { return last_nodes_; }   // (max_nodes > 1: the deferral nodes beneath the last fold)
    // the join for the parent's own aggregation key (built once per parent key)
    const VerifyingKey& join_vk(const VerifyingKey& own_key) {
        if (!join_.circ) {
            if (!own_key.is_aggregation_key()) throw Error(Error::Setup, "deferral: the parent key is not an aggregation key");
            std::vector<zkhip_air> za = own_key.as_airs(), zb = join_child_vk().as_airs();
            const zkhip_params& pb = join_child_vk().params;
            int rc = zkhip_recursion_build_join(&own_key.params, za.data(), za.size(), &pb, zb.data(), zb.size(), &join_.circ);
            if (rc != ZKHIP_OK) throw Error(Error::Setup, std::string("deferral: cannot build the join: ") + zkhip_recursion_last_error(nullptr));
            keygen(join_);
            // what a root under the join must state beneath it: the parent app (its internal commitment is pinned inside the join)
            join_.vk.leaf_commit = own_key.leaf_commit, join_.vk.app_digest = own_key.app_digest, join_.vk.join = true;
        }
        return join_.vk;
    }
    ChildProof prove_join(const VerifyingKey& own_key, const ChildProof& root, const ChildProof& deferral) {
        (void)join_vk(own_key);
        const uint8_t* proofs[2] = {root.proof.data(), deferral.proof.data()};
        const size_t lens[2] = {root.proof.size(), deferral.proof.size()};
        std::vector<const uint32_t*> r0, r1;
        for (const auto& v : root.pvs) r0.push_back(v.data());
        for (const auto& v : deferral.pvs) r1.push_back(v.data());
        const uint32_t* const* pv[2] = {r0.data(), r1.data()};
        std::vector<uint32_t> npv(zkhip_recursion_n_pvs(join_.circ));
        int rc = zkhip_recursion_witness(join_.circ, proofs, lens, pv, 2, npv.data());
        if (rc != ZKHIP_OK) throw Error(Error::GenProof, std::string("deferral join: ") + zkhip_recursion_last_error(join_.circ));
        return prove(join_, std::move(npv));
    }

private:
    DeferralProver() = default;
    struct Node {
        zkhip_recursion* circ = nullptr;
        zkhip_pk* pk = nullptr;
        VerifyingKey vk;
        std::vector<void*> d_traces;
    };
    VerifyingKey child_key_;
    zkhip_params params_{};
    int device_ = 0;
    size_t max_children_ = 4;
    uint32_t region_index_ = 0;
    size_t max_nodes_ = 1;
    zkhip_ctx* ctx_ = nullptr;
    Node def_, fold_, join_;
    std::vector<ChildProof> last_nodes_;
    void check(int rc) const {
        if (rc != ZKHIP_OK) throw Error(Error::GenProof, std::string("deferral: ") + zkhip_last_error(ctx_));
    }
    void keygen(Node& n) {
        n.vk.params = params_;
        std::vector<zkhip_air> na(3);
        for (size_t i = 0; i < 3; i++) {
            if (zkhip_recursion_air(n.circ, i, &na[i]) != ZKHIP_OK) throw Error(Error::Setup, "deferral: zkhip_recursion_air");
            AirDesc d;
            d.width = na[i].width, d.n_pvs = na[i].n_pvs, d.program.assign(na[i].program, na[i].program + na[i].program_len);
            d.has_prep = true, d.prep_log_height = na[i].log_height;
            n.vk.airs.push_back(std::move(d));
            n.vk.heights.push_back(na[i].log_height);
        }
        int rc = zkhip_keygen(ctx_, &n.vk.params, na.data(), 3, &n.pk);
        if (rc != ZKHIP_OK) throw Error(Error::Keygen, std::string("failed to generate STARK proving key: ") + zkhip_last_error(ctx_));
        for (size_t i = 0; i < 3; i++) {
            uint32_t c[8];
            check(zkhip_pk_prep_commitment(ctx_, n.pk, i, c));
            n.vk.airs[i].prep_commit.assign(c, c + 8);
            void* d = nullptr;
            check(zkhip_malloc(ctx_, (na[i].width << na[i].log_height) * 4, &d));
            n.d_traces.push_back(d);
        }
    }
    ChildProof prove(Node& n, std::vector<uint32_t> npv) {
        check(zkhip_recursion_tracegen(ctx_, n.circ, (uint32_t*)n.d_traces[0], (uint32_t*)n.d_traces[1], (uint32_t*)n.d_traces[2]));
        ChildProof out;
        out.pvs.resize(3);
        out.pvs[2] = std::move(npv);
        out.proof.resize(zkhip_proof_size(n.pk));
        const uint32_t* dt[3] = {(const uint32_t*)n.d_traces[0], (const uint32_t*)n.d_traces[1], (const uint32_t*)n.d_traces[2]};
        const uint32_t* pv[3] = {nullptr, nullptr, out.pvs[2].data()};
        size_t len = 0;
        check(zkhip_prove(ctx_, n.pk, dt, pv, out.proof.data(), out.proof.size(), &len));
        out.proof.resize(len);
        if (!n.vk.verify(out)) throw Error(Error::VerifyProof, "deferral: a node proof does not verify");
        return out;
    }
};

}  // namespace scroll_zkvm_hip
