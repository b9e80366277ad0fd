// tree_stream_tsan.cpp -- AggregationProver::TreeStreamT (include/zkhip_aggregation.hpp: the aggregation tree as a stream -- groupers, a
// witness thread and a device thread per level and slot, the greedy fold's queue, the self-verification threads) driven against a prover
// that makes STUB proofs, built with -fsanitize=thread by tests/test_tree_stream_tsan_cpu.py.  A stub proof states the range of segments
// beneath it; the mock's "witness generation" checks what a verifier circuit would: the children are adjacent, in order, finished, of the
// kinds announced, and the slot's witness buffer is free (the buffer protocol between the two threads of a slot).
//   usage: tree_stream_tsan <segments> <greedy 0|1> <slots> <seed> [fail at witness k: the error path -- every thread must still end]
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <random>
#include <thread>

#include "zkhip_aggregation.hpp"

using scroll_zkvm_hip::ChildProof;
using scroll_zkvm_hip::Error;

static void nap(std::mt19937& g, int max_us) { std::this_thread::sleep_for(std::chrono::microseconds(g() % (max_us + 1))); }

struct MockProver {
    struct Stats {
        size_t nodes = 0;
        std::vector<size_t> nodes_per_slot;
        double witness_seconds = 0, tracegen_prove_seconds = 0, verify_seconds = 0, keygen_seconds = 0, build_seconds = 0;
    } stats;
    struct Vk {
        bool verify(const ChildProof& p) const {   // reads every byte of the proof, as a verifier would
            unsigned s = 0;
            for (uint8_t b : p.proof) s += b;
            return p.pvs.size() == 1 && p.pvs[0].size() == 3 && s == sum_of(p.pvs[0]);
        }
    } vk;
    static unsigned sum_of(const std::vector<uint32_t>& v) {
        unsigned s = 0;
        const uint8_t* b = reinterpret_cast<const uint8_t*>(v.data());
        for (size_t i = 0; i < 4 * v.size(); i++) s += b[i];
        return s;
    }
    size_t slots;
    unsigned seed;
    std::vector<std::unique_ptr<std::mutex>> mus;
    std::mutex book;
    std::map<std::pair<size_t, size_t>, int> buffer;   // (level, slot) -> 0 free, 1 holds a witness
    std::atomic<size_t> witnesses{0}, proofs{0};
    size_t fail_at = 0;   // 0 = never
    MockProver(size_t n_slots, unsigned sd) : slots(n_slots), seed(sd) {
        for (size_t i = 0; i < slots; i++) mus.emplace_back(new std::mutex);
    }
    size_t n_slots() const { return slots; }
    std::mutex& slot_mutex(size_t slot) { return *mus.at(slot); }
    bool one_key() const { return true; }
    size_t first_root_layer() const { return 2; }
    size_t arity(size_t level, size_t shape = 0) const { return level == 0 ? (shape == 1 ? 2 : 4) : 5; }
    bool wrapped(size_t) const { return false; }
    ChildProof prove_wrapped(const std::vector<const ChildProof*>&, size_t, size_t = 0) { throw Error(Error::GenProof, "mock: no wrapped shapes"); }
    const Vk& level_vk(size_t, size_t = 0) { return vk; }
    // pvs[0] of a stub proof = [first segment, one past the last, depth (0 = a segment proof)]
    std::vector<uint32_t> witness_node(size_t level, const std::vector<const ChildProof*>& kids, size_t shape = 0, const std::vector<size_t>* kid_shapes = nullptr,
                                       size_t slot = 0, const std::vector<int>* kinds = nullptr) {
        (void)kid_shapes;
        const size_t serial = witnesses.fetch_add(1) + 1;
        std::mt19937 g(seed + 977 * (unsigned)serial);
        if (fail_at && serial == fail_at) throw Error(Error::GenProof, "mock: injected failure at witness " + std::to_string(serial));
        if (kids.empty() || kids.size() > arity(level, shape)) throw Error(Error::GenProof, "mock: a node of " + std::to_string(kids.size()) + " children at level " + std::to_string(level));
        if (kinds && kinds->size() != kids.size()) throw Error(Error::GenProof, "mock: kinds do not match the children");
        {
            std::lock_guard<std::mutex> lk(book);
            int& b = buffer[{level, slot}];
            if (b != 0) throw Error(Error::GenProof, "mock: the witness buffer of level " + std::to_string(level) + " slot " + std::to_string(slot) + " is not free");
            b = 1;
        }
        uint32_t depth = 0;
        for (size_t c = 0; c < kids.size(); c++) {
            const ChildProof& k = *kids[c];
            if (!vk.verify(k)) throw Error(Error::GenProof, "mock: a child does not verify");
            const uint32_t lo = k.pvs[0][0], hi = k.pvs[0][1], d = k.pvs[0][2];
            if (level == 0 && d != 0) throw Error(Error::GenProof, "mock: a leaf node over something that is not a segment proof");
            if (level > 0 && d == 0) throw Error(Error::GenProof, "mock: an internal node over a segment proof");
            if (kinds && ((*kinds)[c] == 0) != (d >= 2)) throw Error(Error::GenProof, "mock: a child is not of the kind announced");
            if (c && kids[c - 1]->pvs[0][1] != lo) throw Error(Error::GenProof, "mock: children are not adjacent");
            if (hi <= lo) throw Error(Error::GenProof, "mock: an empty child");
            depth = std::max(depth, d);
        }
        nap(g, 300);
        return {kids.front()->pvs[0][0], kids.back()->pvs[0][1], depth + 1};
    }
    void upload_witness(size_t level, size_t = 0, size_t slot = 0) {
        if (mus.at(slot)->try_lock()) {
            mus[slot]->unlock();
            throw Error(Error::GenProof, "mock: traces generated without the slot's device lock");
        }
        std::lock_guard<std::mutex> lk(book);
        int& b = buffer[{level, slot}];
        if (b != 1) throw Error(Error::GenProof, "mock: traces generated without a witness");
        b = 0;
    }
    ChildProof prove_uploaded(size_t, std::vector<uint32_t> npv, size_t = 0, size_t = 0) {
        std::mt19937 g(seed + 31 * (unsigned)proofs.fetch_add(1));
        nap(g, 400);
        return stub(std::move(npv));
    }
    static ChildProof stub(std::vector<uint32_t> pv) {
        ChildProof p;
        p.proof.assign(reinterpret_cast<const uint8_t*>(pv.data()), reinterpret_cast<const uint8_t*>(pv.data()) + 4 * pv.size());
        p.pvs.push_back(std::move(pv));
        return p;
    }
};

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atoi(argv[1]) : 37;
    const bool greedy = argc > 2 && atoi(argv[2]);
    const size_t slots = argc > 3 ? atoi(argv[3]) : 3;
    const unsigned seed = argc > 4 ? atoi(argv[4]) : 1;
    MockProver agg(slots, seed);
    agg.fail_at = argc > 5 ? atoi(argv[5]) : 0;
    using Stream = scroll_zkvm_hip::AggregationProver::TreeStreamT<MockProver>;
    std::vector<std::vector<ChildProof>> all;
    ChildProof root;
    try {
        Stream ts(agg, greedy);
        // three lanes push the segment proofs as they "finish": out of order, shapes in runs (a leaf node stops where the shape changes)
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = i;
        std::mt19937 g(seed);
        for (size_t i = 0; i + 1 < n; i++) std::swap(order[i], order[i + g() % std::min<size_t>(3, n - i)]);   // (a lane is at most two segments ahead)
        std::atomic<size_t> next{0};
        std::vector<std::thread> lanes;
        for (int t = 0; t < 3; t++)
            lanes.emplace_back([&, t] {
                std::mt19937 lg(seed * 7 + t);
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= n) return;
                    nap(lg, 200);
                    const size_t i = order[k];
                    ts.push(i, MockProver::stub({(uint32_t)i, (uint32_t)i + 1, 0}), (i / 11) % 3 == 2 ? 1 : 0);
                }
            });
        for (auto& t : lanes) t.join();
        root = ts.finish(n, &all);
    } catch (const std::exception& e) {
        std::printf("FAILED: %s\n", e.what());
        return 1;
    }
    if (root.pvs.size() != 1 || root.pvs[0][0] != 0 || root.pvs[0][1] != n || root.pvs[0][2] < 2) {
        std::printf("FAILED: the root covers [%u, %u) at depth %u, not [0, %zu) above the leaf nodes\n", root.pvs[0][0], root.pvs[0][1], root.pvs[0][2], n);
        return 1;
    }
    size_t leaves = 0, internal = 0, covered = 0;
    for (size_t l = 0; l < all.size(); l++)
        for (const ChildProof& p : all[l]) {
            if (p.pvs[0][2] == 1) leaves++, covered += p.pvs[0][1] - p.pvs[0][0];
            else internal++;
        }
    if (covered != n) {
        std::printf("FAILED: the leaf nodes cover %zu of %zu segments\n", covered, n);
        return 1;
    }
    std::printf("ok: %zu segments, %zu leaf nodes, %zu internal nodes, depth %u, %zu witnesses, %zu node proofs\n", n, leaves, internal, root.pvs[0][2], agg.witnesses.load(),
                agg.proofs.load());
    return 0;
}
