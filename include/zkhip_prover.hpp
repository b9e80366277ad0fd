// zkhip_prover.hpp -- host-side C++ mirror of the reference's prover API over the zkhip C ABI.
//
// The reference's crate `scroll-zkvm-prover` is Rust (no toolchain in this build environment), so
// the host side above include/zkhip.h is written in C++ with the same names, argument meaning and
// error behaviour (INTEGRATION.md shows the Rust shim a maintainer would add instead):
//
//   ProverConfig{path_app_exe, path_app_config}   crates/prover/src/prover/mod.rs:83-88
//   Prover::setup(config, name)                   mod.rs:93-104   (reads files only; keys are lazy)
//   Prover::reset()                               mod.rs:106-108  (drops device-resident keys)
//   Prover::gen_proof_universal(task, with_snark) mod.rs:287-309
//   Prover::gen_proof_stark(..)                   mod.rs:342-413  (prove, encode, SELF-VERIFY)
//   ProvingTask{serialized_witness, aggregated_proofs, fork_name, vk, identifier, input_commits}
//                                                 crates/types/src/task.rs:7-23
//   StarkProof{proof, user_pvs_proof, baseline, deferral_merkle_proofs, stat}, each byte field
//     carried in JSON as base64(bincode(Vec<u8>))   crates/types/src/proof.rs:52-67, utils.rs:20-39
//   StarkProofStat{total_cycles, execution_time_mills, proving_time_mills}   proof.rs:41-48
//   Error::{Io, Setup, Keygen, GenProof, VerifyProof, Custom}               crates/prover/src/error.rs:5-46
//   BatchProver (below Prover): what replaces the reference's SEQUENTIAL loop over the chunks of a batch
//     (crates/integration/src/testers/batch.rs:97-107): a queue of tasks over `inflight` Provers per GPU on every
//     listed GPU, each Prover still proving one task at a time (mod.rs:287 `&mut self`).
//
// What differs, because guest execution and trace generation are outside this path (SURVEY.md 8(f)
// f3): `path_app_exe` is an AIR-set file (the constraint bytecode of every chip) instead
// of an OpenVM vmexe, and `serialized_witness[i]` carries chip i's public values and trace.
// `path_app_config` is the reference's own openvm.toml: the five `[app_fri_params.fri_params]` keys
// are read from it (crates/circuits/chunk-circuit/openvm.toml:1-6).
#pragma once
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "zkhip.h"
#include "zkhip_vm.hpp"

namespace scroll_zkvm_hip {

struct Error : std::runtime_error {
    enum Kind { Io, Setup, Keygen, GenProof, VerifyProof, Custom } kind;
    Error(Kind k, const std::string& m) : std::runtime_error(m), kind(k) {}
};

// ---- wire helpers: base64 + bincode(Vec<u8>) ---------------------------------------------------
inline std::string base64_encode(const std::vector<uint8_t>& in) {
    static const char* T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    out.reserve((in.size() + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 2 < in.size(); i += 3) {
        uint32_t v = (in[i] << 16) | (in[i + 1] << 8) | in[i + 2];
        out += T[v >> 18], out += T[(v >> 12) & 63], out += T[(v >> 6) & 63], out += T[v & 63];
    }
    if (i + 1 == in.size()) {
        uint32_t v = in[i] << 16;
        out += T[v >> 18], out += T[(v >> 12) & 63], out += "==";
    } else if (i + 2 == in.size()) {
        uint32_t v = (in[i] << 16) | (in[i + 1] << 8);
        out += T[v >> 18], out += T[(v >> 12) & 63], out += T[(v >> 6) & 63], out += '=';
    }
    return out;
}
inline std::vector<uint8_t> base64_decode(const std::string& s) {
    auto val = [](char c) -> int {
        if (c >= 'A' && c <= 'Z') return c - 'A';
        if (c >= 'a' && c <= 'z') return c - 'a' + 26;
        if (c >= '0' && c <= '9') return c - '0' + 52;
        if (c == '+') return 62;
        if (c == '/') return 63;
        return -1;
    };
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (char c : s) {
        if (c == '=') break;
        int v = val(c);
        if (v < 0) throw Error(Error::Custom, "invalid base64");
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)(acc >> bits));
        }
    }
    return out;
}
// bincode v1 of a Vec<u8>: u64 little-endian length, then the bytes
inline std::vector<uint8_t> bincode_vec(const std::vector<uint8_t>& v) {
    std::vector<uint8_t> out(8 + v.size());
    uint64_t n = v.size();
    memcpy(out.data(), &n, 8);
    if (n) memcpy(out.data() + 8, v.data(), v.size());
    return out;
}
inline std::vector<uint8_t> unbincode_vec(const std::vector<uint8_t>& b) {
    if (b.size() < 8) throw Error(Error::Custom, "bincode: short buffer");
    uint64_t n;
    memcpy(&n, b.data(), 8);
    if (n != b.size() - 8) throw Error(Error::Custom, "bincode: length mismatch");
    return std::vector<uint8_t>(b.begin() + 8, b.end());
}

struct StarkProofStat {
    uint64_t total_cycles = 0, execution_time_mills = 0, proving_time_mills = 0;
};

struct StarkProof {
    std::vector<uint8_t> proof, user_pvs_proof, baseline, deferral_merkle_proofs;
    StarkProofStat stat;

    std::string to_json() const {
        std::ostringstream os;
        os << "{\"proof\":\"" << base64_encode(bincode_vec(proof)) << "\",\"user_pvs_proof\":\""
           << base64_encode(bincode_vec(user_pvs_proof)) << "\",\"baseline\":\"" << base64_encode(bincode_vec(baseline))
           << "\",\"deferral_merkle_proofs\":\"" << base64_encode(bincode_vec(deferral_merkle_proofs))
           << "\",\"stat\":{\"total_cycles\":" << stat.total_cycles
           << ",\"execution_time_mills\":" << stat.execution_time_mills
           << ",\"proving_time_mills\":" << stat.proving_time_mills << "}}";
        return os.str();
    }
    static StarkProof from_json(const std::string& js) {
        // position just after `"key" :` (whitespace tolerant), or npos
        auto after_key = [&](const char* key) -> size_t {
            std::string k = std::string("\"") + key + "\"";
            size_t p = js.find(k);
            if (p == std::string::npos) return p;
            p += k.size();
            while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n' || js[p] == '\r')) p++;
            if (p >= js.size() || js[p] != ':') return std::string::npos;
            p++;
            while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n' || js[p] == '\r')) p++;
            return p;
        };
        auto str_field = [&](const char* key) -> std::vector<uint8_t> {
            size_t p = after_key(key);
            if (p == std::string::npos) return {};  // `default` fields may be absent
            if (js[p] != '"') throw Error(Error::Custom, "malformed proof json");
            size_t e = js.find('"', p + 1);
            if (e == std::string::npos) throw Error(Error::Custom, "malformed proof json");
            return unbincode_vec(base64_decode(js.substr(p + 1, e - p - 1)));
        };
        auto num_field = [&](const char* key) -> uint64_t {
            size_t p = after_key(key);
            return p == std::string::npos ? 0 : std::stoull(js.substr(p));
        };
        StarkProof sp;
        sp.proof = str_field("proof");
        sp.user_pvs_proof = str_field("user_pvs_proof");
        sp.baseline = str_field("baseline");
        sp.deferral_merkle_proofs = str_field("deferral_merkle_proofs");
        sp.stat.total_cycles = num_field("total_cycles");
        sp.stat.execution_time_mills = num_field("execution_time_mills");
        sp.stat.proving_time_mills = num_field("proving_time_mills");
        if (sp.proof.empty()) throw Error(Error::Custom, "proof json has no `proof` field");
        return sp;
    }
};

struct ProvingTask {
    std::vector<std::vector<uint8_t>> serialized_witness;  // one entry per chip (see witness layout below)
    std::vector<StarkProof> aggregated_proofs;             // unused by leaf (chunk-like) tasks
    std::string fork_name;
    std::vector<uint8_t> vk;
    std::string identifier;
    std::vector<std::array<uint8_t, 32>> input_commits;

    // crates/prover/src/task/mod.rs:13-17,27-38: the guest's input (hint) stream -- every serialized witness as one
    // `write_bytes` item, then the input commitments when there are any.  The stream a zkhip_vm guest reads word by word
    // (environment call 2) frames an item as [byte length][bytes, zero-padded to a word boundary]; the commitments follow as
    // [count][8 words each].
    zkhip_vm::StdIn build_guest_input() const {
        zkhip_vm::StdIn in;
        build_guest_input_inner(in);
        return in;
    }
    void build_guest_input_inner(zkhip_vm::StdIn& in) const {
        auto word = [&](uint32_t v) {
            for (int k = 0; k < 4; k++) in.bytes.push_back((uint8_t)(v >> (8 * k)));
        };
        for (const auto& w : serialized_witness) {
            word((uint32_t)w.size());
            in.bytes.insert(in.bytes.end(), w.begin(), w.end());
            while (in.bytes.size() % 4) in.bytes.push_back(0);
        }
        if (!input_commits.empty()) {
            word((uint32_t)input_commits.size());
            for (const auto& c : input_commits) in.bytes.insert(in.bytes.end(), c.begin(), c.end());
        }
    }
};

// witness layout of chip i (little-endian u32 words):
//   [log_height, n_pvs, pvs..., trace column-major (width columns of 2^log_height canonical words)]
inline std::vector<uint8_t> encode_witness(unsigned log_height, const std::vector<uint32_t>& pvs,
                                           const std::vector<uint32_t>& trace_colmajor) {
    std::vector<uint32_t> w{log_height, (uint32_t)pvs.size()};
    w.insert(w.end(), pvs.begin(), pvs.end());
    w.insert(w.end(), trace_colmajor.begin(), trace_colmajor.end());
    std::vector<uint8_t> out(w.size() * 4);
    memcpy(out.data(), w.data(), out.size());
    return out;
}

// How the guest flow runs (include/zkhip_vm_flow.hpp): fields of the prover's configuration; the ZKHIP_* environment variables are
// overrides read in ONE place (FlowOptions::from_env, which ProverConfig's default uses).
struct FlowOptions {
    unsigned lanes = 3;               // segment provers in flight per device (measured 1 / 2 / 3: DESIGN.md 5)          [ZKHIP_LANES]
    bool verify_segments = false;     // host verification of every segment proof beside the proving (the leaf circuit's witness
                                      // generation checks every child; the root is always verified)                     [ZKHIP_VERIFY_SEGMENTS=1]
    bool one_shape = false;           // every segment under the app's full chip set instead of the smallest shape      [ZKHIP_ONE_SHAPE=1]
    bool lean_shape = true;           // one more shape: the base chips with a small memory system, for segments that stay in their
                                      // registers (zkhip_vm::SegmentShapes::lean_caps)                                   [ZKHIP_NO_LEAN_SHAPE=1 -> false]
    bool agg_nodes_100bit = false;    // node proofs under blow-up 4 / 44 queries instead of the app's parameters         [ZKHIP_AGG_100BIT=1]
    bool per_depth_keys = false;      // round 3's aggregation keys (one per tree depth) instead of ONE key               [ZKHIP_AGG_PER_DEPTH_KEYS=1]
    bool balanced_tree = false;       // the aggregation tree in AggregationPlan's fixed grouping instead of the greedy fold
                                      // (AggregationProver::TreeStream; the fold needs ONE key)                          [ZKHIP_TREE_BALANCED=1]
    bool agg_cli_greedy = false;      // `prove_cli prove-agg` (a tree over segment proofs handed in) with the greedy fold instead of the
                                      // plan's grouping (tests: errors inside the fold)                                  [ZKHIP_AGG_GREEDY=1]
    unsigned internal_arity = 3;      // children of an internal node of the aggregation tree: the reference's default
                                      // (crates/prover/src/prover/mod.rs:57-60).  With the gate chip's Horner rows (second session of round 5)
                                      // the tree's common heights are 2^20 gate rows / 2^17 permutations -- ONE leaf node over two segment
                                      // proofs of the base chips (0.80 M rows, 114 k permutations), an internal node over three node proofs
                                      // (122 k permutations); four or five children need 2^21 / 2^18 for every node of the tree (measured
                                      // 3 / 4 on the mixed and the Fibonacci guest: 5.2 / 4.6 MHz, 17.0 / 14.6 MHz; rounds 4 - 5 ran five
                                      // children at 2^21 / 2^18)                                                         [ZKHIP_INTERNAL_ARITY=n]
    unsigned agg_slots = 3;           // node pipelines (witness thread + device thread, own keys) per device: the node proofs of one
                                      // pipeline are proven one after the other -- with one pipeline the tree lags behind the segment
                                      // lanes and is finished alone on the GPU, a 21 ms proof at a time (measured 1 / 2 / 3 / 4:
                                      // DESIGN.md 15)                                                                     [ZKHIP_AGG_SLOTS=n]
    unsigned deferral_children = 4;   // child proofs per deferral node (<= 8), and deferral nodes per task (<= 8; more than one: every
    unsigned deferral_nodes = 1;      // task's nodes are FOLDED before the join -- 6 x 8 covers the reference's 45 chunks per batch)
                                      //                                                   [ZKHIP_DEFERRAL_CHILDREN, ZKHIP_DEFERRAL_NODES]
    bool trace_tree = false;          // a line per event of the aggregation tree on stderr (a measurement aid)           [ZKHIP_TREE_TRACE=1]
    unsigned wide_in_flight = 0;      // > 0: at most so many segment proofs of WRAPPED shapes (the chunk circuit's 51-chip set) in flight at a time,
                                      // whatever the lanes -- the segment waits at the head of the queue.  Measured on the mixed guest at frames of
                                      // 2^20 (three lanes): 2: 4.9 - 6.2 MHz against 6.0 - 6.5 without (docs/round5_b.md 9); off   [ZKHIP_WIDE_IN_FLIGHT=n]
    bool retry_segments = true;       // a segment proof that fails (a refused trace check, the device self-check, a device error) is made once
                                      // more from the same records before the task is given up; every retry is COUNTED (GuestStark::
                                      // segments_retried, the flow's JSON line).  Off: the first failure ends the task -- what the stress
                                      // loops run with, so that a retried wrong node cannot pass for a clean run          [ZKHIP_NO_RETRY=1 -> false]
    int fail_segment_once = -1;       // TEST: the first attempt at this segment's proof throws (tests of the retry and of its counter)   [ZKHIP_TEST_FAIL_SEGMENT=n]
    int exec_threads = -1;            // record passes of the PARALLEL executor (include/zkhip_vm_exec.hpp: a metered pass cuts the run and keeps
                                      // the memory tree, so many record passes replay the segments side by side; the reference meters first too,
                                      // crates/prover/src/utils/vm.rs:19).  0 = the serial executor on the feeding thread (rounds 3 - 5);
                                      // -1 = auto: a quarter of the CPUs this process may use, at least 2, at most 6       [ZKHIP_EXEC_THREADS=n]
    unsigned exec_threads_or_auto() const {
        if (exec_threads >= 0) return (unsigned)exec_threads;
        return std::min(6u, std::max(2u, zkhip_host_cpus() / 4));
    }
    std::vector<int> devices;         // GPUs of the node the flow spreads over (empty = the prover's device)             [ZKHIP_DEVICES=0,1,..]
    static FlowOptions from_env() {
        FlowOptions o;
        if (const char* e = getenv("ZKHIP_LANES")) o.lanes = std::max(1, atoi(e));
        o.verify_segments = getenv("ZKHIP_VERIFY_SEGMENTS") != nullptr;
        o.one_shape = getenv("ZKHIP_ONE_SHAPE") != nullptr;
        o.lean_shape = getenv("ZKHIP_NO_LEAN_SHAPE") == nullptr;
        o.agg_nodes_100bit = getenv("ZKHIP_AGG_100BIT") != nullptr;
        o.per_depth_keys = getenv("ZKHIP_AGG_PER_DEPTH_KEYS") != nullptr;
        o.balanced_tree = getenv("ZKHIP_TREE_BALANCED") != nullptr;
        o.trace_tree = getenv("ZKHIP_TREE_TRACE") != nullptr;
        o.retry_segments = getenv("ZKHIP_NO_RETRY") == nullptr;
        if (const char* e = getenv("ZKHIP_TEST_FAIL_SEGMENT")) o.fail_segment_once = atoi(e);
        if (const char* e = getenv("ZKHIP_EXEC_THREADS")) o.exec_threads = std::min(64, std::max(0, atoi(e)));
        if (const char* e = getenv("ZKHIP_WIDE_IN_FLIGHT")) o.wide_in_flight = (unsigned)std::max(0, atoi(e));
        o.agg_cli_greedy = getenv("ZKHIP_AGG_GREEDY") != nullptr;
        if (const char* e = getenv("ZKHIP_DEFERRAL_CHILDREN")) o.deferral_children = (unsigned)std::min(8, std::max(1, atoi(e)));
        if (const char* e = getenv("ZKHIP_DEFERRAL_NODES")) o.deferral_nodes = (unsigned)std::min(8, std::max(1, atoi(e)));
        if (const char* e = getenv("ZKHIP_INTERNAL_ARITY")) o.internal_arity = (unsigned)std::min(8, std::max(2, atoi(e)));
        if (const char* e = getenv("ZKHIP_AGG_SLOTS")) o.agg_slots = (unsigned)std::min(8, std::max(1, atoi(e)));
        if (const char* e = getenv("ZKHIP_DEVICES")) {
            std::stringstream ss(e);
            std::string item;
            while (std::getline(ss, item, ','))
                if (!item.empty()) o.devices.push_back(atoi(item.c_str()));
        }
        return o;
    }
};
struct ProverConfig {
    std::string path_app_exe;     // AIR-set file: [0x58414B5A, n_airs, {width, n_pvs, program_len, program...}...]
    std::string path_app_config;  // openvm.toml (FRI parameter block)
    FlowOptions flow = FlowOptions::from_env();
};

struct AirDesc {
    size_t width = 0, n_pvs = 0;
    std::vector<uint32_t> program;
    // preprocessed trace of the chip (fixes its height) and its commitment -- the app's verifying-key material,
    // the analogue of the committed exe / vk the reference checks at crates/verifier/src/verifier.rs:77-80
    bool has_prep = false;
    unsigned prep_log_height = 0;
    std::vector<uint32_t> prep;         // prep_width << prep_log_height canonical words, column-major (may be empty: verify only)
    std::vector<uint32_t> prep_commit;  // 8 canonical words (may be empty until the first keygen)
};
constexpr uint32_t AIRSET_MAGIC = 0x58414B5Au;     // v1: {width, n_pvs, prog_len, prog}
constexpr uint32_t AIRSET_MAGIC_V2 = 0x58414B5Bu;  // v2: v1 + {has_prep, [log_height, prep_len, prep..., has_commit, commit(8)]}

inline std::vector<uint8_t> encode_app_exe(const std::vector<AirDesc>& airs) {
    std::vector<uint32_t> w{AIRSET_MAGIC_V2, (uint32_t)airs.size()};
    for (const auto& a : airs) {
        w.push_back((uint32_t)a.width), w.push_back((uint32_t)a.n_pvs), w.push_back((uint32_t)a.program.size());
        w.insert(w.end(), a.program.begin(), a.program.end());
        w.push_back(a.has_prep ? 1u : 0u);
        if (a.has_prep) {
            w.push_back(a.prep_log_height), w.push_back((uint32_t)a.prep.size());
            w.insert(w.end(), a.prep.begin(), a.prep.end());
            w.push_back(a.prep_commit.size() == 8 ? 1u : 0u);
            if (a.prep_commit.size() == 8) w.insert(w.end(), a.prep_commit.begin(), a.prep_commit.end());
        }
    }
    std::vector<uint8_t> out(w.size() * 4);
    memcpy(out.data(), w.data(), out.size());
    return out;
}

// crates/prover/src/setup.rs:16,88 read_app_exe / read_app_config
inline std::vector<AirDesc> read_app_exe(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(Error::Setup, "failed to read or deserialize " + path + ": cannot open");
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (b.size() < 8 || b.size() % 4) throw Error(Error::Setup, "failed to read or deserialize " + path + ": truncated");
    std::vector<uint32_t> w(b.size() / 4);
    memcpy(w.data(), b.data(), b.size());
    if (w[0] != AIRSET_MAGIC && w[0] != AIRSET_MAGIC_V2) throw Error(Error::Setup, "failed to read or deserialize " + path + ": bad magic");
    const bool v2 = w[0] == AIRSET_MAGIC_V2;
    auto trunc = [&]() { return Error(Error::Setup, "failed to read or deserialize " + path + ": truncated"); };
    if (w[1] > w.size()) throw trunc();
    std::vector<AirDesc> airs(w[1]);
    size_t p = 2;
    for (auto& a : airs) {
        if (p + 3 > w.size()) throw trunc();
        a.width = w[p], a.n_pvs = w[p + 1];
        size_t len = w[p + 2];
        p += 3;
        if (len > w.size() || p + len > w.size()) throw trunc();
        a.program.assign(w.begin() + p, w.begin() + p + len);
        p += len;
        if (!v2) continue;
        if (p + 1 > w.size()) throw trunc();
        a.has_prep = w[p++] != 0;
        if (!a.has_prep) continue;
        if (p + 2 > w.size()) throw trunc();
        a.prep_log_height = w[p];
        size_t plen = w[p + 1];
        p += 2;
        if (a.prep_log_height > 27 || plen > w.size() || p + plen + 1 > w.size()) throw trunc();
        a.prep.assign(w.begin() + p, w.begin() + p + plen);
        p += plen;
        if (w[p++]) {
            if (p + 8 > w.size()) throw trunc();
            a.prep_commit.assign(w.begin() + p, w.begin() + p + 8);
            p += 8;
        }
    }
    return airs;
}
inline zkhip_params read_app_config(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw Error(Error::Setup, "failed to read or deserialize " + path + ": cannot open");
    std::map<std::string, uint32_t> kv;
    std::string line;
    while (std::getline(f, line)) {
        size_t eq = line.find('=');
        if (eq == std::string::npos || line[0] == '[' || line[0] == '#') continue;
        std::string k = line.substr(0, eq), v = line.substr(eq + 1);
        k.erase(0, k.find_first_not_of(" \t")), k.erase(k.find_last_not_of(" \t") + 1);
        try {
            kv[k] = (uint32_t)std::stoul(v);
        } catch (...) {
        }
    }
    auto need = [&](const char* k) -> uint32_t {
        auto it = kv.find(k);
        if (it == kv.end()) throw Error(Error::Setup, "failed to read or deserialize " + path + ": missing " + k);
        return it->second;
    };
    zkhip_params p;
    p.log_blowup = need("log_blowup");
    p.log_final_poly_len = need("log_final_poly_len");
    p.num_queries = need("num_queries");
    p.commit_pow_bits = need("commit_proof_of_work_bits");
    p.query_pow_bits = need("query_proof_of_work_bits");
    return p;
}

class Prover {
public:
    std::string prover_name;
    ProverConfig config;

    // mod.rs:93-104: reads the two files; device keys are built lazily at the first proof
    static Prover setup(const ProverConfig& cfg, const char* name = nullptr, int device = 0) {
        Prover p;
        p.config = cfg;
        p.prover_name = name ? name : "universal";
        p.airs_ = read_app_exe(cfg.path_app_exe);
        p.params_ = read_app_config(cfg.path_app_config);
        p.device_ = device;
        return p;
    }
    Prover(Prover&& o) noexcept { *this = std::move(o); }
    Prover& operator=(Prover&& o) noexcept {
        reset();
        prover_name = std::move(o.prover_name), config = std::move(o.config), airs_ = std::move(o.airs_);
        params_ = o.params_, device_ = o.device_, ctx_ = o.ctx_, pk_ = o.pk_, pk_heights_ = std::move(o.pk_heights_);
        o.ctx_ = nullptr, o.pk_ = nullptr;
        return *this;
    }
    ~Prover() { reset(); }

    // mod.rs:106-108 "Release OpenVM SDK resources": frees the device-resident keys and workspace
    void reset() {
        if (pk_) zkhip_pk_destroy(ctx_, pk_), pk_ = nullptr;
        if (ctx_) zkhip_ctx_destroy(ctx_), ctx_ = nullptr;
        pk_heights_.clear();
    }

    // mod.rs:287-309
    StarkProof gen_proof_universal(const ProvingTask& task, bool with_snark = false) {
        if (with_snark) throw Error(Error::GenProof, "the SNARK (EVM) wrap is outside the HIP backend's path");
        return gen_proof_stark(task);
    }

    // A task whose traces are resident on this Prover's device (the form trace generation on the device produces,
    // SURVEY.md 8(f) f3; also what `upload_witness` returns for a host-side ProvingTask)
    struct DeviceWitness {
        std::string identifier;
        std::vector<unsigned> heights;
        std::vector<std::vector<uint32_t>> pvs;
        std::vector<void*> d_traces;  // column-major Montgomery traces, one per chip
        uint64_t total_cells = 0;
        uint64_t upload_mills = 0;
    };

    // "execute" half of mod.rs:342-413: decode the witness, build keys for its shape, copy the traces to the device
    DeviceWitness upload_witness(const ProvingTask& task) {
        using clk = std::chrono::steady_clock;
        auto t0 = clk::now();
        if (task.serialized_witness.size() != airs_.size())
            throw Error(Error::GenProof, "task " + task.identifier + ": witness count does not match the app's chips");
        DeviceWitness dw;
        dw.identifier = task.identifier;
        dw.pvs.resize(airs_.size());
        std::vector<const uint32_t*> trace_host(airs_.size());
        for (size_t a = 0; a < airs_.size(); a++) {
            const auto& w = task.serialized_witness[a];
            if (w.size() < 8 || w.size() % 4) throw Error(Error::GenProof, "malformed witness");
            const uint32_t* words = reinterpret_cast<const uint32_t*>(w.data());
            unsigned lh = words[0];
            size_t n_pvs = words[1], have = w.size() / 4;
            if (lh > 27 || n_pvs != airs_[a].n_pvs || have != 2 + n_pvs + (airs_[a].width << lh))
                throw Error(Error::GenProof, "witness of chip " + std::to_string(a) + " has the wrong shape");
            if (airs_[a].has_prep && lh != airs_[a].prep_log_height)
                throw Error(Error::GenProof, "witness of chip " + std::to_string(a) + " does not have the height of its preprocessed trace");
            dw.heights.push_back(lh);
            dw.pvs[a].assign(words + 2, words + 2 + n_pvs);
            trace_host[a] = words + 2 + n_pvs;
            dw.total_cells += (uint64_t)airs_[a].width << lh;
        }
        ensure_keys(dw.heights);
        dw.d_traces.assign(airs_.size(), nullptr);
        try {
            for (size_t a = 0; a < airs_.size(); a++) {
                size_t n = airs_[a].width << dw.heights[a];
                check(zkhip_malloc(ctx_, n * 4, &dw.d_traces[a]), Error::GenProof);
                check(zkhip_h2d(ctx_, dw.d_traces[a], trace_host[a], n * 4), Error::GenProof);
                check(zkhip_to_monty(ctx_, (uint32_t*)dw.d_traces[a], n), Error::GenProof);
            }
            check(zkhip_sync(ctx_), Error::GenProof);
        } catch (...) {
            free_witness(dw);
            throw;
        }
        dw.upload_mills = std::chrono::duration_cast<std::chrono::milliseconds>(clk::now() - t0).count();
        return dw;
    }
    void free_witness(DeviceWitness& dw) {
        for (void*& d : dw.d_traces)
            if (d) zkhip_free(ctx_, d), d = nullptr;
    }

    // "prove" half: sdk.prove -> encode -> mandatory self-verify (mod.rs:355-411) from device-resident traces
    StarkProof prove_resident(const DeviceWitness& dw, bool self_verify = true) {
        using clk = std::chrono::steady_clock;
        if (dw.d_traces.size() != airs_.size()) throw Error(Error::GenProof, "device witness does not match the app's chips");
        ensure_keys(dw.heights);
        auto t1 = clk::now();
        StarkProof sp;
        sp.proof.resize(zkhip_proof_size(pk_));
        std::vector<const uint32_t*> dt(airs_.size()), pv(airs_.size());
        for (size_t a = 0; a < airs_.size(); a++) dt[a] = (const uint32_t*)dw.d_traces[a], pv[a] = dw.pvs[a].data();
        size_t len = 0;
        check(zkhip_prove(ctx_, pk_, dt.data(), pv.data(), sp.proof.data(), sp.proof.size(), &len), Error::GenProof);
        sp.proof.resize(len);
        auto t2 = clk::now();
        // user public values, chip by chip (canonical LE words)
        for (const auto& p : dw.pvs) {
            const uint8_t* b = reinterpret_cast<const uint8_t*>(p.data());
            sp.user_pvs_proof.insert(sp.user_pvs_proof.end(), b, b + p.size() * 4);
        }
        // baseline: what a verifier needs besides the app: the per-chip trace heights
        for (unsigned h : dw.heights) sp.baseline.push_back((uint8_t)h);
        sp.stat.total_cycles = dw.total_cells;
        sp.stat.execution_time_mills = dw.upload_mills;
        sp.stat.proving_time_mills = std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
        if (self_verify) verify_stark_proof(sp);  // mandatory self-check, as mod.rs:407-411
        return sp;
    }

    // mod.rs:312-338: execute the guest to get the cycle count (and check its public values); errors become Error::GenProof
    // as in mod.rs:318-319.  `records` receives the per-chip execution records the device trace generators take.
    zkhip_vm::ExecutionResult execute_and_check_with_full_result(const zkhip_vm::Exe& exe, const zkhip_vm::StdIn& stdin_,
                                                                 zkhip_vm::ExecRecords* records = nullptr, uint64_t max_cost = 0) const {
        try {
            const auto t = std::chrono::steady_clock::now();
            zkhip_vm::ExecutionResult r = zkhip_vm::execute_guest(exe, stdin_, max_cost, records);
            last_execution_time_mills_ = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t).count();
            return r;
        } catch (const zkhip_vm::Error& e) {
            throw Error(Error::GenProof, e.what());
        }
    }
    uint64_t execute_and_check(const zkhip_vm::Exe& exe, const zkhip_vm::StdIn& stdin_) const {
        return execute_and_check_with_full_result(exe, stdin_).total_cycle;
    }

    // mod.rs:342-413: execute (when the task comes with its guest: the cycle count and execution time go into the proof's
    // stat, mod.rs:398-404), decode + upload the witness, prove, encode, self-verify
    StarkProof gen_proof_stark(const ProvingTask& task, const zkhip_vm::Exe* guest = nullptr, const zkhip_vm::StdIn* guest_stdin = nullptr) {
        uint64_t cycles = 0, exec_mills = 0;
        if (guest) {
            static const zkhip_vm::StdIn no_input;
            cycles = execute_and_check_with_full_result(*guest, guest_stdin ? *guest_stdin : no_input).total_cycle;
            exec_mills = last_execution_time_mills_;
        }
        DeviceWitness dw = upload_witness(task);
        try {
            StarkProof sp = prove_resident(dw);
            free_witness(dw);
            if (guest) sp.stat.total_cycles = cycles, sp.stat.execution_time_mills = exec_mills;
            return sp;
        } catch (...) {
            free_witness(dw);
            throw;
        }
    }

    // crates/verifier/src/verifier.rs:38-85 for this backend's proofs; throws Error::VerifyProof
    void verify_stark_proof(const StarkProof& sp) const {
        if (sp.baseline.size() != airs_.size()) throw Error(Error::VerifyProof, "baseline does not match the app");
        std::vector<zkhip_air> za(airs_.size());
        std::vector<std::vector<uint32_t>> pvs(airs_.size());
        std::vector<const uint32_t*> pv(airs_.size());
        size_t off = 0;
        for (size_t a = 0; a < airs_.size(); a++) {
            // heights are prover-chosen data: a chip with a preprocessed trace has the height of its table, and no
            // height may leave the field's two-adicity (zkhip_verify re-checks the LogUp bus bound over all of them)
            if (airs_[a].has_prep && sp.baseline[a] != airs_[a].prep_log_height)
                throw Error(Error::VerifyProof, "baseline height of chip " + std::to_string(a) + " does not match its preprocessed trace");
            if (sp.baseline[a] + params_.log_blowup > 27) throw Error(Error::VerifyProof, "baseline height out of range");
            za[a] = make_air(a, sp.baseline[a]);
            if (airs_[a].has_prep && airs_[a].prep_commit.size() != 8)
                throw Error(Error::VerifyProof, "the app holds no commitment for the preprocessed trace of chip " + std::to_string(a));
            if (off + 4 * airs_[a].n_pvs > sp.user_pvs_proof.size()) throw Error(Error::VerifyProof, "short public values");
            pvs[a].resize(airs_[a].n_pvs);
            if (airs_[a].n_pvs) memcpy(pvs[a].data(), sp.user_pvs_proof.data() + off, 4 * airs_[a].n_pvs);
            off += 4 * airs_[a].n_pvs;
            pv[a] = pvs[a].data();
        }
        int rc = zkhip_verify(&params_, za.data(), za.size(), pv.data(), sp.proof.data(), sp.proof.size());
        if (rc != ZKHIP_OK) throw Error(Error::VerifyProof, "failed to verify proof: zkhip_verify returned " + std::to_string(rc));
    }

    const zkhip_params& params() const { return params_; }
    const std::vector<AirDesc>& airs() const { return airs_; }

private:
    Prover() = default;
    std::vector<AirDesc> airs_;
    zkhip_params params_{};
    int device_ = 0;
    zkhip_ctx* ctx_ = nullptr;
    mutable uint64_t last_execution_time_mills_ = 0;
    zkhip_pk* pk_ = nullptr;
    std::vector<unsigned> pk_heights_;

    zkhip_air make_air(size_t a, unsigned lh) const {
        zkhip_air z{airs_[a].program.data(), airs_[a].program.size(), lh, airs_[a].width, airs_[a].n_pvs, nullptr, nullptr};
        if (airs_[a].has_prep) {
            z.prep_trace = airs_[a].prep.empty() ? nullptr : airs_[a].prep.data();
            z.prep_commit = airs_[a].prep_commit.size() == 8 ? airs_[a].prep_commit.data() : nullptr;
        }
        return z;
    }
    void check(int rc, Error::Kind kind) const {
        if (rc != ZKHIP_OK) throw Error(kind, std::string(kind == Error::Keygen ? "failed to generate STARK proving key: " : "failed to generate proof: ") + zkhip_last_error(ctx_));
    }
    // lazily initialised like the reference's OnceLock<Sdk> (mod.rs:78,115-126); keys are per trace shape
    void ensure_keys(const std::vector<unsigned>& heights) {
        if (!ctx_) {
            int rc = zkhip_ctx_create(device_, &ctx_);
            if (rc != ZKHIP_OK) throw Error(Error::Keygen, "no gfx950 device for the HIP backend (zkhip_ctx_create returned " + std::to_string(rc) + ")");
        }
        if (pk_ && pk_heights_ == heights) return;
        if (pk_) zkhip_pk_destroy(ctx_, pk_), pk_ = nullptr;
        std::vector<zkhip_air> za(airs_.size());
        for (size_t a = 0; a < airs_.size(); a++) za[a] = make_air(a, heights[a]);
        check(zkhip_keygen(ctx_, &params_, za.data(), za.size(), &pk_), Error::Keygen);
        pk_heights_ = heights;
        // commitments of the preprocessed traces: adopt them, or -- when the app file already carries them -- insist
        // that the key just generated commits to the same tables (verifier.rs:77-80 does this for the exe commit)
        for (size_t a = 0; a < airs_.size(); a++) {
            if (!airs_[a].has_prep) continue;
            uint32_t c[8];
            check(zkhip_pk_prep_commitment(ctx_, pk_, a, c), Error::Keygen);
            if (airs_[a].prep_commit.size() == 8) {
                if (memcmp(c, airs_[a].prep_commit.data(), 32) != 0)
                    throw Error(Error::Keygen, "preprocessed trace of chip " + std::to_string(a) + " does not match the app's commitment");
            } else {
                airs_[a].prep_commit.assign(c, c + 8);
            }
        }
    }
};

// Task queue over several Provers: `inflight` per GPU on every GPU of `devices`.  Each lane is an ordinary Prover (own
// zkhip context = own HIP stream, own key + workspace) driven by its own host thread, so that the memory-bound stages
// of one proof overlap the VALU-bound hashing of another on the same GPU (bench.py measures the same arrangement) and
// independent segments / chunks spread over the GPUs of a node.  The reference proves the chunks of a
// batch one after another on one device (crates/integration/src/testers/batch.rs:97-107); results keep task order.
class BatchProver {
public:
    struct Stats {
        size_t proofs = 0;
        double seconds = 0, proofs_per_second = 0;
        std::vector<double> lane_seconds;
    };

    static BatchProver setup(const ProverConfig& cfg, unsigned inflight_per_gpu = 3, std::vector<int> devices = {0}) {
        if (inflight_per_gpu == 0 || devices.empty()) throw Error(Error::Setup, "BatchProver needs at least one lane");
        BatchProver b;
        for (int dev : devices)
            for (unsigned k = 0; k < inflight_per_gpu; k++)
                b.lanes_.push_back(Prover::setup(cfg, ("lane-" + std::to_string(dev) + "." + std::to_string(k)).c_str(), dev));
        return b;
    }
    size_t lanes() const { return lanes_.size(); }
    const Prover& lane(size_t i) const { return lanes_.at(i); }
    void reset() {
        for (auto& l : lanes_) l.reset();
    }

    // proves every task (witnesses are uploaded by the lane that takes the task); result i belongs to task i
    std::vector<StarkProof> prove_many(const std::vector<ProvingTask>& tasks, Stats* stats = nullptr) {
        std::vector<StarkProof> out(tasks.size());
        run(tasks.size(), stats, [&](Prover& p, size_t i) { out[i] = p.gen_proof_universal(tasks[i], false); });
        return out;
    }

    // throughput form (what bench.py times): the witness is uploaded ONCE per lane -- traces resident in HBM, as they are
    // when trace generation runs on the device -- and proven `n` times in total.  Every proof is self-verified (mod.rs:407-411)
    // by a pool of host threads BESIDE the lanes: a lane hands its finished proof over and starts the next one at once, so the
    // host-side check (tens of ms of scalar Poseidon2 per 1.1 MB proof) never leaves that lane's GPU stream idle.  The clock
    // stops when the last proof has been verified.
    Stats prove_repeated(const ProvingTask& task, size_t n, std::vector<uint8_t>* last_proof = nullptr) {
        using clk = std::chrono::steady_clock;
        std::vector<Prover::DeviceWitness> dws(lanes_.size());
        parallel_lanes([&](size_t l) { dws[l] = lanes_[l].upload_witness(task); });
        for (size_t l = 0; l < lanes_.size(); l++) (void)lanes_[l].prove_resident(dws[l]);  // first proof: scratch growth, code objects
        std::mutex mu;
        std::condition_variable cv;
        std::deque<StarkProof> queue;
        bool done = false;
        std::string verr;
        size_t verified = 0;
        StarkProof last;
        const Prover& checker = lanes_[0];
        std::vector<std::thread> pool;
        for (size_t v = 0; v < lanes_.size(); v++)
            pool.emplace_back([&]() {
                for (;;) {
                    StarkProof sp;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return done || !queue.empty(); });
                        if (queue.empty()) return;
                        sp = std::move(queue.front());
                        queue.pop_front();
                    }
                    try {
                        checker.verify_stark_proof(sp);
                        std::lock_guard<std::mutex> lk(mu);
                        verified++;
                    } catch (const std::exception& e) {
                        std::lock_guard<std::mutex> lk(mu);
                        if (verr.empty()) verr = e.what();
                    }
                }
            });
        Stats st;
        auto t0 = clk::now();
        std::string perr;
        try {
            run_indexed(n, &st, [&](Prover& p, size_t lane, size_t i) {
                StarkProof sp = p.prove_resident(dws[lane], /*self_verify=*/false);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (i + 1 == n) last = sp;
                    queue.push_back(std::move(sp));
                }
                cv.notify_one();
            });
        } catch (const std::exception& e) {
            perr = e.what();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            done = true;
        }
        cv.notify_all();
        for (auto& t : pool) t.join();
        st.seconds = std::chrono::duration<double>(clk::now() - t0).count();
        st.proofs_per_second = st.seconds > 0 ? n / st.seconds : 0;
        for (size_t l = 0; l < lanes_.size(); l++) lanes_[l].free_witness(dws[l]);
        if (!perr.empty()) throw Error(Error::GenProof, perr);
        if (!verr.empty() || verified != n) throw Error(Error::VerifyProof, verr.empty() ? "not every proof was verified" : verr);
        if (last_proof && n) *last_proof = last.proof;
        return st;
    }

private:
    BatchProver() = default;
    std::vector<Prover> lanes_;

    template <class F>
    void parallel_lanes(F&& f) {
        std::vector<std::thread> th;
        std::vector<std::string> errs(lanes_.size());
        std::vector<int> kinds(lanes_.size(), -1);
        for (size_t l = 0; l < lanes_.size(); l++)
            th.emplace_back([&, l]() {
                try {
                    f(l);
                } catch (const Error& e) {
                    errs[l] = e.what(), kinds[l] = (int)e.kind;
                } catch (const std::exception& e) {
                    errs[l] = e.what(), kinds[l] = (int)Error::Custom;
                }
            });
        for (auto& t : th) t.join();
        for (size_t l = 0; l < lanes_.size(); l++)
            if (kinds[l] >= 0) throw Error((Error::Kind)kinds[l], errs[l]);
    }
    template <class F>
    void run_indexed(size_t n, Stats* stats, F&& f) {
        using clk = std::chrono::steady_clock;
        std::atomic<size_t> next{0};
        std::vector<double> lane_s(lanes_.size(), 0.0);
        auto t0 = clk::now();
        parallel_lanes([&](size_t l) {
            auto tl = clk::now();
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= n) break;
                f(lanes_[l], l, i);
            }
            lane_s[l] = std::chrono::duration<double>(clk::now() - tl).count();
        });
        if (stats) {
            stats->proofs = n;
            stats->seconds = std::chrono::duration<double>(clk::now() - t0).count();
            stats->proofs_per_second = stats->seconds > 0 ? n / stats->seconds : 0;
            stats->lane_seconds = lane_s;
        }
    }
    template <class F>
    void run(size_t n, Stats* stats, F&& f) {
        run_indexed(n, stats, [&](Prover& p, size_t, size_t i) { f(p, i); });
    }
};

// ---- aggregation-tree scheduling (the part that does not need the recursion AIRs) -----------------------
// crates/prover/src/prover/mod.rs:57-60 DEFAULT_AGG_TREE_CONFIG: leaf nodes verify up to 4 segment proofs, internal nodes up to 3
// children, until one root remains.  The recursion (verifier) circuits themselves live in un-vendored OpenVM crates; what can
// be built here is the SHAPE and the SCHEDULE: which node consumes which proofs, and the nodes of a level proven as
// independent tasks over the lanes / GPUs of a BatchProver as soon as the level below is complete.
struct AggregationTreeConfig {
    unsigned num_children_internal = 3, num_children_leaf = 4;
    // ONE aggregation key (mod.rs:147-170 `agg_vk`, crates/verifier/src/verifier.rs:96-111): the root of every tree is a proof of the
    // one internal circuit, so a tree has at least one internal level above its leaves -- a single leaf node is wrapped, as the
    // reference's internal verifier wraps a single leaf proof.  false = the per-depth keys of round 3 (every level hard-wires the key
    // of the level below; the root key depends on the depth).
    bool one_key = true;
};
struct AggregationPlan {
    struct Node {
        std::vector<size_t> children;  // level 0: indices of segment proofs; level l > 0: indices of nodes of level l - 1
    };
    size_t n_segments = 0;
    std::vector<std::vector<Node>> levels;  // levels.back() has exactly one node: the root

    static AggregationPlan build(size_t n_segments, AggregationTreeConfig cfg = {}) {
        if (n_segments == 0 || cfg.num_children_leaf == 0 || cfg.num_children_internal < 2)
            throw Error(Error::Setup, "aggregation plan needs segments, a leaf arity >= 1 and an internal arity >= 2");
        AggregationPlan p;
        p.n_segments = n_segments;
        auto group = [](size_t n, unsigned arity) {
            std::vector<Node> lv;
            for (size_t i = 0; i < n; i += arity) {
                Node nd;
                for (size_t k = i; k < n && k < i + arity; k++) nd.children.push_back(k);
                lv.push_back(std::move(nd));
            }
            return lv;
        };
        p.levels.push_back(group(n_segments, cfg.num_children_leaf));
        while (p.levels.back().size() > 1 || (cfg.one_key && p.levels.size() < 2)) p.levels.push_back(group(p.levels.back().size(), cfg.num_children_internal));
        return p;
    }
    size_t n_nodes() const {
        size_t n = 0;
        for (const auto& l : levels) n += l.size();
        return n;
    }
};

// The tree WITHOUT a fixed shape (one aggregation key: the internal circuit takes leaf-node and internal-node proofs in any mix, so any
// grouping of ADJACENT proofs is a valid node).  FoldLine holds the finished node proofs ("pieces", each over a range [lo, hi) of leaf
// nodes) and says what to fold next: the leftmost `arity` adjacent pieces, whenever there are such; once the number of leaf nodes is
// known, everything has arrived and nothing is in flight, what is left (fewer than `arity` pieces) becomes the root -- a single leaf
// node gets an internal node of its own, so the root is always a proof of the internal circuit.  Every fold but the last is full, so the
// number of internal nodes is the balanced tree's, ceil((m - 1) / (arity - 1)); the SHAPE follows the arrival times: pieces that wait
// fold among themselves (balanced), pieces that find the running fold finished join it (a comb: root = (everything before, the last
// leaf nodes)), which leaves ONE leaf node and the root to do after the last segment proof instead of a node per level.
// Pure bookkeeping (no proofs touched: a piece is a tag); the caller serialises calls.
struct FoldLine {
    struct Fold {
        size_t lo = 0, hi = 0, depth = 0;     // the leaf nodes beneath, the height above them
        std::vector<const void*> kids;        // the pieces' tags, in order
        std::vector<int> kinds;               // per child: 0 internal, j + 1 leaf circuit j
    };
    explicit FoldLine(size_t arity) : arity_(arity < 2 ? 2 : arity) {}
    // a finished piece: a leaf node (kind j + 1, depth 0, [k, k + 1)) or a fold's proof (done)
    void add(size_t lo, size_t hi, int kind, size_t depth, const void* tag) {
        Piece p;
        p.hi = hi, p.kind = kind, p.depth = depth, p.tag = tag;
        line_[lo] = p;
    }
    void set_total(size_t n_leaf_nodes) { total_ = n_leaf_nodes; }
    // the next fold to start, if there is one (its pieces leave the line; it counts as in flight until done())
    bool next(Fold* out) {
        if (root_) return false;
        auto first = line_.end();
        size_t run = 0;
        for (auto it = line_.begin(); it != line_.end(); ++it) {
            auto prev = it;
            if (run && (--prev)->second.hi == it->first) run++;
            else run = 1, first = it;
            if (run == arity_) break;
        }
        size_t take = run == arity_ ? arity_ : 0;
        if (!take && total_ != SIZE_MAX && inflight_ == 0 && !line_.empty()) {
            size_t end = 0;
            bool contiguous = true;
            for (const auto& kv : line_) contiguous = contiguous && kv.first == end, end = kv.second.hi;
            if (contiguous && end == total_) {
                if (line_.size() == 1 && line_.begin()->second.kind == 0) {
                    root_ = line_.begin()->second.tag, depth_ = line_.begin()->second.depth;
                    return false;
                }
                first = line_.begin(), take = line_.size();
            }
        }
        if (!take) return false;
        *out = Fold{};
        out->lo = first->first;
        auto it = first;
        for (size_t c = 0; c < take; c++, ++it) {
            out->kids.push_back(it->second.tag), out->kinds.push_back(it->second.kind);
            out->hi = it->second.hi, out->depth = std::max(out->depth, it->second.depth + 1);
        }
        line_.erase(first, it);
        inflight_++, folds_++;
        return true;
    }
    void done(const Fold& f, const void* tag) {
        inflight_--;
        add(f.lo, f.hi, 0, f.depth, tag);
    }
    const void* root() const { return root_; }   // set by the next() that finds the line reduced to one internal piece over everything
    size_t root_depth() const { return depth_; }
    size_t folds() const { return folds_; }

private:
    struct Piece {
        size_t hi = 0, depth = 0;
        int kind = 0;
        const void* tag = nullptr;
    };
    size_t arity_, total_ = SIZE_MAX, inflight_ = 0, folds_ = 0, depth_ = 0;
    std::map<size_t, Piece> line_;   // by lo; adjacent = consecutive entries with a.hi == b.lo
    const void* root_ = nullptr;
};

// Proves an aggregation plan on a BatchProver set up with the AGGREGATION app: `make_task(level, node, child_proofs)` builds
// the ProvingTask of a node from its children's proofs (for the reference that is the leaf / internal verifier circuit's
// input, crates/integration/src/lib.rs:461-514); the nodes of one level are queued over the lanes like independent
// segments, a level starts when the previous one is complete.  Returns the root proof; `all` (optional) receives every
// level's proofs.
template <class MakeTask>
inline StarkProof prove_aggregation(BatchProver& agg, const AggregationPlan& plan, const std::vector<StarkProof>& segment_proofs,
                                    MakeTask&& make_task, std::vector<std::vector<StarkProof>>* all = nullptr) {
    if (segment_proofs.size() != plan.n_segments) throw Error(Error::GenProof, "aggregation: segment proof count does not match the plan");
    std::vector<StarkProof> below = segment_proofs;
    for (size_t l = 0; l < plan.levels.size(); l++) {
        std::vector<ProvingTask> tasks;
        for (size_t n = 0; n < plan.levels[l].size(); n++) {
            std::vector<const StarkProof*> kids;
            for (size_t c : plan.levels[l][n].children) kids.push_back(&below.at(c));
            ProvingTask t = make_task(l, n, kids);
            if (t.identifier.empty()) t.identifier = "agg-" + std::to_string(l) + "-" + std::to_string(n);
            tasks.push_back(std::move(t));
        }
        below = agg.prove_many(tasks);
        if (all) all->push_back(below);
    }
    return below.at(0);
}

// crates/verifier/src/verifier.rs:20-85 UniversalVerifier: holds only verifying material (the app's AIR programs,
// FRI parameters and preprocessed commitments -- no tables, no device) and checks StarkProofs.
class UniversalVerifier {
public:
    // verifier.rs:28-36 setup(path_vm_config, path_root_committed_exe ...): here the app file + openvm.toml
    static UniversalVerifier setup(const std::string& path_app_exe, const std::string& path_app_config) {
        return UniversalVerifier(Prover::setup(ProverConfig{path_app_exe, path_app_config}, "verifier"));
    }
    // verifier.rs:38-85 verify_stark_proof_with_vk: throws Error::VerifyProof
    void verify_stark_proof(const StarkProof& sp) const { p_.verify_stark_proof(sp); }
    bool verify(const StarkProof& sp) const noexcept {
        try {
            p_.verify_stark_proof(sp);
            return true;
        } catch (...) {
            return false;
        }
    }

private:
    explicit UniversalVerifier(Prover&& p) : p_(std::move(p)) {}
    Prover p_;
};

}  // namespace scroll_zkvm_hip
