// quotient_jit.hpp -- keygen-time code generation for the constraint kernel (K5).
//
// The reference's engines evaluate each chip's constraint DAG with a generic evaluator
// (SURVEY.md 2.3 K5).  On MI355X the AIR set of a proving key is fixed, so at keygen the DAG is
// emitted as straight-line HIP (one SSA value per interior node, trace cells / public values /
// constants / selectors as direct operands) and compiled for gfx950 with hipRTC.  Compared with
// the per-lane interpreter (k_quotient, kept as the fallback) this removes instruction fetch and
// decode, keeps every intermediate in a VGPR instead of LDS, and lets the compiler schedule the
// Montgomery products of independent constraints against each other.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <atomic>
#include <mutex>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "air_compile.hpp"

namespace zk {

// Kernel parameters (all geometry that is fixed at keygen is baked into the source as constants):
//   lde, q, pvs, apow, tw_fwd, zh, inv_zh, tab (pointers), gen, w_n_inv, tw_shift (u32),
//   perm (permutation LDE), lchal (interaction challenges), expo (exposed sum) -- null without interactions,
//   prep (preprocessed LDE) -- null without a preprocessed trace
struct QuotJitParams {
    const uint32_t* lde;
    uint32_t* q;
    const uint32_t* pvs;
    const uint32_t* apow;
    const uint32_t* tw_fwd;
    const uint32_t* zh;
    const uint32_t* inv_zh;
    uint32_t gen, w_n_inv, tw_shift;
};

inline const char* quot_jit_preamble() {
    return R"JIT(
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#define P 0x78000001u
#define NEG_MU 0x77ffffffu
#define ONE 0x0ffffffeu
#define R2 1172168163u
// uniform read-only tables are read through the constant address space so they become scalar loads
typedef const __attribute__((address_space(4))) uint32_t* cptr;
__device__ __forceinline__ uint32_t red(uint32_t x) {
    uint32_t y;
    asm("v_subrev_co_u32 %0, vcc, %2, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "=&v"(y) : "v"(x), "i"(P) : "vcc");
    return y;
}
__device__ __forceinline__ uint32_t mml(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t m = (uint32_t)t * NEG_MU;
    uint64_t s = t + (uint64_t)m * P;
    return (uint32_t)(s >> 32);
}
__device__ __forceinline__ uint32_t mmul(uint32_t a, uint32_t b) { return red(mml(a, b)); }
// signed Montgomery product (babybear.hpp): operands in (-p, p), result in (-0.97 p, 0.97 p), no conditional step; a
// product whose only users are products stays in this form, SC() brings one back to [0, p)
#define MU 0x88000001u
typedef int int32_t;
typedef long long int64_t;
__device__ __forceinline__ int32_t sml(int32_t a, int32_t b) {
    int64_t t = (int64_t)a * b;
    int32_t m = (int32_t)((uint32_t)t * MU);
    int64_t s = t - (int64_t)m * (int32_t)P;
    return (int32_t)(s >> 32);
}
__device__ __forceinline__ uint32_t SC(int32_t d) {
    uint32_t y;
    asm("v_add_co_u32 %0, vcc, %2, %1\n\tv_cndmask_b32 %0, %1, %0, vcc" : "=&v"(y) : "v"(d), "i"(P) : "vcc");
    return y;
}
__device__ __forceinline__ uint32_t madd(uint32_t a, uint32_t b) { return red(a + b); }
__device__ __forceinline__ uint32_t msub(uint32_t a, uint32_t b) {
    uint32_t d, e;
    asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_add_u32 %1, %4, %0\n\tv_cndmask_b32 %0, %0, %1, vcc"
        : "=&v"(d), "=&v"(e) : "v"(a), "v"(b), "i"(P) : "vcc");
    return d;
}
__device__ __forceinline__ uint32_t mneg(uint32_t a) { return a == 0 ? 0 : P - a; }
__device__ __forceinline__ uint32_t minv(uint32_t a) {
    uint32_t r = ONE; uint32_t e = P - 2;
    while (e) { if (e & 1) r = mmul(r, a); a = mmul(a, a); e >>= 1; }
    return r;
}
typedef const __attribute__((address_space(1))) uint32_t* gptr;
// trace cell (column c, this / next row): uniform column base (scalar) + 32-bit per-lane byte offset
#define LD(c, off) (*(gptr)((const __attribute__((address_space(1))) char*)(ldep + (size_t)(c) * M) + (off)))
#define LDP(c, off) (*(gptr)((const __attribute__((address_space(1))) char*)(permp + (size_t)(c) * M) + (off)))
#define LDQ(c, off) (*(gptr)((const __attribute__((address_space(1))) char*)(prepp + (size_t)(c) * M) + (off)))
#define PV(i) ((cptr)pvs)[i]
__device__ __forceinline__ uint32_t mred64(uint64_t t) {
    uint32_t m = (uint32_t)t * NEG_MU;
    uint64_t s = (uint64_t)(uint32_t)t + (uint64_t)m * P;
    uint64_t r = (t >> 32) + (s >> 32);
    if (r >= 2ull * P) r -= 2ull * P;
    return red((uint32_t)r);
}
// acc += alpha^(n-1-k) * v.  The four coefficient products are summed in 64 bits over four
// constraints (4 p^2 < 2^64) and Montgomery-reduced once per group.
// ... and the group sums are banked as split 64-bit halves (hi += t >> 32, lo += low word: 4 full-rate instructions
// per coordinate per group instead of a Montgomery reduction + modular add, 13); FINISH reduces them once.
#define FLUSH { h0 += w0 >> 32; l0 += (uint32_t)w0; h1 += w1 >> 32; l1 += (uint32_t)w1; h2 += w2 >> 32; l2 += (uint32_t)w2; \
    h3 += w3 >> 32; l3 += (uint32_t)w3; w0 = w1 = w2 = w3 = 0; cnt = 0; }
// (hi * 2^32 + lo) * 2^-32 mod p for banks below 2^63
__device__ __forceinline__ uint32_t lazy_reduce(uint64_t hi, uint64_t lo) {
    const uint32_t h = mmul(mred64(hi & 0xffffffffull), R2);
    const uint32_t hh = mmul(mmul(mred64(hi >> 32), R2), R2);
    const uint32_t l = madd(mred64(lo & 0xffffffffull), mmul(mred64(lo >> 32), R2));
    return madd(madd(h, hh), l);
}
#define FINISH { FLUSH acc0 = lazy_reduce(h0, l0); acc1 = lazy_reduce(h1, l1); acc2 = lazy_reduce(h2, l2); acc3 = lazy_reduce(h3, l3); }
#define ACC(k, v) { const uint32_t _v = (v); cptr ap = (cptr)apow + 4 * (k); \
    w0 += (uint64_t)ap[0] * _v; w1 += (uint64_t)ap[1] * _v; w2 += (uint64_t)ap[2] * _v; w3 += (uint64_t)ap[3] * _v; \
    if (++cnt == 4) FLUSH }
)JIT";
}

// ---- shape classes ----------------------------------------------------------------------------
// Constraints of real AIRs repeat a handful of expression shapes (one per limb / per column / per
// bus).  Each constraint is linearised as its own small DAG; constraints whose DAGs are equal up to
// the identity of their leaves (which column, which constant, which public value) form a class.
// The kernel has ONE loop per class: the body is straight-line code for the shape, the leaves come
// from a parameter table read with scalar loads.  Code stays a few KB (instruction-cache resident,
// unlike one straight-line stream for all constraints) and every trace cell of an instance is loaded
// before the arithmetic starts.
struct JitEntry {
    uint32_t op;     // AirOp
    uint32_t a, b;   // local indices for interior nodes; rot for VAR
};
struct JitClass {
    std::vector<JitEntry> entries;           // local post-order DAG of the shape
    std::vector<uint32_t> param_entry;       // entries that take a parameter (VAR col / CONST / PUB)
    std::vector<std::vector<uint32_t>> inst; // per instance: [constraint index, params...]
    size_t table_off = 0;                    // word offset of the class in the parameter table
};

inline bool quot_jit_classify(const AirProgram& p, std::vector<JitClass>* classes, std::string* msg) {
    std::map<std::string, size_t> by_sig;
    size_t expanded = 0;
    for (uint32_t k = 0; k < p.n_cons; k++) {
        std::vector<JitEntry> ent;
        std::vector<uint32_t> params, param_entry;
        std::map<uint32_t, uint32_t> local_of;  // global node -> local index
        // iterative post-order
        std::vector<std::pair<uint32_t, int>> stack{{p.cons[k], 0}};
        while (!stack.empty()) {
            auto [i, st] = stack.back();
            stack.pop_back();
            if (local_of.count(i)) continue;
            const uint32_t op = p.nodes[3 * i], a = p.nodes[3 * i + 1], b = p.nodes[3 * i + 2];
            if (op <= A_TRANS || op >= A_PERM) {
                local_of[i] = (uint32_t)ent.size();
                ent.push_back({op, (op == A_VAR || op == A_PERM || op == A_PREP) ? b : 0u, 0u});
                if (op == A_VAR || op == A_CONST || op == A_PUB || op >= A_PERM) {
                    param_entry.push_back((uint32_t)ent.size() - 1);
                    params.push_back(op == A_CONST ? to_monty(a) : a);
                }
                continue;
            }
            if (st == 0) {
                stack.push_back({i, 1});
                if (op != A_NEG && !local_of.count(b)) stack.push_back({b, 0});
                if (!local_of.count(a)) stack.push_back({a, 0});
            } else {
                local_of[i] = (uint32_t)ent.size();
                ent.push_back({op, local_of[a], op != A_NEG ? local_of[b] : 0u});
            }
        }
        expanded += ent.size();
        std::string sig;
        sig.reserve(ent.size() * 12);
        for (const auto& e : ent) sig += std::to_string(e.op) + "," + std::to_string(e.a) + "," + std::to_string(e.b) + ";";
        auto it = by_sig.find(sig);
        if (it == by_sig.end()) {
            it = by_sig.emplace(sig, classes->size()).first;
            classes->push_back(JitClass());
            classes->back().entries = ent;
            classes->back().param_entry = param_entry;
        }
        std::vector<uint32_t> row{k};
        row.insert(row.end(), params.begin(), params.end());
        (*classes)[it->second].inst.push_back(row);
    }
    if (classes->size() > 96 || expanded > 16 * (size_t)p.n_nodes + 1024) {
        *msg = "AIR does not compress into shape classes (" + std::to_string(classes->size()) + " classes)";
        return false;
    }
    return true;
}

inline void quot_jit_prologue(std::ostringstream& os, unsigned lh, unsigned b, unsigned qd) {
    os << quot_jit_preamble();
    // The trace height is a launch parameter (H = log2 of the LDE height, NQROWS = the first N * qd rows of the LDE, where the
    // quotient lives): one compiled kernel serves an AIR at every height -- continuation segments, tasks of different sizes and
    // re-keying after a reset pay the hipRTC compile once per (AIR, blow-up), not once per height.
    (void)lh, (void)qd;
    os << "#define B " << b << "u\n#define M ((size_t)1 << H)\n";
    os << R"JIT(
extern "C" __global__ __launch_bounds__(256) void quot_jit(const uint32_t* __restrict__ lde, uint32_t* __restrict__ q,
        const uint32_t* __restrict__ pvs, const uint32_t* __restrict__ apow, const uint32_t* __restrict__ tw_fwd,
        const uint32_t* __restrict__ zh_t, const uint32_t* __restrict__ inv_zh_t, const uint32_t* __restrict__ tab,
        uint32_t gen, uint32_t w_n_inv, uint32_t tw_shift, const uint32_t* __restrict__ perm,
        const uint32_t* __restrict__ lchal, const uint32_t* __restrict__ expo, const uint32_t* __restrict__ prep,
        uint32_t H, uint32_t NQROWS) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= NQROWS) return;
    const uint32_t i = __brev(r) >> (32 - H);
    const uint32_t rn = __brev((i + (1u << B)) & ((1u << H) - 1u)) >> (32 - H);
    const uint32_t halfm = 1u << (H - 1);
    const uint32_t wi = i < halfm ? tw_fwd[(size_t)i << tw_shift] : mneg(tw_fwd[(size_t)(i - halfm) << tw_shift]);
    const uint32_t x = mmul(gen, wi);
    const uint32_t zh = zh_t[i & ((1u << B) - 1u)];
    const uint32_t sel_first = mmul(zh, minv(msub(x, ONE)));
    const uint32_t sel_trans = msub(x, w_n_inv);
    const uint32_t sel_last = mmul(zh, minv(sel_trans));
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    uint32_t cnt = 0;
    const uint32_t ro = r << 2, rno = rn << 2;
    const gptr ldep = (gptr)lde;
    const gptr permp = (gptr)perm;
    const gptr prepp = (gptr)prep;
)JIT";
}
inline void quot_jit_epilogue(std::ostringstream& os) {
    os << R"JIT(
    FINISH
    const uint32_t izh = inv_zh_t[i & ((1u << B) - 1u)];
    q[r] = mmul(acc0, izh);
    q[M + r] = mmul(acc1, izh);
    q[2 * M + r] = mmul(acc2, izh);
    q[3 * M + r] = mmul(acc3, izh);
}
)JIT";
}

// Emits the kernel source and the parameter table.
// shared: the SHARED-ROWS form (round 6).  The plain form gives every wave 64 rows of its own and walks all constraints: a wave's working set is
// 64 rows x every column (77 KB for the 300-column chip), 32 waves per CU and 32 CUs per XCD hold 78 MB against 4 MB of L2 -- every re-read of
// a cell (3.9 per cell on that chip) misses the L2 and crosses the fabric to the Infinity Cache (the "43 GB at 7 TB/s" of
// profiles/round05_pmc_traffic.json: FETCH_SIZE counts L2 misses, not HBM bytes).  Here a workgroup of SIXTEEN waves owns ONE block of 64 rows:
// wave w evaluates instances w, w + 16, ... of every class, so the rows in flight per XCD are a sixteenth (4.9 MB) and a cell's re-reads by
// the other waves of the workgroup find it in the L2; the sixteen partial sums meet in LDS.  The two selector inversions are done once per
// workgroup (waves 0 and 1) while the other waves evaluate the classes that need neither.
// waves per row block of the shared-rows form (ZKHIP_JIT_SHARED_WAVES = 4 / 8 / 16 -- at least four: waves 0 .. 3 sum the four coordinates of the
// result; a 2-wave variant failed the parity fuzz for exactly that reason and is not offered; measured on the headline's chip, with the selector
// tables: 16: 5.89 ms, 8: 5.32, 4: 4.99 against the plain form's 5.29 -- profiles/round06_quot_jit_shared_v2.txt; default 4)
inline unsigned quot_shared_waves() {
    static const unsigned v = [] {
        const char* e = getenv("ZKHIP_JIT_SHARED_WAVES");
        const int n = e ? atoi(e) : 4;
        return (unsigned)(n == 8 || n == 16 ? n : 4);
    }();
    return v;
}
inline std::string quot_jit_source(const AirProgram& p, unsigned lh, unsigned b, std::vector<JitClass>& classes,
                                   std::vector<uint32_t>* table, bool shared = false) {
    (void)p;
    std::ostringstream os;
    if (!shared) {
        quot_jit_prologue(os, lh, b, p.qd());
    } else {
        os << quot_jit_preamble();
        os << "#define B " << b << "u\n#define M ((size_t)1 << H)\n#define NW " << quot_shared_waves() << "u\n#define SEL_OFF __SEL_OFF__\n";
        os << R"JIT(
extern "C" __global__ __launch_bounds__(64 * NW, 8) void quot_jit(const uint32_t* __restrict__ lde, uint32_t* __restrict__ q,
        const uint32_t* __restrict__ pvs, const uint32_t* __restrict__ apow, const uint32_t* __restrict__ tw_fwd,
        const uint32_t* __restrict__ zh_t, const uint32_t* __restrict__ inv_zh_t, const uint32_t* __restrict__ tab,
        uint32_t gen, uint32_t w_n_inv, uint32_t tw_shift, const uint32_t* __restrict__ perm,
        const uint32_t* __restrict__ lchal, const uint32_t* __restrict__ expo, const uint32_t* __restrict__ prep,
        uint32_t H, uint32_t NQROWS) {
    __shared__ uint32_t red_s[4u * NW * 64u];   // [wave][coordinate][lane]
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (wave-uniform: the parameter rows stay scalar loads)
    const uint32_t r = blockIdx.x * 64u + lane;   // (NQROWS is a multiple of 64: the form is chosen for tall chips only)
    const uint32_t i = __brev(r) >> (32 - H);
    const uint32_t rn = __brev((i + (1u << B)) & ((1u << H) - 1u)) >> (32 - H);
    const uint32_t halfm = 1u << (H - 1);
    const uint32_t wi = i < halfm ? tw_fwd[(size_t)i << tw_shift] : mneg(tw_fwd[(size_t)(i - halfm) << tw_shift]);
    const uint32_t x = mmul(gen, wi);
    const uint32_t sel_trans = msub(x, w_n_inv);
    // Z_H(x) / (x - 1) and Z_H(x) / (x - w^-1) per LDE row: a table behind the parameter table, generated with the key (k_gen_selectors,
    // csrc/prover.hip) -- sixteen waves would each invert twice, or two of them would hold the others at a barrier
    const uint32_t sel_first = ((gptr)tab)[SEL_OFF + r], sel_last = ((gptr)tab)[SEL_OFF + NQROWS + r];
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    uint32_t cnt = 0;
    const uint32_t ro = r << 2, rno = rn << 2;
    const gptr ldep = (gptr)lde;
    const gptr permp = (gptr)perm;
    const gptr prepp = (gptr)prep;
)JIT";
    }
    // Instances of a class are evaluated UNROLL at a time: the parameter rows (scalar loads) and the trace cells (vector loads) of all of
    // them are requested before the first product, so that a wave waits for memory once per group instead of three times per instance
    // (the rolled loop's body was: parameter row -> wait -> six cell loads -> two more parameters -> wait -> arithmetic -> the row of
    // alpha powers -> wait; its waves were parked 72 % of their cycles: profiles/round05b_pmc_wave_cycles.json).  The order of the
    // accumulation is the rolled loop's, so the sums are too.  ZKHIP_JIT_UNROLL=n (1 .. 8) for measurements; part of the source text, so of
    // every cache key.
    static const unsigned unroll = [] {
        const char* e = getenv("ZKHIP_JIT_UNROLL");
        const int v = e ? atoi(e) : 8;   // (measured on the headline's 300-column chip: 6.07 / 5.90 / 5.94 / 5.33 ms at 1 / 2 / 4 / 8)
        return (unsigned)(v < 1 ? 1 : v > 8 ? 8 : v);
    }();
    for (size_t c = 0; c < classes.size(); c++) {
        JitClass& C = classes[c];
        C.table_off = table->size();
        const size_t stride = 1 + C.param_entry.size();
        for (const auto& row : C.inst) table->insert(table->end(), row.begin(), row.end());
        std::vector<int> pidx(C.entries.size(), -1);
        for (size_t j = 0; j < C.param_entry.size(); j++) pidx[C.param_entry[j]] = (int)j + 1;
        // a product used only by other products keeps the signed form (2 instructions fewer than a reduced one)
        std::vector<uint8_t> only_mul(C.entries.size(), 1), used(C.entries.size(), 0);
        for (size_t e = 0; e < C.entries.size(); e++) {
            const JitEntry& E = C.entries[e];
            if (E.op == A_ADD || E.op == A_SUB || E.op == A_MUL) {
                used[E.a] = used[E.b] = 1;
                if (E.op != A_MUL) only_mul[E.a] = only_mul[E.b] = 0;
            } else if (E.op == A_NEG) {
                used[E.a] = 1, only_mul[E.a] = 0;
            }
        }
        auto stays_signed = [&](size_t e) { return C.entries[e].op == A_MUL && used[e] && only_mul[e] && e + 1 != C.entries.size(); };
        // the trace loads of instance u (its parameter row is pr<u>), then its arithmetic
        auto emit_loads = [&](unsigned u) {
            for (size_t e = 0; e < C.entries.size(); e++) {
                const JitEntry& E = C.entries[e];
                if (E.op == A_VAR || E.op == A_PERM || E.op == A_PREP)
                    os << "        const uint32_t e" << e << "_" << u << " = " << (E.op == A_VAR ? "LD" : E.op == A_PERM ? "LDP" : "LDQ") << "(pr" << u << "[" << pidx[e] << "], "
                       << (E.a ? "rno" : "ro") << ");\n";
            }
        };
        auto emit_arith = [&](unsigned u) {
            auto v = [&](size_t e) { return "e" + std::to_string(e) + "_" + std::to_string(u); };
            for (size_t e = 0; e < C.entries.size(); e++) {
                const JitEntry& E = C.entries[e];
                switch (E.op) {
                    case A_VAR:
                    case A_PREP:
                    case A_PERM: break;
                    case A_CHAL: os << "        const uint32_t " << v(e) << " = ((cptr)lchal)[pr" << u << "[" << pidx[e] << "]];\n"; break;
                    case A_EXPOSED: os << "        const uint32_t " << v(e) << " = ((cptr)expo)[pr" << u << "[" << pidx[e] << "]];\n"; break;
                    case A_PUB: os << "        const uint32_t " << v(e) << " = PV(pr" << u << "[" << pidx[e] << "]);\n"; break;
                    case A_CONST: os << "        const uint32_t " << v(e) << " = pr" << u << "[" << pidx[e] << "];\n"; break;
                    case A_FIRST: os << "        const uint32_t " << v(e) << " = sel_first;\n"; break;
                    case A_LAST: os << "        const uint32_t " << v(e) << " = sel_last;\n"; break;
                    case A_TRANS: os << "        const uint32_t " << v(e) << " = sel_trans;\n"; break;
                    case A_NEG: os << "        const uint32_t " << v(e) << " = mneg(" << v(E.a) << ");\n"; break;
                    case A_MUL:
                        if (stays_signed(e))
                            os << "        const int32_t " << v(e) << " = sml((int32_t)" << v(E.a) << ", (int32_t)" << v(E.b) << ");\n";
                        else
                            os << "        const uint32_t " << v(e) << " = SC(sml((int32_t)" << v(E.a) << ", (int32_t)" << v(E.b) << "));\n";
                        break;
                    default:
                        os << "        const uint32_t " << v(e) << " = " << (E.op == A_ADD ? "madd" : "msub") << "(" << v(E.a) << ", " << v(E.b) << ");\n";
                }
            }
            os << "        ACC(pr" << u << "[0], " << v(C.entries.size() - 1) << ")\n";
        };
        const size_t n_inst = C.inst.size(), U = n_inst >= 2 * unroll ? unroll : 1, n_main = n_inst / U * U;
        if (shared) continue;   // (the shared-rows form emits its loops below, in two groups around the selectors' barrier)
        if (U > 1) {
            os << "    for (uint32_t it = 0; it < " << n_main << "u; it += " << U << "u) {\n";
            for (unsigned u = 0; u < U; u++) os << "        cptr pr" << u << " = (cptr)tab + " << C.table_off << "u + (it + " << u << "u) * " << stride << "u;\n";
            for (unsigned u = 0; u < U; u++) emit_loads(u);
            for (unsigned u = 0; u < U; u++) emit_arith(u);
            os << "    }\n";
        }
        if (n_main < n_inst || U == 1) {
            os << "    for (uint32_t it = " << (U > 1 ? n_main : 0) << "u; it < " << n_inst << "u; it++) {\n";
            os << "        cptr pr0 = (cptr)tab + " << C.table_off << "u + it * " << stride << "u;\n";
            emit_loads(0);
            emit_arith(0);
            os << "    }\n";
        }
    }
    if (shared) {
        // wave w takes instances w, w + NW, ... of a class, SU of them per iteration: an index beyond the class is clamped to its last instance
        // (evaluated again, not accumulated), so the loads of an iteration are unconditional and issue together
        size_t given = 0;
        auto emit_class = [&](size_t c) {
            JitClass& C = classes[c];
            const size_t stride = 1 + C.param_entry.size(), n_inst = C.inst.size();
            std::vector<int> pidx(C.entries.size(), -1);
            for (size_t j = 0; j < C.param_entry.size(); j++) pidx[C.param_entry[j]] = (int)j + 1;
            std::vector<uint8_t> only_mul(C.entries.size(), 1), used(C.entries.size(), 0);
            for (size_t e = 0; e < C.entries.size(); e++) {
                const JitEntry& E = C.entries[e];
                if (E.op == A_ADD || E.op == A_SUB || E.op == A_MUL) {
                    used[E.a] = used[E.b] = 1;
                    if (E.op != A_MUL) only_mul[E.a] = only_mul[E.b] = 0;
                } else if (E.op == A_NEG) {
                    used[E.a] = 1, only_mul[E.a] = 0;
                }
            }
            auto stays_signed = [&](size_t e) { return C.entries[e].op == A_MUL && used[e] && only_mul[e] && e + 1 != C.entries.size(); };
            const unsigned NWv = quot_shared_waves();
            const unsigned per_wave = (unsigned)((n_inst + NWv - 1) / NWv);
            const unsigned SU = std::min<unsigned>(per_wave, std::min<unsigned>(unroll, 4u));
            // (the class's first instance goes to the wave after the one that took the previous class's last: the lists stay level over the
            // classes of a chip, whatever their sizes)
            const unsigned first_wave = (unsigned)(given % NWv);
            given += n_inst;
            os << "    for (uint32_t it = (wv + NW - " << first_wave << "u) % NW; it < " << n_inst << "u; it += " << SU << "u * NW) {\n";
            for (unsigned u = 0; u < SU; u++) {
                os << "        const uint32_t i" << u << " = it + " << u << "u * NW;\n";
                os << "        cptr pr" << u << " = (cptr)tab + " << C.table_off << "u + (i" << u << " < " << n_inst << "u ? i" << u << " : " << (n_inst - 1) << "u) * " << stride << "u;\n";
            }
            for (unsigned u = 0; u < SU; u++)
                for (size_t e = 0; e < C.entries.size(); e++) {
                    const JitEntry& E = C.entries[e];
                    if (E.op == A_VAR || E.op == A_PERM || E.op == A_PREP)
                        os << "        const uint32_t e" << e << "_" << u << " = " << (E.op == A_VAR ? "LD" : E.op == A_PERM ? "LDP" : "LDQ") << "(pr" << u << "[" << pidx[e] << "], "
                           << (E.a ? "rno" : "ro") << ");\n";
                }
            for (unsigned u = 0; u < SU; u++) {
                auto v = [&](size_t e) { return "e" + std::to_string(e) + "_" + std::to_string(u); };
                for (size_t e = 0; e < C.entries.size(); e++) {
                    const JitEntry& E = C.entries[e];
                    switch (E.op) {
                        case A_VAR:
                        case A_PREP:
                        case A_PERM: break;
                        case A_CHAL: os << "        const uint32_t " << v(e) << " = ((cptr)lchal)[pr" << u << "[" << pidx[e] << "]];\n"; break;
                        case A_EXPOSED: os << "        const uint32_t " << v(e) << " = ((cptr)expo)[pr" << u << "[" << pidx[e] << "]];\n"; break;
                        case A_PUB: os << "        const uint32_t " << v(e) << " = PV(pr" << u << "[" << pidx[e] << "]);\n"; break;
                        case A_CONST: os << "        const uint32_t " << v(e) << " = pr" << u << "[" << pidx[e] << "];\n"; break;
                        case A_FIRST: os << "        const uint32_t " << v(e) << " = sel_first;\n"; break;
                        case A_LAST: os << "        const uint32_t " << v(e) << " = sel_last;\n"; break;
                        case A_TRANS: os << "        const uint32_t " << v(e) << " = sel_trans;\n"; break;
                        case A_NEG: os << "        const uint32_t " << v(e) << " = mneg(" << v(E.a) << ");\n"; break;
                        case A_MUL:
                            if (stays_signed(e))
                                os << "        const int32_t " << v(e) << " = sml((int32_t)" << v(E.a) << ", (int32_t)" << v(E.b) << ");\n";
                            else
                                os << "        const uint32_t " << v(e) << " = SC(sml((int32_t)" << v(E.a) << ", (int32_t)" << v(E.b) << "));\n";
                            break;
                        default:
                            os << "        const uint32_t " << v(e) << " = " << (E.op == A_ADD ? "madd" : "msub") << "(" << v(E.a) << ", " << v(E.b) << ");\n";
                    }
                }
                os << "        if (i" << u << " < " << n_inst << "u) ACC(pr" << u << "[0], " << v(C.entries.size() - 1) << ")\n";
            }
            os << "    }\n";
        };
        for (size_t c = 0; c < classes.size(); c++) emit_class(c);
        os << R"JIT(
    FINISH
    red_s[(wv * 4u + 0u) * 64u + lane] = acc0, red_s[(wv * 4u + 1u) * 64u + lane] = acc1;
    red_s[(wv * 4u + 2u) * 64u + lane] = acc2, red_s[(wv * 4u + 3u) * 64u + lane] = acc3;
    __builtin_amdgcn_s_waitcnt(0xc07f); __syncthreads();
    if (wv < 4u) {   // wave c sums coordinate c over the waves
        const uint32_t izh = inv_zh_t[i & ((1u << B) - 1u)];
        uint32_t v = red_s[wv * 64u + lane];
        for (uint32_t w = 1; w < NW; w++) v = madd(v, red_s[(w * 4u + wv) * 64u + lane]);
        q[(size_t)wv * M + r] = mmul(v, izh);
    }
}
)JIT";
        if (table->empty()) table->push_back(0);
        std::string out = os.str();
        const std::string key = "__SEL_OFF__";   // the selector tables start where the parameter table ends
        for (size_t pos = out.find(key); pos != std::string::npos; pos = out.find(key, pos)) out.replace(pos, key.size(), std::to_string(table->size()) + "u");
        return out;
    }
    quot_jit_epilogue(os);
    if (table->empty()) table->push_back(0);
    return os.str();
}

// Third code shape: the class form over an LDS TILE.  The class form reads a trace cell from memory once per constraint that uses it
// (the leaves are table-driven, so nothing can stay in a register between constraints): 3.9x the trace's bytes for the 300-column
// AIR of the headline, at HBM speed.  Here a workgroup of four waves owns 64 consecutive rows: the waves first copy every (column,
// rotation) the constraints use into LDS -- one 256-byte row segment per load, ~150 KB in flight per CU --, then wave w evaluates
// constraints w, w + 4, ... of every class from the tile and the four partial sums meet in LDS.  The trace crosses HBM once.
// (In bit-reversed storage the "next" rows of 64 consecutive rows are 64 consecutive rows too: the offset touches high bits only.)
constexpr unsigned QUOT_TILE_ROWS = 64, QUOT_TILE_WAVES = 8, QUOT_TILE_MAX_LDS_WORDS = 39936;   // 156 KB of the CU's 160
inline bool quot_jit_source_tiled(const AirProgram& p, unsigned lh, unsigned b, std::vector<JitClass>& classes, std::vector<uint32_t>* table,
                                  std::string* src) {
    // Measured on the headline AIR (2^22 x 300, 2^23 LDE rows): HBM traffic of the kernel 43.5 -> ~11 GB, but 17.4 ms
    // against the plain class form's 6.0 ms (27.2 ms with four waves and a rolled load loop): one 84 KB tile per CU leaves nothing
    // to overlap its load with (a second buffer does not fit 160 KB), the three barriers per tile and the selector inversions on
    // one wave sit on the critical path, and the plain form already streams its re-reads at 7 TB/s.  Opt-in: ZKHIP_JIT_TILE=1.
    if (lh + b < 12 || !getenv("ZKHIP_JIT_TILE")) return false;
    // slots: the distinct (column, rotation) pairs of the main trace
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> slot_of;
    size_t n_var_uses = 0;
    for (const JitClass& C : classes) {
        std::vector<int> pidx(C.entries.size(), -1);
        for (size_t j = 0; j < C.param_entry.size(); j++) pidx[C.param_entry[j]] = (int)j + 1;
        for (const auto& row : C.inst)
            for (size_t e = 0; e < C.entries.size(); e++)
                if (C.entries[e].op == A_VAR) {
                    const auto key = std::make_pair(row[(size_t)pidx[e]], C.entries[e].a);
                    if (!slot_of.count(key)) slot_of.emplace(key, (uint32_t)slot_of.size());
                    n_var_uses++;
                }
    }
    const size_t n_slots = slot_of.size(), lds_words = n_slots * QUOT_TILE_ROWS + QUOT_TILE_WAVES * 4 * QUOT_TILE_ROWS + 3 * QUOT_TILE_ROWS;
    // worth it when cells are re-read (uses well above slots) and the tile fits
    if (n_slots < 48 || lds_words > QUOT_TILE_MAX_LDS_WORDS || n_var_uses < 2 * n_slots) return false;
    std::ostringstream os;
    os << quot_jit_preamble();
    (void)p;
    os << "#define B " << b << "u\n#define M ((size_t)1 << H)\n#define NSLOTS " << n_slots << "u\n#define NW " << QUOT_TILE_WAVES << "u\n";
    std::vector<uint32_t> slot_tab(n_slots);
    for (const auto& kv : slot_of) slot_tab[kv.second] = kv.first.first | (kv.first.second ? 0x80000000u : 0u);
    os << R"JIT(
extern "C" __global__ __launch_bounds__(64 * NW) void quot_jit(const uint32_t* __restrict__ lde, uint32_t* __restrict__ q,
        const uint32_t* __restrict__ pvs, const uint32_t* __restrict__ apow, const uint32_t* __restrict__ tw_fwd,
        const uint32_t* __restrict__ zh_t, const uint32_t* __restrict__ inv_zh_t, const uint32_t* __restrict__ tab,
        uint32_t gen, uint32_t w_n_inv, uint32_t tw_shift, const uint32_t* __restrict__ perm,
        const uint32_t* __restrict__ lchal, const uint32_t* __restrict__ expo, const uint32_t* __restrict__ prep,
        uint32_t H, uint32_t NQROWS) {
    __shared__ uint32_t tile[NSLOTS * 64u + 4u * NW * 64u + 3u * 64u];
    uint32_t* const red_s = tile + NSLOTS * 64u;      // [wave][coordinate][lane]
    uint32_t* const sel_s = red_s + 4u * NW * 64u;    // first, trans, last per lane
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const gptr ldep = (gptr)lde;
    const gptr permp = (gptr)perm;
    const gptr prepp = (gptr)prep;
    for (uint32_t t = blockIdx.x; t < (NQROWS >> 6); t += gridDim.x) {
        const uint32_t r = (t << 6) + lane;
        const uint32_t i = __brev(r) >> (32 - H);
        const uint32_t rn = __brev((i + (1u << B)) & ((1u << H) - 1u)) >> (32 - H);
        const uint32_t ro = r << 2, rno = rn << 2;
        // sixteen row segments per wave in flight before the first LDS write (the loop alone keeps one)
        for (uint32_t s0 = wv; s0 < NSLOTS; s0 += 16u * NW) {
            uint32_t v[16];
#pragma unroll
            for (uint32_t u = 0; u < 16u; u++) {
                const uint32_t s = s0 + u * NW;
                if (s < NSLOTS) {
                    const uint32_t cr = ((cptr)tab)[SLOT_TAB + s];
                    v[u] = LD(cr & 0x7fffffffu, (cr >> 31) ? rno : ro);
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 16u; u++) {
                const uint32_t s = s0 + u * NW;
                if (s < NSLOTS) tile[s * 64u + lane] = v[u];
            }
        }
        if (wv == 0) {
            const uint32_t halfm = 1u << (H - 1);
            const uint32_t wi = i < halfm ? tw_fwd[(size_t)i << tw_shift] : mneg(tw_fwd[(size_t)(i - halfm) << tw_shift]);
            const uint32_t x = mmul(gen, wi);
            const uint32_t zh = zh_t[i & ((1u << B) - 1u)];
            const uint32_t st = msub(x, w_n_inv);
            sel_s[lane] = mmul(zh, minv(msub(x, ONE))), sel_s[64u + lane] = st, sel_s[128u + lane] = mmul(zh, minv(st));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); __syncthreads();   // (lds_barrier.hpp: the LDS wait stated in front of the barrier)
        const uint32_t sel_first = sel_s[lane], sel_trans = sel_s[64u + lane], sel_last = sel_s[128u + lane];
        uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
        uint32_t cnt = 0;
)JIT";
    for (size_t c = 0; c < classes.size(); c++) {
        JitClass& C = classes[c];
        C.table_off = table->size();
        const size_t stride = 1 + C.param_entry.size();
        std::vector<int> pidx(C.entries.size(), -1);
        for (size_t j = 0; j < C.param_entry.size(); j++) pidx[C.param_entry[j]] = (int)j + 1;
        for (const auto& row : C.inst) {   // main-trace leaves: the tile's word offset of their slot instead of the column
            std::vector<uint32_t> rw = row;
            for (size_t e = 0; e < C.entries.size(); e++)
                if (C.entries[e].op == A_VAR) rw[(size_t)pidx[e]] = slot_of.at({row[(size_t)pidx[e]], C.entries[e].a}) * QUOT_TILE_ROWS;
            table->insert(table->end(), rw.begin(), rw.end());
        }
        os << "        for (uint32_t it = wv; it < " << C.inst.size() << "u; it += NW) {\n";
        os << "            cptr pr = (cptr)tab + " << C.table_off << "u + it * " << stride << "u;\n";
        for (size_t e = 0; e < C.entries.size(); e++) {
            const JitEntry& E = C.entries[e];
            if (E.op == A_VAR) os << "            const uint32_t e" << e << " = tile[pr[" << pidx[e] << "] + lane];\n";
            else if (E.op == A_PERM || E.op == A_PREP)
                os << "            const uint32_t e" << e << " = " << (E.op == A_PERM ? "LDP" : "LDQ") << "(pr[" << pidx[e] << "], " << (E.a ? "rno" : "ro") << ");\n";
        }
        std::vector<uint8_t> only_mul(C.entries.size(), 1), used(C.entries.size(), 0);
        for (size_t e = 0; e < C.entries.size(); e++) {
            const JitEntry& E = C.entries[e];
            if (E.op == A_ADD || E.op == A_SUB || E.op == A_MUL) {
                used[E.a] = used[E.b] = 1;
                if (E.op != A_MUL) only_mul[E.a] = only_mul[E.b] = 0;
            } else if (E.op == A_NEG) {
                used[E.a] = 1, only_mul[E.a] = 0;
            }
        }
        auto stays_signed = [&](size_t e) { return C.entries[e].op == A_MUL && used[e] && only_mul[e] && e + 1 != C.entries.size(); };
        for (size_t e = 0; e < C.entries.size(); e++) {
            const JitEntry& E = C.entries[e];
            switch (E.op) {
                case A_VAR:
                case A_PREP:
                case A_PERM: break;
                case A_CHAL: os << "            const uint32_t e" << e << " = ((cptr)lchal)[pr[" << pidx[e] << "]];\n"; break;
                case A_EXPOSED: os << "            const uint32_t e" << e << " = ((cptr)expo)[pr[" << pidx[e] << "]];\n"; break;
                case A_PUB: os << "            const uint32_t e" << e << " = PV(pr[" << pidx[e] << "]);\n"; break;
                case A_CONST: os << "            const uint32_t e" << e << " = pr[" << pidx[e] << "];\n"; break;
                case A_FIRST: os << "            const uint32_t e" << e << " = sel_first;\n"; break;
                case A_LAST: os << "            const uint32_t e" << e << " = sel_last;\n"; break;
                case A_TRANS: os << "            const uint32_t e" << e << " = sel_trans;\n"; break;
                case A_NEG: os << "            const uint32_t e" << e << " = mneg(e" << E.a << ");\n"; break;
                case A_MUL:
                    if (stays_signed(e))
                        os << "            const int32_t e" << e << " = sml((int32_t)e" << E.a << ", (int32_t)e" << E.b << ");\n";
                    else
                        os << "            const uint32_t e" << e << " = SC(sml((int32_t)e" << E.a << ", (int32_t)e" << E.b << "));\n";
                    break;
                default:
                    os << "            const uint32_t e" << e << " = " << (E.op == A_ADD ? "madd" : "msub") << "(e" << E.a << ", e" << E.b << ");\n";
            }
        }
        os << "            ACC(pr[0], e" << (C.entries.size() - 1) << ")\n        }\n";
    }
    const size_t slot_tab_off = table->size();
    table->insert(table->end(), slot_tab.begin(), slot_tab.end());
    os << R"JIT(
        FINISH
        red_s[(wv * 4u + 0u) * 64u + lane] = acc0, red_s[(wv * 4u + 1u) * 64u + lane] = acc1;
        red_s[(wv * 4u + 2u) * 64u + lane] = acc2, red_s[(wv * 4u + 3u) * 64u + lane] = acc3;
        __builtin_amdgcn_s_waitcnt(0xc07f); __syncthreads();
        if (wv < 4u) {   // wave c sums coordinate c over the waves
            const uint32_t izh = inv_zh_t[i & ((1u << B) - 1u)];
            uint32_t v = red_s[wv * 64u + lane];
            for (uint32_t w = 1; w < NW; w++) v = madd(v, red_s[(w * 4u + wv) * 64u + lane]);
            q[(size_t)wv * M + r] = mmul(v, izh);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); __syncthreads();   // the tile and the partial sums are rewritten by the next round
    }
}
)JIT";
    std::string out = os.str();
    const std::string key = "SLOT_TAB";
    for (size_t pos = out.find(key); pos != std::string::npos; pos = out.find(key, pos)) out.replace(pos, key.size(), std::to_string(slot_tab_off) + "u");
    *src = out;
    return true;
}

// Second code shape, for AIRs whose constraints SHARE large sub-expressions instead of repeating small shapes (the
// Poseidon2 AIR: 282 constraints over one 2701-node DAG in which every round's S-box outputs feed 16 constraints and the
// partial rounds' lanes are linear expressions that grow for 13 rounds -- expanded per constraint that is 30x the DAG):
// the whole DAG as ONE straight-line block, one SSA value per reachable node, columns / constants / indices as literals,
// each constraint accumulated as soon as its root exists.  The code is larger than the instruction cache, but it is
// fetched strictly sequentially and every wave of a CU walks the same stream.
constexpr size_t QUOT_JIT_FLAT_MAX_NODES = 24576;
// Column indices are read from an identity table with scalar loads rather than written as literals: with a literal
// LLVM rewrites base + c*M + lane_offset as (base + lane_offset) + c*M, i.e. one 64-bit per-lane address per column,
// all of them live at once (309 VGPRs, one wave per SIMD, 11.9 ms for the Poseidon2 AIR at 2^23 rows); through the
// table the column base stays a scalar pair (61 VGPRs).
inline bool quot_jit_source_flat(const AirProgram& p, unsigned lh, unsigned b, std::string* src, std::vector<uint32_t>* table,
                                 std::string* msg) {
    std::vector<uint8_t> live(p.n_nodes, 0);
    for (uint32_t k = 0; k < p.n_cons; k++) live[p.cons[k]] = 1;
    size_t n_live = 0;
    for (uint32_t i = p.n_nodes; i-- > 0;) {  // operands precede their users
        if (!live[i]) continue;
        n_live++;
        const uint32_t op = p.nodes[3 * i], a = p.nodes[3 * i + 1], bb = p.nodes[3 * i + 2];
        if (op == A_ADD || op == A_SUB || op == A_MUL) live[a] = live[bb] = 1;
        else if (op == A_NEG) live[a] = 1;
    }
    if (n_live > QUOT_JIT_FLAT_MAX_NODES) {
        *msg += "; flat form too large (" + std::to_string(n_live) + " nodes)";
        return false;
    }
    std::vector<std::vector<uint32_t>> cons_of(p.n_nodes);
    for (uint32_t k = 0; k < p.n_cons; k++) cons_of[p.cons[k]].push_back(k);
    std::ostringstream os;
    quot_jit_prologue(os, lh, b, p.qd());
    uint32_t max_col = 0;
    for (uint32_t i = 0; i < p.n_nodes; i++) {
        if (!live[i]) continue;
        const uint32_t op0 = p.nodes[3 * i];
        if (op0 == A_VAR || op0 == A_PERM || op0 == A_PREP) max_col = std::max(max_col, p.nodes[3 * i + 1]);
    }
    table->resize((size_t)max_col + 1);
    for (uint32_t c = 0; c <= max_col; c++) (*table)[c] = c;
    os << "    const cptr colv = (cptr)tab;\n";
    auto is_two = [&](uint32_t n) { return p.nodes[3 * n] == A_CONST && p.nodes[3 * n + 1] == 2u; };
    auto is_doubling = [&](uint32_t n) { return p.nodes[3 * n] == A_MUL && (is_two(p.nodes[3 * n + 1]) || is_two(p.nodes[3 * n + 2])); };
    // a product used only by other (true) products keeps the signed form
    std::vector<uint8_t> only_mul(p.n_nodes, 1), used(p.n_nodes, 0);
    for (uint32_t i = 0; i < p.n_nodes; i++) {
        if (!live[i]) continue;
        const uint32_t op = p.nodes[3 * i], a = p.nodes[3 * i + 1], bb = p.nodes[3 * i + 2];
        if (op == A_ADD || op == A_SUB || op == A_MUL) {
            used[a] = used[bb] = 1;
            if (op != A_MUL || is_doubling(i)) only_mul[a] = only_mul[bb] = 0;
        } else if (op == A_NEG) {
            used[a] = 1, only_mul[a] = 0;
        }
    }
    auto stays_signed = [&](uint32_t n) { return p.nodes[3 * n] == A_MUL && !is_doubling(n) && used[n] && only_mul[n] && cons_of[n].empty(); };
    for (uint32_t i = 0; i < p.n_nodes; i++) {
        if (!live[i]) continue;
        const uint32_t op = p.nodes[3 * i], a = p.nodes[3 * i + 1], bb = p.nodes[3 * i + 2];
        os << "    const " << (stays_signed(i) ? "int32_t" : "uint32_t") << " e" << i << " = ";
        switch (op) {
            case A_VAR: os << "LD(colv[" << a << "], " << (bb ? "rno" : "ro") << ")"; break;
            case A_PERM: os << "LDP(colv[" << a << "], " << (bb ? "rno" : "ro") << ")"; break;
            case A_PREP: os << "LDQ(colv[" << a << "], " << (bb ? "rno" : "ro") << ")"; break;
            case A_CHAL: os << "((cptr)lchal)[" << a << "]"; break;
            case A_EXPOSED: os << "((cptr)expo)[" << a << "]"; break;
            case A_PUB: os << "PV(" << a << ")"; break;
            case A_CONST: os << to_monty(a) << "u"; break;
            case A_FIRST: os << "sel_first"; break;
            case A_LAST: os << "sel_last"; break;
            case A_TRANS: os << "sel_trans"; break;
            case A_NEG: os << "mneg(e" << a << ")"; break;
            default: {
                if (op == A_MUL && (is_two(a) || is_two(bb))) {  // doubling is an addition, not a Montgomery product
                    const uint32_t x = is_two(a) ? bb : a;
                    os << "madd(e" << x << ", e" << x << ")";
                } else if (op == A_MUL) {
                    if (stays_signed(i)) os << "sml((int32_t)e" << a << ", (int32_t)e" << bb << ")";
                    else os << "SC(sml((int32_t)e" << a << ", (int32_t)e" << bb << "))";
                } else {
                    os << (op == A_ADD ? "madd" : "msub") << "(e" << a << ", e" << bb << ")";
                }
            }
        }
        os << ";\n";
        for (uint32_t k : cons_of[i]) os << "    ACC(" << k << "u, e" << i << ")\n";
    }
    quot_jit_epilogue(os);
    *src = os.str();
    return true;
}

// Generates the kernel of an AIR and compiles it for gfx950 (or finds its code object in the process-wide / on-disk cache).  No HIP
// call.  (Compiling the chips of an AIR set on several host threads was tried: hipRTC of ROCm 7.2 serialises the compiles of one
// process -- five kernels took 24.1 s on five threads and 24.9 s in sequence -- and -O1 / -O2 compile as long as -O3: the time is the
// backend's on these long basic blocks.)
// `cache_dir` ("" / null = none): compiled code objects kept across processes (zkhip_config.jit_cache_dir)
inline bool quot_jit_code(const AirProgram& p, unsigned lh, unsigned b, std::vector<uint32_t>* table, std::vector<char>* code_out, std::string* msg,
                          unsigned* rows_per_block = nullptr, const char* cache_dir = nullptr) {
    if (rows_per_block) *rows_per_block = 256;
    std::vector<JitClass> classes;
    std::string src;
    // ZKHIP_JIT_FLAT=1 prefers the flat form for every AIR (experiments)
    if (!getenv("ZKHIP_JIT_FLAT") && quot_jit_classify(p, &classes, msg)) {
        if (quot_jit_source_tiled(p, lh, b, classes, table, &src)) {
            if (rows_per_block) *rows_per_block = 0;   // tiles of QUOT_TILE_ROWS rows walked by a fixed number of workgroups
        } else {
            // the shared-rows form: for TALL chips with enough constraints (>= 2^20 LDE rows, >= 64 instances; ZKHIP_JIT_SHARED=0 / 1 forces it
            // off / on for every chip of >= 2^12 LDE rows).  Measured on the headline's 300-column chip: first cut (sixteen waves, selectors
            // inverted by two of them, a barrier in the middle) 6.08 ms against the plain form's 5.32 (profiles/round06_quot_jit_shared_and_lanes.txt);
            // with the selectors in a table, level instance lists and FOUR waves per row block 4.99 ms (profiles/round06_quot_jit_shared_v2.txt)
            // at a third of the plain form's L2 misses.  Bit-exact either way (the parity suites run with it forced).
            size_t n_inst = 0;
            for (const JitClass& C : classes) n_inst += C.inst.size();
            static const int shared_env = getenv("ZKHIP_JIT_SHARED") ? atoi(getenv("ZKHIP_JIT_SHARED")) : -1;
            const bool shared = lh + b >= 12 && (shared_env == 1 || (shared_env < 0 && lh + b >= 20 && n_inst >= 64));
            table->clear();
            src = quot_jit_source(p, lh, b, classes, table, shared);
            if (shared && rows_per_block) *rows_per_block = 64 + 256 * quot_shared_waves();   // (64 rows per workgroup of so many waves)
        }
    } else {
        if (!quot_jit_source_flat(p, lh, b, &src, table, msg)) return false;
    }
    // process-wide cache of compiled code objects keyed by the generated source: several contexts
    // (one per HIP stream) and repeated keygens of the same AIR share one hipRTC compile
    // what else decides the code object, as the source's first line: the caches (this process's map, the files of jit_cache_dir) compare
    // the whole text, so a cache filled under another optimisation level, architecture or hipRTC is never loaded
    static const char* opt_level = getenv("ZKHIP_JIT_OPT") ? getenv("ZKHIP_JIT_OPT") : "-O3";   // (experiments: compile time against kernel time)
    static const std::string key_line = [] {
        int major = 0, minor = 0;
        (void)hiprtcVersion(&major, &minor);
        return "// zkhip quot_jit: --offload-arch=gfx950 " + std::string(opt_level) + " -ffp-contract=off, hipRTC " + std::to_string(major) + "." + std::to_string(minor) + "\n";
    }();
    src = key_line + src;
    static std::mutex cache_mu;
    static std::map<std::string, std::vector<char>> cache;
    std::vector<char> code;
    {
        std::lock_guard<std::mutex> lk(cache_mu);
        auto it = cache.find(src);
        if (it != cache.end()) code = it->second;
    }
    // optional on-disk cache (zkhip_config.jit_cache_dir): code objects keyed by a 128-bit hash of the source, so that a
    // service restarting with the same application skips hipRTC altogether (42 chips: ~90 s of compiles)
    std::string disk_path;
    if (code.empty()) {
        if (const char* dir = (cache_dir && cache_dir[0]) ? cache_dir : nullptr) {
            uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
            for (unsigned char ch : src) {
                h1 = (h1 ^ ch) * 0x100000001b3ull;
                h2 = (h2 + ch) * 0x9e3779b97f4a7c15ull + (h2 >> 29);
            }
            char name[64];
            snprintf(name, sizeof name, "/quot_%016llx%016llx.hsaco", (unsigned long long)h1, (unsigned long long)h2);
            disk_path = std::string(dir) + name;
            if (FILE* f = fopen(disk_path.c_str(), "rb")) {
                // file = [source length u64][source][code]: the source is compared, a hash collision cannot mis-load
                uint64_t sl = 0;
                if (fread(&sl, 8, 1, f) == 1 && sl == src.size()) {
                    std::string stored(sl, 0);
                    if (fread(&stored[0], 1, sl, f) == sl && stored == src) {
                        char buf[65536];
                        size_t n;
                        while ((n = fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + n);
                    }
                }
                fclose(f);
                if (!code.empty()) {
                    std::lock_guard<std::mutex> lk(cache_mu);
                    cache[src] = code;
                }
            }
        }
    }
    if (code.empty()) {
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, src.c_str(), "quot_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
            *msg = "hiprtcCreateProgram failed";
            return false;
        }
        const char* opts[] = {"--offload-arch=gfx950", opt_level, "-ffp-contract=off"};
        hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
        if (rc != HIPRTC_SUCCESS) {
            size_t ls = 0;
            hiprtcGetProgramLogSize(prog, &ls);
            std::string log(ls, 0);
            if (ls) hiprtcGetProgramLog(prog, &log[0]);
            *msg = "hiprtc compile failed: " + log.substr(0, 2000);
            hiprtcDestroyProgram(&prog);
            return false;
        }
        size_t cs = 0;
        hiprtcGetCodeSize(prog, &cs);
        code.resize(cs);
        hiprtcGetCode(prog, code.data());
        hiprtcDestroyProgram(&prog);
        if (!disk_path.empty()) {  // write to a temporary name, then rename: concurrent keygens never see a partial file
            // (... nor do two threads of one process -- the device slots' keys are generated side by side -- share a temporary file)
            static std::atomic<unsigned> tmp_serial{0};
            const std::string tmp = disk_path + ".tmp" + std::to_string((unsigned long long)getpid()) + "_" + std::to_string(tmp_serial.fetch_add(1));
            if (FILE* f = fopen(tmp.c_str(), "wb")) {
                const uint64_t sl = src.size();
                const bool ok = fwrite(&sl, 8, 1, f) == 1 && fwrite(src.data(), 1, src.size(), f) == src.size() &&
                                fwrite(code.data(), 1, code.size(), f) == code.size();
                fclose(f);
                if (!ok || rename(tmp.c_str(), disk_path.c_str()) != 0) remove(tmp.c_str());
            }
        }
        std::lock_guard<std::mutex> lk(cache_mu);
        cache[src] = code;
    }
    *code_out = std::move(code);
    return true;
}

// Compiles (or finds) and loads the module.  Returns false (with a message) on any failure; the caller then keeps the interpreter
// kernel.
inline bool quot_jit_build(const AirProgram& p, unsigned lh, unsigned b, hipModule_t* mod, hipFunction_t* fn,
                           std::vector<uint32_t>* table, std::string* msg, unsigned* rows_per_block = nullptr, const char* cache_dir = nullptr) {
    std::vector<char> code;
    if (!quot_jit_code(p, lh, b, table, &code, msg, rows_per_block, cache_dir)) return false;
    if (hipModuleLoadData(mod, code.data()) != hipSuccess) {
        *msg = "hipModuleLoadData failed";
        return false;
    }
    if (hipModuleGetFunction(fn, *mod, "quot_jit") != hipSuccess) {
        (void)hipModuleUnload(*mod);
        *msg = "hipModuleGetFunction failed";
        return false;
    }
    return true;
}

}  // namespace zk
