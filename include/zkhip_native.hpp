// zkhip_native.hpp -- the NATIVE-FIELD intrinsics of the guest VM: BabyBear arithmetic, its quartic extension, and the cast of a field
// element to bytes (`[app_vm_config.native]` and `[app_vm_config.castf]` of the reference's batch and bundle circuits,
// crates/circuits/batch-circuit/openvm.toml:16,24 and crates/circuits/bundle-circuit/openvm.toml:16,18).  In OpenVM these sections bring the
// chips of openvm-native-circuit (un-vendored): FieldArithmeticChip (ADD / SUB / MUL / DIV over BabyBear), FieldExtensionChip (FE4ADD /
// FE4SUB / BBE4MUL / BBE4DIV over F[X] / (X^4 - 11)) and CastFChip (a field element below 2^30 to four bytes) -- the arithmetic of the
// recursion programs that verify STARK proofs inside the VM.  Their cores exist here since round 2 as stand-alone AIRs
// (zkvm-prover_amd/air.py field_arith_air / field_ext_air / castf_air, zkhip_field_arith_tracegen ...); round 5 puts them on the
// execution and memory buses of the one-statement circuit (include/zkhip_vm_circuit.hpp: native_arith_air, native_ext_air, castf_vm_air).
//
// DEVIATION FROM OPENVM (stated, not hidden: ADVICE round 5).  openvm-native-circuit is not vendored, so this is recollection: besides the three
// chips above, OpenVM's native extension is believed to bring a NATIVE ADDRESS SPACE with its own load / store, branch, jal and range-check
// chips, and the FRI reduced-opening and Poseidon2 verify-batch chips its in-VM STARK verifier runs on.  None of those is built here.  The
// three arithmetic chips sit behind CUSTOM ecalls (9 / 10 / 11) on the RV32 memory, not behind OpenVM's native opcodes: a guest that uses
// openvm `verify_stark` (a native-address-space recursion program) cannot run on this executor.  `[app_vm_config.native]` is therefore
// COVERED IN PART: its field arithmetic, not its instruction set (docs/gaps.md).
//
// Guest interface (environment calls, like the other intrinsics; a native field element is ONE memory word holding its canonical value):
//   a7 = 9   native field:  r = a op b on the 3 words at a0 (a, b, then r's slot); a1 = op: 0 add, 1 sub, 2 mul, 3 div (b != 0)
//   a7 = 10  native ext:    r = a op b on the 12 words at a0 (a[4], b[4], then r[4]), coefficients of 1, X, X^2, X^3; a1 = op as above
//   a7 = 11  castf:         the word at a0 must hold a value below 2^30 (limbs of 8, 8, 8 and 6 bits, OpenVM's CastF); its four bytes
//                           are written to the word at a0 + 4
// Operands are read as field elements whatever 32-bit value the word holds (value mod p); results are written canonical (below p), and
// the chip proves that.  This header: the arithmetic on the host (the interpreter of include/zkhip_vm.hpp and the trace twins use it).
#pragma once
#include <cstdint>

namespace zkhip {
namespace native {

constexpr uint32_t P = 0x78000001u;   // BabyBear
constexpr uint32_t W = 11;            // X^4 = 11
enum Op : uint32_t { OP_ADD, OP_SUB, OP_MUL, OP_DIV, N_OPS };
constexpr uint32_t CALL_ARITH = 9, CALL_EXT = 10, CALL_CASTF = 11;
constexpr uint32_t CASTF_BOUND = 1u << 30;
constexpr uint32_t P_HI = P >> 16;    // 0x7800: a canonical word has hi < P_HI, or hi = P_HI and lo = 0
// what an app's openvm.toml enables: `[app_vm_config.native]` (calls 9, 10), `[app_vm_config.castf]` (call 11), `[app_vm_config.pairing]`
// (no chip: OpenVM's pairing extension is a phantom sub-executor, the final-exponentiation hint -- include/zkhip_vm.hpp phantom kind 2)
struct Enabled {
    bool native = false, castf = false, pairing = false;
};

inline uint32_t fadd(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % P); }
inline uint32_t fsub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + P - b % P) % P); }
inline uint32_t fmul(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a % P) * (b % P) % P); }
inline uint32_t fpow(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    for (a %= P; e; e >>= 1, a = fmul(a, a))
        if (e & 1) r = fmul(r, a);
    return r;
}
inline uint32_t finv(uint32_t a) { return fpow(a, P - 2); }   // (0 for a = 0)

// r = a op b over BabyBear; false for an unknown op or a division by zero
inline bool arith(uint32_t op, uint32_t a, uint32_t b, uint32_t* r) {
    a %= P, b %= P;
    switch (op) {
        case OP_ADD: *r = fadd(a, b); return true;
        case OP_SUB: *r = fsub(a, b); return true;
        case OP_MUL: *r = fmul(a, b); return true;
        case OP_DIV:
            if (b == 0) return false;
            *r = fmul(a, finv(b));
            return true;
        default: return false;
    }
}

inline void ext_mul(const uint32_t x[4], const uint32_t y[4], uint32_t z[4]) {
    uint32_t t[7] = {};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) t[i + j] = fadd(t[i + j], fmul(x[i], y[j]));
    for (int i = 0; i < 4; i++) z[i] = i < 3 ? fadd(t[i], fmul(W, t[i + 4])) : t[3];
}
// inverse in F[X] / (X^4 - 11) through the norm to the quadratic subfield F[X^2] and then to F; false for zero
inline bool ext_inv(const uint32_t y[4], uint32_t out[4]) {
    // y = (y0 + y2 X^2) + X (y1 + y3 X^2) = A + X B with A, B in K = F[Y] / (Y^2 - 11), Y = X^2.  y^-1 = (A - X B) / (A^2 - Y B^2)
    auto kmul = [](const uint32_t a[2], const uint32_t b[2], uint32_t r[2]) {
        const uint32_t r0 = fadd(fmul(a[0], b[0]), fmul(W, fmul(a[1], b[1]))), r1 = fadd(fmul(a[0], b[1]), fmul(a[1], b[0]));
        r[0] = r0, r[1] = r1;
    };
    const uint32_t A[2] = {y[0] % P, y[2] % P}, B[2] = {y[1] % P, y[3] % P};
    uint32_t A2[2], B2[2], YB2[2], D[2];
    kmul(A, A, A2), kmul(B, B, B2);
    YB2[0] = fmul(W, B2[1]), YB2[1] = B2[0];   // Y * (b0 + b1 Y) = 11 b1 + b0 Y
    D[0] = fsub(A2[0], YB2[0]), D[1] = fsub(A2[1], YB2[1]);
    const uint32_t n = fsub(fmul(D[0], D[0]), fmul(W, fmul(D[1], D[1])));   // norm of D to F
    if (n == 0) return false;
    const uint32_t ni = finv(n);
    const uint32_t Di[2] = {fmul(D[0], ni), fmul(fsub(0, D[1]), ni)};   // D^-1 = conj(D) / n
    uint32_t RA[2], RB[2];
    kmul(A, Di, RA), kmul(B, Di, RB);
    out[0] = RA[0], out[2] = RA[1], out[1] = fsub(0, RB[0]), out[3] = fsub(0, RB[1]);
    return true;
}
// r = a op b in the quartic extension; false for an unknown op or a division by zero
inline bool ext_arith(uint32_t op, const uint32_t a[4], const uint32_t b[4], uint32_t r[4]) {
    switch (op) {
        case OP_ADD:
            for (int i = 0; i < 4; i++) r[i] = fadd(a[i] % P, b[i] % P);
            return true;
        case OP_SUB:
            for (int i = 0; i < 4; i++) r[i] = fsub(a[i] % P, b[i]);
            return true;
        case OP_MUL: {
            uint32_t t[4];
            ext_mul(a, b, t);
            for (int i = 0; i < 4; i++) r[i] = t[i];
            return true;
        }
        case OP_DIV: {
            uint32_t inv[4], t[4];
            if (!ext_inv(b, inv)) return false;
            ext_mul(a, inv, t);
            for (int i = 0; i < 4; i++) r[i] = t[i];
            return true;
        }
        default: return false;
    }
}

}  // namespace native
}  // namespace zkhip
