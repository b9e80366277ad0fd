// zkhip_keccak.hpp -- the Keccak-f[1600] AIR (crates/circuits/chunk-circuit/openvm.toml:8-59 lists
// `keccak` among the chunk circuit's extensions: the EVM guest spends most of its cells hashing).  OpenVM's keccak chip wraps
// Plonky3's `p3-keccak-air` (Cargo.lock pins p3-keccak-air 0.4.x; un-vendored); this is that AIR's shape restated from the public
// construction: ONE ROUND PER ROW, 24 rows per permutation, 2633 columns, every constraint of degree <= 3:
//   step_flags[24] | export | preimage[5][5][4] | a[5][5][4] | c[5][64] | c_prime[5][64] | a_prime[5][5][64] | a_prime_prime[5][5][4] |
//   a_prime_prime_0_0_bits[64] | a_prime_prime_prime_0_0_limbs[4]          (64-bit lanes as four 16-bit limbs; arrays indexed [y][x])
// theta:  C'[x][z] = C[x][z] ^ C[x-1][z] ^ C[x+1][z-1];  A = A' ^ C ^ C' limb by limb;  sum_y A'[x][y][z] - C'[x][z] in {0, 2, 4}
// rho/pi: B[x][y][z] = A'[(x + 3y) mod 5][x][z - r]   (a re-indexing of the bits of A')
// chi:    A''[x][y] = B[x][y] ^ (~B[x+1][y] & B[x+2][y]), limbs rebuilt from the bit expressions (degree 3)
// iota:   A'''[0][0] = A''[0][0] ^ RC[round]: A''[0][0] is decomposed into bits, the constant is selected by the step flags
// next row: a = a'' (a''' at [0][0]) unless the permutation ends; the preimage is carried through the 24 rows.
// External parity: tests pin the permutation (and the trace's inputs / outputs) to FIPS 202 through hashlib's SHA3 / and the
// Keccak team's zero-state vector (tests/golden/keccak_kat.json).  Header-only; the device generator is csrc/keccak.hip.
#pragma once
#include <cstdint>
#include <vector>

#include "zkhip_air.hpp"

namespace zkhip {
namespace keccak {
using air::AirBuilder;
using air::Expr;

constexpr size_t NUM_ROUNDS = 24, U64_LIMBS = 4, BITS_PER_LIMB = 16;
constexpr size_t COL_FLAGS = 0, COL_EXPORT = 24, COL_PREIMAGE = 25, COL_A = 125, COL_C = 225, COL_C_PRIME = 545, COL_A_PRIME = 865, COL_A_PP = 2465,
                 COL_A_PP_00_BITS = 2565, COL_A_PPP_00 = 2629, WIDTH = 2633;
constexpr unsigned R[5][5] = {   // rotation offsets r[x][y]
    {0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
constexpr uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                             0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                             0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                             0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

// reference permutation on lanes st[x + 5 y]
inline void keccak_f1600(uint64_t st[25]) {
    auto rol = [](uint64_t v, unsigned r) { return r ? (v << r) | (v >> (64 - r)) : v; };
    for (size_t round = 0; round < NUM_ROUNDS; round++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; x++) {
            const uint64_t d = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) st[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(st[x + 5 * y], R[x][y]);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) st[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        st[0] ^= RC[round];
    }
}

inline size_t col_preimage(size_t y, size_t x, size_t limb) { return COL_PREIMAGE + (y * 5 + x) * U64_LIMBS + limb; }
inline size_t col_a(size_t y, size_t x, size_t limb) { return COL_A + (y * 5 + x) * U64_LIMBS + limb; }
inline size_t col_c(size_t x, size_t z) { return COL_C + x * 64 + z; }
inline size_t col_c_prime(size_t x, size_t z) { return COL_C_PRIME + x * 64 + z; }
inline size_t col_a_prime(size_t y, size_t x, size_t z) { return COL_A_PRIME + (y * 5 + x) * 64 + z; }
inline size_t col_a_pp(size_t y, size_t x, size_t limb) { return COL_A_PP + (y * 5 + x) * U64_LIMBS + limb; }

// AirBuilder(WIDTH, 0)
inline void keccak_f_air(AirBuilder& b) {
    auto xor2 = [](Expr p, Expr q) { return p + q - p * q * 2; };
    auto xor3 = [&](Expr p, Expr q, Expr r) { return xor2(xor2(p, q), r); };
    Expr flag[24];
    for (size_t i = 0; i < 24; i++) flag[i] = b.var(COL_FLAGS + i);
    const Expr exp = b.var(COL_EXPORT), first_step = flag[0], final_step = flag[23], not_final = 1 - final_step;
    // round flags: a one-hot counter that starts at 0 and rotates
    b.when_first_row(flag[0] - 1);
    for (size_t i = 1; i < 24; i++) b.when_first_row(flag[i]);
    for (size_t i = 0; i < 24; i++) b.when_transition(b.next(COL_FLAGS + (i + 1) % 24) - flag[i]);
    b.assert_zero(exp * (exp - 1));
    b.assert_zero(exp * not_final);
    for (size_t y = 0; y < 5; y++)
        for (size_t x = 0; x < 5; x++)
            for (size_t l = 0; l < U64_LIMBS; l++) {
                b.assert_zero(first_step * (b.var(col_preimage(y, x, l)) - b.var(col_a(y, x, l))));
                b.when_transition(not_final * (b.var(col_preimage(y, x, l)) - b.next(col_preimage(y, x, l))));
            }
    // theta
    for (size_t x = 0; x < 5; x++)
        for (size_t z = 0; z < 64; z++) {
            const Expr c = b.var(col_c(x, z)), cp = b.var(col_c_prime(x, z));
            b.assert_zero(c * (c - 1));
            b.assert_zero(cp - xor3(c, b.var(col_c((x + 4) % 5, z)), b.var(col_c((x + 1) % 5, (z + 63) % 64))));
        }
    for (size_t y = 0; y < 5; y++)
        for (size_t x = 0; x < 5; x++)
            for (size_t l = 0; l < U64_LIMBS; l++) {
                Expr sum = b.constant(0);
                for (size_t k = 0; k < BITS_PER_LIMB; k++) {
                    const size_t z = l * BITS_PER_LIMB + k;
                    const Expr ap = b.var(col_a_prime(y, x, z));
                    if (l == 0 && k == 0) (void)ap;
                    sum = sum + xor3(ap, b.var(col_c(x, z)), b.var(col_c_prime(x, z))) * (int64_t)(1u << k);
                }
                b.assert_zero(b.var(col_a(y, x, l)) - sum);
            }
    for (size_t y = 0; y < 5; y++)
        for (size_t x = 0; x < 5; x++)
            for (size_t z = 0; z < 64; z++) {
                const Expr ap = b.var(col_a_prime(y, x, z));
                b.assert_zero(ap * (ap - 1));
            }
    for (size_t x = 0; x < 5; x++)
        for (size_t z = 0; z < 64; z++) {
            Expr sum = b.var(col_a_prime(0, x, z));
            for (size_t y = 1; y < 5; y++) sum = sum + b.var(col_a_prime(y, x, z));
            const Expr diff = sum - b.var(col_c_prime(x, z));
            b.assert_zero(diff * (diff - 2) * (diff - 4));
        }
    // rho, pi, chi
    auto bbit = [&](size_t x, size_t y, size_t z) {
        const size_t a = (x + 3 * y) % 5, bb = x;
        return b.var(col_a_prime(bb, a, (z + 64 - R[a][bb]) % 64));
    };
    for (size_t y = 0; y < 5; y++)
        for (size_t x = 0; x < 5; x++)
            for (size_t l = 0; l < U64_LIMBS; l++) {
                Expr sum = b.constant(0);
                for (size_t k = 0; k < BITS_PER_LIMB; k++) {
                    const size_t z = l * BITS_PER_LIMB + k;
                    const Expr andn = (1 - bbit((x + 1) % 5, y, z)) * bbit((x + 2) % 5, y, z);
                    sum = sum + xor2(bbit(x, y, z), andn) * (int64_t)(1u << k);
                }
                b.assert_zero(b.var(col_a_pp(y, x, l)) - sum);
            }
    // iota on lane (0, 0)
    for (size_t l = 0; l < U64_LIMBS; l++) {
        Expr bits_sum = b.constant(0), out_sum = b.constant(0);
        for (size_t k = 0; k < BITS_PER_LIMB; k++) {
            const size_t z = l * BITS_PER_LIMB + k;
            const Expr bit = b.var(COL_A_PP_00_BITS + z);
            b.assert_zero(bit * (bit - 1));
            bits_sum = bits_sum + bit * (int64_t)(1u << k);
            Expr rc = b.constant(0);
            bool any = false;
            for (size_t r = 0; r < 24; r++)
                if ((RC[r] >> z) & 1) rc = rc + flag[r], any = true;
            out_sum = out_sum + (any ? xor2(bit, rc) : bit) * (int64_t)(1u << k);
        }
        b.assert_zero(b.var(col_a_pp(0, 0, l)) - bits_sum);
        b.assert_zero(b.var(COL_A_PPP_00 + l) - out_sum);
    }
    // the next round starts from this round's output
    for (size_t y = 0; y < 5; y++)
        for (size_t x = 0; x < 5; x++)
            for (size_t l = 0; l < U64_LIMBS; l++) {
                const Expr out = (x == 0 && y == 0) ? b.var(COL_A_PPP_00 + l) : b.var(col_a_pp(y, x, l));
                b.when_transition(not_final * (b.next(col_a(y, x, l)) - out));
            }
}

// The chip inside the VM (include/zkhip_vm_circuit.hpp): the same AIR plus a timestamp column; on its export row every lane's
// (timestamp, lane index, preimage limbs, output limbs) is received from the keccak adapter on `lane_bus` -- 25 receives that tie
// the permutation to the 50 memory words the adapter reads and writes.  AirBuilder(VM_WIDTH, 0)
constexpr size_t COL_TS = WIDTH, VM_WIDTH = WIDTH + 1;
inline void keccak_vm_air(AirBuilder& b, uint32_t lane_bus) {
    keccak_f_air(b);
    const Expr exp = b.var(COL_EXPORT), ts = b.var(COL_TS);
    for (size_t y = 0; y < 5; y++)
        for (size_t x = 0; x < 5; x++) {
            std::vector<Expr> m{ts, b.constant((uint32_t)(x + 5 * y))};
            for (size_t l = 0; l < U64_LIMBS; l++) m.push_back(b.var(col_preimage(y, x, l)));
            for (size_t l = 0; l < U64_LIMBS; l++) m.push_back((x == 0 && y == 0) ? b.var(COL_A_PPP_00 + l) : b.var(col_a_pp(y, x, l)));
            b.push_interaction(lane_bus, m, exp, air::Kind::Receive);
        }
}

}  // namespace keccak
}  // namespace zkhip
