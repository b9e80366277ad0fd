"""The guest execution step (include/zkhip_vm.hpp, mirror of crates/prover/src/utils/vm.rs execute_guest) against an independent
Python RV32IM interpreter (tests/rv32_model.py): instruction count, public values and the per-chip execution records the
device trace generators take; the reference's error behaviour (all-zero public values, metered fall-back)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rv32_model as rv  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VM = os.path.join(ROOT, "zkvm-prover_amd", "vm_cli")
if not os.path.exists(VM):  # host-only tool: one g++ call (normally built by __graft_entry__.build() / make)
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "vm_cli.cpp"), "-o", VM], check=True)
A0, A1, A7, T0, T1, T2, T3, T4, S0, S1, SP = 10, 11, 17, 5, 6, 7, 28, 29, 8, 9, 2


def fib_program():
    """reads n from the input stream, reveals fib(n) (mod 2^32) as public word 0 and n as word 1"""
    return rv.assemble([
        ("addi", A7, 0, 2), ("ecall",),                 # a0 = n
        ("add", S0, A0, 0),
        ("addi", T0, 0, 0), ("addi", T1, 0, 1), ("addi", T2, 0, 0),
        ("label", "loop"), ("bge", T2, S0, "done"),
        ("add", T3, T0, T1), ("add", T0, T1, 0), ("add", T1, T3, 0), ("addi", T2, T2, 1), ("jal", 0, "loop"),
        ("label", "done"),
        ("add", A0, T0, 0), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
        ("add", A0, S0, 0), ("addi", A1, 0, 1), ("ecall",),
        ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)])


def keccak_program(n_calls=2):
    """SHA3-256 of a one-block message through the keccak intrinsic (a7 = 3): the padded block sits in the data segment (keccak_data),
    the guest permutes it in place `n_calls` times (the digest of the FIRST permutation is saved before) and reveals eight digest words."""
    p = rv.li(S0, 0x00400000)
    p += [("add", A0, S0, 0), ("addi", A7, 0, 3), ("ecall",)]
    for k in range(8):                                                    # keep the digest: the next call overwrites the state
        p += [("lw", T0, S0, 4 * k), ("sw", T0, S0, 256 + 4 * k)]
    for _ in range(n_calls - 1):
        p += [("add", A0, S0, 0), ("addi", A7, 0, 3), ("ecall",)]
    p += [("lw", T1, S0, 0)]                                              # a word of the twice-permuted state, folded into word 7
    for k in range(8):
        p += [("lw", A0, S0, 256 + 4 * k)]
        if k == 7:
            p += [("xor", A0, A0, T1)]
        p += [("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def keccak_data(msg):
    assert len(msg) < 136
    blk = bytearray(200)
    blk[:len(msg)] = msg
    blk[len(msg)] ^= 0x06
    blk[135] ^= 0x80
    return bytes(blk)


def sha256_program(n_blocks):
    """SHA-256 of a padded message through the sha256 intrinsic (a7 = 4): the buffer at the data base holds the state (IV) and one
    block; the following blocks sit behind it and are copied into the buffer's block slot one after the other (sha256_data); the eight
    digest words are revealed."""
    p = rv.li(S0, 0x00400000)
    for b in range(n_blocks):
        if b:
            for k in range(16):                                            # next block into the buffer's message slot
                p += [("lw", T0, S0, 96 + 64 * (b - 1) + 4 * k), ("sw", T0, S0, 32 + 4 * k)]
        p += [("add", A0, S0, 0), ("addi", A7, 0, 4), ("ecall",)]
    for k in range(8):
        p += [("lw", A0, S0, 4 * k), ("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def sha256_data(msg):
    """IV | first block | further blocks: words as plain little-endian u32 values of the big-endian message words (FIPS 180-4 5.1.1, 5.2.1)"""
    import struct

    iv = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
    data = msg + b"\x80" + b"\x00" * ((55 - len(msg)) % 64) + (8 * len(msg)).to_bytes(8, "big")
    words = struct.unpack(">%dI" % (len(data) // 4), data)
    return struct.pack("<8I", *iv) + struct.pack("<%dI" % len(words), *words), len(data) // 64


def int256_program():
    """256-bit words through the int256 intrinsic (a7 = 6, a1 = opcode): twelve buffers (b, c, result slot) at the data base, one per
    opcode (add, sub, xor, or, and, mul, the comparisons b < c unsigned, b < c signed, b == c, the shifts left, right, right
    arithmetic; int256_data); the six results' low words and two high words (sub, mul) are revealed, the three comparison bits folded
    into the last one, the left shift's low word and the right shifts' high words into the one before."""
    p = rv.li(S0, 0x00400000)
    for op in range(12):
        p += [("addi", A0, S0, 96 * op), ("addi", A1, 0, op), ("addi", A7, 0, 6), ("ecall",)]
    p += [("lw", T3, S0, 96 * 9 + 64), ("lw", T2, S0, 96 * 10 + 64 + 28), ("xor", T3, T3, T2), ("lw", T2, S0, 96 * 11 + 64 + 28), ("xor", T3, T3, T2)]
    p += [("lw", T1, S0, 96 * 6 + 64), ("lw", T2, S0, 96 * 7 + 64), ("slli", T2, T2, 1), ("or", T1, T1, T2), ("lw", T2, S0, 96 * 8 + 64), ("slli", T2, T2, 2),
          ("or", T1, T1, T2)]
    for k in range(6):
        p += [("lw", A0, S0, 96 * k + 64), ("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    for j, k in enumerate((1, 5)):
        p += [("lw", A0, S0, 96 * k + 64 + 28)]
        p += [("xor", A0, A0, T1 if k == 5 else T3)]
        p += [("addi", A1, 0, 6 + j), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


INT256_OPERANDS = [((1 << 256) - 5, 77), (3, 0xFFFF_FFFF_FFFF_FFFF_FFFF), (0x0123456789ABCDEF << 190 | 0xF0F0, 0xFEDCBA9876543210 << 180 | 0x0FF0),
                   (0xAAAA << 240 | 0x5555, 0x1234 << 240 | 0xAAAA), (0xFFFF << 240 | 0xFF00FF, 0xF0F0 << 240 | 0x0FF0F0),
                   (0xFEDCBA9876543210FEDCBA9876543210FEDCBA98, 0x123456789ABCDEF0123456789ABCDEF0123456789)]


# sltu: 2^256 - 5 < 77 is false; slt: -5 < 77 is true; eq: true
INT256_CMP_OPERANDS = [((1 << 256) - 5, 77), ((1 << 256) - 5, 77), (0x1234 << 200 | 99, 0x1234 << 200 | 99)]
# sll by 13, srl by 100, sra by 77 of a negative word (the amount is c mod 256: 256 + 77)
INT256_SHIFT_OPERANDS = [(0x0123456789ABCDEF0123456789ABCDEF, 13), (0xFEDCBA98 << 224 | 0x1234, 100), (0x87654321 << 224 | 0xABCD, 256 + 77)]


def int256_data():
    return b"".join(int(b).to_bytes(32, "little") + int(c).to_bytes(32, "little") + bytes(32) for b, c in INT256_OPERANDS + INT256_CMP_OPERANDS + INT256_SHIFT_OPERANDS)


# ---- 256-bit branches (round 6; the bigint extension's Rv32BranchEqual256 / Rv32BranchLessThan256, crates/circuits/chunk-circuit/openvm.toml:17-18):
# int256 opcodes 12 beq, 13 bne, 14 bltu, 15 blt, 16 bgeu, 17 bge; a2 = the byte offset from the ecall ----
A2 = 12
# (b, c) per opcode 12 .. 17, twice: first six pairs make the branch TAKEN, the other six NOT taken
BRANCH256_OPERANDS = [(7 << 200 | 5, 7 << 200 | 5), (1 << 255, 1), (3, 1 << 250), ((1 << 256) - 9, 4), (1 << 255, 1 << 254), (6, (1 << 256) - 6),
                      (7 << 200 | 5, 7 << 200 | 4), (99, 99), (1 << 250, 3), (4, (1 << 256) - 9), (1 << 254, 1 << 255), ((1 << 256) - 6, 6)]
BRANCH256_LOOP = 5     # the backward branch: a loop that runs until a 256-bit counter's low word reaches this


def branch256_program():
    """Twelve forward branches (every opcode taken once and not taken once): a taken branch skips the `addi t1, t1, 1` before t1 is shifted
    left (t1 = the mask of the branches NOT taken, branch k at bit 12 - k); then a
    loop closed by a BACKWARD bne256 (a2 negative): t2 counts up in the low word of a 256-bit value until it equals BRANCH256_LOOP.  Reveals
    t1 (the mask of the branches that were NOT taken), t2 (the loop's count), and the comparison word the last forward branch left in its buffer."""
    p = rv.li(S0, 0x00400000) + [("addi", T1, 0, 0)]
    for k in range(12):
        p += [("addi", A0, S0, 96 * k), ("addi", A1, 0, 12 + k % 6), ("addi", A2, 0, 8), ("addi", A7, 0, 6), ("ecall",), ("addi", T1, T1, 1), ("slli", T1, T1, 1)]
    # the loop: buffer 12 = [limit | counter | slot]; counter's low word = t2
    p += [("addi", T2, 0, 0), ("label", "loop"), ("addi", T2, T2, 1), ("sw", T2, S0, 96 * 12 + 32),
          ("addi", A0, S0, 96 * 12), ("addi", A1, 0, 13), ("addi", A2, 0, -24), ("addi", A7, 0, 6), ("ecall",)]     # bne256 limit, counter: back to "loop" (6 instructions up)
    p += [("add", A0, T1, 0), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
          ("add", A0, T2, 0), ("addi", A1, 0, 1), ("addi", A7, 0, 1), ("ecall",),
          ("lw", A0, S0, 96 * 11 + 64), ("addi", A1, 0, 2), ("addi", A7, 0, 1), ("ecall",),
          ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def branch256_data():
    out = b"".join(int(b).to_bytes(32, "little") + int(c).to_bytes(32, "little") + bytes(32) for b, c in BRANCH256_OPERANDS)
    return out + int(BRANCH256_LOOP).to_bytes(32, "little") + bytes(64)


# the six moduli of the reference's chunk circuit, in its order (crates/circuits/chunk-circuit/openvm.toml:20-28): bn254 p, bn254 r,
# secp256k1 p, secp256k1 n, P-256 p, P-256 n
CHUNK_CIRCUIT_MODULI = (
    21888242871839275222246405745257275088696311157297823662689037894645226208583,
    21888242871839275222246405745257275088548364400416034343698204186575808495617,
    115792089237316195423570985008687907853269984665640564039457584007908834671663,
    115792089237316195423570985008687907852837564279074904382605163141518161494337,
    115792089210356248762697446949407573530086143415290314195533631308867097853951,
    115792089210356248762697446949407573529996955224135760342422259061068512044369)


def chunk_circuit_toml(params):
    """the sections of the reference's chunk-circuit openvm.toml this backend reads, in the reference's syntax"""
    return ("[app_fri_params.fri_params]\nlog_blowup = %d\nlog_final_poly_len = %d\nnum_queries = %d\ncommit_proof_of_work_bits = %d\n"
            "query_proof_of_work_bits = %d\n\n[app_vm_config.rv32i]\n\n[app_vm_config.io]\n\n[app_vm_config.keccak]\n\n[app_vm_config.rv32m]\n"
            "range_tuple_checker_sizes = [256, 8192]\n\n[app_vm_config.bigint]\nrange_tuple_checker_sizes = [256, 8192]\n\n[app_vm_config.modular]\n"
            "supported_moduli = [\n" % tuple(params)) + ",\n".join('    "%d"' % m for m in CHUNK_CIRCUIT_MODULI) + "\n]\n\n[app_vm_config.fp2]\nsupported_moduli = [\n    [\"Bn254Fp2\",\"%d\"]\n]\n\n[app_vm_config.sha2]\n\n" % CHUNK_CIRCUIT_MODULI[0] + "".join(
                '[[app_vm_config.ecc.supported_curves]]\nstruct_name = "%s"\nmodulus = "%d"\nscalar = "%d"\na = "%d"\nb = "%d"\n\n' % c for c in CHUNK_CIRCUIT_CURVES)


# crates/circuits/chunk-circuit/openvm.toml:38-59 (struct_name, modulus, scalar, a, b)
CHUNK_CIRCUIT_CURVES = (
    ("Secp256k1Point", 115792089237316195423570985008687907853269984665640564039457584007908834671663,
     115792089237316195423570985008687907852837564279074904382605163141518161494337, 0, 7),
    ("P256Point", 115792089210356248762697446949407573530086143415290314195533631308867097853951,
     115792089210356248762697446949407573529996955224135760342422259061068512044369,
     115792089210356248762697446949407573530086143415290314195533631308867097853948,
     41058363725152142129326129780047268409114441015993725554835256314039467401291),
    ("Bn254G1Affine", 21888242871839275222246405745257275088696311157297823662689037894645226208583,
     21888242871839275222246405745257275088548364400416034343698204186575808495617, 0, 3),
)


def all_extensions_program(with_ecc=False):
    """One guest, four intrinsics: keccak-f on the block at +0, sha256 compression on the buffer at +256, a secp256k1 field product
    (modulus 2 of the chunk circuit's list) at +384, a 256-bit subtraction at +480; reveals two words of each result."""
    p = rv.li(S0, 0x00400000)
    p += [("add", A0, S0, 0), ("addi", A7, 0, 3), ("ecall",)]
    p += [("addi", A0, S0, 256), ("addi", A7, 0, 4), ("ecall",)]
    p += [("addi", A0, S0, 384), ("addi", A1, 0, 2), ("addi", A7, 0, 5), ("ecall",)]
    p += [("addi", A0, S0, 480), ("addi", A1, 0, 1), ("addi", A7, 0, 6), ("ecall",)]
    if with_ecc:   # the double of the secp256k1 generator (curve 0 of the chunk circuit's list) at +576; its low word is folded into word 7
        p += [("addi", A0, S0, 576), ("addi", A1, 0, 8), ("addi", A7, 0, 7), ("ecall",), ("lw", T1, S0, 576 + 128)]
    for k, off in enumerate((0, 4, 256, 260, 384 + 64, 384 + 68, 480 + 64, 480 + 92)):
        p += [("lw", A0, S0, off)]
        if with_ecc and k == 7:
            p += [("xor", A0, A0, T1)]
        p += [("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


ALL_EXT_MSG = b"four intrinsics, one statement"


def all_extensions_data():
    sha, _ = sha256_data(ALL_EXT_MSG)          # IV | one padded block = 96 bytes
    b32 = lambda v: int(v).to_bytes(32, "little")  # noqa: E731
    return (keccak_data(ALL_EXT_MSG) + bytes(56) + sha + bytes(32) + b32(SECP256K1_GX) + b32(SECP256K1_GY) + bytes(32) + b32(5) + b32(7) + bytes(32) +
            b32(SECP256K1_GX) + b32(SECP256K1_GY) + bytes(128))


def modmul_program():
    """The secp256k1 generator's curve equation through the modmul intrinsic (a7 = 5, a1 = modulus index): buffer 0 = (Gy, Gy, .),
    buffer 1 = (Gx, Gx, .), buffer 2 = (Gx^2, Gx, .) at the data base (modmul_data); reveals the low words of y^2 and x^3 (words 0..3 and
    4..7) -- they differ by 7 modulo p -- and one product modulo the second modulus folded into word 7; then x^3 + 7 (modular
    addition), (x^3 + 7) - y^2 (modular subtraction: zero), (x^3 + 7) / y^2 (modular division: one) and the equality tests y^2 = x^3 + 7 (one), 5 = 6 (zero)
    through the same intrinsic with a1 = index + 8 op."""
    p = rv.li(S0, 0x00400000)
    def call(buf, which):
        return [("addi", A0, S0, 96 * buf), ("addi", A1, 0, which), ("addi", A7, 0, 5), ("ecall",)]
    p += call(0, 0) + call(1, 0)
    for k in range(8):                                                     # x^2 into buffer 2's first operand
        p += [("lw", T0, S0, 96 + 64 + 4 * k), ("sw", T0, S0, 192 + 4 * k)]
    p += call(2, 0)
    p += call(3, 1)                                                        # buffer 3 under the other modulus
    for k in range(8):                                                     # x^3 into buffer 4's first operand (its second is 7): x^3 + 7
        p += [("lw", T0, S0, 192 + 64 + 4 * k), ("sw", T0, S0, 384 + 4 * k)]
    p += call(4, 8)                                                        # a1 = 0 + 8 * 1: modulus 0, addition
    for k in range(8):                                                     # (x^3 + 7) - y^2 into buffer 5: must be zero
        p += [("lw", T0, S0, 384 + 64 + 4 * k), ("sw", T0, S0, 480 + 4 * k), ("lw", T0, S0, 64 + 4 * k), ("sw", T0, S0, 480 + 32 + 4 * k)]
    p += call(5, 16)                                                       # subtraction
    for k in range(8):                                                     # (x^3 + 7) / y^2 into buffer 6: must be one
        p += [("lw", T0, S0, 384 + 64 + 4 * k), ("sw", T0, S0, 576 + 4 * k), ("lw", T0, S0, 64 + 4 * k), ("sw", T0, S0, 576 + 32 + 4 * k)]
    p += call(6, 24)                                                       # a1 = 0 + 8 * 3: modulus 0, division
    p += [("lw", T1, S0, 288 + 64), ("lw", T2, S0, 480 + 64), ("or", T1, T1, T2)]   # (the difference's low word is zero: folding it in changes nothing)
    p += [("lw", T2, S0, 576 + 64), ("addi", T2, T2, -1), ("or", T1, T1, T2), ("lw", T2, S0, 576 + 68), ("or", T1, T1, T2)]   # (nor does the quotient minus one)
    for k in range(8):                                                     # is y^2 = (x^3 + 7) - 0?  buffer 7 = (y^2, x^3 + 7, .): the bit must be one
        p += [("lw", T0, S0, 64 + 4 * k), ("sw", T0, S0, 672 + 4 * k), ("lw", T0, S0, 384 + 64 + 4 * k), ("sw", T0, S0, 672 + 32 + 4 * k)]
    p += call(7, 32)                                                       # a1 = 0 + 8 * 4: modulus 0, equality test
    p += call(8, 32)                                                       # buffer 8 = (5, 6, .): the bit must be zero
    p += [("lw", T2, S0, 672 + 64), ("addi", T2, T2, -1), ("or", T1, T1, T2), ("lw", T2, S0, 768 + 64), ("or", T1, T1, T2)]   # (both as expected: nothing changes)
    for k in range(4):
        p += [("lw", A0, S0, 64 + 4 * k), ("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    for k in range(4):
        p += [("lw", A0, S0, 192 + 64 + 4 * k)]
        if k == 3:
            p += [("xor", A0, A0, T1)]
        p += [("addi", A1, 0, 4 + k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


SECP256K1_P = 2**256 - 2**32 - 977
SECP256K1_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
SECP256K1_GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
SECP256K1_GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8


def modmul_data():
    b32 = lambda v: int(v).to_bytes(32, "little")  # noqa: E731
    z32 = bytes(32)
    return (b32(SECP256K1_GY) + b32(SECP256K1_GY) + z32 + b32(SECP256K1_GX) + b32(SECP256K1_GX) + z32 + z32 + b32(SECP256K1_GX) + z32 +
            b32(SECP256K1_N - 2) + b32(SECP256K1_N - 3) + z32 + z32 + b32(7) + z32 + z32 + z32 + z32 + z32 + z32 + z32 + z32 + z32 + z32 + b32(5) + b32(6) + z32)


BN254_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
EC_CURVES = ((SECP256K1_P, 0), (BN254_P, 0))   # (modulus, a): secp256k1, bn254 G1


def ec_program():
    """3 G on secp256k1 through the ecc intrinsic (a7 = 7, a1 = curve index + 8 op): buffer 0 = (G, G, .) doubled, its result copied into
    buffer 1 = (2 G, G, .) and added; buffer 2 = bn254's generator (1, 2) doubled under curve 1.  Reveals the low four words of 3 G's
    abscissa and ordinate, the low word of bn254's 2 G folded into word 7."""
    p = rv.li(S0, 0x00400000)
    def call(buf, which):
        return [("addi", A0, S0, 192 * buf), ("addi", A1, 0, which), ("addi", A7, 0, 7), ("ecall",)]
    p += call(0, 8)                                                        # double on curve 0
    for k in range(16):
        p += [("lw", T0, S0, 128 + 4 * k), ("sw", T0, S0, 192 + 4 * k)]
    p += call(1, 0)                                                        # chord addition on curve 0
    p += call(2, 9)                                                        # double on curve 1
    p += [("lw", T1, S0, 384 + 128)]
    for k in range(4):
        p += [("lw", A0, S0, 192 + 128 + 4 * k), ("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    for k in range(4):
        p += [("lw", A0, S0, 192 + 160 + 4 * k)]
        if k == 3:
            p += [("xor", A0, A0, T1)]
        p += [("addi", A1, 0, 4 + k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def ec_data():
    b32 = lambda v: int(v).to_bytes(32, "little")  # noqa: E731
    g = b32(SECP256K1_GX) + b32(SECP256K1_GY)
    return g + g + bytes(64) + bytes(64) + g + bytes(64) + b32(1) + b32(2) + bytes(128)


# EIP-197's generator of bn254's G2 (x = X0 + X1 u, y = Y0 + Y1 u)
BN254_G2X = (10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634)
BN254_G2Y = (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531)


def fp2_program():
    """The twist equation of bn254 at the published G2 generator through the fp2 intrinsic (a7 = 8, a1 = field index + 8 op): buffer 0 =
    (y, y, .) multiplied, buffer 1 = (x, x, .) multiplied, its result copied into buffer 2 = (x^2, x, .) multiplied: x^3; buffer 3 =
    (3, 9 + u, .) divided; buffer 4 = (x^3, 3 / (9 + u), .) added; buffer 5 = (y^2, x^3 + b', .) subtracted: zero.  Reveals the low words
    of y^2 (both components, two words each) and of x^3 + b' (likewise), the difference's low words folded into the last one."""
    p = rv.li(S0, 0x00400000)
    def call(buf, which):
        return [("addi", A0, S0, 192 * buf), ("addi", A1, 0, which), ("addi", A7, 0, 8), ("ecall",)]
    def copy(src_buf, dst_buf, dst_slot):
        out = []
        for k in range(16):
            out += [("lw", T0, S0, 192 * src_buf + 128 + 4 * k), ("sw", T0, S0, 192 * dst_buf + 64 * dst_slot + 4 * k)]
        return out
    p += call(0, 0) + call(1, 0) + copy(1, 2, 0) + call(2, 0) + call(3, 24) + copy(2, 4, 0) + copy(3, 4, 1) + call(4, 8) + copy(0, 5, 0) + copy(4, 5, 1) + call(5, 16)
    p += [("lw", T1, S0, 192 * 5 + 128), ("lw", T2, S0, 192 * 5 + 160), ("or", T1, T1, T2)]
    for k, off in enumerate((128, 132, 160, 164, 192 * 4 + 128, 192 * 4 + 132, 192 * 4 + 160, 192 * 4 + 164)):
        p += [("lw", A0, S0, off)]
        if k == 7:
            p += [("xor", A0, A0, T1)]
        p += [("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def fp2_data():
    b32 = lambda v: int(v).to_bytes(32, "little")  # noqa: E731
    el = lambda e: b32(e[0]) + b32(e[1])  # noqa: E731
    z64 = bytes(64)
    return (el(BN254_G2Y) + el(BN254_G2Y) + z64 + el(BN254_G2X) + el(BN254_G2X) + z64 + z64 + el(BN254_G2X) + z64 + el((3, 0)) + el((9, 1)) + z64 + z64 * 3 + z64 * 3)




# ---- a CHUNK-LIKE guest: what a chunk circuit's guest spends its instructions on -- register arithmetic, strided
# loads, Keccak-f and SHA-256 calls (MPT hashing), secp256k1 point additions / doublings (ecrecover), modular and 256-bit arithmetic -- in
# PHASES, so that the segments of one run land in all of the app's shapes: plain phases in the lean 22-chip shape, hash phases in the
# 26-chip shape (base + keccak + sha256), full phases in the 51-chip shape of the reference's chunk-circuit configuration.
MIXED_PHASE_ITERATIONS = 128          # iterations per phase; the phase of iteration i is (i >> 7) & 3: 0 plain, 1 and 2 hash, 3 full
MIXED_MIX = {"plain instructions per iteration": 5 * 256 + 4 * 64 + 12, "strided loads per iteration": 64, "keccak_f per hash / full iteration": 1,
             "sha256 blocks per hash / full iteration": 1, "secp256k1 add + double per full iteration": 2, "modular mul + div per full iteration": 2,
             "int256 add + mul per full iteration": 2}
_MX = dict(KK=0x000, SHA=0x100, ECA=0x200, ECD=0x300, MM=0x400, MD=0x480, I1=0x500, I2=0x580, ARR=0x1000)


def mixed_chunk_program():
    """reads the iteration count n from the input stream; every iteration runs a register-only loop (256 x 5 instructions) and a strided
    load loop (64 x 4, stride 68 bytes); iterations of a hash phase add one Keccak-f call and one SHA-256 block; iterations of a full phase
    add Q <- Q + G and D <- 2 D on secp256k1 (curve 0), a <- a b and r = a / b modulo secp256k1's p (modulus 2) and a 256-bit addition and
    multiplication.  Reveals words of every running state (the checks of tests/test_vm2_cpu.py compare them with the Python model)."""
    S2, S3, S4, T5 = 18, 19, 20, 30
    X = _MX
    p = [("addi", A7, 0, 2), ("ecall",), ("add", S1, A0, 0)]                       # s1 = n
    p += rv.li(S0, 0x00400000) + rv.li(S3, 0x00400000 + X["ARR"])
    p += [("addi", S2, 0, 0), ("addi", S4, 0, 0),
          ("label", "outer"), ("bge", S2, S1, "done"),
          # ---- every phase: a register-only loop, then strided loads
          ("addi", T0, 0, 0), ("addi", T1, 0, 1), ("addi", T2, 0, 256),
          ("label", "fib"), ("add", T3, T0, T1), ("add", T0, T1, 0), ("add", T1, T3, 0), ("addi", T2, T2, -1), ("bne", T2, 0, "fib"),
          ("add", S4, S4, T0),
          ("add", T4, S3, 0), ("addi", T2, 0, 64),
          ("label", "lds"), ("lw", T3, T4, 0), ("add", S4, S4, T3), ("addi", T4, T4, 68), ("addi", T2, T2, -1), ("bne", T2, 0, "lds")]
    # the phase: (i >> 7) & 3   (bne over the addi/ecall blocks: branch targets are labels)
    p += [("srli", T5, S2, 7), ("andi", T5, T5, 3), ("beq", T5, 0, "next"),
          # ---- hash phases: Keccak-f in place, one SHA-256 compression (the state chains)
          ("addi", A0, S0, X["KK"]), ("addi", A7, 0, 3), ("ecall",),
          ("addi", A0, S0, X["SHA"]), ("addi", A7, 0, 4), ("ecall",),
          ("addi", T0, 0, 3), ("bne", T5, T0, "next"),
          # ---- full phase: Q <- Q + G
          ("addi", A0, S0, X["ECA"]), ("addi", A1, 0, 0), ("addi", A7, 0, 7), ("ecall",)]
    for k in range(16):
        p += [("lw", T0, S0, X["ECA"] + 128 + 4 * k), ("sw", T0, S0, X["ECA"] + 4 * k)]
    p += [("addi", A0, S0, X["ECD"]), ("addi", A1, 0, 8), ("addi", A7, 0, 7), ("ecall",)]          # D <- 2 D
    for k in range(16):
        p += [("lw", T0, S0, X["ECD"] + 128 + 4 * k), ("sw", T0, S0, X["ECD"] + 4 * k)]
    p += [("addi", A0, S0, X["MM"]), ("addi", A1, 0, 2), ("addi", A7, 0, 5), ("ecall",)]           # a <- a b mod secp256k1 p
    for k in range(8):
        p += [("lw", T0, S0, X["MM"] + 64 + 4 * k), ("sw", T0, S0, X["MM"] + 4 * k), ("sw", T0, S0, X["MD"] + 4 * k)]
    p += [("addi", A0, S0, X["MD"]), ("addi", A1, 0, 2 + 8 * 3), ("addi", A7, 0, 5), ("ecall",)]   # r = a / b
    p += [("addi", A0, S0, X["I1"]), ("addi", A1, 0, 0), ("addi", A7, 0, 6), ("ecall",)]           # 256-bit a = b + c; b <- a
    for k in range(8):
        p += [("lw", T0, S0, X["I1"] + 64 + 4 * k), ("sw", T0, S0, X["I1"] + 4 * k)]
    p += [("addi", A0, S0, X["I2"]), ("addi", A1, 0, 5), ("addi", A7, 0, 6), ("ecall",)]           # 256-bit a = b c; b <- a
    for k in range(8):
        p += [("lw", T0, S0, X["I2"] + 64 + 4 * k), ("sw", T0, S0, X["I2"] + 4 * k)]
    p += [("label", "next"), ("addi", S2, S2, 1), ("jal", 0, "outer"), ("label", "done")]
    reveal = [("add", A0, S4, 0)] , [("lw", A0, S0, X["KK"])], [("lw", A0, S0, X["SHA"])], [("lw", A0, S0, X["ECA"])], [("lw", A0, S0, X["ECD"] + 32)], \
             [("lw", A0, S0, X["MM"])], [("lw", A0, S0, X["MD"] + 64), ("lw", T0, S0, X["I1"]), ("xor", A0, A0, T0)], [("lw", A0, S0, X["I2"])]
    for k, ld in enumerate(reveal):
        p += list(ld) + [("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def mixed_chunk_data():
    import struct

    b32 = lambda v: int(v).to_bytes(32, "little")  # noqa: E731
    X = _MX
    two_g = (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5, 0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A)   # 2 G (published)
    d = bytearray(X["ARR"] + 64 * 68 + 64)
    d[X["KK"]:X["KK"] + 200] = keccak_data(b"scroll-zkvm mixed guest")
    d[X["SHA"]:X["SHA"] + 96] = sha256_data(b"mixed")[0][:96]
    d[X["ECA"]:X["ECA"] + 128] = b32(two_g[0]) + b32(two_g[1]) + b32(SECP256K1_GX) + b32(SECP256K1_GY)                 # Q = 2 G, G
    d[X["ECD"]:X["ECD"] + 64] = b32(SECP256K1_GX) + b32(SECP256K1_GY)                                                  # D = G
    d[X["MM"]:X["MM"] + 64] = b32(SECP256K1_GX) + b32(SECP256K1_GY)                                                    # a, b
    d[X["MD"] + 32:X["MD"] + 64] = b32(SECP256K1_GY)                                                                   # the divisor
    d[X["I1"]:X["I1"] + 64] = b32(0x0123456789ABCDEF << 128 | 77) + b32((1 << 255) + 12345)
    d[X["I2"]:X["I2"] + 64] = b32(3) + b32(0x10001)
    for k in range(64 * 17 + 16):
        struct.pack_into("<I", d, X["ARR"] + 4 * k, (2654435761 * (k + 1)) & 0xFFFFFFFF)
    return bytes(d)


# ---- the pairing extension (`[app_vm_config.pairing] supported_curves = ["Bn254"]`, crates/circuits/chunk-circuit/openvm.toml:35-36): no chip,
# a phantom sub-executor that leaves the final-exponentiation witness in the hint stream (include/zkhip_pairing.hpp) ----
def pairing_hint_program(n_hint_words=192):
    """phantom kind 2 on the buffer [curve | f] at the data base; then the hinted words (c and the scaling factor: 192 for Bn254, 288 for
    Bls12_381) are read one by one (a7 = 2) and XOR-folded into eight words (word k into fold k mod 8), which are revealed"""
    S2 = 18
    p = rv.li(S0, 0x00400000) + [("phantom", 2, S0), ("addi", S2, 0, 0), ("addi", T2, 0, n_hint_words),
                                 ("label", "rd"), ("addi", A7, 0, 2), ("ecall",),
                                 ("andi", T0, S2, 7), ("slli", T0, T0, 2), ("add", T0, T0, S0), ("lw", T1, T0, 0x300), ("xor", T1, T1, A0), ("sw", T1, T0, 0x300),
                                 ("addi", S2, S2, 1), ("bne", S2, T2, "rd")]
    for k in range(8):
        p += [("lw", A0, S0, 0x300 + 4 * k), ("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def pairing_hint_data(f_sextic, curve=0):
    """f_sextic: six (a, b) pairs of integers; curve 0 = Bn254 (32-byte elements), 1 = Bls12_381 (48-byte)"""
    nb = 48 if curve else 32
    out = int(curve).to_bytes(4, "little") + b"".join(int(a).to_bytes(nb, "little") + int(b).to_bytes(nb, "little") for a, b in f_sextic)
    return out + bytes(0x340 - len(out))

# ---- the native / castf sections of the reference's batch and bundle circuits (crates/circuits/batch-circuit/openvm.toml:16,24;
# bundle-circuit/openvm.toml:16,18): BabyBear arithmetic, its quartic extension, the cast of a field element to bytes ----
BABYBEAR = 2013265921
NATIVE_OPERANDS = [(5, BABYBEAR - 3), (7, 9), (BABYBEAR - 2, BABYBEAR - 5), (1000, 7),            # add (wraps), sub (negative), mul, div
                   (BABYBEAR - 1, 1), (0x80000005, 0xFFFFFFFF)]                                    # add to the top canonical word p - 1 + ... ; operands that are not reduced
NATIVE_OPS = [0, 1, 2, 3, 0, 2]
NATIVE_EXT_OPERANDS = [((1, 2, 3, 4), (BABYBEAR - 1, 5, 0, 7)), ((0, 0, 0, 0), (1, 1, 1, 1)), ((3, 1, 4, 1), (5, 9, 2, 6)), ((2, 7, 1, 8), (2, 7, 1, 8)),
                       ((2, 7, 1, 8), (2, 8, 1, BABYBEAR - 8))]   # add sub mul div (x / x) div
CASTF_VALUES = [0x3FFFFFFF, 0x00C0FFEE]


def native_program():
    """BabyBear arithmetic through the native intrinsics: six field operations (a7 = 9: buffers of 3 words), five extension operations (a7 =
    10: buffers of 12 words: add, sub, mul, x / x, x / y), two casts (a7 = 11: buffers of 2 words).  Reveals: the results of the
    first four field operations, the XOR of the last two, the extension product's and quotient's first coefficients, the two cast words
    folded into one."""
    p = rv.li(S0, 0x00400000)
    for k, op in enumerate(NATIVE_OPS):
        p += [("addi", A0, S0, 12 * k), ("addi", A1, 0, op), ("addi", A7, 0, 9), ("ecall",)]
    e0 = 12 * len(NATIVE_OPS)
    for k, op in enumerate((0, 1, 2, 3, 3)):
        p += [("addi", A0, S0, e0 + 48 * k), ("addi", A1, 0, op), ("addi", A7, 0, 10), ("ecall",)]
    c0 = e0 + 48 * 5
    for k in range(len(CASTF_VALUES)):
        p += [("addi", A0, S0, c0 + 8 * k), ("addi", A7, 0, 11), ("ecall",)]
    for k in range(4):
        p += [("lw", A0, S0, 12 * k + 8), ("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("lw", A0, S0, 12 * 4 + 8), ("lw", T0, S0, 12 * 5 + 8), ("xor", A0, A0, T0), ("addi", A1, 0, 4), ("addi", A7, 0, 1), ("ecall",)]
    p += [("lw", A0, S0, e0 + 48 * 2 + 32), ("addi", A1, 0, 5), ("addi", A7, 0, 1), ("ecall",)]
    p += [("lw", A0, S0, e0 + 48 * 3 + 32), ("addi", A1, 0, 6), ("addi", A7, 0, 1), ("ecall",)]
    p += [("lw", A0, S0, c0 + 4), ("lw", T0, S0, c0 + 12), ("xor", A0, A0, T0), ("addi", A1, 0, 7), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def native_data():
    import struct

    out = b"".join(struct.pack("<3I", b, c, 0xDEADBEEF) for b, c in NATIVE_OPERANDS)            # (the result slots hold junk: they are overwritten)
    out += b"".join(struct.pack("<12I", *x, *y, 11, 22, 33, 44) for x, y in NATIVE_EXT_OPERANDS)
    out += b"".join(struct.pack("<2I", v, 0x55AA55AA) for v in CASTF_VALUES)
    return out

# ---- the reference's BATCH circuit: BLS12-381 (crates/circuits/batch-circuit/openvm.toml:18-36) -- a base field above 2^256: operands of
# 48 bytes, limb chips of 48 limbs ----
BLS12_381_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS12_381_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
# the standard generators of G1 (y^2 = x^3 + 4) and G2 (y^2 = x^3 + 4 (1 + u) over Fp[u] / (u^2 + 1)); test_limbs48_cpu checks both equations
BLS12_381_G1 = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
                0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
BLS12_381_G2X = (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
                 0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e)
BLS12_381_G2Y = (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
                 0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)
BATCH_CIRCUIT_MODULI = (BLS12_381_P, BLS12_381_R)
BATCH_CIRCUIT_CURVES = (("Bls12_381G1Affine", BLS12_381_P, BLS12_381_R, 0, 4),)


def batch_circuit_toml(params):
    """the sections of the reference's batch-circuit openvm.toml, in the reference's syntax and order (since round 5 castf and native bring their
    chips; pairing brings the phantom hint of include/zkhip_pairing.hpp -- built for Bn254, refused for this file's Bls12_381)"""
    return ("[app_fri_params.fri_params]\nlog_blowup = %d\nlog_final_poly_len = %d\nnum_queries = %d\ncommit_proof_of_work_bits = %d\n"
            "query_proof_of_work_bits = %d\n\n[app_vm_config.rv32i]\n\n[app_vm_config.rv32m]\n\n[app_vm_config.io]\n\n[app_vm_config.keccak]\n\n"
            "[app_vm_config.castf]\n\n[app_vm_config.modular]\nsupported_moduli = [\n" % tuple(params)) + "".join('    "%d",\n' % m for m in BATCH_CIRCUIT_MODULI) + (
                "]\n[app_vm_config.native]\n[app_vm_config.pairing]\nsupported_curves = [\"Bls12_381\"]\n[app_vm_config.sha2]\n[app_vm_config.fp2]\nsupported_moduli = [\n"
                "    [\"Bls12_381Fp2\",\"%d\"]\n]\n" % BLS12_381_P) + "".join(
                    '[[app_vm_config.ecc.supported_curves]]\nstruct_name = "%s"\nmodulus = "%d"\nscalar = "%d"\na = "%d"\nb = "%d"\n' % c for c in BATCH_CIRCUIT_CURVES)


BLS_MOD, BLS_EC, BLS_FP2 = 0x00400000, 0x00400400, 0x00400800   # the three regions of bls_data


def bls_program():
    """BLS12-381 through the three limb intrinsics with 48-byte operands (modulus 0 = p; modulus 1 = r: 32-byte operands beside them).
    modular (stride 144): y^2, x^2, x^3, x^3 + 4, [y^2 = x^3 + 4] at the G1 generator; one product modulo r (stride 96).
    ecc (stride 288): 2 G, then 3 G = 2 G + G.  fp2 (stride 288): y^2, x^2, x^3, x^3 + (4 + 4 u), y^2 - that at the G2 generator.
    Reveals: 0 the equality bit, 1 / 2 the low words of 3 G, 3 / 4 the low words of y^2 and of x^3 + 4 (1 + u) in Fp2 (first component),
    5 every word of their difference OR-ed together (zero), 6 the low word of the product modulo r, 7 the low word of y^2 modulo p."""
    R_MOD, R_EC, R_FP2 = S0, S1, 18
    p = rv.li(R_MOD, BLS_MOD) + rv.li(R_EC, BLS_EC) + rv.li(R_FP2, BLS_FP2)

    def call(n, reg, off, which):
        return [("addi", A0, reg, off), ("addi", A1, 0, which), ("addi", A7, 0, n), ("ecall",)]

    def copy(reg, src, dst, n_words):
        out = []
        for k in range(n_words):
            out += [("lw", T0, reg, src + 4 * k), ("sw", T0, reg, dst + 4 * k)]
        return out
    # modular: buffers of 144 bytes (a | b | r)
    p += call(5, R_MOD, 0, 0) + call(5, R_MOD, 144, 0) + copy(R_MOD, 144 + 96, 288, 12) + call(5, R_MOD, 288, 0)
    p += copy(R_MOD, 288 + 96, 432, 12) + call(5, R_MOD, 432, 8)                                  # x^3 + 4
    p += copy(R_MOD, 96, 576, 12) + copy(R_MOD, 432 + 96, 576 + 48, 12) + call(5, R_MOD, 576, 32)   # [y^2 = x^3 + 4]
    p += call(5, R_MOD, 720, 1)                                                                   # (r - 2)(r - 3) mod r, 32-byte operands
    # ecc: buffers of 288 bytes (x1 y1 | x2 y2 | x3 y3)
    p += call(7, R_EC, 0, 8) + copy(R_EC, 192, 288, 24) + call(7, R_EC, 288, 0)
    # fp2: buffers of 288 bytes (a0 a1 | b0 b1 | r0 r1)
    p += call(8, R_FP2, 0, 0) + call(8, R_FP2, 288, 0) + copy(R_FP2, 288 + 192, 576, 24) + call(8, R_FP2, 576, 0)
    p += copy(R_FP2, 576 + 192, 864, 24) + call(8, R_FP2, 864, 8)                                 # x^3 + (4 + 4 u)
    p += copy(R_FP2, 192, 1152, 24) + copy(R_FP2, 864 + 192, 1152 + 96, 24) + call(8, R_FP2, 1152, 16)   # y^2 - (x^3 + 4 + 4 u)
    p += [("addi", T1, 0, 0)]
    for k in range(24):
        p += [("lw", T2, R_FP2, 1152 + 192 + 4 * k), ("or", T1, T1, T2)]
    for k, (reg, off) in enumerate(((R_MOD, 576 + 96), (R_EC, 288 + 192), (R_EC, 288 + 240), (R_FP2, 192), (R_FP2, 864 + 192), (None, 0), (R_MOD, 720 + 64), (R_MOD, 96))):
        p += [("add", A0, T1, 0)] if reg is None else [("lw", A0, reg, off)]
        p += [("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
    p += [("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def bls_data():
    b48, b32 = (lambda v: int(v).to_bytes(48, "little")), (lambda v: int(v).to_bytes(32, "little"))  # noqa: E731
    gx, gy = BLS12_381_G1
    z48 = bytes(48)
    mod = (b48(gy) + b48(gy) + z48) + (b48(gx) + b48(gx) + z48) + (z48 + b48(gx) + z48) + (z48 + b48(4) + z48) + (z48 * 3)
    mod += b32(BLS12_381_R - 2) + b32(BLS12_381_R - 3) + bytes(32)
    mod += bytes(BLS_EC - BLS_MOD - len(mod))
    g = b48(gx) + b48(gy)
    ec = (g + g + bytes(96)) + (bytes(96) + g + bytes(96))
    ec += bytes(BLS_FP2 - BLS_EC - len(ec))
    el = lambda e: b48(e[0]) + b48(e[1])  # noqa: E731
    z96 = bytes(96)
    f = (el(BLS12_381_G2Y) + el(BLS12_381_G2Y) + z96) + (el(BLS12_381_G2X) + el(BLS12_381_G2X) + z96) + (z96 + el(BLS12_381_G2X) + z96) + (z96 + el((4, 4)) + z96) + z96 * 3
    return mod + ec + f


# ---- phantom instructions (OpenVM's algebra extension: ModularPhantom::{HintNonQr, HintSqrt}): advice the circuit does not see ----
PHANTOM_MODULI = (SECP256K1_P, SECP256K1_N, BLS12_381_P)   # (n = 1 mod 4: the square root takes Tonelli - Shanks' loop)


def phantom_cases():
    """(modulus index, x): a square modulo p, a value modulo n (square or not: Python decides), the smallest non-residue modulo BLS12-381's p"""
    z = next(c for c in range(2, 100) if pow(c, (BLS12_381_P - 1) // 2, BLS12_381_P) == BLS12_381_P - 1)
    return ((0, SECP256K1_GY * SECP256K1_GY % SECP256K1_P), (1, 0xDEADBEEF12345), (2, z))


def phantom_layout():
    """per case a region of 1024 bytes: H = [index | x] at 0, B0 = (r | r | .) at 256, B1 = (x | z | .) at 448, B2 = (r^2 | x | .) at 640,
    B3 = (r^2 | x z | .) at 832 -- the three-operand buffers of the modular intrinsic (3 nb bytes, nb = 32 or 48)"""
    return dict(H=0, B0=256, B1=448, B2=640, B3=832, stride=1024)


def phantom_program():
    """For every case: ask for the square root (phantom 1) and for the non-residue (phantom 0), read both from the hint stream, then CHECK
    them with the modular intrinsic: r^2 (mul), x z (mul), [r^2 = x], [r^2 = x z] (is_eq).  Reveals per case s | eq1 << 1 | eq2 << 2
    (a square: 3, a non-square: 4), and the low word of the first root."""
    L = phantom_layout()
    p = []
    for k, (mi, _) in enumerate(phantom_cases()):
        nw = 8 if PHANTOM_MODULI[mi] < 1 << 256 else 12
        nb = 4 * nw
        p += rv.li(S0, 0x00400000 + k * L["stride"])
        p += [("phantom", 1, S0), ("addi", A7, 0, 2), ("ecall",), ("add", S1, A0, 0)]                 # s
        for j in range(nw):                                                                             # r into B0's two operands
            p += [("addi", A7, 0, 2), ("ecall",), ("sw", A0, S0, L["B0"] + 4 * j), ("sw", A0, S0, L["B0"] + nb + 4 * j)]
        p += [("phantom", 0, S0)]
        for j in range(nw):                                                                             # z into B1's second operand
            p += [("addi", A7, 0, 2), ("ecall",), ("sw", A0, S0, L["B1"] + nb + 4 * j)]
        for j in range(nw):                                                                             # x into B1's first, B2's second operand
            p += [("lw", T0, S0, L["H"] + 4 + 4 * j), ("sw", T0, S0, L["B1"] + 4 * j), ("sw", T0, S0, L["B2"] + nb + 4 * j)]
        call = lambda off, op: [("addi", A0, S0, off), ("addi", A1, 0, mi + 8 * op), ("addi", A7, 0, 5), ("ecall",)]  # noqa: E731
        p += call(L["B0"], 0) + call(L["B1"], 0)
        for j in range(nw):                                                                             # r^2 into B2 / B3, x z into B3
            p += [("lw", T0, S0, L["B0"] + 2 * nb + 4 * j), ("sw", T0, S0, L["B2"] + 4 * j), ("sw", T0, S0, L["B3"] + 4 * j),
                  ("lw", T0, S0, L["B1"] + 2 * nb + 4 * j), ("sw", T0, S0, L["B3"] + nb + 4 * j)]
        p += call(L["B2"], 4) + call(L["B3"], 4)
        p += [("lw", T0, S0, L["B2"] + 2 * nb), ("slli", T0, T0, 1), ("lw", T1, S0, L["B3"] + 2 * nb), ("slli", T1, T1, 2), ("or", T0, T0, T1), ("or", A0, T0, S1),
              ("addi", A1, 0, k), ("addi", A7, 0, 1), ("ecall",)]
        if k == 0:
            p += [("lw", A0, S0, L["B0"]), ("addi", A1, 0, 7), ("addi", A7, 0, 1), ("ecall",)]
    p += [("fence",), ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def phantom_data():
    L = phantom_layout()
    out = b""
    for mi, x in phantom_cases():
        nb = 32 if PHANTOM_MODULI[mi] < 1 << 256 else 48
        region = bytearray(L["stride"])
        region[0:4] = mi.to_bytes(4, "little")
        region[4:4 + nb] = x.to_bytes(nb, "little")
        out += bytes(region)
    return out


DEFERRAL_DATA_BYTES = 64   # the batch-like guest's data segment: the child app's exe commitment (8 words), vm commitment (8 words)


def deferral_guest_program(deferral_base=0x00402000):
    """A guest that DEFERS the verification of its children (the reference's batch guest: crates/types/circuit/src/lib.rs:137-154
    `verify_stark::<0>(input_commit, &expected)` once per aggregated proof).  Input stream (ProvingTask::build_guest_input): one witness
    item = the children's 32 public-value bytes each, then the input commitments [count | 8 words each].  For child k it writes the claim
    [input commitment | exe commitment | vm commitment | public values] -- the program commitments are CONSTANTS of this guest (its data
    segment, as the reference compiles crates/circuits/*-circuit/*_commit.rs into the parent) -- into the deferral region at word
    32 + 32 k, then the number of claims at word 0.  Reveals word 0 = the number of children, word 1 = the sum of their first public words."""
    p = rv.li(S0, deferral_base) + rv.li(S1, 0x00400000)
    p += [("addi", A7, 0, 2), ("ecall",), ("srli", 18, A0, 5),           # s2 (x18) = byte length of the witness / 32 = n
          ("addi", T0, S0, 128), ("addi", T1, 0, 0), ("addi", 19, 0, 0),  # t0 = claim pointer, t1 = k, s3 (x19) = the running sum
          ("label", "pvs"), ("bge", T1, 18, "pvs_done")]
    for j in range(8):
        p += [("addi", A7, 0, 2), ("ecall",), ("sw", A0, T0, 96 + 4 * j)]
        if j == 0:
            p += [("add", 19, 19, A0)]
    for j in range(16):
        p += [("lw", T3, S1, 4 * j), ("sw", T3, T0, 32 + 4 * j)]
    p += [("addi", T0, T0, 128), ("addi", T1, T1, 1), ("jal", 0, "pvs"), ("label", "pvs_done"),
          ("addi", A7, 0, 2), ("ecall",),                                  # the count of input commitments (= n)
          ("addi", T0, S0, 128), ("addi", T1, 0, 0),
          ("label", "ics"), ("bge", T1, 18, "ics_done")]
    for j in range(8):
        p += [("addi", A7, 0, 2), ("ecall",), ("sw", A0, T0, 4 * j)]
    p += [("addi", T0, T0, 128), ("addi", T1, T1, 1), ("jal", 0, "ics"), ("label", "ics_done"),
          ("sw", 18, S0, 0),
          ("add", A0, 18, 0), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
          ("add", A0, 19, 0), ("addi", A1, 0, 1), ("ecall",),
          ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def deferral_guest_stdin(child_pvs, input_commits):
    """ProvingTask::build_guest_input for one witness item (the children's public values) + the input commitments (8 words each)."""
    w = b"".join(bytes(x) for x in child_pvs)
    out = len(w).to_bytes(4, "little") + w + bytes(-len(w) % 4)
    out += len(input_commits).to_bytes(4, "little")
    for c in input_commits:
        out += b"".join(int(x).to_bytes(4, "little") for x in c)
    return out


def mixed_program():
    """every instruction class: a checksum over a table in memory with multiplies, comparisons, shifts, division, bytes"""
    p = [("addi", A7, 0, 2), ("ecall",), ("add", S0, A0, 0)]          # seed
    p += rv.li(S1, 0x00400000)                                           # table base
    p += rv.li(T4, 0x9E3779B9)
    p += [("addi", T2, 0, 0),
          ("label", "fill"), ("slti", T3, T2, 64), ("beq", T3, 0, "filled"),
          ("mul", S0, S0, T4), ("xori", S0, S0, 0x5A5), ("slli", T3, T2, 2), ("add", T3, T3, S1), ("sw", S0, T3, 0),
          ("addi", T2, T2, 1), ("jal", 0, "fill"),
          ("label", "filled"), ("addi", T2, 0, 0), ("addi", T0, 0, 0),
          ("label", "sum"), ("sltiu", T3, T2, 64), ("beq", T3, 0, "summed"),
          ("slli", T3, T2, 2), ("add", T3, T3, S1), ("lw", T1, T3, 0), ("lbu", T3, T3, 1),
          ("sltu", 12, T1, T0), ("slt", 13, T1, T0), ("sub", T0, T0, 12), ("add", T0, T0, 13),
          ("and", 14, T1, T4), ("or", 15, T1, T3), ("xor", T0, T0, 14), ("add", T0, T0, 15),
          ("srli", 14, T1, 7), ("sra", 15, T1, T3), ("add", T0, T0, 14), ("sub", T0, T0, 15),
          ("ori", 16, T3, 1), ("divu", 14, T1, 16), ("rem", 15, T1, 16), ("div", 19, T1, 13), ("remu", 20, T1, 12), ("xor", 14, 14, 19), ("add", 15, 15, 20), ("mulhu", 16, T1, T4), ("mulh", 18, T1, T4), ("mulhsu", 19, T1, T4), ("xor", 16, 16, 19),
          ("add", T0, T0, 14), ("add", T0, T0, 15), ("xor", T0, T0, 16), ("add", T0, T0, 18),
          ("sb", T0, S1, 3), ("lh", 14, S1, 2), ("add", T0, T0, 14),
          ("sh", T0, S1, 6), ("lhu", 15, S1, 6), ("lb", 16, S1, 7), ("add", T0, T0, 15), ("xor", T0, T0, 16),
          ("blt", T1, T0, "s1"), ("addi", T0, T0, 3), ("label", "s1"), ("bgeu", T1, T4, "s2"), ("xori", T0, T0, 0x11), ("label", "s2"),
          ("bltu", T3, T1, "s3"), ("addi", T0, T0, 1), ("label", "s3"), ("bge", 14, T3, "s4"), ("sub", T0, T0, T3), ("label", "s4"),
          ("bne", 14, T3, "s5"), ("addi", T0, T0, 5), ("label", "s5"),
          ("auipc", 17, 0), ("jalr", 18, 17, 13), ("addi", T0, T0, 7), ("add", T0, T0, 18),   # jalr lands on the add (12 + 1, low bit cleared)
          ("auipc", 17, 1), ("addi", 17, 17, -2038), ("addi", 17, 17, -2038), ("jalr", 0, 17, -4),   # 4096 - 4076 - 4 = +16 from the auipc: the next line
          ("xori", T0, T0, 0x2A),
          ("addi", T2, T2, 1), ("jal", 0, "sum"),
          ("label", "summed"),
          ("add", A0, T0, 0), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
          ("andi", A0, T0, 0x7F), ("addi", A1, 0, 7), ("ecall",),
          ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def run_cli(tmp_path, program, stdin=b"", max_cost=0, data=b""):
    exe = tmp_path / "exe.bin"
    exe.write_bytes(rv.exe_bytes(program, data=data))
    inp = tmp_path / "stdin.bin"
    inp.write_bytes(stdin)
    out = tmp_path / "rec"
    out.mkdir(exist_ok=True)
    r = subprocess.run([VM, "exec", str(exe), str(inp), str(out)] + ([str(max_cost)] if max_cost else []), capture_output=True, text=True)
    rec = {}
    if r.returncode == 0:
        for name in ("pc_index", "alu_op", "alu_b", "alu_c", "lt_op", "lt_b", "lt_c", "mul_b", "mul_c", "shift_op", "shift_b", "shift_c", "beq_op", "beq_a", "beq_b", "beq_imm", "blt_op", "blt_a", "blt_b", "blt_imm", "mulh_op", "mulh_b", "mulh_c", "div_op", "div_b", "div_c", "ls_case", "ls_read", "ls_prev", "jal_op", "jal_pc", "jal_imm", "auipc_pc", "auipc_imm", "jalr_pc", "jalr_rs1", "jalr_imm", "acc_as", "acc_ptr", "acc_prev_data",
                     "acc_prev_ts", "acc_data", "acc_ts", "acc_is_read", "bnd_as", "bnd_ptr", "bnd_initial", "bnd_final", "bnd_final_ts"):
            rec[name] = np.fromfile(out / (name + ".u32"), dtype=np.uint32)
    return r, (json.loads(r.stdout) if r.returncode == 0 else None), rec


def check_memory_log(acc, bnd):
    """The offline memory-checking property itself: replaying the log from the boundary's initial values, every access consumes
    exactly the state the previous access of its cell left, and the last state of every cell is the boundary's final one."""
    state = {(a, p): (init, 0) for a, p, init, fin, ts in bnd}
    for k, (a, p, pd, pts, d, ts, is_read) in enumerate(acc):
        assert ts == k + 1 and state[(a, p)] == (pd, pts) and pts < ts and d < 65536
        if is_read:
            assert d == pd
        state[(a, p)] = (d, ts)
    assert all(state[(a, p)] == (fin, ts) for a, p, init, fin, ts in bnd)


def check_against_model(js, rec, model):
    m = model["records"]
    assert js["total_cycle"] == model["instret"] == len(m["pc_index"])
    assert bytes(js["public_values"]) == model["pvs"]
    assert rec["pc_index"].tolist() == m["pc_index"]
    assert list(zip(rec["alu_op"].tolist(), rec["alu_b"].tolist(), rec["alu_c"].tolist())) == m["alu"]
    assert list(zip(rec["lt_op"].tolist(), rec["lt_b"].tolist(), rec["lt_c"].tolist())) == m["lt"]
    assert list(zip(rec["mul_b"].tolist(), rec["mul_c"].tolist())) == m["mul"]
    assert list(zip(rec["shift_op"].tolist(), rec["shift_b"].tolist(), rec["shift_c"].tolist())) == m["shifts"]
    assert list(zip(rec["beq_op"].tolist(), rec["beq_a"].tolist(), rec["beq_b"].tolist(), rec["beq_imm"].tolist())) == m["beq"]
    assert list(zip(rec["blt_op"].tolist(), rec["blt_a"].tolist(), rec["blt_b"].tolist(), rec["blt_imm"].tolist())) == m["blt"]
    assert list(zip(rec["mulh_op"].tolist(), rec["mulh_b"].tolist(), rec["mulh_c"].tolist())) == m["mulhs"]
    assert list(zip(rec["div_op"].tolist(), rec["div_b"].tolist(), rec["div_c"].tolist())) == m["divs"]
    assert list(zip(rec["ls_case"].tolist(), rec["ls_read"].tolist(), rec["ls_prev"].tolist())) == m["ls"]
    assert list(zip(rec["jal_op"].tolist(), rec["jal_pc"].tolist(), rec["jal_imm"].tolist())) == m["jal"]
    assert list(zip(rec["auipc_pc"].tolist(), rec["auipc_imm"].tolist())) == m["auipc"]
    assert list(zip(rec["jalr_pc"].tolist(), rec["jalr_rs1"].tolist(), rec["jalr_imm"].tolist())) == m["jalr"]
    for k in ("shift", "branch", "jump", "load_store", "mulh", "divrem", "lui_auipc", "ecall"):
        assert js["records"][k] == m[k], k
    assert js["records"]["executed"] == model["instret"]
    # the memory log and the boundary records
    acc = list(zip(*(rec["acc_" + k].tolist() for k in ("as", "ptr", "prev_data", "prev_ts", "data", "ts", "is_read"))))
    assert acc == m["acc"] and js["records"]["memory_accesses"] == len(acc)
    bnd = list(zip(*(rec["bnd_" + k].tolist() for k in ("as", "ptr", "initial", "final", "final_ts"))))
    assert bnd == m["bnd"] and js["records"]["cells_touched"] == len(bnd)
    check_memory_log(acc, bnd)


@pytest.mark.parametrize("n", [0, 1, 10, 47, 1000])
def test_fibonacci_guest(tmp_path, n):
    prog = fib_program()
    stdin = int(n).to_bytes(4, "little")
    if n == 0:
        # fib(0) = 0 and n = 0: every public value is zero -> the reference's sanity check refuses the run
        r, js, _ = run_cli(tmp_path, prog, stdin)
        assert r.returncode != 0 and "public_values are all 0s" in r.stderr
        return
    r, js, rec = run_cli(tmp_path, prog, stdin)
    assert r.returncode == 0, r.stderr
    model = rv.run(prog, stdin)
    check_against_model(js, rec, model)
    a, b = 0, 1
    for _ in range(n):
        a, b = b, (a + b) & 0xFFFFFFFF
    assert int.from_bytes(bytes(js["public_values"][:4]), "little") == a and js["public_values"][4] == n & 255


@pytest.mark.parametrize("seed", [1, 0xDEADBEEF, 12345])
def test_mixed_guest_every_instruction_class(tmp_path, seed):
    prog = mixed_program()
    stdin = int(seed).to_bytes(4, "little")
    r, js, rec = run_cli(tmp_path, prog, stdin)
    assert r.returncode == 0, r.stderr
    model = rv.run(prog, stdin)
    check_against_model(js, rec, model)
    assert js["records"]["mul"] == 64 and js["records"]["divrem"] == 256 and js["records"]["mulh"] == 192 and js["records"]["load_store"] > 300


def test_metered_fallback_and_guest_failure(tmp_path):
    prog = fib_program()
    stdin = (5000).to_bytes(4, "little")
    r, js, rec = run_cli(tmp_path, prog, stdin, max_cost=1000)   # the metered run exceeds its cost bound
    assert r.returncode == 0 and js["total_cycle"] == 2**64 - 1  # the plain executor reports no cycle count (vm.rs:41-46)
    assert len(rec["pc_index"]) == rv.run(prog, stdin)["instret"]
    # non-zero exit code
    bad = rv.assemble([("addi", A0, 0, 3), ("addi", A7, 0, 93), ("ecall",)])
    r, _, _ = run_cli(tmp_path, bad)
    assert r.returncode != 0 and "exited with code 3" in r.stderr
    # a jump out of the program
    r, _, _ = run_cli(tmp_path, rv.assemble([("jal", 0, 64)]))
    assert r.returncode != 0 and "pc outside the program" in r.stderr


def test_elf_guest(tmp_path):
    """The same guest as an ELF image (what the reference's `exe` is before transpilation): text and data segments, bss, an entry
    point that is not the first instruction."""
    table = bytes(range(1, 33))
    prog = [("jal", 0, "start"),                                  # never executed: the entry point skips it
            ("label", "start")] + rv.li(S1, 0x00400000) + [
        ("addi", T2, 0, 0), ("addi", T0, 0, 0),
        ("label", "loop"), ("slti", T3, T2, 32), ("beq", T3, 0, "done"),
        ("add", T3, S1, T2), ("lbu", T1, T3, 0), ("add", T0, T0, T1), ("sb", T0, T3, 64),   # running sums into the bss
        ("addi", T2, T2, 1), ("jal", 0, "loop"),
        ("label", "done"), ("lw", A0, S1, 92), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
        ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    words = rv.assemble(prog)
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(words, data=table, bss=64, entry_offset=4))
    out = tmp_path / "rec"
    out.mkdir()
    r = subprocess.run([VM, "exec", str(exe), "-", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    js = json.loads(r.stdout)
    model = rv.run(words[1:], pc_base=0x00200004, data=table + bytes(64))
    assert js["total_cycle"] == model["instret"] and bytes(js["public_values"]) == model["pvs"]
    sums = np.cumsum(np.frombuffer(table, np.uint8).astype(np.int64)) & 255
    assert bytes(js["public_values"][:4]) == bytes(int(v) for v in sums[28:32])
    # not RISC-V / truncated
    bad = bytearray(exe.read_bytes())
    bad[18] = 3
    (tmp_path / "x86.elf").write_bytes(bad)
    r = subprocess.run([VM, "exec", str(tmp_path / "x86.elf"), "-", "-"], capture_output=True, text=True)
    assert r.returncode != 0 and "RISC-V" in r.stderr
    (tmp_path / "cut.elf").write_bytes(exe.read_bytes()[:100])
    r = subprocess.run([VM, "exec", str(tmp_path / "cut.elf"), "-", "-"], capture_output=True, text=True)
    assert r.returncode != 0


REC_NAMES = ("pc_index", "alu_op", "alu_b", "alu_c", "lt_op", "lt_b", "lt_c", "mul_b", "mul_c", "shift_op", "shift_b", "shift_c", "beq_op", "beq_a", "beq_b", "beq_imm", "blt_op", "blt_a", "blt_b", "blt_imm", "mulh_op", "mulh_b", "mulh_c", "div_op", "div_b", "div_c", "ls_case", "ls_read", "ls_prev", "jal_op", "jal_pc", "jal_imm", "auipc_pc", "auipc_imm", "jalr_pc", "jalr_rs1", "jalr_imm", "acc_as", "acc_ptr", "acc_prev_data",
             "acc_prev_ts", "acc_data", "acc_ts", "acc_is_read", "bnd_as", "bnd_ptr", "bnd_initial", "bnd_final", "bnd_final_ts")


def run_segments(tmp_path, program, stdin, segment_instr, max_segments=64):
    exe = tmp_path / "exe.bin"
    exe.write_bytes(rv.exe_bytes(program))
    inp = tmp_path / "stdin.bin"
    inp.write_bytes(stdin)
    out = tmp_path / "segs"
    out.mkdir(exist_ok=True)
    for k in range(max_segments):
        (out / ("seg-%d" % k)).mkdir(exist_ok=True)
    r = subprocess.run([VM, "exec-segments", str(exe), str(inp), str(out), str(segment_instr)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    js = json.loads(r.stdout)
    segs = [{n: np.fromfile(out / ("seg-%d" % k) / (n + ".u32"), dtype=np.uint32) for n in REC_NAMES} for k in range(len(js["segments"]))]
    return js, segs


@pytest.mark.parametrize("segment_instr", [100, 1000, 10**6])
def test_continuation_segments(tmp_path, segment_instr):
    """The run cut into segments: the segments' instruction streams concatenate to the whole run's, every segment's memory log is a
    consistent history from ITS boundary's initial values, and a cell's final value in one segment is its initial value in the next
    segment that touches it."""
    prog = mixed_program()
    stdin = (4242).to_bytes(4, "little")
    js, segs = run_segments(tmp_path, prog, stdin, segment_instr)
    model = rv.run(prog, stdin)
    assert js["total_cycle"] == model["instret"] == sum(js["segments"]) and bytes(js["public_values"]) == model["pvs"]
    assert all(n == segment_instr for n in js["segments"][:-1]) and 0 < js["segments"][-1] <= segment_instr
    assert np.concatenate([s["pc_index"] for s in segs]).tolist() == model["records"]["pc_index"]
    assert sum(len(s["alu_op"]) for s in segs) == len(model["records"]["alu"])
    state = {}
    for s in segs:
        acc = list(zip(*(s["acc_" + k].tolist() for k in ("as", "ptr", "prev_data", "prev_ts", "data", "ts", "is_read"))))
        bnd = list(zip(*(s["bnd_" + k].tolist() for k in ("as", "ptr", "initial", "final", "final_ts"))))
        check_memory_log(acc, bnd)
        for a, p, init, fin, ts in bnd:
            if (a, p) in state:
                assert state[(a, p)] == init      # the link between consecutive segment proofs
            state[(a, p)] = fin
    # the last values agree with the un-segmented run's boundary
    whole = {(a, p): fin for a, p, init, fin, ts in model["records"]["bnd"]}
    assert state == whole


def test_interpreter_survives_arbitrary_programs(tmp_path):
    """Random instruction words, random mutations of a real program, truncated / corrupt images: under AddressSanitizer + UBSan the
    interpreter either finishes or reports an error -- it never crashes, reads out of bounds or hangs (instruction limit)."""
    exe_san = str(tmp_path / "vm_cli_san")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tools", "vm_cli.cpp"), "-o", exe_san], check=True)
    rng = np.random.default_rng(2024)
    base = mixed_program()
    env = dict(os.environ, ZKHIP_VM_MAX_INSTR="20000", ASAN_OPTIONS="detect_leaks=0")
    outcomes = {0: 0, 1: 0}
    for case in range(250):
        kind = case % 5
        if kind == 0:
            words = [int(x) for x in rng.integers(0, 1 << 32, size=int(rng.integers(1, 40)), dtype=np.uint64)]
        elif kind == 1:   # valid opcodes, random fields
            ops = [0x37, 0x17, 0x6F, 0x67, 0x63, 0x03, 0x23, 0x13, 0x33, 0x73]
            words = [(int(rng.integers(0, 1 << 25)) << 7) | int(rng.choice(ops)) for _ in range(int(rng.integers(1, 60)))]
        else:             # a real program with a few words / bits changed
            words = list(base)
            for _ in range(int(rng.integers(1, 6))):
                k = int(rng.integers(0, len(words)))
                words[k] = (words[k] ^ (1 << int(rng.integers(0, 32)))) if kind < 4 else int(rng.integers(0, 1 << 32, dtype=np.uint64))
        raw = rv.exe_bytes(words) if case % 7 else rv.elf_bytes(words)
        if case % 11 == 0:
            raw = raw[:int(rng.integers(1, len(raw)))]           # truncated image
        elif case % 13 == 0:
            raw = bytearray(raw)
            raw[int(rng.integers(0, min(60, len(raw))))] ^= 0xFF   # corrupt header
        f = tmp_path / "fuzz.bin"
        f.write_bytes(bytes(raw))
        inp = tmp_path / "in.bin"
        inp.write_bytes(rng.integers(0, 256, size=64, dtype=np.uint8).tobytes())
        r = subprocess.run([exe_san, "exec", str(f), str(inp), "-"], capture_output=True, text=True, env=env, timeout=60)
        assert r.returncode in (0, 1), (case, r.returncode, r.stderr[-2000:])
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (case, r.stderr[-2000:])
        outcomes[r.returncode] += 1
    assert outcomes[1] > 50   # most mutants fail cleanly; some still finish


def test_build_guest_input_stream(tmp_path):
    """ProvingTask::build_guest_input (crates/prover/src/task/mod.rs:13-38) in the C++ mirror: witnesses as length-framed items,
    then the input commitments; a guest that walks the whole stream (sum of all words, number of items) sees what the Python
    model sees on the same bytes."""
    exe = str(tmp_path / "guest_input_cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "guest_input_cpp.cpp"), "-o", exe,
                    "-L", os.path.join(ROOT, "zkvm-prover_amd"), "-lzkhip", "-Wl,-rpath," + os.path.join(ROOT, "zkvm-prover_amd"), "-pthread"], check=True)
    # guest: read 3 items [len][words...], then the commitments [count][8 words each]; checksum += every word read
    p = [("addi", S0, 0, 0), ("addi", S1, 0, 0),                       # s0 = checksum, s1 = items seen
         ("label", "item"), ("slti", T3, S1, 3), ("beq", T3, 0, "commits"),
         ("addi", A7, 0, 2), ("ecall",), ("add", S0, S0, A0),          # length
         ("addi", T0, A0, 3), ("srli", T0, T0, 2),                     # words to read
         ("label", "w"), ("beq", T0, 0, "next"), ("ecall",), ("add", S0, S0, A0), ("addi", T0, T0, -1), ("jal", 0, "w"),
         ("label", "next"), ("addi", S1, S1, 1), ("jal", 0, "item"),
         ("label", "commits"), ("addi", A7, 0, 2), ("ecall",), ("add", S0, S0, A0), ("slli", T0, A0, 3),
         ("label", "cw"), ("beq", T0, 0, "out"), ("ecall",), ("add", S0, S0, A0), ("addi", T0, T0, -1), ("jal", 0, "cw"),
         ("label", "out"), ("add", A0, S0, 0), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
         ("addi", A0, S1, 0), ("addi", A1, 0, 1), ("ecall",),
         ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    words = rv.assemble(p)
    guest = tmp_path / "guest.bin"
    guest.write_bytes(rv.exe_bytes(words))
    r = subprocess.run([exe, str(guest)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    stream_hex, pv_hex, cycles = r.stdout.split()
    stream = bytes.fromhex(stream_hex)
    w1, w3 = bytes([1, 2, 3, 4, 5]), bytes([0xFF, 0xFE, 0xFD, 0xFC, 0xFB, 0xFA, 0xF9, 0xF8])
    commit = bytes((3 * i + 1) & 255 for i in range(32))
    expect = (5).to_bytes(4, "little") + w1 + bytes(3) + (0).to_bytes(4, "little") + (8).to_bytes(4, "little") + w3 + (2).to_bytes(4, "little") + commit * 2
    assert stream == expect
    model = rv.run(words, stream)
    assert bytes.fromhex(pv_hex) == model["pvs"] and int(cycles) == model["instret"]
    total = sum(int.from_bytes(stream[i:i + 4], "little") for i in range(0, len(stream), 4)) & 0xFFFFFFFF
    assert int.from_bytes(model["pvs"][:4], "little") == total and model["pvs"][4] == 3


def test_text_segment_must_lie_below_two_to_the_30(tmp_path):
    """The chips hold a pc as one BabyBear element (and compose pc + 4 from 30 bits of limbs, OpenVM's PC_BITS): an image linked
    at the usual bare-metal 0x80000000 is refused by the loader with a clear message instead of by a trace generator later."""
    exe = tmp_path / "high.elf"
    exe.write_bytes(rv.elf_bytes(fib_program(), text_vaddr=0x80000000))
    inp = tmp_path / "stdin.bin"
    inp.write_bytes((5).to_bytes(4, "little"))
    r = subprocess.run([VM, "exec", str(exe), str(inp), "-"], capture_output=True, text=True)
    assert r.returncode != 0 and "below 2^30" in r.stderr


def test_deferral_guest_writes_its_claims(tmp_path):
    """The batch-like guest on the executor: public values (number of children, sum of their first public words) == the Python model's."""
    prog = deferral_guest_program()
    rng = np.random.default_rng(3)
    pvs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(3)]
    ics = [rng.integers(0, 2013265921, 8, dtype=np.uint32) for _ in range(3)]
    data = b"".join(int(x).to_bytes(4, "little") for x in range(100, 116))
    stdin = deferral_guest_stdin(pvs, ics)
    exe = tmp_path / "guest.elf"
    exe.write_bytes(rv.elf_bytes(prog, data=data))
    (tmp_path / "stdin.bin").write_bytes(stdin)
    r = subprocess.run([VM, "exec", str(exe), str(tmp_path / "stdin.bin"), "-"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    js = json.loads(r.stdout)
    model = rv.run(prog, stdin, data=data)
    assert js["total_cycle"] == model["instret"] and bytes(js["public_values"]) == model["pvs"]
    total = sum(int.from_bytes(p[:4], "little") for p in pvs) & 0xFFFFFFFF
    assert int.from_bytes(model["pvs"][:4], "little") == 3 and int.from_bytes(model["pvs"][4:8], "little") == total
