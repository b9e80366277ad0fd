#!/usr/bin/env python3
"""Known-answer vectors for the 256-bit ALU -> tests/golden/int256_kat.json.  Source of truth OUTSIDE this repository: Python's
arbitrary-precision integers (a = (b op c) mod 2^256 for op = add, sub, xor, or, and, mul): random operands and the edge cases of the carry
chain (all ones, wrap-around, borrow through every limb)."""
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
M = 1 << 256
OPS = {0: lambda b, c: (b + c) % M, 1: lambda b, c: (b - c) % M, 2: lambda b, c: b ^ c, 3: lambda b, c: b | c, 4: lambda b, c: b & c,
       5: lambda b, c: (b * c) % M}


def main():
    rnd = random.Random(20260404)
    pairs = [(0, 0), (M - 1, 1), (0, 1), (M - 1, M - 1), (1 << 255, 1 << 255), (0xFF, 0x01), ((1 << 128) - 1, 1), (1 << 128, 1), (M - 1, 0)]
    pairs += [(rnd.getrandbits(256), rnd.getrandbits(256)) for _ in range(23)]
    cases = [{"op": op, "b": hex(b), "c": hex(c), "a": hex(f(b, c))} for op, f in OPS.items() for b, c in pairs]
    # comparisons (a generator of their own: the cases above keep their values): 6 b < c unsigned, 7 b < c signed (two's complement), 8 b == c
    rnd_cmp = random.Random(20261003)
    sgn = lambda v: v - M if v >> 255 else v  # noqa: E731
    cpairs = [(0, 0), (0, 1), (1, 0), (M - 1, 0), (0, M - 1), (M - 1, M - 1), (1 << 255, (1 << 255) - 1), ((1 << 255) - 1, 1 << 255), (1 << 255, 1 << 255),
              (1 << 255, M - 1), (5 << 248, 5 << 248 | 1), (0x80 << 248, 0x7F << 248), (0xFF << 248, 0x80 << 248)]
    for _ in range(12):
        b = rnd_cmp.getrandbits(256)
        cpairs += [(b, rnd_cmp.getrandbits(256)), (b, b ^ (1 << rnd_cmp.randrange(256))), (b, b)]
    cmp_cases = [{"op": op, "b": hex(b), "c": hex(c), "a": hex(int(b < c) if op == 6 else int(sgn(b) < sgn(c)) if op == 7 else int(b == c))}
                 for op in (6, 7, 8) for b, c in cpairs]
    # shifts by c mod 256 (a generator of their own): 9 b << s, 10 b >> s, 11 arithmetic
    rnd_sh = random.Random(20261004)
    spairs = [(M - 1, 0), (M - 1, 1), (M - 1, 255), (1, 255), (1 << 255, 255), (1 << 255, 8), (1 << 255, 7), (0x80 << 248 | 0x55, 9), (0x7F << 248 | 0xAA, 9),
              (M - 1, 256 + 4), (12345, (7 << 200) | 17), (0, 200)]
    for _ in range(10):
        b = rnd_sh.getrandbits(256)
        spairs += [(b, rnd_sh.randrange(256)), (b | 1 << 255, rnd_sh.randrange(256)), (b, 8 * rnd_sh.randrange(32))]
    sh = {9: lambda b, s: (b << s) % M, 10: lambda b, s: b >> s, 11: lambda b, s: (sgn(b) >> s) % M}
    shift_cases = [{"op": op, "b": hex(b), "c": hex(c), "a": hex(sh[op](b, c % 256))} for op in (9, 10, 11) for b, c in spairs]
    with open(os.path.join(HERE, "int256_kat.json"), "w") as f:
        json.dump({"about": "a = (b op c) mod 2^256 from Python integers (generator: tests/golden/gen_int256_kat.py); op: 0 add 1 sub 2 xor 3 or 4 and 5 mul (low 256 bits)",
                   "cases": cases, "cmp": cmp_cases, "shift": shift_cases}, f, indent=0)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
