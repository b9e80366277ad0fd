"""A tiny RV32IM assembler and an independent Python interpreter (test infrastructure): the checker of include/zkhip_vm.hpp.
Follows the RISC-V unprivileged specification (RV32I base + M); environment calls as documented in zkhip_vm.hpp
(a7 = 93 exit, 1 reveal, 2 read input word)."""
import struct

M32 = 0xFFFFFFFF


def _r(op, f3, f7, rd, rs1, rs2):
    return (f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op


def _i(op, f3, rd, rs1, imm):
    return ((imm & 0xFFF) << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op


def _s(f3, rs1, rs2, imm):
    return (((imm >> 5) & 0x7F) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | ((imm & 31) << 7) | 0x23


def _b(f3, rs1, rs2, imm):
    return (((imm >> 12) & 1) << 31) | (((imm >> 5) & 63) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (((imm >> 1) & 15) << 8) | (((imm >> 11) & 1) << 7) | 0x63


def _j(rd, imm):
    return (((imm >> 20) & 1) << 31) | (((imm >> 1) & 1023) << 21) | (((imm >> 11) & 1) << 20) | (((imm >> 12) & 255) << 12) | (rd << 7) | 0x6F


R_OPS = {"add": (0, 0), "sub": (0, 0x20), "sll": (1, 0), "slt": (2, 0), "sltu": (3, 0), "xor": (4, 0), "srl": (5, 0), "sra": (5, 0x20),
         "or": (6, 0), "and": (7, 0), "mul": (0, 1), "mulh": (1, 1), "mulhsu": (2, 1), "mulhu": (3, 1), "div": (4, 1), "divu": (5, 1),
         "rem": (6, 1), "remu": (7, 1)}
I_OPS = {"addi": 0, "slti": 2, "sltiu": 3, "xori": 4, "ori": 6, "andi": 7}
B_OPS = {"beq": 0, "bne": 1, "blt": 4, "bge": 5, "bltu": 6, "bgeu": 7}
L_OPS = {"lb": 0, "lh": 1, "lw": 2, "lbu": 4, "lhu": 5}
S_OPS = {"sb": 0, "sh": 1, "sw": 2}


def assemble(lines):
    """lines: tuples like ("addi", rd, rs1, imm), ("beq", rs1, rs2, "label"), ("label", "name"), ("jal", rd, "label"),
    ("lw", rd, rs1, imm), ("sw", rs2, rs1, imm), ("lui", rd, imm20), ("ecall",).  Returns the instruction words."""
    labels, pc = {}, 0
    for ln in lines:
        if ln[0] == "label":
            labels[ln[1]] = pc
        else:
            pc += 4
    out, pc = [], 0
    for ln in lines:
        op = ln[0]
        if op == "label":
            continue
        tgt = lambda v: (labels[v] - pc) if isinstance(v, str) else v  # noqa: E731
        if op in R_OPS:
            f3, f7 = R_OPS[op]
            w = _r(0x33, f3, f7, ln[1], ln[2], ln[3])
        elif op in I_OPS:
            w = _i(0x13, I_OPS[op], ln[1], ln[2], ln[3])
        elif op in ("slli", "srli", "srai"):
            w = _r(0x13, 1 if op == "slli" else 5, 0x20 if op == "srai" else 0, ln[1], ln[2], ln[3] & 31)
        elif op in B_OPS:
            w = _b(B_OPS[op], ln[1], ln[2], tgt(ln[3]))
        elif op in L_OPS:
            w = _i(0x03, L_OPS[op], ln[1], ln[2], ln[3])
        elif op in S_OPS:
            w = _s(S_OPS[op], ln[2], ln[1], ln[3])
        elif op == "lui":
            w = ((ln[2] & 0xFFFFF) << 12) | (ln[1] << 7) | 0x37
        elif op == "auipc":
            w = ((ln[2] & 0xFFFFF) << 12) | (ln[1] << 7) | 0x17
        elif op == "jal":
            w = _j(ln[1], tgt(ln[2]))
        elif op == "jalr":
            w = _i(0x67, 0, ln[1], ln[2], ln[3])
        elif op == "ecall":
            w = 0x73
        elif op == "fence":
            w = 0x0FF0000F
        elif op == "phantom":   # ("phantom", kind, rs1): a FENCE-coded no-op whose sub-executor leaves advice in the hint stream (fm = 0101)
            w = (5 << 28) | ((ln[1] & 0xFF) << 20) | (ln[2] << 15) | 0x0F
        else:
            raise ValueError(op)
        out.append(w & M32)
        pc += 4
    return out


def li(rd, value):
    """load a 32-bit constant: lui + addi"""
    value &= M32
    lo = value & 0xFFF
    if lo >= 0x800:
        lo -= 0x1000
    hi = ((value - lo) >> 12) & 0xFFFFF
    return [("lui", rd, hi), ("addi", rd, rd, lo)]


def sdiv(a, b):
    q = abs(a) // abs(b)   # RISC-V division truncates towards zero
    return q if (a < 0) == (b < 0) else -q


def srem(a, b):
    r = abs(a) % abs(b)    # the remainder has the sign of the dividend
    return r if a >= 0 else -r


def sx(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def _sqrts(a, p):
    """square roots of a modulo the odd prime p (a a square): Tonelli - Shanks, written for this model; yields the root the product's
    sub-executor computes first (the algorithm is deterministic: same non-residue, same steps)"""
    if a == 0:
        yield 0
        return
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = next(c for c in range(2, 1000) if pow(c, (p - 1) // 2, p) == p - 1)
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c, t, r = i, b * b % p, t * b * b % p, r * b % p
    yield r


BABYBEAR = 2013265921


def ext4_mul(a, b):
    """(a b) mod (X^4 - 11) over BabyBear, schoolbook with Python integers"""
    t = [0] * 7
    for i in range(4):
        for j in range(4):
            t[i + j] += a[i] * b[j]
    return [(t[k] + 11 * (t[k + 4] if k < 3 else 0)) % BABYBEAR for k in range(4)]


def ext4_pow(a, e):
    r = [1, 0, 0, 0]
    while e:
        if e & 1:
            r = ext4_mul(r, a)
        a = ext4_mul(a, a)
        e >>= 1
    return r


def run(program, stdin=b"", pc_base=0x00200000, data_base=0x00400000, memory_bytes=1 << 20, data=b"", max_instr=10**7, moduli=(), curves=(), fp2=()):
    """Independent interpreter.  Returns dict(instret, pvs (32 bytes), exit_code, records)."""
    x = [0] * 32
    x[2] = data_base + memory_bytes
    mem = bytearray(memory_bytes)
    mem[:len(data)] = data
    pvs = bytearray(32)
    pc, n, pos = pc_base, 0, 0
    hints = []   # the hint stream (phantom instructions fill it, the input ecall drains it first)
    rec = dict(pc_index=[], alu=[], lt=[], mul=[], shift=0, branch=0, jump=0, load_store=0, mulh=0, divrem=0, lui_auipc=0, ecall=0,
               acc=[], bnd=[], shifts=[], beq=[], blt=[], jal=[], auipc=[], jalr=[], mulhs=[], ls=[], divs=[])
    s32 = lambda v: sx(v, 32)  # noqa: E731
    # memory log over 16-bit cells: address space 1 = registers (cell 2 i + half), 2 = memory (halfword index); entries
    # (as, ptr, prev_data, prev_ts, data, ts, is_read), timestamps 1, 2, ...
    cells, clock = {}, [0]

    def touch(as_, ptr, current, write, value):
        c = cells.setdefault((as_, ptr), [current, current, 0])   # initial, data, ts
        clock[0] += 1
        prev = (c[1], c[2])
        if write:
            c[1] = value
        c[2] = clock[0]
        rec["acc"].append((as_, ptr, prev[0], prev[1], c[1], clock[0], 0 if write else 1))

    def rr(i):
        touch(1, 2 * i, x[i] & 0xFFFF, False, 0)
        touch(1, 2 * i + 1, x[i] >> 16, False, 0)
        return x[i]

    def rw(i, v):
        v = 0 if i == 0 else v & M32
        touch(1, 2 * i, x[i] & 0xFFFF, True, v & 0xFFFF)
        touch(1, 2 * i + 1, x[i] >> 16, True, v >> 16)
        x[i] = v

    def cellv(c):
        return int.from_bytes(mem[2 * c:2 * c + 2], "little")

    def finish():
        for (as_, ptr) in sorted(cells):
            c = cells[(as_, ptr)]
            rec["bnd"].append((as_, ptr, c[0], c[1], c[2]))
    while True:
        assert n < max_instr
        idx = (pc - pc_base) // 4
        w = program[idx]
        rec["pc_index"].append(idx)
        op, rd, f3, rs1, rs2, f7 = w & 0x7F, (w >> 7) & 31, (w >> 12) & 7, (w >> 15) & 31, (w >> 20) & 31, w >> 25
        a = rr(rs1) if op in (0x67, 0x63, 0x03, 0x23, 0x13, 0x33) else 0
        b = rr(rs2) if op in (0x63, 0x23, 0x33) else 0
        imm_i = sx(w >> 20, 12)
        nxt, val, wr = pc + 4, 0, False

        def alu(o, p, q):
            rec["alu"].append((o, p, q))
            return [p + q, p - q, p ^ q, p | q, p & q][o] & M32

        def lt(o, p, q):
            rec["lt"].append((o, p, q))
            return int(s32(p) < s32(q)) if o == 0 else int(p < q)

        def shift(k, p, s):
            rec["shift"] += 1
            rec["shifts"].append((k, p, s))
            s &= 31
            return [(p << s) & M32, p >> s, (s32(p) >> s) & M32][k]

        if op == 0x0F:   # FENCE: no operation; with fm = 0101 a phantom instruction (advice for the guest, nothing the circuit sees)
            if w >> 28 == 5:
                kind, ptr = (w >> 20) & 0xFF, x[(w >> 15) & 31]
                off = ptr - data_base
                if kind == 2:   # the pairing extension's final-exponentiation witness (tests/pairing_util.py): [curve | f] -> c, u
                    import pairing_util as pu

                    curve, = struct.unpack_from("<I", mem, off)
                    assert curve in (0, 1)
                    ew = 12 if curve else 8                      # words per base-field element
                    wds = struct.unpack_from("<%dI" % (12 * ew), mem, off + 4)
                    el = lambda k: sum(wds[ew * k + i] << (32 * i) for i in range(ew))  # noqa: E731
                    cv = pu.Bls12_381 if curve else pu
                    c_, u_ = cv.final_exp_hint(cv.from_sextic([(el(2 * k), el(2 * k + 1)) for k in range(6)]))
                    for v in (c_, u_):
                        for a_, b_ in cv.to_sextic(v):
                            for e_ in (a_, b_):
                                hints.extend((e_ >> (32 * i)) & M32 for i in range(ew))
                    rec["alu"].append((0, 0, 0))
                    pc, n = pc + 4, n + 1
                    continue
                p_ = moduli[struct.unpack_from("<I", mem, off)[0]]
                nb = 32 if p_ < 1 << 256 else 48
                z_ = next(c for c in range(2, 1000) if pow(c, (p_ - 1) // 2, p_) == p_ - 1)
                words_of = lambda v: [(v >> (32 * i)) & M32 for i in range(nb // 4)]  # noqa: E731
                if kind == 0:
                    hints.extend(words_of(z_))
                elif kind == 1:
                    xv = int.from_bytes(mem[off + 4:off + 4 + nb], "little") % p_
                    sq = xv == 0 or pow(xv, (p_ - 1) // 2, p_) == 1
                    tgt = xv if sq else xv * z_ % p_
                    r_ = next(c for c in _sqrts(tgt, p_))
                    hints.extend([int(sq)] + words_of(r_))
                else:
                    raise ValueError("phantom kind")
            rec["alu"].append((0, 0, 0))
        elif op == 0x37:
            val, wr = w & 0xFFFFF000, True
            rec["lui_auipc"] += 1
            rec["jal"].append((1, pc, w >> 12))
        elif op == 0x17:
            val, wr = (pc + (w & 0xFFFFF000)) & M32, True
            rec["lui_auipc"] += 1
            rec["auipc"].append((pc, w >> 12))
        elif op == 0x6F:
            imm = sx(((w >> 31) << 20) | (((w >> 12) & 255) << 12) | (((w >> 20) & 1) << 11) | (((w >> 21) & 1023) << 1), 21)
            val, wr, nxt = pc + 4, True, (pc + imm) & M32
            rec["jump"] += 1
            rec["jal"].append((0, pc, imm % 2013265921))
        elif op == 0x67:
            val, wr, nxt = pc + 4, True, (a + imm_i) & M32 & ~1
            rec["jump"] += 1
            rec["jalr"].append((pc, a, w >> 20))
        elif op == 0x63:
            imm = sx(((w >> 31) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 63) << 5) | (((w >> 8) & 15) << 1), 13)
            t = {0: a == b, 1: a != b, 4: s32(a) < s32(b), 5: s32(a) >= s32(b), 6: a < b, 7: a >= b}[f3]
            if t:
                nxt = (pc + imm) & M32
            rec["branch"] += 1
            if f3 < 2:
                rec["beq"].append((f3, a, b, imm % 2013265921))
            else:
                rec["blt"].append(({4: 0, 6: 1, 5: 2, 7: 3}[f3], a, b, imm % 2013265921))
        elif op == 0x03:
            addr = (a + imm_i) & M32
            o = addr - data_base
            size = {0: 1, 1: 2, 2: 4, 4: 1, 5: 2}[f3]
            v = int.from_bytes(mem[o:o + size], "little")
            for c in range(o >> 1, ((o + size - 1) >> 1) + 1):
                touch(2, c, cellv(c), False, 0)
            val = (sx(v, 8 * size) & M32) if f3 in (0, 1) else v
            wr = True
            rec["load_store"] += 1
            wo, off = o & ~3, o & 3
            case = {2: 0, 5: 1 + off // 2, 4: 3 + off, 1: 14 + off // 2, 0: 16 + off}[f3]
            rec["ls"].append((case, int.from_bytes(bytes(mem[wo:wo + 4]).ljust(4, b"\0"), "little"), x[rd]))
        elif op == 0x23:
            imm = sx(((w >> 25) << 5) | ((w >> 7) & 31), 12)
            o = ((a + imm) & M32) - data_base
            size = 1 << f3
            cs = list(range(o >> 1, ((o + size - 1) >> 1) + 1))
            before = [cellv(c) for c in cs]
            wo, off = o & ~3, o & 3
            rec["ls"].append(({2: 7, 1: 8 + off // 2, 0: 10 + off}[f3], b, int.from_bytes(bytes(mem[wo:wo + 4]).ljust(4, b"\0"), "little")))
            mem[o:o + size] = (b & ((1 << (8 * size)) - 1)).to_bytes(size, "little")
            for c, bv in zip(cs, before):
                touch(2, c, bv, True, cellv(c))
            rec["load_store"] += 1
        elif op == 0x13:
            c = imm_i & M32
            wr = True
            if f3 == 0:
                val = alu(0, a, c)
            elif f3 == 2:
                val = lt(0, a, c)
            elif f3 == 3:
                val = lt(1, a, c)
            elif f3 == 4:
                val = alu(2, a, c)
            elif f3 == 6:
                val = alu(3, a, c)
            elif f3 == 7:
                val = alu(4, a, c)
            elif f3 == 1:
                val = shift(0, a, rs2)
            else:
                val = shift(2 if f7 else 1, a, rs2)
        elif op == 0x33:
            wr = True
            if f7 == 1:
                sa, sb = s32(a), s32(b)
                if f3 == 0:
                    val = (a * b) & M32
                    rec["mul"].append((a, b))
                elif f3 == 1:
                    val = ((sa * sb) >> 32) & M32
                elif f3 == 2:
                    val = ((sa * b) >> 32) & M32
                elif f3 == 3:
                    val = ((a * b) >> 32) & M32
                elif f3 == 4:
                    val = M32 if b == 0 else (a if (a == 0x80000000 and b == M32) else sdiv(sa, sb) & M32)
                elif f3 == 5:
                    val = M32 if b == 0 else a // b
                elif f3 == 6:
                    val = a if b == 0 else (0 if (a == 0x80000000 and b == M32) else srem(sa, sb) & M32)
                else:
                    val = a if b == 0 else a % b
                if 1 <= f3 <= 3:
                    rec["mulh"] += 1
                    rec["mulhs"].append((f3 - 1, a, b))
                if f3 >= 4:
                    rec["divrem"] += 1
                    rec["divs"].append((f3 - 4, a, b))
            else:
                if f3 == 0:
                    val = alu(1 if f7 else 0, a, b)
                elif f3 == 1:
                    val = shift(0, a, b)
                elif f3 == 2:
                    val = lt(0, a, b)
                elif f3 == 3:
                    val = lt(1, a, b)
                elif f3 == 4:
                    val = alu(2, a, b)
                elif f3 == 5:
                    val = shift(2 if f7 else 1, a, b)
                elif f3 == 6:
                    val = alu(3, a, b)
                else:
                    val = alu(4, a, b)
        elif op == 0x73:
            rec["ecall"] += 1
            call = rr(17)
            if call == 93:
                code = rr(10)
                n += 1
                finish()
                return dict(instret=n, pvs=bytes(pvs), exit_code=code, records=rec)
            if call == 1:
                word = rr(10)
                k = rr(11)
                pvs[4 * k:4 * k + 4] = struct.pack("<I", word)
            elif call == 2:
                if hints:   # what a phantom instruction left comes before the input stream
                    rw(10, hints.pop(0))
                else:
                    rw(10, struct.unpack_from("<I", stdin, pos)[0])
                    pos += 4
            elif call == 3:   # Keccak-f[1600] in place on the 200 bytes at a0 (the memory log of this model does not cover it)
                off = rr(10) - data_base
                assert off % 4 == 0 and 0 <= off and off + 200 <= len(mem)
                mem[off:off + 200] = keccak_f1600_bytes(bytes(mem[off:off + 200]))
            elif call == 6:   # a = b op c modulo 2^256 on the 24 words at a0 (b | c | a), op = a1: add sub xor or and
                off = rr(10) - data_base
                op_ = rr(11)
                assert off % 4 == 0 and 0 <= off and off + 96 <= len(mem) and op_ < 18
                b_, c_ = int.from_bytes(mem[off:off + 32], "little"), int.from_bytes(mem[off + 32:off + 64], "little")
                sg = lambda v: v - (1 << 256) if v >> 255 else v  # noqa: E731
                if op_ >= 12:   # 256-bit branches: 12 beq, 13 bne, 14 bltu, 15 blt, 16 bgeu, 17 bge -- the comparison's 0 / 1 goes to a, pc += a2 if taken
                    cmp_ = [int(b_ == c_), int(b_ == c_), int(b_ < c_), int(sg(b_) < sg(c_)), int(b_ < c_), int(sg(b_) < sg(c_))][op_ - 12]
                    mem[off + 64:off + 96] = cmp_.to_bytes(32, "little")
                    off_ = rr(12)
                    assert off_ % 4 == 0
                    if bool(cmp_) != (op_ in (13, 16, 17)):
                        nxt = (pc + off_) & M32
                else:
                    a_ = [(b_ + c_) % (1 << 256), (b_ - c_) % (1 << 256), b_ ^ c_, b_ | c_, b_ & c_, (b_ * c_) % (1 << 256), int(b_ < c_), int(sg(b_) < sg(c_)),
                          int(b_ == c_), (b_ << (c_ % 256)) % (1 << 256), b_ >> (c_ % 256), (sg(b_) >> (c_ % 256)) % (1 << 256)][op_]
                    mem[off + 64:off + 96] = a_.to_bytes(32, "little")
            elif call == 5:   # r = a b mod moduli[a1] on the 24 words at a0 (a | b | r, little-endian)
                off = rr(10) - data_base
                sel = rr(11)                 # a1 = modulus index + 8 * operation (0 mul, 1 add, 2 sub, 3 div, 4 is_eq)
                p_, mop = moduli[sel & 7], sel >> 3
                nb = 32 if p_ < 1 << 256 else 48   # bytes per operand: a modulus above 2^256 (BLS12-381's base field) has 48-byte operands
                assert off % 4 == 0 and 0 <= off and off + 3 * nb <= len(mem) and mop < 5
                a_, b_ = int.from_bytes(mem[off:off + nb], "little"), int.from_bytes(mem[off + nb:off + 2 * nb], "little")
                res = int((a_ - b_) % p_ == 0) if mop == 4 else a_ * pow(b_, -1, p_) % p_ if mop == 3 else [a_ * b_ % p_, (a_ + b_) % p_, (a_ - b_) % p_][mop]
                mem[off + 2 * nb:off + 3 * nb] = res.to_bytes(nb, "little")
            elif call == 7:   # (x3, y3) = p1 + p2 (op 0) or 2 p1 (op 1) on curves[a1 & 7] = (modulus, a), 48 words at a0 (p1 | p2 | p3)
                off = rr(10) - data_base
                sel = rr(11)
                (p_, a_), eop = curves[sel & 7], sel >> 3
                nb = 32 if p_ < 1 << 256 else 48
                assert off % 4 == 0 and 0 <= off and off + 6 * nb <= len(mem) and eop < 2
                x1, y1, x2, y2 = (int.from_bytes(mem[off + nb * k:off + nb * k + nb], "little") for k in range(4))
                lam = (3 * x1 * x1 + a_) * pow(2 * y1, -1, p_) % p_ if eop else (y2 - y1) * pow(x2 - x1, -1, p_) % p_
                x3 = (lam * lam - x1 - (x1 if eop else x2)) % p_
                mem[off + 4 * nb:off + 5 * nb] = x3.to_bytes(nb, "little")
                mem[off + 5 * nb:off + 6 * nb] = ((lam * (x1 - x3) - y1) % p_).to_bytes(nb, "little")
            elif call == 8:   # r = a op b in Fp[u] / (u^2 + 1) over fp2[a1 & 7], 48 words at a0 (a0 a1 | b0 b1 | r0 r1); op 0 mul 1 add 2 sub 3 div
                off = rr(10) - data_base
                sel = rr(11)
                p_, fop = fp2[sel & 7], sel >> 3
                nb = 32 if p_ < 1 << 256 else 48
                assert off % 4 == 0 and 0 <= off and off + 6 * nb <= len(mem) and fop < 4
                a0_, a1_, b0_, b1_ = (int.from_bytes(mem[off + nb * k:off + nb * k + nb], "little") for k in range(4))
                if fop == 3:   # divide: multiply by the conjugate over the norm
                    nrm = pow((b0_ * b0_ + b1_ * b1_) % p_, -1, p_)
                    b0_, b1_ = b0_ * nrm % p_, -b1_ * nrm % p_
                if fop in (0, 3):
                    r0_, r1_ = (a0_ * b0_ - a1_ * b1_) % p_, (a0_ * b1_ + a1_ * b0_) % p_
                else:
                    sg = 1 if fop == 1 else -1
                    r0_, r1_ = (a0_ + sg * b0_) % p_, (a1_ + sg * b1_) % p_
                mem[off + 4 * nb:off + 5 * nb] = r0_.to_bytes(nb, "little")
                mem[off + 5 * nb:off + 6 * nb] = r1_.to_bytes(nb, "little")
            elif call == 4:   # SHA-256 compression on the 24 words at a0: state[8] <- compress(state, block[16])
                off = rr(10) - data_base
                assert off % 4 == 0 and 0 <= off and off + 96 <= len(mem)
                buf = list(struct.unpack_from("<24I", mem, off))
                struct.pack_into("<8I", mem, off, *sha256_compress(buf[:8], buf[8:]))
            elif call in (9, 10):   # native field (3 words: a | b | r) / its quartic extension X^4 = 11 (12 words); a1 = op: add sub mul div
                off = rr(10) - data_base
                fop = rr(11)
                nwd = 3 if call == 9 else 12
                assert off % 4 == 0 and 0 <= off and off + 4 * nwd <= len(mem) and fop < 4
                wds = list(struct.unpack_from("<%dI" % nwd, mem, off))
                if call == 9:
                    a_, b_ = wds[0] % BABYBEAR, wds[1] % BABYBEAR
                    r_ = [(a_ + b_) % BABYBEAR, (a_ - b_) % BABYBEAR, a_ * b_ % BABYBEAR, a_ * pow(b_, -1, BABYBEAR) % BABYBEAR if fop == 3 else 0][fop]
                    struct.pack_into("<I", mem, off + 8, r_)
                else:
                    a_, b_ = [v % BABYBEAR for v in wds[:4]], [v % BABYBEAR for v in wds[4:8]]
                    if fop == 3:
                        b_ = ext4_pow(b_, BABYBEAR ** 4 - 2)   # the inverse by Fermat: independent of the C++ norm formula
                    r_ = ext4_mul(a_, b_) if fop in (2, 3) else [(u + (v if fop == 0 else -v)) % BABYBEAR for u, v in zip(a_, b_)]
                    struct.pack_into("<4I", mem, off + 32, *r_)
            elif call == 11:   # castf: the word at a0 (below 2^30) copied to the word at a0 + 4
                off = rr(10) - data_base
                assert off % 4 == 0 and 0 <= off and off + 8 <= len(mem)
                v, = struct.unpack_from("<I", mem, off)
                assert v < 1 << 30
                struct.pack_into("<I", mem, off + 4, v)
            else:
                raise ValueError("ecall %d" % call)
        else:
            raise ValueError("illegal instruction %08x" % w)
        if wr:
            rw(rd, val)
        pc = nxt
        n += 1


def sha256_constants():
    """K_t = the first 32 bits of the fractional parts of the cube roots of the first 64 primes (FIPS 180-4 4.2.2), derived here."""
    primes, c = [], 2
    while len(primes) < 64:
        if all(c % q for q in primes):
            primes.append(c)
        c += 1

    def icbrt(n):
        x = int(round(n ** (1 / 3)))
        while x ** 3 > n:
            x -= 1
        while (x + 1) ** 3 <= n:
            x += 1
        return x
    return [icbrt(q << 96) & 0xFFFFFFFF for q in primes]


def sha256_compress(h, m):
    """FIPS 180-4 6.2.2 on words"""
    k = sha256_constants()
    rr_ = lambda v, n: ((v >> n) | (v << (32 - n))) & M32  # noqa: E731
    w = list(m)
    for t in range(16, 64):
        s0 = rr_(w[t - 15], 7) ^ rr_(w[t - 15], 18) ^ (w[t - 15] >> 3)
        s1 = rr_(w[t - 2], 17) ^ rr_(w[t - 2], 19) ^ (w[t - 2] >> 10)
        w.append((w[t - 16] + s0 + w[t - 7] + s1) & M32)
    a, b, c, d, e, f, g, hh = h
    for t in range(64):
        t1 = (hh + (rr_(e, 6) ^ rr_(e, 11) ^ rr_(e, 25)) + ((e & f) ^ (~e & M32 & g)) + k[t] + w[t]) & M32
        t2 = ((rr_(a, 2) ^ rr_(a, 13) ^ rr_(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & M32
        a, b, c, d, e, f, g, hh = (t1 + t2) & M32, a, b, c, (d + t1) & M32, e, f, g
    return [(x + y) & M32 for x, y in zip(h, (a, b, c, d, e, f, g, hh))]


def keccak_f1600_bytes(state200):
    """Keccak-f[1600] on 200 bytes (25 little-endian lanes, lane x + 5 y), written from FIPS 202 section 3 for this model."""
    rc = []
    r = 1
    for _ in range(24):               # round constants from the LFSR of section 3.2.5
        c = 0
        for j in range(7):
            if r & 1:
                c |= 1 << ((1 << j) - 1)
            r = ((r << 1) ^ (0x71 if r & 0x80 else 0)) & 0xFF
        rc.append(c)
    rot = [[0] * 5 for _ in range(5)]
    xx, yy = 1, 0
    for t in range(24):               # rotation offsets (t + 1)(t + 2) / 2 along the walk (x, y) -> (y, 2 x + 3 y)
        rot[xx][yy] = ((t + 1) * (t + 2) // 2) % 64
        xx, yy = yy, (2 * xx + 3 * yy) % 5
    m64 = (1 << 64) - 1
    rol = lambda v, k: ((v << k) | (v >> (64 - k))) & m64 if k else v  # noqa: E731
    a = [[int.from_bytes(state200[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], rot[x][y])
        a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & m64 & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc[rnd]
    return b"".join(a[x][y].to_bytes(8, "little") for y in range(5) for x in range(5))


def exe_bytes(program, pc_base=0x00200000, data_base=0x00400000, memory_bytes=1 << 20, data=b""):
    """the vm_cli exe format"""
    return struct.pack("<III", 0x58455A4B, pc_base, len(program)) + struct.pack("<%dI" % len(program), *program) + \
        struct.pack("<III", data_base, memory_bytes, len(data)) + bytes(data)


def elf_bytes(program, text_vaddr=0x00200000, data=b"", data_vaddr=0x00400000, bss=0, entry_offset=0):
    """A minimal ELF32 little-endian RISC-V executable: one PF_X PT_LOAD segment (the program), one PF_R|PF_W segment (data + bss)."""
    text = struct.pack("<%dI" % len(program), *program)
    n_ph = 2 if (data or bss) else 1
    ehsize, phsize = 52, 32
    off_text = ehsize + n_ph * phsize
    off_data = off_text + len(text)
    eh = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8) + struct.pack("<HHIIIIIHHHHHH", 2, 243, 1, text_vaddr + entry_offset, ehsize, 0, 0,
                                                                  ehsize, phsize, n_ph, 40, 0, 0)
    ph = struct.pack("<IIIIIIII", 1, off_text, text_vaddr, text_vaddr, len(text), len(text), 5, 4)
    if n_ph == 2:
        ph += struct.pack("<IIIIIIII", 1, off_data, data_vaddr, data_vaddr, len(data), len(data) + bss, 6, 4)
    return eh + ph + text + bytes(data)
