"""Independent big-int Python model of the arithmetic spec.

Written separately from oracle/*.c (different language, different algorithms:
Python ints, naive O(n^2) DFT, list-based sponge) so that agreement between the
two is evidence that both follow the same published definitions.  It is used
only by tests/golden/gen_golden.py to produce the committed fixtures and by a
few small CPU tests.  PARITY UNPINNED against p3 0.4.3 (no upstream vectors
are available offline; see oracle/zk_oracle.h).
"""
P = 2013265921
GEN_2_27 = 0x1A427A41


def two_adic_generator(bits):
    g = GEN_2_27
    for _ in range(bits, 27):
        g = g * g % P
    return g


def inv(a):
    return pow(a, P - 2, P)


# ---- extension F[x]/(x^4 - 11) -------------------------------------------
def ext_mul(a, b):
    c = [0] * 7
    for i in range(4):
        for j in range(4):
            c[i + j] += a[i] * b[j]
    return [(c[k] + 11 * (c[k + 4] if k + 4 < 7 else 0)) % P for k in range(4)]


def ext_pow(a, e):
    r = [1, 0, 0, 0]
    while e:
        if e & 1:
            r = ext_mul(r, a)
        a = ext_mul(a, a)
        e >>= 1
    return r


def ext_inv(a):
    return ext_pow(a, P ** 4 - 2)


# ---- DFT -------------------------------------------------------------------
def dft_naive(xs, inverse=False):
    n = len(xs)
    g = two_adic_generator(n.bit_length() - 1)
    if inverse:
        g = inv(g)
    out = [sum(x * pow(g, i * j, P) for j, x in enumerate(xs)) % P for i in range(n)]
    if inverse:
        ni = inv(n)
        out = [o * ni % P for o in out]
    return out


def coset_lde_naive(xs, added_bits, shift):
    n = len(xs)
    coeffs = dft_naive(xs, inverse=True)
    coeffs = [c * pow(shift, i, P) % P for i, c in enumerate(coeffs)] + [0] * ((n << added_bits) - n)
    return dft_naive(coeffs)


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


# ---- Poseidon2 -------------------------------------------------------------
def grain_constants(n_bits=31, t=16, r_f=8, r_p=13):
    bits = []

    def put(v, n):
        bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    put(1, 2), put(0, 4), put(n_bits, 12), put(t, 12), put(r_f, 10), put(r_p, 10)
    bits.extend([1] * 30)
    assert len(bits) == 80

    def step():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0)
        bits.append(nb)
        return nb

    for _ in range(160):
        step()

    def next_bit():
        nb = step()
        while nb == 0:
            step()
            nb = step()
        return step()

    out = []
    while len(out) < r_f * t + r_p:
        v = 0
        for _ in range(n_bits):
            v = (v << 1) | next_bit()
        if v < P:
            out.append(v)
    return out


RC = grain_constants()
_i2 = inv(2)
DIAG = [P - 2, 1, 2, _i2, 3, 4, P - _i2, P - 3, P - 4, pow(_i2, 8, P), pow(_i2, 2, P), pow(_i2, 3, P),
        pow(_i2, 27, P), P - pow(_i2, 8, P), P - pow(_i2, 4, P), P - pow(_i2, 27, P)]
M4 = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]


def external_linear(s):
    blocks = []
    for b in range(0, 16, 4):
        x = s[b:b + 4]
        blocks.append([sum(M4[i][j] * x[j] for j in range(4)) % P for i in range(4)])
    sums = [sum(blk[k] for blk in blocks) % P for k in range(4)]
    return [(blocks[i // 4][i % 4] + sums[i % 4]) % P for i in range(16)]


def internal_linear(s):
    sm = sum(s) % P
    return [(s[i] * DIAG[i] + sm) % P for i in range(16)]


def permute(s):
    s = external_linear(list(s))
    for r in range(4):
        s = external_linear([pow((s[i] + RC[r * 16 + i]) % P, 7, P) for i in range(16)])
    for r in range(13):
        s[0] = pow((s[0] + RC[64 + r]) % P, 7, P)
        s = internal_linear(s)
    for r in range(4):
        s = external_linear([pow((s[i] + RC[77 + r * 16 + i]) % P, 7, P) for i in range(16)])
    return s


def hash_slice(xs):
    s = [0] * 16
    for i in range(0, len(xs), 8):
        chunk = xs[i:i + 8]
        s[:len(chunk)] = chunk
        s = permute(s)
    return s[:8]


def compress(l, r):
    return permute(list(l) + list(r))[:8]


def merkle_root(mats):
    """mats: list of (log_height, rows) with rows = list of row lists."""
    lh = max(m[0] for m in mats)

    def rows_at(level, i):
        out = []
        for (h, rows) in mats:
            if h == level:
                out += rows[i]
        return out

    layer = [hash_slice(rows_at(lh, i)) for i in range(1 << lh)]
    level = lh
    while level > 0:
        level -= 1
        nxt = [compress(layer[2 * i], layer[2 * i + 1]) for i in range(1 << level)]
        if any(h == level for h, _ in mats):
            nxt = [compress(nxt[i], hash_slice(rows_at(level, i))) for i in range(1 << level)]
        layer = nxt
    return layer[0]


class Challenger:
    def __init__(self):
        self.state = [0] * 16
        self.inp = []
        self.out = []

    def _duplex(self):
        self.state[:len(self.inp)] = self.inp
        self.inp = []
        self.state = permute(self.state)
        self.out = self.state[:8]

    def observe(self, vals):
        for v in vals:
            self.out = []
            self.inp.append(v)
            if len(self.inp) == 8:
                self._duplex()

    def sample(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def sample_ext(self):
        return [self.sample() for _ in range(4)]

    def sample_bits(self, bits):
        return self.sample() & ((1 << bits) - 1)

    def clone(self):
        c = Challenger()
        c.state, c.inp, c.out = list(self.state), list(self.inp), list(self.out)
        return c

    def grind(self, bits):
        w = 0
        while True:
            c = self.clone()
            c.observe([w])
            if c.sample_bits(bits) == 0:
                self.state, self.inp, self.out = c.state, c.inp, c.out
                return w
            w += 1


def fri_fold(vals, beta):
    """vals: list of 2n ext elements (bit-reversed order); returns n ext elements."""
    n = len(vals) // 2
    h = n.bit_length() - 1
    g = two_adic_generator(h + 1)
    out = []
    for i in range(n):
        x = pow(g, bitrev(i, h), P)
        e0, e1 = vals[2 * i], vals[2 * i + 1]
        c = inv((-2 * x) % P)
        d = [((e1[k] - e0[k]) * c) % P for k in range(4)]
        bx = [(beta[0] - x) % P] + list(beta[1:])
        t = ext_mul(bx, d)
        out.append([(e0[k] + t[k]) % P for k in range(4)])
    return out


# ---- LogUp / sum-check building blocks ------------------------------------------------------------
def ext_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def logup_running_sum(den, num):
    acc, out = [0, 0, 0, 0], []
    for d, m in zip(den, num):
        inv_d = ext_inv(d)
        acc = ext_add(acc, [x * m % P for x in inv_d])
        out.append(acc)
    return out


def mle_fold(vals, r):
    out = []
    for i in range(len(vals) // 2):
        a, b = vals[2 * i], vals[2 * i + 1]
        t = ext_mul(r, [(y - x) % P for x, y in zip(a, b)])
        out.append(ext_add(a, t))
    return out


def sumcheck_round(tables):
    k, n_half = len(tables), len(tables[0]) // 2
    out = []
    for t in range(k + 1):
        acc = [0, 0, 0, 0]
        for i in range(n_half):
            prod = [1, 0, 0, 0]
            for f in tables:
                a, b = f[2 * i], f[2 * i + 1]
                prod = ext_mul(prod, [(x + t * (y - x)) % P for x, y in zip(a, b)])
            acc = ext_add(acc, prod)
        out.append(acc)
    return out
