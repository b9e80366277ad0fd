"""The pairing extension's final-exponentiation hint, restated with Python integers (TEST INFRASTRUCTURE; include/zkhip_pairing.hpp is the
product's).  `[app_vm_config.pairing] supported_curves = ["Bn254"]` (crates/circuits/chunk-circuit/openvm.toml:35-36) brings no chip in
OpenVM: the extension is a PHANTOM sub-executor (PairingPhantom::HintFinalExp, un-vendored openvm-pairing) that leaves a residue witness in
the hint stream, so that the guest checks a pairing equation with ONE exponentiation by lambda instead of the final exponentiation
(Novakovic, Eagen: "On Proving Pairings", 2024):  for f with f^((p^12 - 1) / r) = 1 there are c and u, u a power of a fixed 27th root of
unity, with   c^lambda = f u,   lambda = 6 x + 2 + p - p^2 + p^3.

Fp12 is Fp[w] / (w^12 - 18 w^6 + 82) (w^6 = 9 + u, u^2 = -1): twelve coefficients, schoolbook products.  The I/O layout is OpenVM's
SexticExtField<Fp2>: six Fp2 coefficients c_i = a_i + b_i u of w^i, i.e. flat coefficient k < 6 is a_k - 9 b_k and coefficient k + 6 is b_k.
Everything here is computed from the curve's parameter x alone."""

X = 4965661367192848881
P = 36 * X**4 + 36 * X**3 + 24 * X**2 + 6 * X + 1
R = 36 * X**4 + 36 * X**3 + 18 * X**2 + 6 * X + 1
N = P**12 - 1
H = N // R
LAMBDA = 6 * X + 2 + P - P**2 + P**3
assert N % R == 0 and LAMBDA % R == 0 and N % 27 == 0 and N % 81 != 0
D27 = 27
U_ORDER = N // (R * D27)


def mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [v % P for v in t[:12]]


ONE = [1] + [0] * 11


def power(a, e):
    r, b = ONE, a
    while e:
        if e & 1:
            r = mul(r, b)
        b = mul(b, b)
        e >>= 1
    return r


def from_sextic(c):
    """six (a_i, b_i) pairs -> twelve flat coefficients"""
    return [(c[k][0] - 9 * c[k][1]) % P for k in range(6)] + [c[k][1] % P for k in range(6)]


def to_sextic(f):
    return [((f[k] + 9 * f[k + 6]) % P, f[k + 6]) for k in range(6)]


TAU = power([0, 1] + [0] * 10, N // 27)          # a generator of the 27-part: (the class of w)^((p^12 - 1) / 27)
assert power(TAU, 9) != ONE and power(TAU, 27) == ONE


def final_exp_hint(f):
    """(c, u) with c^lambda = f u for f in the subgroup of order (p^12 - 1) / r; the rule that makes the pair unique is the product's:
    u = tau^j for the smallest j in {0, 1, 2} that makes f u a cube; c = c_U tau^k with c_U the lambda-th root in the part of order coprime
    to 3 r and k the smallest exponent with (tau^k)^lambda = the 27-part of f u."""
    assert power(f, H) == ONE, "f is not in the subgroup a Miller loop's output lies in"
    for j in range(3):
        u = power(TAU, j)
        y = mul(f, u)
        if power(y, N // 3) == ONE:
            break
    else:
        raise AssertionError("no cubic residue among f, f tau, f tau^2")
    rd = R * D27
    proj_u = rd * pow(rd, -1, U_ORDER)                     # the projector onto the part of order U_ORDER
    c_u = power(y, proj_u * pow(LAMBDA, -1, U_ORDER) % N)
    ru = R * U_ORDER
    y_t = power(y, ru * pow(ru, -1, D27) % N)              # the 27-part of y
    for k in range(27):
        if power(TAU, k * LAMBDA % 27) == y_t:
            return mul(c_u, power(TAU, k)), u
    raise AssertionError("the 27-part has no lambda-th root")


def sample_f(seed):
    """an element of the subgroup of order (p^12 - 1) / r: g^r for a pseudo-random g"""
    import hashlib

    g = [int.from_bytes(hashlib.sha256(b"zkhip pairing hint %d %d" % (seed, k)).digest(), "big") % P for k in range(12)]
    return power(g, R)


# ---- BLS12-381 (`[app_vm_config.pairing] supported_curves = ["Bls12_381"]`, crates/circuits/batch-circuit/openvm.toml:25-26) ----------------
# c^lambda = f s with lambda = p + |x|.  (p^12 - 1) / r = 27 * ((|x| + 1) / 3) * C, C coprime to the other two factors and to lambda: the
# scaling factor s takes f to its component of order C (f s = f^e with e = 1 mod C and e = 0 mod the rest) and c is the lambda-th root there
# (gnark's / OpenVM's construction).  Fp12 = Fp[w] / (w^12 - 2 w^6 + 2) (w^6 = 1 + u, u^2 = -1).
class Bls12_381:
    X = -0xd201000000010000
    P = (X - 1)**2 * (X**4 - X**2 + 1) // 3 + X
    R = X**4 - X**2 + 1
    N = P**12 - 1
    H = N // R
    LAMBDA = P - X
    SMALL = 27 * (abs(X - 1) // 3)
    C = H // SMALL
    ONE = [1] + [0] * 11

    @classmethod
    def mul(cls, a, b):
        t = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    t[i + j] += x * y
        for k in range(22, 11, -1):
            c = t[k]
            if c:
                t[k - 6] += 2 * c
                t[k - 12] -= 2 * c
        return [v % cls.P for v in t[:12]]

    @classmethod
    def power(cls, a, e):
        r, b = cls.ONE, a
        while e:
            if e & 1:
                r = cls.mul(r, b)
            b = cls.mul(b, b)
            e >>= 1
        return r

    @classmethod
    def from_sextic(cls, c):
        return [(c[k][0] - c[k][1]) % cls.P for k in range(6)] + [c[k][1] % cls.P for k in range(6)]

    @classmethod
    def to_sextic(cls, f):
        return [((f[k] + f[k + 6]) % cls.P, f[k + 6]) for k in range(6)]

    @classmethod
    def final_exp_hint(cls, f):
        assert cls.H % cls.SMALL == 0 and cls.LAMBDA % cls.R == 0
        assert cls.power(f, cls.H) == cls.ONE, "f is not in the subgroup a Miller loop's output lies in"
        e = cls.SMALL * pow(cls.SMALL, -1, cls.C) % cls.H
        s = cls.power(f, (e - 1) % cls.H)
        c = cls.power(f, e * pow(cls.LAMBDA, -1, cls.C) % cls.H)
        return c, s

    @classmethod
    def sample_f(cls, seed):
        import hashlib

        g = [int.from_bytes(hashlib.sha512(b"zkhip pairing hint bls %d %d" % (seed, k)).digest(), "big") % cls.P for k in range(12)]
        return cls.power(g, cls.R)
