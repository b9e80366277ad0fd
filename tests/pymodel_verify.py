"""A third, independent verifier of this repository's STARK proofs, in pure Python big-int arithmetic on top of
tests/pymodel.py (naive field / Poseidon2 / challenger).  TEST INFRASTRUCTURE: written from the protocol
description in DESIGN.md section 4, sharing no code with oracle/stark.c or zkvm-prover_amd/csrc/verifier.hip; used by
tests/test_pymodel_verify.py to check proofs produced by the oracle (and, through the parity tests, by the HIP prover).

Covers: multi-AIR mixed heights, public values, preprocessed traces (commitment in the verifying key), bus
interactions (LogUp permutation phase), quotient chunks, two-adic FRI with per-round and query proof-of-work."""
import pymodel as m

P = m.P
AIR_MAGIC, PREP_MAGIC, LOGUP_MAGIC = 0x31414B5A, 0x50504B5A, 0x554C4B5A
PROOF_MAGIC, PROTO_TAG, GEN = 0x31504B5A, 0x5A4B4831, 31
MAX_FIELDS = 32
(VAR, PUB, CONST, FIRST, LAST, TRANS, ADD, SUB, MUL, NEG, PERM, CHAL, EXPOSED, PREP) = range(14)


class Reject(Exception):
    pass


def need(cond, why):
    if not cond:
        raise Reject(why)


# ---- extension field helpers (elements are lists of 4 ints) ------------------------------------
def e_from(a):
    return [a % P, 0, 0, 0]


def e_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def e_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def e_scale(a, s):
    return [x * s % P for x in a]


def e_mul(a, b):
    return m.ext_mul(a, b)


def e_inv(a):
    return m.ext_inv(a)


def e_pow(a, e):
    return m.ext_pow(a, e)


ZERO, ONE = [0, 0, 0, 0], [1, 0, 0, 0]


# ---- AIR program ------------------------------------------------------------------------------------
class Program:
    def __init__(self, words, width):
        w = [int(x) for x in words]
        need(len(w) >= 4 and w[0] == AIR_MAGIC, "bad AIR magic")
        self.n_nodes, self.n_cons, self.n_pvs = w[1], w[2], w[3]
        q = 4 + 3 * self.n_nodes
        need(q + self.n_cons <= len(w), "short program")
        self.nodes = [tuple(w[4 + 3 * i:7 + 3 * i]) for i in range(self.n_nodes)]
        self.cons = w[q:q + self.n_cons]
        q += self.n_cons
        self.prep_width, self.ints = 0, []
        if q + 2 <= len(w) and w[q] == PREP_MAGIC:
            self.prep_width = w[q + 1]
            need(self.prep_width > 0, "empty preprocessed section")
            q += 2
        if q != len(w):
            need(q + 2 <= len(w) and w[q] == LOGUP_MAGIC, "trailing words")
            n_int = w[q + 1]
            q += 2
            groups = []
            need(n_int > 0, "empty interaction section")
            for _ in range(n_int):
                need(q + 4 <= len(w), "short interaction")
                bus, sign, count, nf = w[q:q + 4]
                q += 4
                need(sign <= 1 and 1 <= nf <= MAX_FIELDS and q + nf <= len(w) and bus < P - 1, "bad interaction")
                self.ints.append((bus, sign, count, w[q:q + nf]))
                q += nf
                need(q < len(w), "short interaction")
                prev = groups[-1] if groups else 0
                need(w[q] == prev or (groups and w[q] == prev + 1), "group order")
                groups.append(w[q])
                q += 1
            need(q == len(w), "trailing words")
        self.perm_width = 4 * (groups[-1] + 2) if self.ints else 0
        deg = []
        for i, (op, a, b) in enumerate(self.nodes):
            if op == VAR:
                need(a < width and b <= 1, "VAR")
                deg.append(1)
            elif op == PREP:
                need(a < self.prep_width and b <= 1, "PREP")
                deg.append(1)
            elif op == PERM:
                need(a < self.perm_width and b <= 1, "PERM")
                deg.append(1)
            elif op == PUB:
                need(a < self.n_pvs, "PUB")
                deg.append(0)
            elif op == CONST:
                need(a < P, "CONST")
                deg.append(0)
            elif op in (FIRST, LAST):
                deg.append(1)
            elif op == TRANS:
                deg.append(0)
            elif op == CHAL:
                need(self.ints and a < 4 * (1 + MAX_FIELDS), "CHAL")
                deg.append(0)
            elif op == EXPOSED:
                need(self.ints and a < 4, "EXPOSED")
                deg.append(0)
            elif op in (ADD, SUB):
                need(a < i and b < i, "operand order")
                deg.append(max(deg[a], deg[b]))
            elif op == MUL:
                need(a < i and b < i, "operand order")
                deg.append(deg[a] + deg[b])
            elif op == NEG:
                need(a < i, "operand order")
                deg.append(deg[a])
            else:
                raise Reject("unknown op")
        need(all(c < self.n_nodes for c in self.cons), "constraint index")
        self.max_degree = max([deg[c] for c in self.cons], default=0)
        # quotient chunks: the quotient of degree-d constraints has degree < (d - 1) N -> next power of two of max(d, 2) - 1
        self.qd = 1
        while self.qd + 1 < max(self.max_degree, 2):
            self.qd *= 2
        # bus operands are expressions of the current row only
        for (_, _, count, fields) in self.ints:
            stack = [count] + list(fields)
            while stack:
                i = stack.pop()
                need(i < self.n_nodes, "operand index")
                op, a, b = self.nodes[i]
                if op in (VAR, PREP):
                    need(b == 0, "operand reads the next row")
                elif op in (ADD, SUB, MUL):
                    stack += [a, b]
                elif op == NEG:
                    stack.append(a)
                else:
                    need(op in (PUB, CONST), "operand is not row-local")


def verify_opening(root, mats, index, rows, path):
    """mats: list of (log_height, width) in commitment order; rows: their opened rows; path: sibling digests
    bottom-up.  Mixed-height rule: a matrix of height 2^h is hashed into the node layer of size 2^h."""
    lh = max(h for h, _ in mats)

    def rows_at(level):
        out = []
        for (h, _), r in zip(mats, rows):
            if h == level:
                out += r
        return out

    cur = m.hash_slice(rows_at(lh))
    for l in range(lh):
        sib = path[l]
        cur = m.compress(cur, sib) if ((index >> l) & 1) == 0 else m.compress(sib, cur)
        level = lh - l - 1
        if any(h == level for h, _ in mats):
            cur = m.compress(cur, m.hash_slice(rows_at(level)))
    return cur == list(root)


def verify(params, airs, proof_words):
    """params = (log_blowup, log_final_poly_len, num_queries, commit_pow_bits, query_pow_bits);
    airs: dicts with program, log_height, width, n_pvs, pvs[, prep_commit].  Raises Reject, returns True."""
    b, lfp, n_queries, cpow, qpow = params
    need(0 <= lfp <= 8 and b >= 1, "parameters")
    pr = [int(x) for x in proof_words]
    need(all(x < P for x in pr), "non-canonical word")
    progs = [Program(a["program"], a["width"]) for a in airs]
    for a, pg in zip(airs, progs):
        need(pg.n_pvs == a["n_pvs"] == len(a["pvs"]), "public values")
        need(pg.qd <= (1 << b), "degree")
        if pg.prep_width:
            need(a.get("prep_commit") is not None, "missing preprocessed commitment")
    hmax = max(a["log_height"] + b for a in airs)
    need(all(a["log_height"] >= lfp for a in airs), "a trace shorter than the final polynomial")
    n_layers = hmax - b - lfp
    lu = [i for i, pg in enumerate(progs) if pg.ints]
    pp = [i for i, pg in enumerate(progs) if pg.prep_width]
    # committed matrices in opening order: (air, kind, log_height(trace), width, n_pts)
    cm = [(i, "main", a["log_height"], a["width"], 2) for i, a in enumerate(airs)]
    cm += [(i, "prep", airs[i]["log_height"], progs[i].prep_width, 2) for i in pp]
    cm += [(i, "perm", airs[i]["log_height"], progs[i].perm_width, 2) for i in lu]
    cm += [(i, "quot%d" % j, a["log_height"], 4, 1) for i, a in enumerate(airs) for j in range(progs[i].qd)]
    pos = [0]

    def take(n):
        need(pos[0] + n <= len(pr), "truncated proof")
        out = pr[pos[0]:pos[0] + n]
        pos[0] += n
        return out

    hdr = take(4)
    need(hdr == [PROOF_MAGIC + (1 if lu else 0) + (2 if pp else 0), len(airs), hmax, n_layers], "header")
    root_main = take(8)
    root_perm, exposed = None, {}
    if lu:
        root_perm = take(8)
        for i in lu:
            exposed[i] = take(4)
    root_quot = take(8)
    opened = {}
    for (i, kind, lh, w, npts) in cm:
        opened[(i, kind)] = [[take(4) for _ in range(w)] for _ in range(npts)]
    fri_roots, fri_pows = [], []
    for _ in range(n_layers):
        fri_roots.append(take(8))
        fri_pows.append(take(1)[0])
    fin = [take(4) for _ in range(1 << lfp)]          # coefficients of the final polynomial (degree < 2^lfp)
    qpow_w = take(1)[0]

    # ---- transcript ----
    ch = m.Challenger()
    ch.observe([PROTO_TAG, len(airs), b, lfp, n_queries, cpow, qpow])
    for a, pg in zip(airs, progs):
        ch.observe([a["log_height"], a["width"], a["n_pvs"]])
        ch.observe(m.hash_slice([int(x) for x in a["program"]]))
        if pg.prep_width:
            ch.observe([int(x) for x in a["prep_commit"]])
        ch.observe([int(x) for x in a["pvs"]])
    ch.observe(root_main)
    chal = [0] * (4 * (1 + MAX_FIELDS))
    if lu:
        gamma, beta = ch.sample_ext(), ch.sample_ext()
        chal[0:4] = gamma
        cur = list(beta)
        for i in range(1, MAX_FIELDS + 1):
            chal[4 * i:4 * i + 4] = cur
            cur = e_mul(cur, beta)
        ch.observe(root_perm)
        tot = ZERO
        for i in lu:
            ch.observe(exposed[i])
            tot = e_add(tot, exposed[i])
        need(tot == ZERO, "bus sums do not cancel")
    alpha = ch.sample_ext()
    ch.observe(root_quot)
    zeta = ch.sample_ext()
    for key in [(i, kind) for (i, kind, _, _, _) in cm]:
        for pt in opened[key]:
            for v in pt:
                ch.observe(v)
    alpha_f = ch.sample_ext()

    # ---- constraints at zeta ----
    for i, (a, pg) in enumerate(zip(airs, progs)):
        lh = a["log_height"]
        n = 1 << lh
        zn = e_pow(zeta, n)
        zh = e_sub(zn, ONE)
        w_inv = m.inv(m.two_adic_generator(lh))
        is_first = e_mul(zh, e_inv(e_sub(zeta, ONE)))
        is_trans = e_sub(zeta, e_from(w_inv))
        is_last = e_mul(zh, e_inv(is_trans))
        loc, nxt = opened[(i, "main")]
        vals = []
        for (op, x, y) in pg.nodes:
            if op == VAR:
                vals.append(nxt[x] if y else loc[x])
            elif op == PREP:
                vals.append(opened[(i, "prep")][1 if y else 0][x])
            elif op == PERM:
                vals.append(opened[(i, "perm")][1 if y else 0][x])
            elif op == PUB:
                vals.append(e_from(int(a["pvs"][x])))
            elif op == CONST:
                vals.append(e_from(x))
            elif op == FIRST:
                vals.append(is_first)
            elif op == LAST:
                vals.append(is_last)
            elif op == TRANS:
                vals.append(is_trans)
            elif op == CHAL:
                vals.append(e_from(chal[x]))
            elif op == EXPOSED:
                vals.append(e_from(exposed[i][x]))
            elif op == ADD:
                vals.append(e_add(vals[x], vals[y]))
            elif op == SUB:
                vals.append(e_sub(vals[x], vals[y]))
            elif op == MUL:
                vals.append(e_mul(vals[x], vals[y]))
            else:
                vals.append(e_sub(ZERO, vals[x]))
        acc = ZERO
        for c in pg.cons:
            acc = e_add(e_mul(acc, alpha), vals[c])
        lhs = e_mul(acc, e_inv(zh))
        # the quotient lives on the first qd of the 2^b cosets s_j * H of the LDE domain; chunk j is opened as 4 base polynomials
        nch = pg.qd
        w_m = m.two_adic_generator(lh + b)
        shifts = [GEN * pow(w_m, m.bitrev(j, b), P) % P for j in range(nch)]
        rhs = ZERO
        for j in range(nch):
            zps = ONE
            for k in range(nch):
                if k == j:
                    continue
                num = e_sub(e_pow(e_scale(zeta, m.inv(shifts[k])), n), ONE)
                den = (pow(shifts[j] * m.inv(shifts[k]) % P, n, P) - 1) % P
                zps = e_mul(zps, e_scale(num, m.inv(den)))
            chunk = opened[(i, "quot%d" % j)][0]
            v = ZERO
            for k in range(4):
                basis = [0, 0, 0, 0]
                basis[k] = 1
                v = e_add(v, e_mul(basis, chunk[k]))
            rhs = e_add(rhs, e_mul(v, zps))
        need(lhs == rhs, "constraints do not hold at zeta (AIR %d)" % i)

    # ---- FRI transcript ----
    betas = []
    for l in range(n_layers):
        ch.observe(fri_roots[l])
        ch.observe([fri_pows[l]])
        need(ch.sample_bits(cpow) == 0, "commit-phase proof of work")
        betas.append(ch.sample_ext())
    ch.observe([w for c in fin for w in c])
    ch.observe([qpow_w])
    need(ch.sample_bits(qpow) == 0, "query proof of work")

    # ---- batches (one commitment each) ----
    batches = [(root_main, [c for c in cm if c[1] == "main"])]
    batches += [(airs[i]["prep_commit"], [c for c in cm if c[0] == i and c[1] == "prep"]) for i in pp]
    if lu:
        batches.append((root_perm, [c for c in cm if c[1] == "perm"]))
    batches.append((root_quot, [c for c in cm if c[1].startswith("quot")]))

    for _ in range(n_queries):
        idx = ch.sample_bits(hmax)
        ro, num_reduced = {}, {}
        for root, mats in batches:
            bh = max(lh + b for (_, _, lh, _, _) in mats)
            rows = [take(w) for (_, _, _, w, _) in mats]
            path = [take(8) for _ in range(bh)]
            need(verify_opening(root, [(lh + b, w) for (_, _, lh, w, _) in mats], idx >> (hmax - bh), rows, path), "input opening")
            for (i, kind, lh, w, npts), row in zip(mats, rows):
                h = lh + b
                ih = idx >> (hmax - h)
                x = GEN * pow(m.two_adic_generator(h), m.bitrev(ih, h), P) % P
                apow, cur = [], ONE
                for _k in range(w):
                    apow.append(cur)
                    cur = e_mul(cur, alpha_f)
                rrow = ZERO
                for k in range(w):
                    rrow = e_add(rrow, e_scale(apow[k], row[k]))
                for pt in range(npts):
                    z = zeta if pt == 0 else e_scale(zeta, m.two_adic_generator(lh))
                    ry = ZERO
                    for k in range(w):
                        ry = e_add(ry, e_mul(apow[k], opened[(i, kind)][pt][k]))
                    off = e_pow(alpha_f, num_reduced.get(h, 0))
                    u = e_mul(e_mul(e_sub(ry, rrow), e_inv(e_sub(z, e_from(x)))), off)
                    ro[h] = e_add(ro.get(h, ZERO), u)
                    num_reduced[h] = num_reduced.get(h, 0) + w
        ev = ro.get(hmax, ZERO)
        for l in range(n_layers):
            log_len = hmax - l
            il = idx >> l
            sib = take(4)
            path = [take(8) for _ in range(log_len - 1)]
            pair = [ev, sib] if (il & 1) == 0 else [sib, ev]
            need(verify_opening(fri_roots[l], [(log_len - 1, 8)], il >> 1, [pair[0] + pair[1]], path), "FRI opening")
            xx = pow(m.two_adic_generator(log_len), m.bitrev(il >> 1, log_len - 1), P)
            # fold: e0 + (beta - x) * (e1 - e0) / (-2x)
            c = (-m.inv(2 * xx % P)) % P
            d = e_scale(e_sub(pair[1], pair[0]), c)
            ev = e_add(pair[0], e_mul(e_sub(betas[l], e_from(xx)), d))
            if (log_len - 1) in ro:
                ev = e_add(ev, e_mul(e_mul(betas[l], betas[l]), ro[log_len - 1]))
        # the folded value is the final polynomial at this query's point of the last domain (size 2^(b + lfp))
        xf = pow(m.two_adic_generator(b + lfp), m.bitrev(idx >> n_layers, b + lfp), P) if lfp else 0
        want = ZERO
        for c in reversed(fin):
            want = e_add(e_scale(want, xf), c)
        need(ev == want, "final value")
    need(pos[0] == len(pr), "trailing proof words")
    return True
