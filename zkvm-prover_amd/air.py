"""AIR description for libzkhip: a symbolic-expression builder that serialises to the constraint
bytecode consumed by zkhip_keygen / zkhip_verify (format: DESIGN.md "AIR bytecode"), plus the
synthetic AIRs BASELINE.json's configs are measured on (a Fibonacci AIR and a
~300-column degree-3 AIR standing in for the 42 OpenVM chips of the chunk circuit,
AGENTS.md:183-185) with their trace generators (numpy for tests, torch for full-size benches).

Host-side product code: no arithmetic of the proving path lives here, only the description of
WHAT is proven and the synthetic witness data.
"""
import numpy as np

P = 2013265921
AIR_MAGIC = 0x31414B5A
OP_VAR, OP_PUB, OP_CONST, OP_FIRST, OP_LAST, OP_TRANS, OP_ADD, OP_SUB, OP_MUL, OP_NEG = range(10)
# leaves of the after-challenge (LogUp) phase: a base column of the permutation matrix, a coordinate
# of the interaction challenges, a coordinate of the AIR's exposed cumulative sum
OP_PERM, OP_CHAL, OP_EXPOSED = 10, 11, 12
OP_PREP = 13                              # a cell of the AIR's preprocessed trace (committed at keygen)
CACHED_MAGIC = 0x43414B5A                 # section [CACHED_MAGIC, cached_width]: leading main columns committed on their own
PREP_MAGIC = 0x50504B5A                   # section [PREP_MAGIC, prep_width] after the constraints
LOGUP_MAGIC = 0x554C4B5A
LOGUP_MAX_FIELDS = 32                     # challenge vector = gamma, beta^1 .. beta^32 (4 coordinates each)
N_CHAL = 4 * (1 + LOGUP_MAX_FIELDS)
EXT_W = 11                                # x^4 = 11


class Expr:
    __slots__ = ("b", "idx", "deg")

    def __init__(self, b, idx, deg):
        self.b, self.idx, self.deg = b, idx, deg

    def _lift(self, o):
        return o if isinstance(o, Expr) else self.b.const(o)

    def __add__(self, o):
        o = self._lift(o)
        return self.b._node(OP_ADD, self.idx, o.idx, max(self.deg, o.deg))

    __radd__ = __add__

    def __sub__(self, o):
        o = self._lift(o)
        return self.b._node(OP_SUB, self.idx, o.idx, max(self.deg, o.deg))

    def __rsub__(self, o):
        return self._lift(o) - self

    def __mul__(self, o):
        o = self._lift(o)
        return self.b._node(OP_MUL, self.idx, o.idx, self.deg + o.deg)

    __rmul__ = __mul__

    def __neg__(self):
        return self.b._node(OP_NEG, self.idx, 0, self.deg)


class AirBuilder:
    """Mirrors the shape of p3-air's AirBuilder: main-trace variables with rotation 0/1, public
    values, is_first_row / is_last_row / is_transition selectors, assert_zero."""

    def __init__(self, width, n_pvs=0, prep_width=0, cached_width=0):
        """cached_width: the first cached_width main columns form a CACHED main partition (OpenVM-v1 `cached_mains`): they
        are committed in a tree of their own (e.g. the program ROM, whose commitment is reused), the rest joins the common
        main commitment.  Constraints address the main trace as before."""
        self.width, self.n_pvs, self.prep_width, self.cached_width = width, n_pvs, prep_width, cached_width
        assert 0 <= cached_width < width
        self.nodes, self.cons, self._cache = [], [], {}

    def _node(self, op, a=0, b=0, deg=0):
        key = (op, a, b)
        if key not in self._cache:
            self.nodes.append(key)
            self._cache[key] = Expr(self, len(self.nodes) - 1, deg)
        return self._cache[key]

    def var(self, col, rot=0):
        assert 0 <= col < self.width and rot in (0, 1)
        return self._node(OP_VAR, col, rot, 1)

    def next(self, col):
        return self.var(col, 1)

    def prep(self, col, rot=0):
        """Cell of the preprocessed trace (fixed at keygen, like p3's `preprocessed` window)."""
        assert 0 <= col < self.prep_width and rot in (0, 1)
        return self._node(OP_PREP, col, rot, 1)

    def pub(self, i):
        assert 0 <= i < self.n_pvs
        return self._node(OP_PUB, i, 0, 0)

    def const(self, v):
        return self._node(OP_CONST, int(v) % P, 0, 0)

    def is_first_row(self):
        return self._node(OP_FIRST, 0, 0, 1)

    def is_last_row(self):
        return self._node(OP_LAST, 0, 0, 1)

    def is_transition(self):
        return self._node(OP_TRANS, 0, 0, 0)

    def assert_zero(self, e):
        self.cons.append(e.idx)

    # ---- bus interactions (LogUp) ---------------------------------------------------------------
    def perm(self, col, rot=0):
        return self._node(OP_PERM, col, rot, 1)

    def chal(self, i):
        return self._node(OP_CHAL, i, 0, 0)

    def exposed(self, i):
        return self._node(OP_EXPOSED, i, 0, 0)

    def _row_local(self, idx):
        """True if node idx is an expression of the CURRENT row only (main / preprocessed cells with rotation 0,
        public values, constants, + - * neg): what a bus message may be built from."""
        stack, seen = [idx], set()
        while stack:
            i = stack.pop()
            if i in seen:
                continue
            seen.add(i)
            op, a, b = self.nodes[i]
            if op in (OP_VAR, OP_PREP):
                if b != 0:
                    return False
            elif op in (OP_PUB, OP_CONST):
                pass
            elif op in (OP_ADD, OP_SUB, OP_MUL):
                stack += [a, b]
            elif op == OP_NEG:
                stack.append(a)
            else:
                return False
        return True

    def push_interaction(self, bus, fields, count, kind):
        """fields / count: expressions of the current row (e.g. `b.var(0) + 256 * b.var(1)`, `is_valid * x`),
        like the bus messages of OpenVM chips.  kind 'send' adds count/denominator, 'receive' subtracts it.
        Degree budget: 1 + deg(field) and deg(count) must stay <= 2^log_blowup + 1."""
        assert kind in ("send", "receive") and 0 <= bus < (1 << 20) and 1 <= len(fields) <= LOGUP_MAX_FIELDS
        fields = [f if isinstance(f, Expr) else self.const(f) for f in fields]
        count = count if isinstance(count, Expr) else self.const(count)
        for e in fields + [count]:
            assert self._row_local(e.idx), "interaction operands must be expressions of the current row"
        if not hasattr(self, "interactions"):
            self.interactions = []
        self.interactions.append((bus, 0 if kind == "send" else 1, count, fields))

    # extension-field expressions: 4 base coordinates (None = zero), x^4 = 11
    def _ext_mul(self, a, b):
        out = [None] * 4
        for i in range(4):
            for j in range(4):
                if a[i] is None or b[j] is None:
                    continue
                t = a[i] * b[j]
                if i + j >= 4:
                    t = t * EXT_W
                m = (i + j) % 4
                out[m] = t if out[m] is None else out[m] + t
        return out

    def _ext_add(self, a, b):
        return [x if y is None else (y if x is None else x + y) for x, y in zip(a, b)]

    def finalize_interactions(self):
        """Appends the LogUp constraints (as base-field constraints on the coordinates of the extension values) for
        the interactions pushed so far.  Interactions are packed greedily into GROUPS as far as the constraint-degree
        budget allows (like OpenVM's interaction chunking): group g owns base columns 4g..4g+3 of the permutation
        matrix and holds phi_g = sum_{j in g} sign_j*count_j / den_j, constrained by
            phi_g * prod_j den_j = sum_j sign_j*count_j * prod_{k != j} den_k;
        the running sum of all phi_g owns the last 4 columns.  With degree-1 message fields and a degree budget of 3
        two interactions share a column group: half the permutation columns to extend, hash and open."""
        ints = getattr(self, "interactions", [])
        if not ints or getattr(self, "_logup_done", False):
            return
        self._logup_done = True
        budget = getattr(self, "max_constraint_degree", 3)
        dens, dds = [], []
        for (bus, sign, count, fields) in ints:
            # denominator coordinates d_k = gamma_k + [k==0](bus+1) + sum_i beta^(i+1)_k * f_i
            d = []
            for k in range(4):
                e = self.chal(k)
                if k == 0:
                    e = e + (bus + 1)
                for i, f in enumerate(fields):
                    e = e + self.chal(4 * (i + 1) + k) * f
                d.append(e)
            dens.append(d)
            dds.append(max([f.deg for f in fields] + [0]))
        groups = []
        for j in range(len(ints)):
            placed = False
            if groups and len(groups[-1]) < 4:
                g = groups[-1] + [j]
                dsum = sum(dds[k] for k in g)
                if 1 + dsum <= budget and all(ints[k][2].deg + dsum - dds[k] <= budget for k in g):
                    groups[-1] = g
                    placed = True
            if not placed:
                groups.append([j])
        self.interaction_groups = [0] * len(ints)
        for gi, g in enumerate(groups):
            for j in g:
                self.interaction_groups[j] = gi
        n_grp = len(groups)
        for gi, g in enumerate(groups):
            phi = [self.perm(4 * gi + k) for k in range(4)]
            lhs = phi
            for j in g:
                lhs = self._ext_mul(lhs, dens[j])
            rhs = [None] * 4
            for j in g:
                bus, sign, count, fields = ints[j]
                term = [count if sign == 0 else -count, None, None, None]
                for k in g:
                    if k != j:
                        term = self._ext_mul(term, dens[k])
                rhs = self._ext_add(rhs, term)
            for m in range(4):
                e = lhs[m] if rhs[m] is None else lhs[m] - rhs[m]
                self.assert_zero(e)
        s_loc = [self.perm(4 * n_grp + k) for k in range(4)]
        s_nxt = [self.perm(4 * n_grp + k, 1) for k in range(4)]
        for k in range(4):
            row_sum, nxt_sum = None, None
            for gi in range(n_grp):
                row_sum = self.perm(4 * gi + k) if row_sum is None else row_sum + self.perm(4 * gi + k)
                nxt_sum = self.perm(4 * gi + k, 1) if nxt_sum is None else nxt_sum + self.perm(4 * gi + k, 1)
            self.when_first_row(s_loc[k] - row_sum)
            self.when_transition(s_nxt[k] - s_loc[k] - nxt_sum)
            self.when_last_row(s_loc[k] - self.exposed(k))

    def when_first_row(self, e):
        self.assert_zero(self.is_first_row() * e)

    def when_last_row(self, e):
        self.assert_zero(self.is_last_row() * e)

    def when_transition(self, e):
        self.assert_zero(self.is_transition() * e)

    def max_degree(self):
        deg = {}
        for i, (op, a, b) in enumerate(self.nodes):
            if op in (OP_VAR, OP_PERM, OP_PREP, OP_FIRST, OP_LAST):
                deg[i] = 1
            elif op in (OP_PUB, OP_CONST, OP_TRANS, OP_CHAL, OP_EXPOSED):
                deg[i] = 0
            elif op in (OP_ADD, OP_SUB):
                deg[i] = max(deg[a], deg[b])
            elif op == OP_MUL:
                deg[i] = deg[a] + deg[b]
            else:
                deg[i] = deg[a]
        return max((deg[c] for c in self.cons), default=0)

    def program(self):
        self.finalize_interactions()
        words = [AIR_MAGIC, len(self.nodes), len(self.cons), self.n_pvs]
        for n in self.nodes:
            words.extend(n)
        words.extend(self.cons)
        if self.prep_width:
            words += [PREP_MAGIC, self.prep_width]
        if self.cached_width:
            words += [CACHED_MAGIC, self.cached_width]
        ints = getattr(self, "interactions", [])
        if ints:
            # trailing section: [LOGUP_MAGIC, n_int, {bus, sign, count node, n_fields, field nodes, group}]
            # (groups are numbered 0.. in order; interaction j adds its term to permutation column group `group`)
            words += [LOGUP_MAGIC, len(ints)]
            for j, (bus, sign, count, fields) in enumerate(ints):
                words += [bus, sign, count.idx, len(fields)] + [f.idx for f in fields] + [self.interaction_groups[j]]
        return np.array(words, dtype=np.uint32)


def with_cached_width(program, cached_width):
    """The same AIR program with its first `cached_width` main columns declared a cached main partition (inserted after the
    constraints / the preprocessed section, before the interactions)."""
    w = [int(x) for x in program]
    pos = 4 + 3 * w[1] + w[2]
    if pos + 2 <= len(w) and w[pos] == PREP_MAGIC:
        pos += 2
    assert not (pos + 2 <= len(w) and w[pos] == CACHED_MAGIC), "program already has a cached partition"
    return np.array(w[:pos] + [CACHED_MAGIC, cached_width] + w[pos:], dtype=np.uint32)


def quotient_chunks(program):
    """Number of quotient chunk matrices of an AIR program: next_pow2(max(constraint degree, 2) - 1), the rule of the
    reference's engine (its stored v1 proofs have 1 or 4 chunks per AIR at blow-up 4).  Degrees in units of the trace length:
    trace / permutation / preprocessed cells, is_first, is_last 1; is_transition, constants, public values, challenges 0."""
    w = [int(x) for x in program]
    n_nodes, n_cons = w[1], w[2]
    deg = []
    for i in range(n_nodes):
        op, a, b = w[4 + 3 * i: 7 + 3 * i]
        if op in (OP_VAR, OP_PERM, OP_PREP, OP_FIRST, OP_LAST):
            deg.append(1)
        elif op in (OP_ADD, OP_SUB):
            deg.append(max(deg[a], deg[b]))
        elif op == OP_MUL:
            deg.append(deg[a] + deg[b])
        elif op == OP_NEG:
            deg.append(deg[a])
        else:
            deg.append(0)
    d = max([deg[c] for c in w[4 + 3 * n_nodes: 4 + 3 * n_nodes + n_cons]], default=0)
    qd = 1
    while qd + 1 < max(d, 2):
        qd *= 2
    return qd


# ---- reference evaluation of a program on a trace (numpy; host-side witness sanity check) ----
def check_trace(program, trace, pvs, prep=None):
    """trace: [width, n] canonical uint32 (prep: [prep_width, n] or None).  Returns the list of violated
    (constraint, row)."""
    w = [int(x) for x in program]
    n_nodes, n_cons = w[1], w[2]
    nodes = [tuple(w[4 + 3 * i: 7 + 3 * i]) for i in range(n_nodes)]
    cons = w[4 + 3 * n_nodes:4 + 3 * n_nodes + n_cons]
    n = trace.shape[1]
    t = trace.astype(np.int64)
    rows = np.arange(n)
    vals = [None] * n_nodes
    for i, (op, a, b) in enumerate(nodes):
        if op == OP_VAR:
            vals[i] = np.roll(t[a], -1) if b else t[a]
        elif op == OP_PUB:
            vals[i] = np.full(n, int(pvs[a]), dtype=np.int64)
        elif op == OP_CONST:
            vals[i] = np.full(n, a, dtype=np.int64)
        elif op == OP_FIRST:
            vals[i] = (rows == 0).astype(np.int64)
        elif op == OP_LAST:
            vals[i] = (rows == n - 1).astype(np.int64)
        elif op == OP_TRANS:
            vals[i] = (rows != n - 1).astype(np.int64)
        elif op == OP_ADD:
            vals[i] = None if vals[a] is None or vals[b] is None else (vals[a] + vals[b]) % P
        elif op == OP_SUB:
            vals[i] = None if vals[a] is None or vals[b] is None else (vals[a] - vals[b]) % P
        elif op == OP_MUL:
            vals[i] = None if vals[a] is None or vals[b] is None else (vals[a] * vals[b]) % P
        elif op == OP_NEG:
            vals[i] = None if vals[a] is None else (-vals[a]) % P
        elif op == OP_PREP and prep is not None:
            pt = np.asarray(prep).astype(np.int64)
            vals[i] = np.roll(pt[a], -1) if b else pt[a]
        else:
            vals[i] = None  # LogUp-phase leaf: only the main-trace constraints are checked here
    bad = []
    for k, c in enumerate(cons):
        if vals[c] is None:
            continue
        nz = np.nonzero(vals[c])[0]
        if len(nz):
            bad.append((k, int(nz[0])))
    return bad


# ---- Fibonacci AIR (SURVEY.md 8(d) cfg #1/#4 (B)) ---------------------------------------------
def fibonacci_air():
    """2 columns (a, b); pvs = (a0, b0, b_last)."""
    b = AirBuilder(2, 3)
    a0, b0 = b.var(0), b.var(1)
    b.when_first_row(a0 - b.pub(0))
    b.when_first_row(b0 - b.pub(1))
    b.when_transition(b.next(0) - b0)
    b.when_transition(b.next(1) - (a0 + b0))
    b.when_last_row(b0 - b.pub(2))
    return b


def fibonacci_trace(log_n, a0=0, b0=1):
    n = 1 << log_n
    tr = np.zeros((2, n), dtype=np.uint32)
    a, bb = a0 % P, b0 % P
    for i in range(n):
        tr[0, i], tr[1, i] = a, bb
        a, bb = bb, (a + bb) % P
    return tr, np.array([a0 % P, b0 % P, int(tr[1, n - 1])], dtype=np.uint32)


# ---- synthetic wide AIR (SURVEY.md 8(d) cfg #4 (A)) --------------------------------------------
class SyntheticAir:
    """`width` columns: `n_free` free (random) columns, the rest defined one constraint each from
    earlier columns, so a satisfying trace can be generated column by column in parallel over rows:
      local     : col_j(r)   = c1*x*y*z + c2*u*v + c3          (degree 3)
      transition: col_j(r+1) = c1*x(r)*y(r) + c2*u(r) + c3     (enforced on rows 0..n-2; row 0 = pv)
    plus booleanity constraints b*(b-1) on the first free columns and first-/last-row boundary
    constraints against public values.  Deterministic in `seed`."""

    def __init__(self, width=300, n_free=60, n_bool=16, n_boundary=8, seed=0, degree=3):
        """degree=5: the local definitions read c1*x*y*z*u*v + c2*u*v + c3 (constraint degree 5: four quotient chunks, what
        most chips of the reference's stored proofs have at blow-up 4)."""
        assert 4 <= n_free < width and n_bool <= n_free and degree in (3, 5)
        rng = np.random.default_rng(seed)
        self.width, self.n_free, self.n_bool, self.degree = width, n_free, n_bool, degree
        self.defs = []  # per derived column: (kind, cols..., coeffs...)
        n_trans = 0
        for j in range(n_free, width):
            kind = "trans" if (j % 4 == 3) else "local"
            if kind == "local":
                cols = rng.integers(0, j, size=5).tolist()  # any earlier column
            else:
                cols = rng.integers(0, n_free, size=3).tolist()  # free columns only -> row-parallel
                n_trans += 1
            coef = rng.integers(1, P, size=3).tolist()
            self.defs.append((kind, cols, coef))
        self.trans_cols = [n_free + i for i, d in enumerate(self.defs) if d[0] == "trans"]
        self.boundary_cols = rng.choice(np.arange(n_free, width), size=n_boundary, replace=False).tolist()
        self.n_pvs = len(self.trans_cols) + 2 * n_boundary
        b = AirBuilder(width, self.n_pvs)
        for c in range(n_bool):
            x = b.var(c)
            b.assert_zero(x * (x - 1))
        pv = 0
        for j, (kind, cols, coef) in zip(range(n_free, width), self.defs):
            if kind == "local":
                x, y, z, u, v = (b.var(c) for c in cols)
                top = x * y * z * u * v if degree == 5 else x * y * z
                b.assert_zero(b.var(j) - (b.const(coef[0]) * top + b.const(coef[1]) * u * v + coef[2]))
            else:
                x, y, u = (b.var(c) for c in cols)
                b.when_transition(b.next(j) - (b.const(coef[0]) * x * y + b.const(coef[1]) * u + coef[2]))
                b.when_first_row(b.var(j) - b.pub(pv))
                pv += 1
        for c in self.boundary_cols:
            b.when_first_row(b.var(c) - b.pub(pv))
            b.when_last_row(b.var(c) - b.pub(pv + 1))
            pv += 2
        assert pv == self.n_pvs
        self.builder = b

    def program(self):
        return self.builder.program()

    def gen_trace(self, log_n, seed=0, xp=None, device=None):
        """Returns (trace [width, n] canonical, pvs).  xp='torch' generates on `device` with int64
        tensor ops (full-size benches); default numpy."""
        n = 1 << log_n
        if xp == "torch":
            import torch

            g = torch.Generator(device=device)
            g.manual_seed(seed)
            tr = torch.empty((self.width, n), dtype=torch.int64, device=device)
            tr[:self.n_free] = torch.randint(0, P, (self.n_free, n), generator=g, device=device, dtype=torch.int64)
            tr[:self.n_bool] %= 2
            first = torch.randint(0, P, (len(self.trans_cols),), generator=g, device=device, dtype=torch.int64)
            roll = lambda v: torch.roll(v, 1)
        else:
            rng = np.random.default_rng(seed)
            tr = np.empty((self.width, n), dtype=np.int64)
            tr[:self.n_free] = rng.integers(0, P, size=(self.n_free, n), dtype=np.int64)
            tr[:self.n_bool] %= 2
            first = rng.integers(0, P, size=len(self.trans_cols), dtype=np.int64)
            roll = lambda v: np.roll(v, 1)
        ti = 0
        for j, (kind, cols, coef) in zip(range(self.n_free, self.width), self.defs):
            if kind == "local":
                x, y, z, u, v = (tr[c] for c in cols)
                top = x * y % P * z % P
                if self.degree == 5:
                    top = top * u % P * v % P
                tr[j] = (coef[0] * top % P + coef[1] * (u * v % P) % P + coef[2]) % P
            else:
                x, y, u = (tr[c] for c in cols)
                nxt = (coef[0] * (x * y % P) % P + coef[1] * u % P + coef[2]) % P
                col = roll(nxt)
                col[0] = first[ti]
                tr[j] = col
                ti += 1
        pvs = [int(first[i]) for i in range(len(self.trans_cols))]
        for c in self.boundary_cols:
            pvs += [int(tr[c][0]), int(tr[c][n - 1])]
        pvs = np.array(pvs, dtype=np.uint32)
        if xp == "torch":
            return tr.to(torch.int32), pvs  # values < 2^31: bit pattern == canonical u32
        return tr.astype(np.uint32), pvs


# ---- lookup demo AIRs for the LogUp phase ----------------------------------------------------------
def lookup_sender_air(width=3, bus=7):
    """Every row sends the pair (col0, col1) once on `bus`; col2 is unconstrained filler."""
    b = AirBuilder(width, 0)
    b.assert_zero(b.var(0) * (b.var(0) - 1) * 0 + b.var(2) - b.var(2))  # keeps a main-trace constraint around
    b.push_interaction(bus, [b.var(0), b.var(1)], 1, "send")
    return b


def lookup_table_air(bus=7):
    """Table of (key, value) pairs with a multiplicity column: receives each pair `mult` times."""
    b = AirBuilder(3, 0)
    b.push_interaction(bus, [b.var(0), b.var(1)], b.var(2), "receive")
    return b


def lookup_traces(log_n_sender, log_n_table, seed=0, sender_width=3):
    """A satisfying pair of traces: the table holds distinct (key, key^2+1) rows, the sender picks rows
    of the table at random, the multiplicity column counts the picks."""
    rng = np.random.default_rng(seed)
    nt, ns = 1 << log_n_table, 1 << log_n_sender
    keys = rng.permutation(1 << 20)[:nt].astype(np.int64)
    vals = (keys * keys + 1) % P
    pick = rng.integers(0, nt, size=ns)
    sender = np.zeros((sender_width, ns), dtype=np.uint32)
    sender[0], sender[1] = keys[pick], vals[pick]
    sender[2:] = rng.integers(0, P, size=(sender_width - 2, ns))
    table = np.zeros((3, nt), dtype=np.uint32)
    table[0], table[1] = keys, vals
    table[2] = np.bincount(pick, minlength=nt)
    return sender, table


def limb_air(bus=13):
    """Compound bus messages: the AIR sends the 16-bit value lo + 256*hi (two byte columns) gated by a boolean
    `is_valid` column, and receives the recomposed value from a third column -- the shape of OpenVM's
    byte-decomposition chips.  Columns: lo, hi, value, is_valid."""
    b = AirBuilder(4, 0)
    lo, hi, val, ok = b.var(0), b.var(1), b.var(2), b.var(3)
    b.assert_zero(ok * (ok - 1))
    b.assert_zero(ok * (lo + hi * 256 - val))
    b.push_interaction(bus, [lo + hi * 256, ok * val], ok, "send")
    b.push_interaction(bus, [val, val * ok], ok * 1, "receive")
    return b


def limb_trace(log_n, seed=0):
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    lo, hi = rng.integers(0, 256, n), rng.integers(0, 256, n)
    ok = rng.integers(0, 2, n)
    val = np.where(ok == 1, lo + 256 * hi, rng.integers(0, P, n))
    return np.stack([lo, hi, val, ok]).astype(np.uint32)


def bus_mix_air(width=6):
    """Exercises every interaction operand form in one AIR: a constant field, a public-value field, a
    constant multiplicity and a column multiplicity; the AIR sends and receives the same messages, so
    it balances on its own.  pvs = (tag,)."""
    b = AirBuilder(width, 1)
    b.assert_zero(b.var(0) * b.var(1) - b.var(2))
    b.push_interaction(3, [b.var(0), 5, b.pub(0)], b.var(3), "send")
    b.push_interaction(3, [b.var(0), 5, b.pub(0)], b.var(3), "receive")
    b.push_interaction(9, [b.var(1)], 1, "send")
    b.push_interaction(9, [b.var(4)], 1, "receive")
    b.push_interaction(11, [b.var(0), b.var(1), b.var(2), b.var(3), b.var(4), b.var(5), 7, b.pub(0)], 2, "send")
    b.push_interaction(11, [b.var(0), b.var(1), b.var(2), b.var(3), b.var(4), b.var(5), 7, b.pub(0)], b.const(2), "receive")
    return b


def bus_mix_trace(log_n, seed=0, width=6):
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    t = rng.integers(0, P, size=(width, n)).astype(np.int64)
    t[2] = t[0] * t[1] % P
    t[4] = t[1][rng.permutation(n)]          # bus 9: column 4 is a permutation of column 1
    return t.astype(np.uint32), np.array([rng.integers(0, P)], dtype=np.uint32)


# ---- preprocessed-trace demo: a range-check table whose keys are fixed at keygen ---------------
def range_table_air(bus=5):
    """Preprocessed column 0 = 0..N-1 (the table); main column 0 = multiplicity.  Receives (key) mult times."""
    b = AirBuilder(1, 0, prep_width=1)
    # the table is what it claims to be: first key 0, keys increase by one (checked on the preprocessed column,
    # which also exercises next-row access to it)
    b.when_first_row(b.prep(0))
    b.when_transition(b.prep(0, 1) - b.prep(0) - 1)
    b.push_interaction(bus, [b.prep(0)], b.var(0), "receive")
    return b


def range_user_air(width=4, bus=5):
    """Every row sends its column 0 (must lie in the table's range) once; column 1 = column 0 squared."""
    b = AirBuilder(width, 0)
    b.assert_zero(b.var(0) * b.var(0) - b.var(1))
    b.push_interaction(bus, [b.var(0)], 1, "send")
    return b


def range_traces(log_n_user, log_n_table, seed=0, user_width=4):
    rng = np.random.default_rng(seed)
    nu, nt = 1 << log_n_user, 1 << log_n_table
    user = rng.integers(0, P, size=(user_width, nu)).astype(np.int64)
    user[0] = rng.integers(0, nt, size=nu)
    user[1] = user[0] * user[0] % P
    prep = np.arange(nt, dtype=np.uint32).reshape(1, nt)
    mult = np.bincount(user[0], minlength=nt).astype(np.uint32).reshape(1, nt)
    return user.astype(np.uint32), mult, prep


# ---- three more periphery chips whose traces are generated on the device (csrc/tracegen_tables.hip) ----
def range_tuple_table_air(size_x=256, size_y=8192, bus=6):
    """OpenVM RangeTupleCheckerChip<2> (sizes [256, 8192] in crates/circuits/chunk-circuit/openvm.toml): preprocessed columns
    (x, y) enumerate all tuples, row x * size_y + y; the main trace is the multiplicity column.  Receives (x, y) mult times."""
    b = AirBuilder(1, 0, prep_width=2)
    b.when_first_row(b.prep(0))
    b.when_first_row(b.prep(1))
    # y counts up and wraps to 0 when x steps: (y' - y - 1) * y' = 0 and x' = x + [y' == 0]
    b.when_transition((b.prep(1, 1) - b.prep(1) - 1) * b.prep(1, 1))
    b.when_transition((b.prep(0, 1) - b.prep(0)) * (b.prep(0, 1) - b.prep(0) - 1))
    b.push_interaction(bus, [b.prep(0), b.prep(1)], b.var(0), "receive")
    return b


def range_tuple_prep(size_x=256, size_y=8192):
    i = np.arange(size_x * size_y, dtype=np.uint32)
    return np.stack([i // size_y, i % size_y]).astype(np.uint32)


def var_range_table_air(bus=7):
    """OpenVM VariableRangeCheckerChip: one table for every range check x < 2^bits with bits <= max_bits.  Preprocessed columns
    (value, bits), row 2^bits - 1 + value (var_range_prep); the main trace is the multiplicity column.  Receives (value, bits)."""
    b = AirBuilder(1, 0, prep_width=2)
    b.push_interaction(bus, [b.prep(0), b.prep(1)], b.var(0), "receive")
    return b


def var_range_prep(max_bits):
    """[2, 2^(max_bits + 1)]: rows (value, bits) for bits = 0 .. max_bits in order, then one row (0, max_bits + 1) that pads the
    height to a power of two (as the reference chip's table does)."""
    r = np.arange(1 << (max_bits + 1), dtype=np.uint64) + 1
    bits = np.floor(np.log2(r)).astype(np.uint32)
    value = (r - (np.uint64(1) << bits.astype(np.uint64))).astype(np.uint32)
    return np.stack([value, bits]).astype(np.uint32)


def var_range_user_air(bus=7):
    """Every row sends (col0, col1): col0 < 2^col1; col2 = col0 * col1 keeps a main constraint around."""
    b = AirBuilder(3, 0)
    b.assert_zero(b.var(0) * b.var(1) - b.var(2))
    b.push_interaction(bus, [b.var(0), b.var(1)], 1, "send")
    return b


CASTF_WIDTH = 6


def castf_air(bus=7):
    """The core of OpenVM's native CASTF chip (native `CastFCoreAir`: a field element to four little-endian limbs of 8, 8, 8 and 6
    bits): columns x | limb[4] | is_valid; x = sum limb_i 2^(8 i), every limb range-checked through the variable range checker
    ((limb, 8) three times, (limb_3, 6)) -- below 2^30 the decomposition is unique."""
    b = AirBuilder(CASTF_WIDTH, 0)
    x, limb, ok = b.var(0), [b.var(1 + i) for i in range(4)], b.var(5)
    b.assert_zero(ok * (ok - 1))
    b.assert_zero(ok * (limb[0] + limb[1] * 256 + limb[2] * 65536 + limb[3] * 16777216 - x))
    for i in range(4):
        b.push_interaction(bus, [limb[i], 8 if i < 3 else 6], ok, "send")
    return b


def range_tuple_user_air(bus=6):
    """Every row sends the tuple (col0, col1) once; col2 = col0 * col1 keeps a main constraint around."""
    b = AirBuilder(3, 0)
    b.assert_zero(b.var(0) * b.var(1) - b.var(2))
    b.push_interaction(bus, [b.var(0), b.var(1)], 1, "send")
    return b


def bitwise_lookup_air(bits=8, bus=9):
    """OpenVM BitwiseOperationLookupChip<bits>: preprocessed (x, y, x ^ y) over all pairs, row (x << bits) + y; main trace =
    two multiplicity columns.  Receives (x, y, 0, 0) for range requests and (x, y, x ^ y, 1) for XOR requests."""
    b = AirBuilder(2, 0, prep_width=3)
    b.push_interaction(bus, [b.prep(0), b.prep(1), 0, 0], b.var(0), "receive")
    b.push_interaction(bus, [b.prep(0), b.prep(1), b.prep(2), 1], b.var(1), "receive")
    return b


def bitwise_lookup_prep(bits=8):
    i = np.arange(1 << (2 * bits), dtype=np.uint32)
    x, y = i >> bits, i & ((1 << bits) - 1)
    return np.stack([x, y, x ^ y]).astype(np.uint32)


def bitwise_user_air(bus=9):
    """Columns x, y, z, op (boolean): op = 1 rows claim z = x ^ y, op = 0 rows range-check (x, y) and hold z = 0."""
    b = AirBuilder(4, 0)
    op = b.var(3)
    b.assert_zero(op * (op - 1))
    b.assert_zero((1 - op) * b.var(2))
    b.push_interaction(bus, [b.var(0), b.var(1), b.var(2), op], 1, "send")
    return b


RV32_ALU_WIDTH = 18


def rv32_alu_core_air(bus=9):
    """The core of OpenVM's RV32 base ALU chip (rv32im `BaseAluCoreAir`): columns a[4] | b[4] | c[4] | is_add is_sub is_xor is_or
    is_and | is_valid, 8-bit limbs, a = result.  ADD / SUB are checked through their carry chains (carry_i = (b_i + c_i + carry_{i-1}
    - a_i) / 256 must be boolean; SUB with a and b exchanged); the bitwise opcodes -- and the range of every result limb -- through the
    bitwise-operation lookup: per limb the row sends (x, y, x ^ y, 1) with (x, y) = (b_i, c_i) for XOR / OR / AND and (a_i, a_i) for
    ADD / SUB, where x ^ y = a_i (XOR), 2 a_i - b_i - c_i (OR), b_i + c_i - 2 a_i (AND), 0 (ADD / SUB)."""
    b = AirBuilder(RV32_ALU_WIDTH, 0)
    a_, b_, c_ = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)], [b.var(8 + i) for i in range(4)]
    f_add, f_sub, f_xor, f_or, f_and = (b.var(12 + i) for i in range(5))
    ok = b.var(17)
    for f in (f_add, f_sub, f_xor, f_or, f_and, ok):
        b.assert_zero(f * (f - 1))
    b.assert_zero(f_add + f_sub + f_xor + f_or + f_and - ok)
    inv256 = pow(256, -1, P)
    carry_add, carry_sub = None, None
    for i in range(4):
        carry_add = (b_[i] + c_[i] - a_[i] + (carry_add if carry_add is not None else 0)) * inv256
        carry_sub = (a_[i] + c_[i] - b_[i] + (carry_sub if carry_sub is not None else 0)) * inv256
        b.assert_zero(f_add * (carry_add * (carry_add - 1)))
        b.assert_zero(f_sub * (carry_sub * (carry_sub - 1)))
    bitwise = f_xor + f_or + f_and
    for i in range(4):
        x = bitwise * b_[i] + (1 - bitwise) * a_[i]
        y = bitwise * c_[i] + (1 - bitwise) * a_[i]
        z = f_xor * a_[i] + f_or * (a_[i] * 2 - b_[i] - c_[i]) + f_and * (b_[i] + c_[i] - a_[i] * 2)
        b.push_interaction(bus, [x, y, z, 1], ok, "send")
    return b


RV32_LT_WIDTH = 18


def rv32_lt_core_air(bus=9):
    """The core of OpenVM's RV32 less-than chip (rv32im `LessThanCoreAir<4, 8>`: SLT / SLTU): columns b[4] | c[4] | cmp | is_slt
    is_sltu | b_msb_f c_msb_f | marker[4] | diff_val.  b_msb_f / c_msb_f are the top limbs as field elements (limb, or limb - 256
    for a negative SLT operand: the difference to the limb is 0 or 256, and 0 for SLTU); scanning from the top, limbs are equal
    until the marked one, where (c_i - b_i) * (2 cmp - 1) = diff_val; no marker means b = c and cmp = 0.  Two RANGE requests go to
    the bitwise lookup: (b_msb_f + 128 is_slt, c_msb_f + 128 is_slt) puts the signed limbs into [-128, 127] (the unsigned ones into
    [0, 255]), (diff_val - 1, 0) makes the marked difference 1..255, i.e. of the claimed sign."""
    b = AirBuilder(RV32_LT_WIDTH, 0)
    bl, cl = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)]
    cmp, slt, sltu, bm, cm = b.var(8), b.var(9), b.var(10), b.var(11), b.var(12)
    mk, dv = [b.var(13 + i) for i in range(4)], b.var(17)
    ok = slt + sltu
    for f in (slt, sltu, ok, cmp) + tuple(mk):
        b.assert_zero(f * (f - 1))
    for limb, f in ((bl[3], bm), (cl[3], cm)):
        d = limb - f
        b.assert_zero(d * (d - 256))
        b.assert_zero((1 - slt) * d)
    sign = cmp * 2 - 1
    prefix = None
    for i in (3, 2, 1, 0):
        diff = ((cm if i == 3 else cl[i]) - (bm if i == 3 else bl[i])) * sign
        prefix = mk[i] if prefix is None else prefix + mk[i]
        b.assert_zero((1 - prefix) * diff)
        b.assert_zero(mk[i] * (dv - diff))
    b.assert_zero(prefix * (prefix - 1))
    b.assert_zero((1 - prefix) * cmp)
    b.assert_zero((1 - ok) * prefix)   # padding rows carry no marker (and no request)
    b.push_interaction(bus, [bm + slt * 128, cm + slt * 128, 0, 0], ok, "send")
    b.push_interaction(bus, [dv - 1, 0, 0, 0], prefix, "send")
    return b


RV32_BRANCH_EQ_WIDTH = 17


def rv32_branch_eq_core_air():
    """The core of OpenVM's RV32 branch-equal chip (rv32im `BranchEqualCoreAir<4>`: BEQ / BNE): columns a[4] | b[4] | taken | imm |
    is_beq is_bne | diff_inv_marker[4] | pc_inc.  eq := taken for BEQ, 1 - taken for BNE; eq forces every limb pair equal, and
    eq + sum_i (a_i - b_i) marker_i = 1 forces a difference somewhere when eq = 0 (the prover puts the inverse of the first non-zero
    difference into its marker).  pc_inc = imm if the branch is taken, else 4.  No lookups: the operand limbs come range-checked
    from memory in OpenVM's design."""
    b = AirBuilder(RV32_BRANCH_EQ_WIDTH, 0)
    a_, b_ = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)]
    taken, imm, beq, bne = b.var(8), b.var(9), b.var(10), b.var(11)
    mk, inc = [b.var(12 + i) for i in range(4)], b.var(16)
    ok = beq + bne
    for f in (beq, bne, ok, taken):
        b.assert_zero(f * (f - 1))
    eq = taken * beq + (1 - taken) * bne
    total = eq
    for i in range(4):
        d = a_[i] - b_[i]
        b.assert_zero(eq * d)
        total = total + d * mk[i]
    b.assert_zero(ok * (total - 1))
    b.assert_zero(ok * (inc - taken * imm - (1 - taken) * 4))
    return b


RV32_BRANCH_LT_WIDTH = 23


def rv32_branch_lt_core_air(bus=9):
    """The core of OpenVM's RV32 branch-less-than chip (rv32im `BranchLessThanCoreAir<4, 8>`: BLT / BLTU / BGE / BGEU): columns
    a[4] | b[4] | cmp_lt | taken | imm | is_blt is_bltu is_bge is_bgeu | a_msb_f b_msb_f | marker[4] | diff_val | pc_inc.  cmp_lt is
    a < b by the comparison of rv32_lt_core_air() (signed for BLT / BGE); taken = cmp_lt for the less-than opcodes, 1 - cmp_lt for
    the greater-or-equal ones; pc_inc = imm if taken else 4.  The same two range requests as the less-than chip."""
    b = AirBuilder(RV32_BRANCH_LT_WIDTH, 0)
    al, bl = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)]
    cmp, taken, imm = b.var(8), b.var(9), b.var(10)
    blt, bltu, bge, bgeu = (b.var(11 + i) for i in range(4))
    am, bm = b.var(15), b.var(16)
    mk, dv, inc = [b.var(17 + i) for i in range(4)], b.var(21), b.var(22)
    signed = blt + bge
    ge = bge + bgeu
    ok = blt + bltu + bge + bgeu
    for f in (blt, bltu, bge, bgeu, ok, cmp, taken) + tuple(mk):
        b.assert_zero(f * (f - 1))
    b.assert_zero(taken - (cmp + ge - cmp * ge * 2))
    for limb, f in ((al[3], am), (bl[3], bm)):
        d = limb - f
        b.assert_zero(d * (d - 256))
        b.assert_zero((1 - signed) * d)
    sign = cmp * 2 - 1
    prefix = None
    for i in (3, 2, 1, 0):
        diff = ((bm if i == 3 else bl[i]) - (am if i == 3 else al[i])) * sign
        prefix = mk[i] if prefix is None else prefix + mk[i]
        b.assert_zero((1 - prefix) * diff)
        b.assert_zero(mk[i] * (dv - diff))
    b.assert_zero(prefix * (prefix - 1))
    b.assert_zero((1 - prefix) * cmp)
    b.assert_zero((1 - ok) * prefix)
    b.assert_zero(ok * (inc - taken * imm - (1 - taken) * 4))
    b.push_interaction(bus, [am + signed * 128, bm + signed * 128, 0, 0], ok, "send")
    b.push_interaction(bus, [dv - 1, 0, 0, 0], prefix, "send")
    return b


RV32_JAL_LUI_WIDTH = 9


def rv32_jal_lui_core_air(bus=9):
    """The core of OpenVM's RV32 JAL / LUI chip (rv32im `Rv32JalLuiCoreAir`): columns pc | imm | rd[4] | is_jal is_lui | pc_inc.
    LUI: rd = imm << 12 with the 20-bit immediate, i.e. rd_0 = 0 and rd_1 + 2^8 rd_2 + 2^16 rd_3 = 16 imm.  JAL: rd = pc + 4
    (composed from the limbs; the top limb below 2^6 as pc < 2^30), pc_inc = imm (the offset as a field element); LUI steps by 4.
    Range requests to the bitwise lookup: (rd_0, rd_1), (rd_2, rd_3) per row and (4 rd_3, 0) for JAL."""
    b = AirBuilder(RV32_JAL_LUI_WIDTH, 0)
    pc, imm = b.var(0), b.var(1)
    rd = [b.var(2 + i) for i in range(4)]
    jal, lui, inc = b.var(6), b.var(7), b.var(8)
    ok = jal + lui
    for f in (jal, lui, ok):
        b.assert_zero(f * (f - 1))
    b.assert_zero(lui * rd[0])
    b.assert_zero(lui * (rd[1] + rd[2] * 256 + rd[3] * 65536 - imm * 16))
    b.assert_zero(jal * (rd[0] + rd[1] * 256 + rd[2] * 65536 + rd[3] * 16777216 - pc - 4))
    b.assert_zero(ok * (inc - jal * imm - lui * 4))
    b.push_interaction(bus, [rd[0], rd[1], 0, 0], ok, "send")
    b.push_interaction(bus, [rd[2], rd[3], 0, 0], ok, "send")
    b.push_interaction(bus, [rd[3] * 4, 0, 0, 0], jal, "send")
    return b


RV32_AUIPC_WIDTH = 14


def rv32_auipc_core_air(bus=9):
    """The core of OpenVM's RV32 AUIPC chip (rv32im `Rv32AuipcCoreAir`): columns pc | imm | pc_limb[4] | imm_limb[3] | rd[4] | is_valid.
    rd = pc + (imm << 12) mod 2^32: the pc and 16 imm (= bytes 1..3 of imm << 12) are decomposed into 8-bit limbs, rd_0 = pc_limb_0,
    and limbs 1..3 add with boolean carries.  Five range requests: (pc_0, pc_1), (pc_2, 4 pc_3), (imm_0, imm_1), (imm_2, rd_1),
    (rd_2, rd_3); the second keeps the pc limbs below 2^30, so that their sum cannot be pc + p."""
    b = AirBuilder(RV32_AUIPC_WIDTH, 0)
    pc, imm = b.var(0), b.var(1)
    pl, il, rd = [b.var(2 + i) for i in range(4)], [b.var(6 + i) for i in range(3)], [b.var(9 + i) for i in range(4)]
    ok = b.var(13)
    b.assert_zero(ok * (ok - 1))
    b.assert_zero(ok * (pl[0] + pl[1] * 256 + pl[2] * 65536 + pl[3] * 16777216 - pc))
    b.assert_zero(ok * (il[0] + il[1] * 256 + il[2] * 65536 - imm * 16))
    b.assert_zero(ok * (rd[0] - pl[0]))
    inv256 = pow(256, -1, P)
    carry = None
    for i in range(1, 4):
        carry = (pl[i] + il[i - 1] - rd[i] + (carry if carry is not None else 0)) * inv256
        b.assert_zero(ok * (carry * (carry - 1)))
    for x, y in ((pl[0], pl[1]), (pl[2], pl[3] * 4), (il[0], il[1]), (il[2], rd[1]), (rd[2], rd[3])):
        b.push_interaction(bus, [x, y, 0, 0], ok, "send")
    return b


RV32_JALR_WIDTH = 20


def rv32_jalr_core_air(bus=9):
    """The core of OpenVM's RV32 JALR chip (rv32im `Rv32JalrCoreAir`): columns pc | imm | imm_limb[2] | imm_sign | rs1[4] | rd[4] |
    t[4] | lsb | to_pc | is_valid.  imm is the raw 12-bit immediate = imm_limb_0 + 2^8 imm_limb_1 with imm_sign its bit 11
    (imm_limb_1 - 8 imm_sign in [0, 8)); t = rs1 + sign-extended imm mod 2^32, limb by limb with boolean carries; to_pc = t with its
    lowest bit (lsb) cleared; rd = pc + 4.  Five range requests: (imm_0, 32 (imm_1 - 8 sign)), ((t_0 - lsb) / 2, t_1), (t_2, 4 t_3),
    (rd_0, rd_1), (rd_2, 4 rd_3)."""
    b = AirBuilder(RV32_JALR_WIDTH, 0)
    pc, imm = b.var(0), b.var(1)
    il, sign = [b.var(2), b.var(3)], b.var(4)
    rs, rd, t = [b.var(5 + i) for i in range(4)], [b.var(9 + i) for i in range(4)], [b.var(13 + i) for i in range(4)]
    lsb, to_pc, ok = b.var(17), b.var(18), b.var(19)
    for f in (ok, sign, lsb):
        b.assert_zero(f * (f - 1))
    b.assert_zero(ok * (il[0] + il[1] * 256 - imm))
    ext = [il[0], il[1] + sign * 240, sign * 255, sign * 255]
    inv256 = pow(256, -1, P)
    carry = None
    for i in range(4):
        carry = (rs[i] + ext[i] - t[i] + (carry if carry is not None else 0)) * inv256
        b.assert_zero(ok * (carry * (carry - 1)))
    b.assert_zero(ok * (t[0] + t[1] * 256 + t[2] * 65536 + t[3] * 16777216 - lsb - to_pc))
    b.assert_zero(ok * (rd[0] + rd[1] * 256 + rd[2] * 65536 + rd[3] * 16777216 - pc - 4))
    inv2 = pow(2, -1, P)
    b.push_interaction(bus, [il[0], (il[1] - sign * 8) * 32, 0, 0], ok, "send")
    b.push_interaction(bus, [(t[0] - lsb) * inv2, t[1], 0, 0], ok, "send")
    b.push_interaction(bus, [t[2], t[3] * 4, 0, 0], ok, "send")     # t below 2^30: to_pc cannot alias a target beyond the field
    b.push_interaction(bus, [rd[0], rd[1], 0, 0], ok, "send")
    b.push_interaction(bus, [rd[2], rd[3] * 4, 0, 0], ok, "send")
    return b


RV32_SHIFT_WIDTH = 32


def rv32_shift_core_air(bus=9):
    """The core of OpenVM's RV32 shift chip (rv32im `ShiftCoreAir<4, 8>`: SLL / SRL / SRA), 8-bit limbs.  Columns
    a[4] | b[4] | c0 | is_sll is_srl is_sra | bit_marker[8] | limb_marker[4] | carry[4] | sign | q | mult_left | mult_right.
    The shift amount is c0 mod 32 = bit_shift + 8 limb_shift (one-hot markers; q = c0 >> 5).  With mult = 2^bit_shift:
      left :  a[i] + 256 carry[k] = b[k] mult + carry[k - 1]           (k = i - limb_shift; a[i] = 0 below the limb shift)
      right:  a[i] mult + carry[k] = b[k] + 256 carry[k + 1]           (k = i + limb_shift; carry[4] = sign (mult - 1); a[i] = 255 sign above)
    mult_left / mult_right are columns (mult gated by the opcode) so that every constraint stays of degree 3.  Lookups (bitwise
    table): carry[i] < mult as the range pair (carry[i], mult - 1 - carry[i]); the result limbs pairwise; (q, 32 q); and for SRA the
    sign bit of b[3] as the XOR (b[3], 128, b[3] + 128 - 256 sign)."""
    b = AirBuilder(RV32_SHIFT_WIDTH, 0)
    a_, b_ = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)]
    c0, sll, srl, sra = b.var(8), b.var(9), b.var(10), b.var(11)
    bm, lm = [b.var(12 + i) for i in range(8)], [b.var(20 + i) for i in range(4)]
    cy = [b.var(24 + i) for i in range(4)]
    sign, q, ml, mr = b.var(28), b.var(29), b.var(30), b.var(31)
    ok = sll + srl + sra
    right = srl + sra
    for f in [sll, srl, sra, ok, sign] + bm + lm:
        b.assert_zero(f * (f - 1))
    sbm, slm, mult, bs, ls = bm[0], lm[0], bm[0], None, None
    for i in range(1, 8):
        sbm = sbm + bm[i]
        mult = mult + bm[i] * (1 << i)
        bs = bm[i] * i if bs is None else bs + bm[i] * i
    for j in range(1, 4):
        slm = slm + lm[j]
        ls = lm[j] * j if ls is None else ls + lm[j] * j
    b.assert_zero(sbm - ok)
    b.assert_zero(slm - ok)
    b.assert_zero(c0 - bs - ls * 8 - q * 32)
    b.assert_zero(ml - sll * mult)
    b.assert_zero(mr - right * mult)
    b.assert_zero(sign * (1 - sra))
    for j in range(4):
        for i in range(4):
            if i < j:
                b.assert_zero(lm[j] * (a_[i] * sll))
            else:
                k = i - j
                exp = b_[k] * ml - cy[k] * sll * 256
                if k > 0:
                    exp = exp + cy[k - 1] * sll
                b.assert_zero(lm[j] * (a_[i] * sll - exp))
            if i + j > 3:
                b.assert_zero(lm[j] * (a_[i] * right - sign * right * 255))
            else:
                k = i + j
                nxt = sign * (mr - right) if k == 3 else cy[k + 1] * right
                b.assert_zero(lm[j] * (a_[i] * mr - nxt * 256 - (b_[k] - cy[k]) * right))
    for i in range(4):
        b.push_interaction(bus, [cy[i], ml + mr - 1 - cy[i], 0, 0], ok, "send")
    b.push_interaction(bus, [a_[0], a_[1], 0, 0], ok, "send")
    b.push_interaction(bus, [a_[2], a_[3], 0, 0], ok, "send")
    b.push_interaction(bus, [q, q * 32, 0, 0], ok, "send")
    b.push_interaction(bus, [b_[3], 128, b_[3] + 128 - sign * 256, 1], sra, "send")
    return b


RV32_MUL_WIDTH = 13


def rv32_mul_core_air(bus=6):
    """The core of OpenVM's RV32 multiplication chip (rv32im `MultiplicationCoreAir`): columns a[4] | b[4] | c[4] | is_valid, 8-bit
    limbs, a = low 32 bits of b * c.  carry_i = (sum_{k<=i} b_k c_{i-k} + carry_{i-1} - a_i) / 256 is an EXPRESSION (degree 2); the row
    sends (a_i, carry_i) to the range-tuple checker (limb < 256, carry < 8192 with the reference's sizes), which is what makes a the
    product's limbs."""
    b = AirBuilder(RV32_MUL_WIDTH, 0)
    a_, b_, c_ = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)], [b.var(8 + i) for i in range(4)]
    ok = b.var(12)
    b.assert_zero(ok * (ok - 1))
    inv256 = pow(256, -1, P)
    carry = None
    for i in range(4):
        acc = carry if carry is not None else 0
        for k in range(i + 1):
            acc = b_[k] * c_[i - k] + acc
        carry = (acc - a_[i]) * inv256
        b.push_interaction(bus, [a_[i], carry], ok, "send")
    return b


RV32_MULH_WIDTH = 21


def rv32_mulh_core_air(tuple_bus=6, bitwise_bus=9):
    """The core of OpenVM's RV32 high-multiplication chip (rv32im `MulHCoreAir<4, 8>`: MULH / MULHSU / MULHU): columns a[4] | b[4] |
    c[4] | a_mul[4] | b_sign c_sign | is_mulh is_mulhsu is_mulhu.  The operands are extended to eight limbs by their sign limbs
    (255 sign) and multiplied schoolbook-wise: limbs 0..3 of the product are a_mul, limbs 4..7 the result a; all eight carries are
    expressions, and the row sends (limb, carry) to the range-tuple checker (carry < 2048) as the multiplication chip does.  Signs:
    MULHU has none, MULHSU only b's; 2 (x_3 - 128 sign) in byte range ties a sign to its operand's top limb (bitwise lookup)."""
    b = AirBuilder(RV32_MULH_WIDTH, 0)
    a_, b_, c_ = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)], [b.var(8 + i) for i in range(4)]
    am = [b.var(12 + i) for i in range(4)]
    bs, cs = b.var(16), b.var(17)
    mulh, mulhsu, mulhu = b.var(18), b.var(19), b.var(20)
    ok = mulh + mulhsu + mulhu
    for f in (mulh, mulhsu, mulhu, ok, bs, cs):
        b.assert_zero(f * (f - 1))
    b.assert_zero(mulhu * bs)
    b.assert_zero((mulhu + mulhsu) * cs)
    b_ext, c_ext = bs * 255, cs * 255
    inv256 = pow(256, -1, P)
    carry = None
    for i in range(4):
        acc = carry if carry is not None else 0
        for k in range(i + 1):
            acc = b_[k] * c_[i - k] + acc
        carry = (acc - am[i]) * inv256
        b.push_interaction(tuple_bus, [am[i], carry], ok, "send")
    for j in range(4):
        acc = carry
        for k in range(j + 1, 4):
            acc = b_[k] * c_[4 + j - k] + acc
        for k in range(j + 1):
            acc = b_[k] * c_ext + c_[k] * b_ext + acc
        carry = (acc - a_[j]) * inv256
        b.push_interaction(tuple_bus, [a_[j], carry], ok, "send")
    b.push_interaction(bitwise_bus, [(b_[3] - bs * 128) * 2, 0, 0, 0], mulh + mulhsu, "send")
    b.push_interaction(bitwise_bus, [(c_[3] - cs * 128) * 2, 0, 0, 0], mulh, "send")
    return b


RV32_DIVREM_WIDTH = 41


def rv32_divrem_core_air(tuple_bus=6, bitwise_bus=9):
    """The core of an RV32 division chip (the job of rv32im `DivRemCoreAir<4, 8>`: DIV / DIVU / REM / REMU), columns
    b[4] | c[4] | q[4] | r[4] | c_abs[4] | r_abs[4] | b_sign c_sign q_sign r_sign | k_c k_r | zero_divisor c_sum_inv | marker[4] | diff |
    is_div is_divu is_rem is_remu.  Three statements:
      * b = c q + r over the integers: the four values are sign-extended to eight limbs by their sign columns and
        c q + r - b is multiplied out limb by limb; each of the eight carries is an expression sent with a limb of q (low half) or
        of r (high half) to the range-tuple checker, which also makes those limbs bytes.  q_sign is free: the identity together
        with the bound below admits exactly one value of q, and in the overflow case DIV(-2^31, -1) that value is +2^31
        (limbs 0x80000000 with q_sign = 0), the result RISC-V prescribes;
      * |r| < |c| unless c = 0, and r has b's sign or is zero: c_abs, r_abs are the magnitudes (x + x_abs = 2^32 in two 16-bit
        halves with a carry bit when the sign is set, equal limbs otherwise), compared like the less-than chip (marker at the
        most significant differing limb, diff = c_abs_i - r_abs_i in 1..255); b_sign (1 - r_sign) r_i = 0;
      * c = 0 (zero_divisor, with c_sum_inv witnessing a non-zero limb sum otherwise) forces q = 0xffffffff; r = b then follows
        from the identity.
    b_sign / c_sign are tied to the top limbs through the bitwise lookup (signed opcodes), the unsigned opcodes have no signs."""
    b = AirBuilder(RV32_DIVREM_WIDTH, 0)
    bl, cl, ql, rl = ([b.var(4 * g + i) for i in range(4)] for g in range(4))
    ca, ra = [b.var(16 + i) for i in range(4)], [b.var(20 + i) for i in range(4)]
    b_sign, c_sign, q_sign, r_sign = (b.var(24 + i) for i in range(4))
    kc, kr, zd, cinv = b.var(28), b.var(29), b.var(30), b.var(31)
    mk, diff = [b.var(32 + i) for i in range(4)], b.var(36)
    div, divu, rem, remu = (b.var(37 + i) for i in range(4))
    ok = div + divu + rem + remu
    signed = div + rem
    for f in (div, divu, rem, remu, ok, b_sign, c_sign, q_sign, r_sign, kc, kr, zd) + tuple(mk):
        b.assert_zero(f * (f - 1))
    for f in (b_sign, c_sign, q_sign, r_sign):
        b.assert_zero((1 - signed) * f)
    b.assert_zero((1 - ok) * zd)
    # ---- c = 0
    csum = cl[0] + cl[1] + cl[2] + cl[3]
    for i in range(4):
        b.assert_zero(zd * cl[i])
        b.assert_zero(zd * (ql[i] - 255))
    b.assert_zero((ok - zd) * (csum * cinv - 1))
    # ---- b = c q + r
    b_ext, c_ext, q_ext, r_ext = b_sign * 255, c_sign * 255, q_sign * 255, r_sign * 255
    inv256 = pow(256, -1, P)
    carry = None
    for i in range(4):
        acc = carry if carry is not None else 0
        for k in range(i + 1):
            acc = cl[k] * ql[i - k] + acc
        carry = (acc + rl[i] - bl[i]) * inv256
        b.push_interaction(tuple_bus, [ql[i], carry], ok, "send")
    for j in range(4):
        acc = carry
        for k in range(j + 1, 4):
            acc = cl[k] * ql[4 + j - k] + acc
        for k in range(j + 1):
            acc = cl[k] * q_ext + ql[k] * c_ext + acc
        carry = (acc + r_ext - b_ext) * inv256
        b.push_interaction(tuple_bus, [rl[j], carry], ok, "send")
    # ---- magnitudes
    for x, xa, sg, k in ((cl, ca, c_sign, kc), (rl, ra, r_sign, kr)):
        for i in range(4):
            b.assert_zero((1 - sg) * (x[i] - xa[i]))
        b.assert_zero(sg * (x[0] + x[1] * 256 + xa[0] + xa[1] * 256 - k * 65536))
        b.assert_zero(sg * (x[2] + x[3] * 256 + xa[2] + xa[3] * 256 + k - 65536))
    b.assert_zero(r_sign * (1 - b_sign))
    for i in range(4):
        b.assert_zero(b_sign * (1 - r_sign) * rl[i])
    # ---- |r| < |c|
    prefix = None
    for i in (3, 2, 1, 0):
        d = ca[i] - ra[i]
        prefix = mk[i] if prefix is None else prefix + mk[i]
        b.assert_zero((1 - zd - prefix) * d)
        b.assert_zero(mk[i] * (diff - d))
    b.assert_zero(prefix - (ok - zd))
    b.push_interaction(bitwise_bus, [(bl[3] - b_sign * 128) * 2, (cl[3] - c_sign * 128) * 2, 0, 0], signed, "send")
    b.push_interaction(bitwise_bus, [ca[0], ca[1], 0, 0], ok, "send")
    b.push_interaction(bitwise_bus, [ca[2], ca[3], 0, 0], ok, "send")
    b.push_interaction(bitwise_bus, [ra[0], ra[1], 0, 0], ok, "send")
    b.push_interaction(bitwise_bus, [ra[2], ra[3], 0, 0], ok, "send")
    b.push_interaction(bitwise_bus, [diff - 1, 0, 0, 0], prefix, "send")
    return b


RV32_LOADSTORE_WIDTH = 33
# (kind, shift) of the 20 cases of the load/store chip, in flag order: LW, LHU x2, LBU x4, SW, SH x2, SB x4, LH x2, LB x4
RV32_LOADSTORE_CASES = ([("lw", 0), ("lhu", 0), ("lhu", 2)] + [("lbu", s) for s in range(4)] + [("sw", 0), ("sh", 0), ("sh", 2)] +
                        [("sb", s) for s in range(4)] + [("lh", 0), ("lh", 2)] + [("lb", s) for s in range(4)])


def rv32_loadstore_core_air(bus=9):
    """The cores of OpenVM's RV32 load/store chips in one AIR (rv32im `LoadStoreCoreAir<4>`: LW LHU LBU SW SH SB, and
    `LoadSignExtendCoreAir<4, 8>`: LH LB): columns read[4] | prev[4] | write[4] | case flag[20] | sign.  One flag per (opcode, byte
    offset inside the aligned word) in the order of RV32_LOADSTORE_CASES; read = the aligned memory word (loads) or the register
    (stores), prev = what the destination held, write = what it holds afterwards: each limb of write is the flag-selected limb of
    read / prev, 0, or 255 sign.  For LH / LB, sign is the top bit of the loaded value: 2 (top limb - 128 sign) goes to the
    bitwise lookup's range column."""
    b = AirBuilder(RV32_LOADSTORE_WIDTH, 0)
    rd_, pv, wr = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)], [b.var(8 + i) for i in range(4)]
    fl = [b.var(12 + i) for i in range(20)]
    sign = b.var(32)
    ok = fl[0]
    for f in fl[1:]:
        ok = ok + f
    for f in fl + [ok, sign]:
        b.assert_zero(f * (f - 1))
    signed = fl[14]
    for f in fl[15:]:
        signed = signed + f
    b.assert_zero(sign * (1 - signed))
    ext = sign * 255
    for i in range(4):
        acc = None
        for f, (kind, s) in zip(fl, RV32_LOADSTORE_CASES):
            if kind in ("lw", "sw"):
                t = rd_[i]
            elif kind in ("lhu", "lh"):
                t = rd_[s + i] if i < 2 else (ext if kind == "lh" else None)
            elif kind in ("lbu", "lb"):
                t = rd_[s] if i == 0 else (ext if kind == "lb" else None)
            elif kind == "sh":
                t = rd_[i - s] if s <= i < s + 2 else pv[i]
            else:
                t = rd_[0] if i == s else pv[i]
            if t is None:
                continue
            acc = f * t if acc is None else acc + f * t
        b.assert_zero(wr[i] - acc)
    top = None
    for f, (kind, s) in zip(fl[14:], RV32_LOADSTORE_CASES[14:]):
        t = f * rd_[s + 1 if kind == "lh" else s]
        top = t if top is None else top + t
    b.push_interaction(bus, [(top - sign * 128) * 2, 0, 0, 0], signed, "send")
    return b


MEMORY_ACCESS_WIDTH = 10


def memory_access_air(range_bus=5, memory_bus=1):
    """One row per access of a 16-bit memory cell (the access side of OpenVM's offline memory checking; memory_boundary_air() is
    the other side): columns as | ptr | prev_data | prev_ts | data | ts | is_read | is_valid | gap_lo | gap_hi.  A valid row
    receives (as, ptr, prev_data, prev_ts) and sends (as, ptr, data, ts) on the memory bus; a read leaves the value as it was;
    time moves forward: ts - prev_ts - 1 = gap_lo + 2^16 gap_hi, the limbs and the cell value range-checked to 16 bits and 8 gap_hi
    too: gap_hi < 2^13, the gap stays below 2^29 and cannot stand for a negative difference modulo p (timestamps below 2^29, as
    OpenVM's timestamp_max_bits; the generators refuse larger ones)."""
    b = AirBuilder(MEMORY_ACCESS_WIDTH, 0)
    as_, ptr, pd, pts, d, ts, rd, ok, lo, hi = (b.var(i) for i in range(10))
    b.assert_zero(ok * (ok - 1))
    b.assert_zero(rd * (rd - 1))
    b.assert_zero((1 - ok) * rd)
    b.assert_zero(rd * (d - pd))
    b.assert_zero(ok * (ts - pts - 1 - lo - hi * (1 << 16)))
    b.push_interaction(range_bus, [lo], ok, "send")
    b.push_interaction(range_bus, [hi], ok, "send")
    b.push_interaction(range_bus, [hi * 8], ok, "send")
    b.push_interaction(range_bus, [d], ok, "send")
    b.push_interaction(memory_bus, [as_, ptr, pd, pts], ok, "receive")
    b.push_interaction(memory_bus, [as_, ptr, d, ts], ok, "send")
    return b


PROGRAM_FIELDS = 9
PROGRAM_BUS = 8


def program_air(bus=PROGRAM_BUS):
    """OpenVM ProgramAir: the program (pc, opcode, operands a..g) is a CACHED main partition of width 9, the common main is the
    execution-frequency column; every instruction is received `frequency` times on the program bus.  Degree 2: one quotient chunk
    (the first AIR of the reference's stored proofs: cached width 9, common width 1, after-challenge width 8, one chunk)."""
    b = AirBuilder(PROGRAM_FIELDS + 1, 0, cached_width=PROGRAM_FIELDS)
    b.push_interaction(bus, [b.var(c) for c in range(PROGRAM_FIELDS)], b.var(PROGRAM_FIELDS), "receive")
    return b


def exec_frame_air(bus=PROGRAM_BUS):
    """One row per executed instruction: its nine program fields and is_valid; a valid row sends the instruction on the program
    bus (in OpenVM every instruction chip's adapter does; this stand-alone chip keeps the pair self-contained)."""
    b = AirBuilder(PROGRAM_FIELDS + 1, 0)
    ok = b.var(PROGRAM_FIELDS)
    b.assert_zero(ok * (ok - 1))
    b.push_interaction(bus, [b.var(c) for c in range(PROGRAM_FIELDS)], ok, "send")
    return b


MEMORY_BOUNDARY_WIDTH = 8


def memory_boundary_air(pointer_bits=27, range_bus=5, memory_bus=1):
    """OpenVM VolatileBoundaryChip: columns as, ptr, initial, final, final_ts, is_valid, gap_lo, gap_hi.  Valid rows come
    first and are strictly sorted by key = as * 2^pointer_bits + ptr: key' - key - 1 = gap_lo + 2^16 gap_hi with both limbs
    and 8 gap_hi sent to the range checker (the gap stays below 2^29: with address spaces below 4 and pointers below 2^27 the keys
    stay below 2^29 too, so the difference cannot wrap around p and two rows cannot carry one key); each valid row sends (as, ptr,
    initial, 0) and receives (as, ptr, final, final_ts) on the memory bus (the initial / final memory states of the
    offline-checking argument)."""
    b = AirBuilder(MEMORY_BOUNDARY_WIDTH, 0)
    as_, ptr, init, fin, ts, ok, lo, hi = (b.var(i) for i in range(8))
    ok_n = b.var(5, 1)
    b.assert_zero(ok * (ok - 1))
    b.when_transition(ok_n * (1 - ok))  # no valid row after an invalid one
    key = as_ * (1 << pointer_bits) + ptr
    key_n = b.var(0, 1) * (1 << pointer_bits) + b.var(1, 1)
    b.when_transition(ok_n * (key_n - key - 1 - lo - hi * (1 << 16)))
    b.push_interaction(range_bus, [lo], ok, "send")
    b.push_interaction(range_bus, [hi], ok, "send")
    b.push_interaction(range_bus, [hi * 8], ok, "send")
    b.push_interaction(memory_bus, [as_, ptr, init, 0], ok, "send")
    b.push_interaction(memory_bus, [as_, ptr, fin, ts], ok, "receive")
    return b


# ---- a chunk-circuit-shaped AIR set: many chips of different heights that talk over buses ------------
class ChipSet:
    """`n_chips` SyntheticAir chips (the reference's chunk circuit has 42 OpenVM chips, AGENTS.md:183-185) of mixed
    heights whose widths sum to ~`total_width`, each with
      * a bus of its own on which it sends and receives the compound message (c0 + 2*c1, c2) gated by a boolean column,
      * one message per row to a shared range table with PREPROCESSED keys (bus 5): the value of a boolean column,
    plus the range-table chip itself (multiplicities = how often each key was sent).  Deterministic in `seed`."""

    RANGE_BUS = 5

    def __init__(self, n_chips=42, log_max=16, log_min=6, total_width=300, seed=0, log_table=4):
        rng = np.random.default_rng(seed)
        w = rng.integers(6, 2 * total_width // n_chips - 5, size=n_chips)
        self.widths = [int(x) for x in w]
        # a few tall chips, a long tail of short ones (like CPU / memory chips vs. small ALU chips)
        hs = np.sort(rng.integers(log_min, log_max + 1, size=n_chips))[::-1].copy()
        hs[: max(1, n_chips // 10)] = log_max
        self.heights = [int(h) for h in rng.permutation(hs)]
        self.log_table = log_table
        self.chips = []
        for i, wi in enumerate(self.widths):
            sa = SyntheticAir(width=wi, n_free=max(4, wi // 3), n_bool=2, n_boundary=1, seed=seed * 1000 + i)
            b = sa.builder
            msg = [b.var(2) + b.var(3) * 2, b.var(sa.n_free)]
            b.push_interaction(100 + i, msg, b.var(0), "send")
            b.push_interaction(100 + i, msg, b.var(0), "receive")
            b.push_interaction(self.RANGE_BUS, [b.var(1)], 1, "send")
            self.chips.append(sa)
        self.table = range_table_air(self.RANGE_BUS)

    def gen(self, seed=0):
        """Returns the list of AIR dicts (program, shapes, trace, pvs[, prep]) with balanced buses."""
        airs, counts = [], np.zeros(1 << self.log_table, dtype=np.int64)
        for i, (sa, h) in enumerate(zip(self.chips, self.heights)):
            tr, pv = sa.gen_trace(h, seed=seed * 7919 + i)
            counts[:2] += np.bincount(tr[1].astype(np.int64), minlength=2)[:2]
            airs.append(dict(program=sa.program(), log_height=h, width=sa.width, n_pvs=len(pv), trace=tr, pvs=pv))
        nt = 1 << self.log_table
        airs.append(dict(program=self.table.program(), log_height=self.log_table, width=1, n_pvs=0,
                         trace=(counts % P).astype(np.uint32).reshape(1, nt), pvs=np.zeros(0, np.uint32),
                         prep=np.arange(nt, dtype=np.uint32).reshape(1, nt)))
        return airs


class ReferenceShapedSet:
    """An AIR set with the SHAPE of the chunk proof the reference stores (crates/verifier/testdata/proofs/chunk-proof-feynman.json,
    decoded by tests/refproof_v1.py; the shape is in tests/golden/ref_v1_vectors.json): 17 AIRs of 2^1 .. 2^21 rows, common main
    widths 1 .. 398, one cached main partition of width 9 (the program chip), two preprocessed traces (widths 1 and 2), and as
    many interactions per AIR as give the reference's after-challenge widths (two interactions share a permutation column group,
    so an after-challenge matrix of 4 * (g + 1) base columns takes g send / receive pairs).  The chips' CONSTRAINTS are
    synthetic (SyntheticAir with degree-5 definitions: four quotient chunks per AIR, as in the stored proof); the first AIR
    is program-chip-like (nine cached columns received on a bus with the frequency column as multiplicity, degree 2: one
    chunk) and the last a preprocessed tuple table (one chunk): 62 quotient matrices in all, the stored proof's count.
    What is real is every dimension a prover's cost depends on.  `shrink` subtracts from every log-height (tests)."""

    LOG_DEGREES = [17, 1, 6, 18, 19, 18, 13, 17, 19, 18, 21, 17, 18, 17, 19, 15, 18]
    MAIN_WIDTHS = [1, 5, 23, 12, 11, 13, 17, 398, 27, 38, 29, 12, 23, 27, 21, 6, 1]
    AFTER_CHALLENGE_WIDTHS = [8, 12, 16, 12, 12, 12, 12, 160, 44, 20, 20, 16, 16, 24, 24, 8, 8]
    CACHED = {0: 9}          # AIR index -> cached main width
    PREP = {1: 1, 16: 2}     # AIR index -> preprocessed width
    TABLE, TABLE_USER, TABLE_BUS = 16, 10, 6

    def __init__(self, shrink=0, seed=0, log_degrees=None):
        """log_degrees: the heights of another of the eight stored proofs (they are proofs of one circuit, 17 AIRs of the same widths,
        at different heights; tests/golden/ref_v1_vectors.json `shapes`)."""
        self.heights = [max(1, d - shrink) for d in (log_degrees or self.LOG_DEGREES)]
        self.heights[1] = 1
        self.chips = []
        for i, (w, acw) in enumerate(zip(self.MAIN_WIDTHS, self.AFTER_CHALLENGE_WIDTHS)):
            groups = acw // 4 - 1
            if i == self.TABLE:
                lh = self.heights[i]
                self.table_sizes = (1 << (lh // 2), 1 << (lh - lh // 2))
                self.chips.append(range_tuple_table_air(self.table_sizes[0], self.table_sizes[1], bus=self.TABLE_BUS))
                continue
            if i == 0:
                # program chip: the cached partition holds the program, the common column its execution frequencies
                b = AirBuilder(w + self.CACHED[0], 0, cached_width=self.CACHED[0])
                b.push_interaction(100, [b.var(c) for c in range(self.CACHED[0])], b.var(self.CACHED[0]), "receive")
                b.width_total = w + self.CACHED[0]
                self.chips.append(b)
                continue
            width = w + self.CACHED.get(i, 0)
            sa = SyntheticAir(width=width, n_free=max(4, width // 3), n_bool=2, n_boundary=1, seed=seed * 1000 + i, degree=5)
            b = sa.builder
            b.cached_width = self.CACHED.get(i, 0)
            if i in self.PREP:
                b.prep_width = self.PREP[i]
                for c in range(b.prep_width):
                    b.assert_zero(b.prep(c) * (b.prep(c) - 1))
            rng = np.random.default_rng(seed * 7 + i)
            n_pairs = groups - 1 if i == self.TABLE_USER else groups
            for k in range(n_pairs):
                c = [int(x) for x in rng.integers(0, width, size=3)]
                msg = [b.var(c[0]) + b.var(c[1]) * (k + 2), b.var(c[2])]
                b.push_interaction(100 + i, msg, b.var(0), "send")
                b.push_interaction(100 + i, msg, b.var(0), "receive")
            if i == self.TABLE_USER:
                b.push_interaction(self.TABLE_BUS, [b.var(0), b.var(1)], 1, "send")  # (bool, bool): always in the table
            self.chips.append(sa)

    def gen(self, seed=0):
        airs, counts = [], None
        for i, (chip, h) in enumerate(zip(self.chips, self.heights)):
            if i == self.TABLE:
                airs.append(None)
                continue
            if i == 0:
                # nothing sends on the program bus here: every frequency is zero (the bus balances trivially)
                tr = np.random.default_rng(seed * 7919).integers(0, P, size=(chip.width_total, 1 << h)).astype(np.uint32)
                tr[chip.width_total - 1] = 0
                airs.append(dict(program=chip.program(), log_height=h, width=chip.width_total, n_pvs=0, trace=tr, pvs=np.zeros(0, np.uint32)))
                continue
            tr, pv = chip.gen_trace(h, seed=seed * 7919 + i)
            a = dict(program=chip.program(), log_height=h, width=chip.width, n_pvs=len(pv), trace=tr, pvs=pv)
            if i in self.PREP:
                a["prep"] = (np.arange(self.PREP[i] << h, dtype=np.uint32).reshape(self.PREP[i], 1 << h) % 2).astype(np.uint32)
            if i == self.TABLE_USER:
                sx, sy = self.table_sizes
                counts = np.bincount(tr[0].astype(np.int64) * sy + tr[1].astype(np.int64), minlength=sx * sy)
            airs.append(a)
        sx, sy = self.table_sizes
        airs[self.TABLE] = dict(program=self.chips[self.TABLE].program(), log_height=self.heights[self.TABLE], width=1, n_pvs=0,
                                trace=(counts % P).astype(np.uint32).reshape(1, sx * sy), pvs=np.zeros(0, np.uint32),
                                prep=range_tuple_prep(sx, sy))
        return airs


def program_bus_air(bus=2):
    """A 12-field bus message, the width of OpenVM's program / execution buses (pc, opcode, operands a..g, ...):
    the AIR sends its whole row and receives it back, gated by a boolean."""
    b = AirBuilder(13, 0)
    ok = b.var(12)
    b.assert_zero(ok * (ok - 1))
    msg = [b.var(i) for i in range(11)] + [b.var(0) + b.var(1) * 3]
    b.push_interaction(bus, msg, ok, "send")
    b.push_interaction(bus, msg, ok, "receive")
    return b


def program_bus_trace(log_n, seed=0):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, P, size=(13, 1 << log_n)).astype(np.uint32)
    t[12] %= 2
    return t


# ---------------------------------------------------------------------------------------------------------------
# Poseidon2 AIR: one permutation per row (the shape of p3-poseidon2-air 0.4.3 as OpenVM instantiates it for BabyBear:
# width 16, S-box degree 7 with ONE committed register x^3 per S-box so that every constraint has degree <= 3;
# Cargo.lock: openvm-poseidon2-air / p3-poseidon2-air).  The reference stack uses this AIR wherever the VM hashes
# (memory Merkle tree, recursion transcript); here it is the first chip whose TRACE IS GENERATED ON THE DEVICE
# (zkhip_poseidon2_air_tracegen).
#
# Columns (298): inputs[16] | 4 x { sbox[16] (= (state+rc)^3), post[16] } | 13 x { sbox, post_sbox } | 4 x { sbox[16], post[16] }
# A full round constrains  sbox_i = (s_i + rc_i)^3  and  post = M_ext(sbox_i^2 * (s_i + rc_i));  a partial round constrains
# sbox = (s_0 + rc)^3, post_sbox = sbox^2 * (s_0 + rc), then applies the internal linear layer to EXPRESSIONS (lanes 1..15 stay
# uncommitted linear combinations until the next full round).  The initial external layer acts on the inputs as expressions.
POSEIDON2_AIR_WIDTH = 16 + 4 * 32 + 13 * 2 + 4 * 32


def poseidon2_round_constants():
    """The 141 canonical round constants (Grain LFSR for field=prime, x^7, n=31, t=16, R_F=8, R_P=13: SURVEY.md A.3;
    the same generator as csrc/gen_poseidon2_rc.py, which emits them in Montgomery form for the kernels)."""
    bits = []
    for v, n in ((1, 2), (0, 4), (31, 12), (16, 12), (8, 10), (13, 10)):
        bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]
    bits += [1] * 30

    def step():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        del bits[0]
        bits.append(nb)
        return nb

    def next_bit():
        while True:
            a, c = step(), step()
            if a:
                return c

    for _ in range(160):
        step()
    out = []
    while len(out) < 141:
        v = 0
        for _ in range(31):
            v = (v << 1) | next_bit()
        if v < P:
            out.append(v)
    return out


def poseidon2_internal_diag():
    i2 = pow(2, P - 2, P)
    v = [-2, 1, 2, i2, 3, 4, -i2, -3, -4, pow(i2, 8, P), pow(i2, 2, P), pow(i2, 3, P), pow(i2, 27, P),
         -pow(i2, 8, P), -pow(i2, 4, P), -pow(i2, 27, P)]
    return [x % P for x in v]


def _p2_external(s):
    """circ(2 M4, M4, M4, M4) with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on a list of 16 expressions."""
    out = [None] * 16
    for blk in range(0, 16, 4):
        x0, x1, x2, x3 = s[blk:blk + 4]
        t01, t23 = x0 + x1, x2 + x3
        t0123 = t01 + t23
        t01123, t01233 = t0123 + x1, t0123 + x3
        out[blk + 3] = t01233 + x0 * 2
        out[blk + 1] = t01123 + x2 * 2
        out[blk + 0] = t01123 + t01
        out[blk + 2] = t01233 + t23
    sums = [(out[k] + out[4 + k]) + (out[8 + k] + out[12 + k]) for k in range(4)]
    return [out[i] + sums[i % 4] for i in range(16)]


def poseidon2_air(bus=None, out_lanes=8):
    """AirBuilder of the Poseidon2 AIR above: 298 columns, 282 constraints of degree 3, no public values.
    With `bus` the chip serves compression requests like OpenVM's Poseidon2 periphery chip: one more column `mult`
    (column 298) and the interaction receive(bus, inputs[0..16] ++ outputs[0..8], mult) -- a 24-field message."""
    b = AirBuilder(POSEIDON2_AIR_WIDTH + (1 if bus is not None else 0), 0)
    rc, diag = poseidon2_round_constants(), poseidon2_internal_diag()
    col = 16
    state = _p2_external([b.var(i) for i in range(16)])

    def full_round(state, rcs, col):
        outs = []
        for i in range(16):
            y = state[i] + rcs[i]
            reg = b.var(col + i)
            b.assert_zero(reg - y * y * y)
            outs.append(reg * reg * y)
        lin = _p2_external(outs)
        post = [b.var(col + 16 + i) for i in range(16)]
        for i in range(16):
            b.assert_zero(post[i] - lin[i])
        return post, col + 32

    for r in range(4):
        state, col = full_round(state, rc[16 * r:16 * r + 16], col)
    for r in range(13):
        y = state[0] + rc[64 + r]
        reg, post = b.var(col), b.var(col + 1)
        b.assert_zero(reg - y * y * y)
        b.assert_zero(post - reg * reg * y)
        col += 2
        state = [post] + state[1:]
        total = state[0]
        for i in range(1, 16):
            total = total + state[i]
        state = [state[i] * diag[i] + total for i in range(16)]
    for r in range(4):
        state, col = full_round(state, rc[77 + 16 * r:77 + 16 * r + 16], col)
    assert col == POSEIDON2_AIR_WIDTH
    if bus is not None:
        msg = [b.var(i) for i in range(16)] + state[:out_lanes]   # 8 lanes serve compressions, 16 a sponge (duplex_air)
        b.push_interaction(bus, msg, b.var(POSEIDON2_AIR_WIDTH), "receive")
    return b


MMCS_PATH_WIDTH = 39


def mmcs_path_air(hash_bus, claims_bus):
    """In-circuit verification of mixed-height Merkle (MMCS) openings -- a piece of the recursion circuit (the aggregation nodes
    of the reference verify their children's openings this way): columns root[8] | parent[8] | a[8] | b[8] | bit | is_inj |
    is_first | is_last | is_real | idx | lvl.  A path is walked from the ROOT down, one row per compression:
      * every row claims parent = compress(a, b) on `hash_bus` (the Poseidon2 chip receives it, air.poseidon2_air(bus));
      * a sibling row (is_inj = 0) descends into a (bit = 0) or b (bit = 1): the next row's parent; idx' = 2 idx + bit, lvl' = lvl + 1;
      * an injection row (is_inj = 1, where shorter matrices join the tree: node = compress(inner, digest of their rows))
        descends into a and reports b on `claims_bus` as (root, lvl, idx, digest) -- the row digest of the matrices of height
        2^lvl at row idx;
      * the last row of a path reports the child it would descend into: the leaf digest, at (lvl, idx) = (tree height, index).
    is_first rows start from parent = root; root is carried down the path, so every claim names its commitment."""
    b = AirBuilder(MMCS_PATH_WIDTH, 0)
    root, par = [b.var(i) for i in range(8)], [b.var(8 + i) for i in range(8)]
    a_, b_ = [b.var(16 + i) for i in range(8)], [b.var(24 + i) for i in range(8)]
    bit, inj, first, last, real, idx, lvl = (b.var(32 + i) for i in range(7))
    n_root, n_par = [b.next(i) for i in range(8)], [b.next(8 + i) for i in range(8)]
    n_bit, n_inj, n_first, n_real, n_idx, n_lvl = b.next(32), b.next(33), b.next(34), b.next(36), b.next(37), b.next(38)
    for f in (bit, inj, first, last, real):
        b.assert_zero(f * (f - 1))
    b.assert_zero(inj * bit)
    b.assert_zero(first * (1 - real))
    b.assert_zero(last * (1 - real))
    b.assert_zero(last * inj)
    link = real - last                      # 1 where the next row continues this path
    b.when_first_row(real - first)
    b.when_last_row(link)
    b.when_transition(link * (1 - n_real))
    b.when_transition(link * n_first)
    b.when_transition(last * n_real * (1 - n_first))
    b.when_transition((1 - real) * n_real)
    for i in range(8):
        b.assert_zero(first * (par[i] - root[i]))
        b.when_transition(link * (n_root[i] - root[i]))
        b.when_transition(link * (n_par[i] - a_[i] - bit * (b_[i] - a_[i])))
    b.assert_zero(first * (idx - bit))
    b.assert_zero(first * (lvl - 1 + inj))
    b.when_transition(link * (n_idx - idx * (2 - n_inj) - n_bit))
    b.when_transition(link * (n_lvl - lvl - 1 + n_inj))
    b.push_interaction(hash_bus, a_ + b_ + par, real, "send")
    b.push_interaction(claims_bus, root + [lvl, idx] + b_, inj, "send")
    b.push_interaction(claims_bus, root + [lvl, idx] + [a_[i] + bit * (b_[i] - a_[i]) for i in range(8)], last, "send")
    return b


def mmcs_claims_air(claims_bus):
    """The receiving side of mmcs_path_air's claims, as a table: columns root[8] | lvl | idx | digest[8] | mult -- row = "the matrices
    of height 2^lvl committed under root have digest `digest` at row idx", received mult times.  (In a recursion circuit the
    sponge over the opened values produces these; as a table it lets the claims be stated and checked.)"""
    b = AirBuilder(19, 0)
    b.push_interaction(claims_bus, [b.var(i) for i in range(18)], b.var(18), "receive")
    return b


DUPLEX_WIDTH = 50


def duplex_air(hash_bus, io_bus):
    """The Fiat-Shamir transcript in-circuit (p3 `DuplexChallenger<F, Poseidon2, 16, 8>`, the challenger of the reference's STARK
    configuration) -- a third piece of the recursion circuit: one row per duplexing, columns st_in[16] | st_out[16] | f[8] | s[8] |
    seq | is_real.  f is the prefix of rate lanes the observed values overwrote before the permutation, the other lanes carry
    over from the previous row's st_out (zero in the first row); st_out = Poseidon2(st_in) is requested from the Poseidon2 chip
    over `hash_bus` (a 32-field message: air.poseidon2_air(bus, out_lanes=16)); s is the suffix of output lanes that were
    sampled afterwards (the challenger pops from the end).  Every observed value goes out on `io_bus` as (seq, lane, value, 0),
    every sampled one as (seq, lane, value, 1): whoever feeds or uses the transcript is bound to it there."""
    b = AirBuilder(DUPLEX_WIDTH, 0)
    st_in, st_out = [b.var(i) for i in range(16)], [b.var(16 + i) for i in range(16)]
    f, s_ = [b.var(32 + i) for i in range(8)], [b.var(40 + i) for i in range(8)]
    seq, real = b.var(48), b.var(49)
    n_in, n_f, n_seq, n_real = [b.next(i) for i in range(16)], [b.next(32 + i) for i in range(8)], b.next(48), b.next(49)
    for x in f + s_ + [real]:
        b.assert_zero(x * (x - 1))
    for j in range(8):
        b.assert_zero((1 - real) * f[j])
        b.assert_zero((1 - real) * s_[j])
    for j in range(7):
        b.assert_zero(f[j + 1] * (1 - f[j]))
        b.assert_zero(s_[j] * (1 - s_[j + 1]))
    b.when_transition((1 - real) * n_real)
    b.when_first_row(seq)
    b.when_transition(n_real * (n_seq - seq - 1))
    for j in range(16):
        if j < 8:
            b.when_first_row((1 - f[j]) * st_in[j])
            b.when_transition(n_real * (1 - n_f[j]) * (n_in[j] - st_out[j]))
        else:
            b.when_first_row(st_in[j])
            b.when_transition(n_real * (n_in[j] - st_out[j]))
    b.push_interaction(hash_bus, st_in + st_out, real, "send")
    for j in range(8):
        b.push_interaction(io_bus, [seq, j, st_in[j], 0], f[j], "send")
    for j in range(8):
        b.push_interaction(io_bus, [seq, j, st_out[j], 1], s_[j], "send")
    return b


def duplex_io_air(io_bus):
    """The other side of duplex_air's io bus as a table: columns seq | lane | value | kind | mult."""
    b = AirBuilder(5, 0)
    b.push_interaction(io_bus, [b.var(i) for i in range(4)], b.var(4), "receive")
    return b


FIELD_ARITH_WIDTH = 8
FIELD_EXT_WIDTH = 20


def _ext_mul_exprs(x, y):
    """coordinates of x * y in F[X] / (X^4 - 11) for lists of four expressions"""
    W = 11
    return [x[0] * y[0] + (x[1] * y[3] + x[2] * y[2] + x[3] * y[1]) * W,
            x[0] * y[1] + x[1] * y[0] + (x[2] * y[3] + x[3] * y[2]) * W,
            x[0] * y[2] + x[1] * y[1] + x[2] * y[0] + x[3] * y[3] * W,
            x[0] * y[3] + x[1] * y[2] + x[2] * y[1] + x[3] * y[0]]


def field_arith_air():
    """The core of OpenVM's NATIVE field-arithmetic chip (native `FieldArithmeticCoreAir`: the base-field ADD / SUB / MUL / DIV of the
    recursion programs the aggregation circuits run): columns a | b | c | is_add is_sub is_mul is_div | divisor_inv, a = b op c;
    division is a c = b with c divisor_inv = 1."""
    b = AirBuilder(FIELD_ARITH_WIDTH, 0)
    a_, b_, c_ = b.var(0), b.var(1), b.var(2)
    add, sub, mul, div, inv = b.var(3), b.var(4), b.var(5), b.var(6), b.var(7)
    ok = add + sub + mul + div
    for f in (add, sub, mul, div, ok):
        b.assert_zero(f * (f - 1))
    b.assert_zero(add * (a_ - b_ - c_))
    b.assert_zero(sub * (a_ - b_ + c_))
    b.assert_zero(mul * (a_ - b_ * c_))
    b.assert_zero(div * (b_ - a_ * c_))
    b.assert_zero(div * (c_ * inv - 1))
    return b


def field_ext_air():
    """The core of OpenVM's native field-extension chip (native `FieldExtensionCoreAir`: FE4ADD / FE4SUB / BBE4MUL / BBE4DIV over
    F[X] / (X^4 - 11)): columns x[4] | y[4] | z[4] | is_add is_sub is_mul is_div | divisor_inv[4], z = x op y; division is
    z = x * divisor_inv with y * divisor_inv = 1."""
    b = AirBuilder(FIELD_EXT_WIDTH, 0)
    x, y, z = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)], [b.var(8 + i) for i in range(4)]
    add, sub, mul, div = b.var(12), b.var(13), b.var(14), b.var(15)
    inv = [b.var(16 + i) for i in range(4)]
    ok = add + sub + mul + div
    for f in (add, sub, mul, div, ok):
        b.assert_zero(f * (f - 1))
    xy, xi, yi = _ext_mul_exprs(x, y), _ext_mul_exprs(x, inv), _ext_mul_exprs(y, inv)
    for i in range(4):
        b.assert_zero(add * (z[i] - x[i] - y[i]))
        b.assert_zero(sub * (z[i] - x[i] + y[i]))
        b.assert_zero(mul * (z[i] - xy[i]))
        b.assert_zero(div * (z[i] - xi[i]))
        b.assert_zero(div * (yi[i] - (1 if i == 0 else 0)))
    return b


FRI_FOLD_WIDTH = 19
DOMAIN_POINT_BITS = 26
DOMAIN_POINT_WIDTH = 2 + 2 * DOMAIN_POINT_BITS


def fri_fold_air(point_bus=None):
    """One FRI folding step per row, in-circuit (p3 `TwoAdicFriFolding::fold_row`, arity 2) -- another piece of the recursion
    circuit: columns e0[4] | e1[4] | beta[4] | x_inv | folded[4] | is_real | k.  e0, e1 are the sibling evaluations at x and -x,
    x_inv = 1 / x in the base field; folded = (e0 + e1) / 2 + beta (e0 - e1) x_inv / 2 in the quartic extension (X^4 = 11),
    stated as 2 folded = (e0 + e1) + x_inv (beta * (e0 - e1)), coordinate by coordinate (degree 3).  k is the pair's index in
    its layer; with `point_bus` the row sends (k, x_inv) there and domain_point_air() must hold that pair: x = g^bitrev(k)."""
    b = AirBuilder(FRI_FOLD_WIDTH, 0)
    e0, e1, beta = [b.var(i) for i in range(4)], [b.var(4 + i) for i in range(4)], [b.var(8 + i) for i in range(4)]
    xinv, folded, real, k = b.var(12), [b.var(13 + i) for i in range(4)], b.var(17), b.var(18)
    b.assert_zero(real * (real - 1))
    d = [e0[i] - e1[i] for i in range(4)]
    W = 11
    prod = [beta[0] * d[0] + (beta[1] * d[3] + beta[2] * d[2] + beta[3] * d[1]) * W,
            beta[0] * d[1] + beta[1] * d[0] + (beta[2] * d[3] + beta[3] * d[2]) * W,
            beta[0] * d[2] + beta[1] * d[1] + beta[2] * d[0] + beta[3] * d[3] * W,
            beta[0] * d[3] + beta[1] * d[2] + beta[2] * d[1] + beta[3] * d[0]]
    for i in range(4):
        b.assert_zero(folded[i] * 2 - e0[i] - e1[i] - xinv * prod[i])    # all-zero padding rows satisfy it as they are
    if point_bus is not None:
        b.push_interaction(point_bus, [k, xinv], real, "send")
    return b


def domain_point_inverse_roots():
    """W_j^-1 for j = 0 .. 25 with W_j the generator of the subgroup of order 2^(j + 2): bit j of a pair index k contributes W_j
    to x = g^bitrev(k) WHATEVER the layer's size is (g_{m+1}^(2^(m-1-j)) has order 2^(j+2) for every m > j)."""
    g27 = 0x1A427A41
    return [pow(pow(g27, 1 << (27 - (j + 2)), P), P - 2, P) for j in range(DOMAIN_POINT_BITS)]


def domain_point_air(point_bus):
    """The evaluation point of a FRI pair from its index, in-circuit: columns k | bit[26] | acc[26] | mult.  k = sum bit_j 2^j;
    acc_j = acc_(j-1) (1 + bit_j (W_j^-1 - 1)) runs through the bits (degree 2), acc_25 = x^-1 for x = g^bitrev(k) (a pair index has at most 26 bits: the largest layer has 2^27 points).  The row
    receives (k, x^-1) `mult` times from the rows of fri_fold_air(point_bus) that fold the pair k (of any layer: see
    domain_point_inverse_roots)."""
    b = AirBuilder(DOMAIN_POINT_WIDTH, 0)
    k = b.var(0)
    bit = [b.var(1 + j) for j in range(DOMAIN_POINT_BITS)]
    acc = [b.var(1 + DOMAIN_POINT_BITS + j) for j in range(DOMAIN_POINT_BITS)]
    mult = b.var(1 + 2 * DOMAIN_POINT_BITS)
    winv = domain_point_inverse_roots()
    total = None
    for j in range(DOMAIN_POINT_BITS):
        b.assert_zero(bit[j] * (bit[j] - 1))
        term = bit[j] * (1 << j)
        total = term if total is None else total + term
        factor = bit[j] * (winv[j] - 1) + 1
        b.assert_zero(acc[j] - (factor if j == 0 else acc[j - 1] * factor))
    b.assert_zero(k - total)
    b.push_interaction(point_bus, [k, acc[DOMAIN_POINT_BITS - 1]], mult, "receive")
    return b


def hasher_user_air(bus):
    """A chip that needs 2-to-1 compressions (a Merkle-path checker, say): columns left[8] | right[8] | out[8] | is_real;
    it sends (left, right, out) on `bus` when is_real = 1 and relies on the Poseidon2 chip to receive it, i.e. to
    have a row whose permutation maps left ++ right to out ++ (anything)."""
    b = AirBuilder(25, 0)
    real = b.var(24)
    b.assert_zero(real * (real - 1))
    b.push_interaction(bus, [b.var(i) for i in range(24)], real, "send")
    return b


def hasher_user_trace(log_n, left, right, out):
    """Trace of hasher_user_air: rows 0..len(left) are real requests (canonical uint32 arrays [n][8]), the rest padding."""
    n = len(left)
    t = np.zeros((25, 1 << log_n), dtype=np.uint32)
    t[0:8, :n], t[8:16, :n], t[16:24, :n] = np.asarray(left).T, np.asarray(right).T, np.asarray(out).T
    t[24, :n] = 1
    return t
