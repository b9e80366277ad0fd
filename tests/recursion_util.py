"""Shared helpers of the verifier-circuit tests: small child AIR sets with a chained state, the node circuit's traces
restated in numpy (the CPU twin of zkhip_recursion_tracegen: a gather of the wire values; the Poseidon2 chip through the
oracle's trace generator) and the node statement recomputed independently with the oracle's Poseidon2."""
import numpy as np

import oracle_lib as ora
from zkvm_prover_amd import air

NOPV = np.zeros(0, np.uint32)
P = ora.P


def counter_air():
    """1 column x, x' = x + 1; public values (first, last): a segment whose state is a counter."""
    b = air.AirBuilder(1, 2)
    x = b.var(0)
    b.when_first_row(x - b.pub(0))
    b.when_transition(b.next(0) - x - 1)
    b.when_last_row(x - b.pub(1))
    return b


def counter_segment(start, log_n=3, seed=0):
    """One 'segment' = a counter chip (state start -> start + 2^log_n - 1), a lookup pair (bus traffic) and a Fibonacci chip
    (unchained public values).  The NEXT segment starts where this one ends."""
    n = 1 << log_n
    tr = ((start + np.arange(n, dtype=np.int64)) % P).astype(np.uint32).reshape(1, n)
    pv = np.array([start % P, (start + n - 1) % P], np.uint32)
    s, t = air.lookup_traces(4, 3, seed=seed)
    ftr, fpv = air.fibonacci_trace(3, a0=seed, b0=1)
    return [dict(program=counter_air().program(), log_height=log_n, width=1, n_pvs=2, trace=tr, pvs=pv),
            dict(program=air.lookup_sender_air().program(), log_height=4, width=3, n_pvs=0, trace=s, pvs=NOPV),
            dict(program=air.fibonacci_air().program(), log_height=3, width=2, n_pvs=3, trace=ftr, pvs=fpv),
            dict(program=air.lookup_table_air().program(), log_height=3, width=3, n_pvs=0, trace=t, pvs=NOPV)]


COUNTER_STMT = dict(start=[(0, 0)], end=[(0, 1)])


def verifying(params, airs):
    out = []
    for a in airs:
        v = {k: a[k] for k in ("program", "log_height", "width", "n_pvs")}
        if a.get("prep") is not None:
            v["prep_commit"] = a["prep_commit"] if a.get("prep_commit") is not None else ora.prep_commit(params, a)
        out.append(v)
    return out


def node_traces(rc):
    """The three chips' traces (canonical) from the circuit's last witness: what zkhip_recursion_tracegen writes."""
    W = rc.wires()
    na = rc.airs()
    g, p2 = na[0], na[1]
    gt = np.zeros((28, 1 << g["log_height"]), np.uint32)
    for s in range(4):
        gt[4 * s:4 * s + 4] = W[g["prep"][s]].T
    # Horner rows (preprocessed column 18 = the row's flag, 19 + j = coordinate j of slot b is taken): the values between the steps in slots 4 .. 6,
    # t <- t * d + b[j] for j = 3, 2, 1 where taken (the last step, j = 0, ends in slot c: a wire), restated with Python integers
    rows = np.nonzero(g["prep"][18])[0]
    if len(rows):
        t = gt[0:4, rows].astype(object)
        al = gt[12:16, rows].astype(object)
        for i in range(3):
            j = 3 - i
            take = g["prep"][19 + j][rows] == 1
            prod = [(t[0] * al[0] + 11 * (t[1] * al[3] + t[2] * al[2] + t[3] * al[1])) % P,
                    (t[0] * al[1] + t[1] * al[0] + 11 * (t[2] * al[3] + t[3] * al[2])) % P,
                    (t[0] * al[2] + t[1] * al[1] + t[2] * al[0] + 11 * t[3] * al[3]) % P,
                    (t[0] * al[3] + t[1] * al[2] + t[2] * al[1] + t[3] * al[0]) % P]
            prod[0] = (prod[0] + gt[4 + j, rows].astype(object)) % P
            t = np.array([np.where(take, prod[k], t[k]) for k in range(4)], dtype=object)
            gt[16 + 4 * i:20 + 4 * i, rows] = t.astype(np.uint32)
    inp = np.zeros((rc.n_perms, 16), np.uint32)
    for j in range(4):
        inp[:, 4 * j:4 * j + 4] = W[p2["prep"][j][:rc.n_perms]]
    pt = ora.poseidon2_air_trace(inp, p2["log_height"])
    return na, [gt, pt, np.zeros((1, 1), np.uint32)]


def node_instance(rc, node_pvs):
    na, traces = node_traces(rc)
    return [dict(a, trace=t, pvs=v) for a, t, v in zip(na, traces, [NOPV, NOPV, node_pvs])]


def sponge(vals):
    st = np.zeros(16, np.uint32)
    vals = [int(v) for v in vals]
    for i in range(0, len(vals), 8):
        chunk = vals[i:i + 8]
        st[:len(chunk)] = chunk
        st = ora.permute(st)
    return st[:8].copy()


def compress(l, r):
    return ora.permute(np.concatenate([np.asarray(l, np.uint32), np.asarray(r, np.uint32)]))[:8].copy()


def leaf_accumulator(children_pvs):
    """acc = compress(... compress(compress(0, H(pvs_0)), H(pvs_1)) ...), H = the sponge over all public values of a child."""
    acc = np.zeros(8, np.uint32)
    for pvs in children_pvs:
        flat = [int(x) for p in pvs for x in p]
        acc = compress(acc, sponge(flat) if flat else np.zeros(8, np.uint32))
    return acc


def internal_accumulator(child_accs):
    acc = np.zeros(8, np.uint32)
    for a in child_accs:
        acc = compress(acc, a)
    return acc


def one_key_circuits(params, app_vk, stmt, arity_leaf=4, arity_internal=3, node_params=None):
    """The two circuits of ONE aggregation key (zkvm-prover_amd/aggregate.py one_key_circuits; include/zkhip_aggregation.hpp
    `AggregationProver::build_one_key` in C++): the leaf circuit for the app's key with the uniform public-value layout, the uniform
    internal circuit for the node AIR set, both padded to the smallest common heights."""
    from zkvm_prover_amd import aggregate

    leafs, internal, _ = aggregate.one_key_circuits(params, [app_vk], stmt, arity_leaf, arity_internal, node_params)
    leaf = leafs[0]
    assert leaf.log_heights() == internal.log_heights()
    for a, b in zip(leaf.airs(), internal.airs()):
        assert (a["program"] == b["program"]).all() and a["width"] == b["width"] and a["n_pvs"] == b["n_pvs"]
    return leaf, internal


def node_key_commits(params, node_airs):
    """The three preprocessed commitments of a node key (oracle) and their digest (zkhip_recursion_key_commit)."""
    import zkvm_prover_amd as z

    pcs = [np.asarray(ora.prep_commit(params, dict(a, pvs=NOPV, trace=None)), np.uint32) for a in node_airs]
    return pcs, z.key_commit(pcs)


# ---- a toy "guest flow" for the deferral circuits: segments whose public values are a (pc, memory root) state ----
def state_air():
    """1 column; 18 public values: start = (pc, memory root[8]), end = (pc, memory root[8]).  The column carries the start pc."""
    b = air.AirBuilder(1, 18)
    x = b.var(0)
    b.when_first_row(x - b.pub(0))
    b.when_transition(b.next(0) - x)
    return b


def state_segment(start, end, log_n=3):
    pv = np.array(list(start) + list(end), np.uint32)
    tr = np.full((1, 1 << log_n), int(start[0]), np.uint32)
    return [dict(program=state_air().program(), log_height=log_n, width=1, n_pvs=18, trace=tr, pvs=pv)]


STATE_STMT = dict(start=[(0, k) for k in range(9)], end=[(0, 9 + k) for k in range(9)])


def block(cells):
    return compress(cells, np.zeros(8, np.uint32))


def memory_root_with_public_values(pv_bytes, rng):
    """A final memory root in which the 32 public-value bytes open (address space 3, blocks 0 and 1 of the 2^28-leaf tree, as
    include/zkhip_vm_flow.hpp `check_public_values`): random siblings above the block pair.  Returns (root, cells[16], siblings[27][8])."""
    cells = np.array([pv_bytes[2 * j] | (pv_bytes[2 * j + 1] << 8) for j in range(16)], np.uint32)
    cur = compress(block(cells[:8]), block(cells[8:]))
    sibs = rng.integers(0, P, size=(27, 8), dtype=np.uint64).astype(np.uint32)
    idx = (3 << 26) >> 1
    for l in range(27):
        cur = compress(sibs[l], cur) if idx & 1 else compress(cur, sibs[l])
        idx >>= 1
    return cur, cells, sibs


def merkle_of_blocks(cells):
    """root of the subtree over consecutive memory blocks (8 cells each; a power of two of them)"""
    level = [block(cells[8 * i:8 * i + 8]) for i in range(len(cells) // 8)]
    while len(level) > 1:
        level = [compress(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
    return level[0]


def memory_root_with_public_values_and_region(pv_bytes, region_cells, region_base, rng):
    """A final memory root in which BOTH the 32 public-value bytes (address space 3, blocks 0 and 1) and a guest's 8 KiB deferral region
    (address space 2, the 512 blocks from byte address `region_base`: include/zkhip_vm_flow.hpp `check_deferral_region`) open: random
    siblings up to the two address spaces' subtrees, which are siblings one level below the root's right child.
    Returns (root, pv cells[16], pv siblings[27][8], region siblings[19][8], region index)."""
    region_cells = np.asarray(region_cells, np.uint32)
    assert region_cells.size == 4096
    cells = np.array([pv_bytes[2 * j] | (pv_bytes[2 * j + 1] << 8) for j in range(16)], np.uint32)
    pv_sibs = rng.integers(0, P, size=(27, 8), dtype=np.uint64).astype(np.uint32)
    reg_sibs = rng.integers(0, P, size=(19, 8), dtype=np.uint64).astype(np.uint32)
    region_index = ((2 << 26) | (region_base // 16)) >> 9
    cur, idx = merkle_of_blocks(region_cells), region_index
    for l in range(17):
        cur = compress(reg_sibs[l], cur) if idx & 1 else compress(cur, reg_sibs[l])
        idx >>= 1
    assert idx == 2
    a2 = cur
    cur, idx = compress(block(cells[:8]), block(cells[8:])), (3 << 26) >> 1
    for l in range(25):
        cur = compress(pv_sibs[l], cur) if idx & 1 else compress(cur, pv_sibs[l])
        idx >>= 1
    assert idx == 3
    a3 = cur
    s0 = pv_sibs[26]
    reg_sibs[17], reg_sibs[18], pv_sibs[25] = a3, s0, a2
    root = compress(s0, compress(a2, a3))
    return root, cells, pv_sibs, reg_sibs, region_index


def deferral_region_cells(claims):
    """The 4096 cells of a deferral region holding `claims` (each the five chunks of deferral_claim): word 0 = their number, claim k = the
    32 words from word 32 + 32 k -- three commitments (a field element per word), then the public values (two cells per word)."""
    words = np.zeros(2048, np.uint32)
    words[0] = len(claims)
    for k, (ic, exe, vm, c0, c1) in enumerate(claims):
        w = words[32 + 32 * k:64 + 32 * k]
        w[0:8], w[8:16], w[16:24] = ic, exe, vm
        pv = np.concatenate([c0, c1]).astype(np.uint32)
        w[24:32] = pv[0::2] | (pv[1::2] << 16)
    cells = np.zeros(4096, np.uint32)
    cells[0::2], cells[1::2] = words & 0xFFFF, words >> 16
    return cells


def deferral_claim(root_pvs, cells):
    """The five chunks a deferral node chains for a child root with statement `root_pvs` (50 words; a join's: 58): what a parent guest states about it."""
    root_pvs = np.asarray(root_pvs, np.uint32)
    app, pc_start, root0, lc = root_pvs[:8], root_pvs[8], root_pvs[9:17], root_pvs[34:42]
    input_commit = sponge(root_pvs)
    exe_commit = compress(root0, np.array([pc_start, 0, 0, 0, 0, 0, 0, 0], np.uint32))
    vm_commit = compress(app, lc)
    return [input_commit, exe_commit, vm_commit, np.asarray(cells[:8], np.uint32), np.asarray(cells[8:], np.uint32)]


def deferral_chain(acc, claims):
    for chunks in claims:
        for c in chunks:
            acc = compress(acc, c)
    return acc


def one_key_circuits_shapes(params, app_vks, stmt, arity_leaf=4, arity_internal=3):
    """ONE aggregation key over several SHAPES of one app (per-proof chip presence: a segment carries only the chips it used, the app has
    one segment key and one leaf circuit per shape): the leaf circuits state one app id -- the digest of the LAST (full) shape --, share
    one height set with the internal circuit, and the internal circuit takes the list of their commitments."""
    from zkvm_prover_amd import aggregate

    leafs, internal, app_id = aggregate.one_key_circuits(params, app_vks, stmt, arity_leaf, arity_internal)
    for l in leafs:
        assert l.log_heights() == internal.log_heights()
    return leafs, internal, app_id


class OracleAggregator:
    """zkvm-prover_amd/aggregate.py TreeAggregator with the ORACLE as the node prover (CPU: the circuits' witnesses run on the host anyway;
    only the node proofs need a prover): folds gathered proofs of one app key to a root under one aggregation key."""

    def __init__(self, params, app_vk, stmt=None):
        from zkvm_prover_amd import aggregate

        self.params = params
        leafs, self.internal, _ = aggregate.one_key_circuits(params, [app_vk], stmt)
        self.leaf = leafs[0]
        self.leaf_commits, self.leaf_commit = node_key_commits(params, self.leaf.airs())
        self.internal_commits, self.internal_commit = node_key_commits(params, self.internal.airs())

    def root_vk(self):
        return verifying(self.params, [dict(a, prep_commit=c) for a, c in zip(self.internal.airs(), self.internal_commits)])

    def prove_leaf(self, group):
        st, npv = self.leaf.witness([p for p, _ in group], [pv for _, pv in group])
        assert st == 0, self.leaf.last_error()
        return ora.stark_prove(self.params, node_instance(self.leaf, npv)).tobytes(), npv

    def prove_internal(self, group, children_are_leaves):
        commits = self.leaf_commits if children_are_leaves else self.internal_commits
        st, npv = self.internal.witness([p for p, _ in group], [[NOPV, NOPV, pv] for _, pv in group], prep_commits=[commits] * len(group),
                                        is_leaf=[1 if children_are_leaves else 0] * len(group), leaf_commit=self.leaf_commit, internal_commit=self.internal_commit)
        assert st == 0, self.internal.last_error()
        return ora.stark_prove(self.params, node_instance(self.internal, npv)).tobytes(), npv

    def aggregate(self, proofs, pvs):
        from zkvm_prover_amd import aggregate

        return aggregate.fold_tree(list(zip(proofs, pvs)), self.prove_leaf, self.prove_internal)
