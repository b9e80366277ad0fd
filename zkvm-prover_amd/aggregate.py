"""Aggregation of gathered proofs under ONE key, over the C ABI (the Python twin of include/zkhip_aggregation.hpp AggregationProver).

What it replaces in the reference: the loop that proves the chunks of a batch one after the other and hands their proofs on
(crates/integration/src/testers/batch.rs:97-107), and the SDK's leaf / internal aggregation provers behind `sdk.prove`
(crates/prover/src/prover/mod.rs:47-60 tree arity 4 / 3, :147-170 one aggregation key).  SURVEY.md 8(e): the proofs of the units that
were sharded over the GPUs are gathered on rank 0 and folded there: leaf nodes verify up to four gathered proofs, internal nodes up to
three node proofs, the root is a proof of the internal circuit whatever the number of proofs.

`fold_tree` is the tree's shape as plain Python (no GPU: tests/test_shard_gloo.py folds stub proofs with it); `TreeAggregator` is the
real thing on a device context."""
import time

import numpy as np

from . import _binding as z

NOPV = np.zeros(0, np.uint32)


def fold_tree(items, prove_leaf, prove_internal, arity_leaf=4, arity_internal=3):
    """Folds `items` (proofs of the app) to ONE root: leaf nodes over groups of `arity_leaf` items, then internal nodes over groups of
    `arity_internal` nodes until one is left -- and at least one internal level (the root is a proof of the internal circuit whatever
    the number of items: one aggregation key).  prove_leaf(group) / prove_internal(group, children_are_leaves) return a node.
    Returns (root, levels: list of lists of nodes)."""
    if not items:
        raise ValueError("aggregation: no proofs")
    level = [prove_leaf(items[i:i + arity_leaf]) for i in range(0, len(items), arity_leaf)]
    levels = [level]
    leaves = True
    while len(level) > 1 or leaves:
        level = [prove_internal(level[i:i + arity_internal], leaves) for i in range(0, len(level), arity_internal)]
        levels.append(level)
        leaves = False
    return level[0], levels


def one_key_circuits(params, app_vks, stmt, arity_leaf=4, arity_internal=3, node_params=None):
    """The circuits of ONE aggregation key: a leaf circuit per app shape (uniform public-value layout; several shapes state one app id,
    the digest of the last = full shape) and the uniform internal circuit, all padded to the smallest common heights (a fixed point:
    the internal circuit verifies proofs of its own height).  Returns (leaf circuits, internal circuit, app id or None)."""
    node_params = node_params or params
    app_id = z.vk_digest(params, app_vks[-1]) if len(app_vks) > 1 else None
    leafs = [z.RecursionCircuit(params, vk, arity_leaf, stmt=stmt, uniform=True, app_id=app_id) for vk in app_vks]
    child = [{k: a[k] for k in ("program", "log_height", "width", "n_pvs")} for a in leafs[0].airs()]
    H = [max(l.log_heights()[i] for l in leafs) for i in (0, 1)]
    while True:
        for c, h in zip(child, H + [0]):
            c["log_height"] = h
        internal = z.RecursionCircuit(node_params, child, arity_internal, stmt="uniform", min_log_height=H, n_leaf_shapes=len(leafs))
        hh = internal.log_heights()[:2]
        if hh == H:
            break
        H = hh
    for l in leafs:
        l.pad(*H)
    return leafs, internal, app_id


class TreeAggregator:
    """Folds proofs of ONE app key (params, verifying AIRs) on a device context; node proofs under `node_params` (default: the app's)."""

    def __init__(self, ctx, params, app_vk, stmt=None, node_params=None):
        t0 = time.perf_counter()
        self.ctx, self.params, self.node_params = ctx, tuple(params), tuple(node_params or params)
        leafs, self.internal, _ = one_key_circuits(self.params, [app_vk], stmt, node_params=self.node_params)
        self.leaf = leafs[0]
        t1 = time.perf_counter()
        self.leaf_pk = z.ProvingKey(ctx, self.node_params, self.leaf.airs())
        self.internal_pk = z.ProvingKey(ctx, self.node_params, self.internal.airs())
        self.leaf_commits = [self.leaf_pk.prep_commitment(i) for i in range(3)]
        self.internal_commits = [self.internal_pk.prep_commitment(i) for i in range(3)]
        self.leaf_commit, self.internal_commit = z.key_commit(self.leaf_commits), z.key_commit(self.internal_commits)
        self.build_s, self.keygen_s = t1 - t0, time.perf_counter() - t1

    def root_vk(self):
        """The aggregation key: the internal circuit's verifying AIRs (+ what a root states beneath it: leaf_commit, internal_commit)."""
        return self.internal_pk.verifying_airs()

    def _prove(self, circ, pk, npv):
        return pk.prove(circ.tracegen(self.ctx), [NOPV, NOPV, npv])

    def prove_leaf(self, group):
        st, npv = self.leaf.witness([p for p, _ in group], [pv for _, pv in group])
        if st != 0:
            raise z.ZkhipError("aggregation: " + self.leaf.last_error())
        return self._prove(self.leaf, self.leaf_pk, npv), npv

    def prove_internal(self, group, children_are_leaves):
        commits = self.leaf_commits if children_are_leaves else self.internal_commits
        st, npv = self.internal.witness([p for p, _ in group], [[NOPV, NOPV, pv] for _, pv in group], prep_commits=[commits] * len(group),
                                        is_leaf=[1 if children_are_leaves else 0] * len(group), leaf_commit=self.leaf_commit, internal_commit=self.internal_commit)
        if st != 0:
            raise z.ZkhipError("aggregation: " + self.internal.last_error())
        return self._prove(self.internal, self.internal_pk, npv), npv

    def verify_root(self, root, rpv):
        """True iff `root` verifies under the aggregation key and states this key's two commitments (what aggregate() insists on)."""
        return (z.verify(self.node_params, self.root_vk(), [NOPV, NOPV, rpv], root) == 0 and rpv[-16:-8].tolist() == self.leaf_commit.tolist()
                and rpv[-8:].tolist() == self.internal_commit.tolist())

    def aggregate(self, proofs, pvs):
        """proofs: list of proof bytes of the app; pvs[i][a]: public values of AIR a of proof i.  Returns (root proof, root public values, levels)."""
        (root, rpv), levels = fold_tree(list(zip(proofs, pvs)), self.prove_leaf, self.prove_internal)
        if z.verify(self.node_params, self.root_vk(), [NOPV, NOPV, rpv], root) != 0:
            raise z.ZkhipError("aggregation: the root does not verify under the aggregation key")
        if rpv[-16:-8].tolist() != self.leaf_commit.tolist() or rpv[-8:].tolist() != self.internal_commit.tolist():
            raise z.ZkhipError("aggregation: the root does not state this key's commitments")
        return root, rpv, levels
