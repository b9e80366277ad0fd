"""CPU: the verifier circuit of the aggregation layer (zkhip_recursion_*, zkvm-prover_amd/csrc/recursion.hip).  Child proofs
come from the oracle prover (their bytes equal the HIP prover's: tests/test_gpu_stark.py); building the circuit and running
its witness need no GPU.
  * the circuit's assertions hold exactly when the product's verifier accepts (every word of a proof tampered in turn);
  * the wire values, gathered into the three chips' traces, satisfy every constraint, and the oracle proves the node and both
    verifiers accept it -- while the traces of a tampered child do NOT give a verifying proof;
  * the node statement (state chain, accumulator, child-vk digest) against an independent restatement;
  * a second level: an internal circuit verifies proofs of the leaf circuit."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import recursion_util as ru

PARAMS = (1, 0, 4, 3, 3)
NOPV = ru.NOPV


def prove_children(ora, starts, params=PARAMS):
    kids = [ru.counter_segment(s, seed=i) for i, s in enumerate(starts)]
    proofs = [ora.stark_prove(params, k) for k in kids]
    pvs = [[a["pvs"] for a in k] for k in kids]
    return kids, proofs, pvs


@pytest.fixture(scope="module")
def leaf(ora):
    kids, proofs, pvs = prove_children(ora, [5, 12, 19])
    vk = ru.verifying(PARAMS, kids[0])
    for k, p, pv in zip(kids, proofs, pvs):
        assert z.verify(PARAMS, vk, pv, p.tobytes()) == 0
    rc = z.RecursionCircuit(PARAMS, vk, 4, stmt=ru.COUNTER_STMT)
    return dict(kids=kids, proofs=proofs, pvs=pvs, vk=vk, rc=rc)


def test_circuit_accepts_exactly_what_the_verifier_accepts(ora, leaf):
    rc, proofs, pvs, vk = leaf["rc"], leaf["proofs"], leaf["pvs"], leaf["vk"]
    st, _ = rc.witness([p.tobytes() for p in proofs], pvs)
    assert st == 0, rc.last_error()
    base = [p.tobytes() for p in proofs]
    for pos in range(len(proofs[1])):
        bad = proofs[1].copy()
        bad[pos] = (int(bad[pos]) + 1) % ora.P
        v = z.verify(PARAMS, vk, pvs[1], bad.tobytes())
        st, _ = rc.witness([base[0], bad.tobytes(), base[2]], pvs)
        assert (v == 0) == (st == 0), "word %d: verifier %d, circuit %d" % (pos, v, st)
        assert st != 0
    # a public value of a child that is not part of the chained state (the Fibonacci chip's)
    pv2 = [[x.copy() for x in c] for c in pvs]
    pv2[2][2][1] = (int(pv2[2][2][1]) + 1) % ora.P
    assert rc.witness(base, pv2)[0] == -7
    # wrong size / non-canonical word
    assert rc.witness([base[0][:-4]], pvs[:1])[0] == -3
    bad = proofs[0].copy()
    bad[100] = ora.P
    assert rc.witness([bad.tobytes()], pvs[:1])[0] == -7


def test_statement_chain_flags_and_accumulator(ora, leaf):
    rc, proofs, pvs = leaf["rc"], leaf["proofs"], leaf["pvs"]
    base = [p.tobytes() for p in proofs]
    assert rc.n_pvs == 8 + 1 + 1 + 8 and rc.n_state == 1
    for n in (1, 2, 3):
        st, npv = rc.witness(base[:n], pvs[:n])
        assert st == 0, rc.last_error()
        assert npv[:8].tolist() == rc.child_vk_digest().tolist()
        assert int(npv[8]) == 5 and int(npv[9]) == 5 + 7 * n   # start of child 0, end of the last PRESENT child
        assert npv[10:].tolist() == ru.leaf_accumulator(pvs[:n]).tolist()
    # a broken hand-over: child 1 does not start where child 0 ends
    kids, proofs2, pvs2 = prove_children(ora, [5, 13])
    st, _ = rc.witness([p.tobytes() for p in proofs2], pvs2)
    assert st == -7
    # the same two proofs are fine on their own
    assert rc.witness([proofs2[1].tobytes()], pvs2[1:])[0] == 0
    # more children than the circuit was built for
    assert rc.witness(base + base[:2], pvs + pvs[:2])[0] == -3


def test_node_traces_satisfy_the_airs_and_prove(ora, leaf):
    rc, proofs, pvs = leaf["rc"], leaf["proofs"], leaf["pvs"]
    st, npv = rc.witness([p.tobytes() for p in proofs], pvs)
    assert st == 0
    node = ru.node_instance(rc, npv)
    for a in node[:2]:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []
    # the Poseidon2 chip's outputs are the wires the circuit says they are
    W, p2 = rc.wires(), node[1]
    for j in range(4):
        assert (p2["trace"][282 + 4 * j:286 + 4 * j, :rc.n_perms].T == W[p2["prep"][4 + j][:rc.n_perms]]).all()
    proof = ora.stark_prove(PARAMS, node)
    assert ora.stark_verify(PARAMS, node, proof) == 0
    nvk = ru.verifying(PARAMS, node)
    assert z.verify(PARAMS, nvk, [a["pvs"] for a in node], proof.tobytes()) == 0
    # other public values than the circuit computed: refused (the public-value chip binds them to the wires)
    wrong = npv.copy()
    wrong[9] = (int(wrong[9]) + 1) % ora.P
    assert z.verify(PARAMS, nvk, [NOPV, NOPV, wrong], proof.tobytes()) != 0
    leaf["node"] = dict(instance=node, proof=proof, vk=nvk, pvs=npv)


def test_horner_rows_bind_their_steps(ora, leaf):
    """The gate chip's Horner rows (csrc/recursion.hip Builder::hstep: the reduced opening of a query walks the coordinates of the packed
    opened rows, up to four steps per row): the circuit has them, every step is constrained -- a changed value between two steps, a changed
    coordinate that a step takes, or a changed result breaks a constraint of exactly that row --, a coordinate the row does NOT take is free,
    and on other rows the three extra slots must not matter to the gate constraint."""
    rc, proofs, pvs = leaf["rc"], leaf["proofs"], leaf["pvs"]
    st, npv = rc.witness([p.tobytes() for p in proofs], pvs)
    assert st == 0
    g = ru.node_instance(rc, npv)[0]
    prep, tr = g["prep"], g["trace"]
    rows = np.nonzero(prep[18])[0]
    assert len(rows) > 100 and tr.shape[0] == 28
    full = [r for r in rows if all(prep[19 + j][r] for j in range(4))]
    part = [r for r in rows if not all(prep[19 + j][r] for j in range(4))]
    assert full and part   # (matrix widths that are no multiples of four leave partly taken values)

    def bad_rows(t):
        return {r for (_, r) in air.check_trace(g["program"], t, g["pvs"], g["prep"])}

    assert bad_rows(tr) == set()
    r = int(full[len(full) // 2])
    for col in (16, 21, 26, 8, 4 + 2, 0, 12):   # the three values between the steps, the result, a taken coordinate, the start, alpha
        t = tr.copy()
        t[col, r] = (int(t[col, r]) + 1) % ora.P
        assert bad_rows(t) == {r}, col
    r = int(part[0])
    j = [j for j in range(4) if not prep[19 + j][r]][0]
    t = tr.copy()
    t[4 + j, r] = (int(t[4 + j, r]) + 1) % ora.P   # (the wire bus would notice: the chip's own constraints do not)
    assert bad_rows(t) == set()
    other = int(np.nonzero(prep[18] == 0)[0][5])
    t = tr.copy()
    t[20, other] = 7
    assert bad_rows(t) == set()


def test_traces_of_a_tampered_child_do_not_prove(ora, leaf):
    rc, proofs, pvs = leaf["rc"], leaf["proofs"], leaf["pvs"]
    bad = proofs[0].copy()
    lay = z.proof_layout(PARAMS, leaf["vk"])
    bad[lay["opened"] + 5] = (int(bad[lay["opened"] + 5]) + 1) % ora.P
    st, npv = rc.witness([bad.tobytes(), proofs[1].tobytes()], pvs[:2])
    assert st == -7
    node = ru.node_instance(rc, npv)
    assert any(air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) for a in node[:2])
    proof = ora.stark_prove(PARAMS, node)
    assert ora.stark_verify(PARAMS, node, proof) != 0
    assert z.verify(PARAMS, ru.verifying(PARAMS, node), [a["pvs"] for a in node], proof.tobytes()) != 0


def test_internal_level_verifies_leaf_proofs(ora, leaf):
    """Two leaf proofs (3 + 2 segments) under an internal circuit: the state chain and the vk digest go up, a leaf proof with
    another child-vk digest or a broken chain is refused."""
    if "node" not in leaf:
        test_node_traces_satisfy_the_airs_and_prove(ora, leaf)
    rc = leaf["rc"]
    n0 = leaf["node"]
    kids, proofs, pvs = prove_children(ora, [26, 33])
    st, npv1 = rc.witness([p.tobytes() for p in proofs], pvs)
    assert st == 0
    node1 = ru.node_instance(rc, npv1)
    proof1 = ora.stark_prove(PARAMS, node1)
    irc = z.RecursionCircuit(PARAMS, n0["vk"], 3, stmt="node")
    assert irc.n_state == 1 and irc.n_pvs == 18
    child_pvs = [[NOPV, NOPV, n0["pvs"]], [NOPV, NOPV, npv1]]
    st, top = irc.witness([n0["proof"].tobytes(), proof1.tobytes()], child_pvs)
    assert st == 0, irc.last_error()
    assert top[:8].tolist() == rc.child_vk_digest().tolist()          # still the APP's digest
    assert int(top[8]) == 5 and int(top[9]) == 40
    assert top[10:].tolist() == ru.internal_accumulator([n0["pvs"][10:], npv1[10:]]).tolist()
    inode = ru.node_instance(irc, top)
    for a in inode[:2]:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []
    # children in the wrong order: the chain breaks
    st, _ = irc.witness([proof1.tobytes(), n0["proof"].tobytes()], child_pvs[::-1])
    assert st == -7
    # a child claiming other public values than its proof binds
    lie = n0["pvs"].copy()
    lie[9] = 25
    assert irc.witness([n0["proof"].tobytes(), proof1.tobytes()], [[NOPV, NOPV, lie], child_pvs[1]])[0] == -7


def test_circuit_covers_cached_preprocessed_and_final_poly(ora):
    """A child AIR set with cached main partitions, a preprocessed table, blow-up 4 and a final polynomial of 4 coefficients."""
    from test_cached_main_cpu import cached_case

    params = (2, 2, 3, 2, 2)
    airs = cached_case()
    proof = ora.stark_prove(params, airs)
    pvs = [a["pvs"] for a in airs]
    vk = ru.verifying(params, airs)
    assert z.verify(params, vk, pvs, proof.tobytes()) == 0
    rc = z.RecursionCircuit(params, vk, 1)
    st, _ = rc.witness([proof.tobytes()], [pvs])
    assert st == 0, rc.last_error()
    rng = np.random.default_rng(1)
    for pos in sorted(set(range(4, 80)) | set(rng.integers(4, len(proof), 150).tolist())):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % ora.P
        assert rc.witness([bad.tobytes()], [pvs])[0] == -7, pos
    node = ru.node_instance(rc, rc.witness([proof.tobytes()], [pvs])[1])
    for a in node[:2]:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []


def test_build_refuses_inconsistent_keys(ora, leaf):
    vk = [dict(v) for v in leaf["vk"]]
    vk[0]["log_height"] = 27
    with pytest.raises(z.ZkhipError):
        z.RecursionCircuit(PARAMS, vk, 2)
    with pytest.raises(z.ZkhipError):
        z.RecursionCircuit(PARAMS, leaf["vk"], 2, stmt=dict(start=[(0, 5)], end=[(0, 1)]))   # no such public value
    with pytest.raises(z.ZkhipError):
        z.RecursionCircuit(PARAMS, leaf["vk"], 2, stmt="node")                                # last AIR is not a public-value chip


def test_one_aggregation_key_self_recursive_internal_circuit(ora):
    """ONE aggregation key (crates/prover/src/prover/mod.rs:147-170, crates/verifier/src/verifier.rs:96-111: one agg vk): the leaf and
    the internal circuit share programs and (padded) heights; the internal circuit takes its child's preprocessed commitments as values
    and verifies proofs of the leaf circuit AND of itself.  Six segments -> three leaf nodes -> two internal nodes -> a root that is a
    proof of the SAME internal key; the root states (leaf commitment, internal commitment); a child of another key, a child whose kind
    is misstated and nodes that state another commitment pair have no witness."""
    starts = [5, 12, 19, 26, 33, 40]
    kids = [ru.counter_segment(s, seed=i) for i, s in enumerate(starts)]
    proofs = [ora.stark_prove(PARAMS, k).tobytes() for k in kids]
    pvs = [[a["pvs"] for a in k] for k in kids]
    vk = ru.verifying(PARAMS, kids[0])
    leaf, internal = ru.one_key_circuits(PARAMS, vk, ru.COUNTER_STMT)
    assert leaf.n_pvs == internal.n_pvs == 8 + 1 + 1 + 8 + 16 and internal.n_state == 1
    leaf_pcs, LC = ru.node_key_commits(PARAMS, leaf.airs())
    int_pcs, IC = ru.node_key_commits(PARAMS, internal.airs())
    assert LC.tolist() != IC.tolist()

    def leaf_node(group):
        st, npv = leaf.witness([proofs[i] for i in group], [pvs[i] for i in group])
        assert st == 0, leaf.last_error()
        assert (npv[-16:] == 0).all()
        return ora.stark_prove(PARAMS, ru.node_instance(leaf, npv)).tobytes(), npv

    def internal_node(circ, children, kinds):
        st, npv = circ.witness([c[0] for c in children], [[NOPV, NOPV, c[1]] for c in children],
                               prep_commits=[leaf_pcs if k else int_pcs for k in kinds], is_leaf=kinds, leaf_commit=LC, internal_commit=IC)
        assert st == 0, circ.last_error()
        node = ru.node_instance(circ, npv)
        for a in node[:2]:
            assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []
        return ora.stark_prove(PARAMS, node).tobytes(), npv

    L = [leaf_node(g) for g in ([0, 1, 2], [3, 4], [5])]
    n0 = internal_node(internal, L[:2], [1, 1])
    fork = internal.fork()                                    # a second user of the same circuit (another level of the tree)
    n1 = internal_node(fork, L[2:], [1])
    root = internal_node(internal, [n0, n1], [0, 0])          # the internal circuit verifies proofs of ITSELF
    rpv = root[1]
    assert rpv[:8].tolist() == leaf.child_vk_digest().tolist()
    assert int(rpv[8]) == 5 and int(rpv[9]) == 47
    assert rpv[10:18].tolist() == ru.internal_accumulator([ru.internal_accumulator([L[0][1][10:18], L[1][1][10:18]]), ru.internal_accumulator([L[2][1][10:18]])]).tolist()
    assert rpv[18:26].tolist() == LC.tolist() and rpv[26:].tolist() == IC.tolist()
    # the root, the middle nodes: all proofs under the one internal key
    ivk = ru.verifying(PARAMS, [dict(a, prep_commit=c) for a, c in zip(internal.airs(), int_pcs)])
    for pr, npv in (n0, n1, root):
        assert z.verify(PARAMS, ivk, [NOPV, NOPV, npv], pr) == 0
    # a leaf proof and a node proof side by side
    mixed = internal_node(fork, [n0, L[2]], [0, 1])
    assert mixed[1][8:10].tolist() == [5, 47]

    def status(kinds, preps, lc=LC, ic=IC, ch=(n0, n1)):
        return internal.witness([c[0] for c in ch], [[NOPV, NOPV, c[1]] for c in ch], prep_commits=preps, is_leaf=kinds, leaf_commit=lc,
                                internal_commit=ic)[0]

    assert status([0, 0], [int_pcs, int_pcs]) == 0
    assert status([1, 0], [int_pcs, int_pcs]) == -7            # the kind is misstated
    assert status([0, 0], [leaf_pcs, int_pcs]) == -7           # a proof under another key than the one handed in
    bad = IC.copy()
    bad[0] ^= 1
    assert status([0, 0], [int_pcs, int_pcs], ic=bad) == -7    # the children's commitments do not hash to the stated internal commitment
    bad = LC.copy()
    bad[3] ^= 1
    assert status([0, 0], [int_pcs, int_pcs], lc=bad) == -7    # internal children state another leaf commitment
    assert status([1, 1], [leaf_pcs, leaf_pcs], lc=bad, ch=(L[0], L[1])) == -7
    # the per-key entry point refuses a uniform circuit and the other way round
    assert internal.witness([n0[0]], [[NOPV, NOPV, n0[1]]])[0] == -3
    assert leaf.witness([proofs[0]], [pvs[0]], prep_commits=[leaf_pcs], is_leaf=[1], leaf_commit=LC, internal_commit=IC)[0] == -3


def test_one_key_over_several_shapes_of_one_app(ora):
    """PER-PROOF CHIP PRESENCE (the reference proves only the chips a segment used, AGENTS.md:183-185): the app has one segment key -- and
    one leaf circuit -- per shape.  Here shape A = four chips, shape B = the same without the Fibonacci chip.  Segments A A B A -> leaf
    nodes [A A] [B] [A] (a leaf node takes proofs of ONE shape) -> one internal node over leaf proofs of both circuits: the root states the
    sponge of the two leaf commitments and ONE app id; a leaf proof of a circuit outside the list, a misstated shape have no witness."""
    starts = [5, 12, 19, 26]
    full = [ru.counter_segment(s, seed=i) for i, s in enumerate(starts)]
    segs = [full[0], full[1], [full[2][0], full[2][1], full[2][3]], full[3]]        # segment 2 carries no Fibonacci chip
    shape_of = [0, 0, 1, 0]
    vks = [ru.verifying(PARAMS, segs[2]), ru.verifying(PARAMS, full[0])]             # shapes: B (index 0), A = the full set (index 1, last)
    shape_idx = [1, 1, 0, 1]
    proofs = [ora.stark_prove(PARAMS, s).tobytes() for s in segs]
    pvs = [[a["pvs"] for a in s] for s in segs]
    leafs, internal, app_id = ru.one_key_circuits_shapes(PARAMS, vks, ru.COUNTER_STMT)
    assert app_id.tolist() == z.vk_digest(PARAMS, vks[1]).tolist() != z.vk_digest(PARAMS, vks[0]).tolist()
    keys = [ru.node_key_commits(PARAMS, l.airs()) for l in leafs]
    int_pcs, IC = ru.node_key_commits(PARAMS, internal.airs())
    LCs = [k[1] for k in keys]
    LC_list = ru.sponge(np.concatenate(LCs))

    def leaf_node(shape, group):
        st, npv = leafs[shape].witness([proofs[i] for i in group], [pvs[i] for i in group])
        assert st == 0, leafs[shape].last_error()
        assert npv[:8].tolist() == app_id.tolist()                                   # both circuits state the one app id
        return ora.stark_prove(PARAMS, ru.node_instance(leafs[shape], npv)).tobytes(), npv, shape

    assert leafs[1].witness([proofs[1], proofs[2]], [pvs[1], pvs[2]])[0] != 0       # a leaf node takes proofs of ONE shape
    L = [leaf_node(1, [0, 1]), leaf_node(0, [2]), leaf_node(1, [3])]
    assert shape_of and shape_idx

    def status(children, kinds, lcs=LCs, ic=IC):
        return internal.witness([c[0] for c in children], [[NOPV, NOPV, c[1]] for c in children],
                                prep_commits=[keys[k - 1][0] if k else int_pcs for k in kinds], is_leaf=kinds, leaf_commit=np.concatenate(lcs), internal_commit=ic)

    st, rpv = status(L, [2, 1, 2])
    assert st == 0, internal.last_error()
    assert rpv[:8].tolist() == app_id.tolist() and int(rpv[8]) == 5 and int(rpv[9]) == 33
    assert rpv[18:26].tolist() == LC_list.tolist() and rpv[26:34].tolist() == IC.tolist()
    node = ru.node_instance(internal, rpv)
    for a in node[:2]:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []
    root = ora.stark_prove(PARAMS, node).tobytes()
    # ... and the internal circuit verifies its own proof beside a leaf proof, stating the same list
    st, top = internal.witness([root], [[NOPV, NOPV, rpv]], prep_commits=[int_pcs], is_leaf=[0], leaf_commit=np.concatenate(LCs), internal_commit=IC)
    assert st == 0 and top[18:26].tolist() == LC_list.tolist()
    # refusals: a shape misstated, the list in another order (another statement: the children then fail their selector), a circuit outside the list
    assert status(L, [2, 2, 2])[0] == -7
    assert status(L, [1, 1, 2])[0] == -7
    assert status(L, [2, 1, 2], lcs=LCs[::-1])[0] == -7
    other = LCs[0].copy()
    other[2] ^= 1
    assert status(L, [2, 1, 2], lcs=[other, LCs[1]])[0] == -7
    st, top2 = internal.witness([root], [[NOPV, NOPV, rpv]], prep_commits=[int_pcs], is_leaf=[0], leaf_commit=np.concatenate([other, LCs[1]]), internal_commit=IC)
    assert st == -7                                                                  # an internal child states another list


def test_a_wide_shape_enters_the_tree_through_a_wrapper(ora):
    """A shape whose leaf circuit is too large for the tree's common heights (the reference's chunk-circuit configuration: 51 chips, a leaf
    circuit of 2^25 gate rows against 2^22) keeps a leaf circuit of its own, natural size; a WRAPPER -- a circuit of the common size that
    verifies ONE proof of that leaf circuit and restates its public values in the one-key layout -- takes its place in the list of leaf
    commitments.  Here the 'wide' shape is the four-chip set, the narrow one the three-chip set."""
    full = [ru.counter_segment(s, seed=i) for i, s in enumerate([5, 12, 19])]
    narrow_seg = [full[1][0], full[1][1], full[1][3]]
    vk_narrow, vk_wide = ru.verifying(PARAMS, narrow_seg), ru.verifying(PARAMS, full[0])
    app_id = z.vk_digest(PARAMS, vk_wide)
    big = z.RecursionCircuit(PARAMS, vk_wide, 1, stmt=ru.COUNTER_STMT, uniform=True, app_id=app_id)          # natural size, one proof per node
    big_pcs, _ = ru.node_key_commits(PARAMS, big.airs())
    big_vk = ru.verifying(PARAMS, [dict(a, prep_commit=c) for a, c in zip(big.airs(), big_pcs)])
    wrap = z.RecursionCircuit(PARAMS, big_vk, 1, stmt="node", uniform=True)
    assert wrap.n_pvs == big.n_pvs == 34
    narrow = z.RecursionCircuit(PARAMS, vk_narrow, 4, stmt=ru.COUNTER_STMT, uniform=True, app_id=app_id)
    # the common heights: over the narrow leaf, the wrapper and the internal circuit -- NOT the wide leaf
    child = [{k: a[k] for k in ("program", "log_height", "width", "n_pvs")} for a in narrow.airs()]
    H = [max(c.log_heights()[i] for c in (narrow, wrap)) for i in (0, 1)]
    while True:
        for c, h in zip(child, H + [0]):
            c["log_height"] = h
        internal = z.RecursionCircuit(PARAMS, child, 3, stmt="uniform", min_log_height=H, n_leaf_shapes=2)
        if internal.log_heights()[:2] == H:
            break
        H = internal.log_heights()[:2]
    narrow.pad(*H), wrap.pad(*H)
    keys = [ru.node_key_commits(PARAMS, c.airs()) for c in (narrow, wrap)]
    int_pcs, IC = ru.node_key_commits(PARAMS, internal.airs())
    segs = [full[0], narrow_seg, full[2]]                                                                      # wide, narrow, wide
    proofs = [ora.stark_prove(PARAMS, s).tobytes() for s in segs]
    pvs = [[a["pvs"] for a in s] for s in segs]

    def through_wrapper(i):
        st, npv = big.witness([proofs[i]], [pvs[i]])
        assert st == 0, big.last_error()
        bp = ora.stark_prove(PARAMS, ru.node_instance(big, npv)).tobytes()
        st, wpv = wrap.witness([bp], [[NOPV, NOPV, npv]])
        assert st == 0, wrap.last_error()
        assert wpv[:10].tolist() == npv[:10].tolist() and (wpv[-16:] == 0).all()                              # app id, start, end restated
        assert wpv[10:18].tolist() == ru.internal_accumulator([npv[10:18]]).tolist()
        return ora.stark_prove(PARAMS, ru.node_instance(wrap, wpv)).tobytes(), wpv

    st, npv1 = narrow.witness([proofs[1]], [pvs[1]])
    assert st == 0
    L = [through_wrapper(0), (ora.stark_prove(PARAMS, ru.node_instance(narrow, npv1)).tobytes(), npv1), through_wrapper(2)]
    kinds = [2, 1, 2]
    st, rpv = internal.witness([c[0] for c in L], [[NOPV, NOPV, c[1]] for c in L], prep_commits=[keys[k - 1][0] for k in kinds], is_leaf=kinds,
                               leaf_commit=np.concatenate([keys[0][1], keys[1][1]]), internal_commit=IC)
    assert st == 0, internal.last_error()
    assert rpv[:8].tolist() == app_id.tolist() and int(rpv[8]) == 5 and int(rpv[9]) == 26
    # a proof of the wide leaf circuit itself (not wrapped) is not a child of the internal circuit: another height set altogether
    st, npv0 = big.witness([proofs[0]], [pvs[0]])
    bp0 = ora.stark_prove(PARAMS, ru.node_instance(big, npv0)).tobytes()
    assert internal.witness([bp0], [[NOPV, NOPV, npv0]], prep_commits=[big_pcs], is_leaf=[2], leaf_commit=np.concatenate([keys[0][1], keys[1][1]]), internal_commit=IC)[0] != 0
