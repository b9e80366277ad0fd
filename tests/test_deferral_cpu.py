"""CPU: the deferral circuits of the aggregation layer (zkvm-prover_amd/csrc/recursion.hip, zkhip_recursion_stmt.child_is_node = 3 and
zkhip_recursion_build_join) -- what the reference builds in `Prover::enable_deferral` (crates/prover/src/prover/mod.rs:200-282: child agg
vk -> VerifyProver -> deferral prover) for a guest that calls `verify_stark(input_commit, expected)` (crates/types/circuit/src/lib.rs:137-154).
Children here are toy guest flows (segments whose public values are a (pc, memory root) state, proven by the oracle, folded under ONE
aggregation key): building the circuits and running their witnesses need no GPU.
  * the deferral node verifies root proofs of the child app, opens each child's 32 public-value bytes in its final memory root, and
    chains the claims; its accumulator == the independent restatement (tests/recursion_util.py deferral_claim / deferral_chain);
  * node traces satisfy the AIRs, the oracle proves the node, the product's verifier accepts it;
  * the join verifies (a root under key A, the deferral node's proof) and states [root statement | deferral accumulator];
  * a flipped byte in a child proof, a wrong opening, a child that did not exit with code 0 have no witness; reordered children give
    another accumulator; a chain that does not start at zero cannot be joined;
  * BUNDLE OVER BATCHES: a deferral node with region_index != 0 takes JOIN proofs -- it opens the child guest's deferral region in the
    child's final memory root and chains its claims in the circuit; a wrong claim count, a cell that does not open and a region whose
    claims are not the ones the child's own deferral node verified have no witness."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import recursion_util as ru

PARAMS = (1, 0, 4, 3, 3)
NOPV = ru.NOPV
REGION_BASE = 0x00402000


@pytest.fixture(scope="module")
def flows(ora):
    rng = np.random.default_rng(5)

    def dig():
        return rng.integers(0, ora.P, 8, dtype=np.uint64).astype(np.uint32)

    image, mid = dig(), dig()
    seg0 = ru.state_segment([0x200000] + list(image), [0x200040] + list(mid))
    leaf, internal = ru.one_key_circuits(PARAMS, ru.verifying(PARAMS, seg0), ru.STATE_STMT)
    leaf_pcs, LC = ru.node_key_commits(PARAMS, leaf.airs())
    int_pcs, IC = ru.node_key_commits(PARAMS, internal.airs())
    ivk = ru.verifying(PARAMS, [dict(a, prep_commit=c) for a, c in zip(internal.airs(), int_pcs)])

    def child_flow(pv_bytes, pc_end=0, region=None):
        extra = {}
        if region is None:
            root1, cells, sibs = ru.memory_root_with_public_values(pv_bytes, rng)
        else:   # a guest that itself deferred: its final memory also holds its claims
            root1, cells, sibs, rsibs, ridx = ru.memory_root_with_public_values_and_region(pv_bytes, region, REGION_BASE, rng)
            extra = dict(region=np.asarray(region, np.uint32), region_sibs=rsibs, region_index=ridx)
        segs = [ru.state_segment([0x200000] + list(image), [0x200040] + list(mid)), ru.state_segment([0x200040] + list(mid), [pc_end] + list(root1))]
        proofs = [ora.stark_prove(PARAMS, s).tobytes() for s in segs]
        st, npv = leaf.witness(proofs, [[a["pvs"] for a in s] for s in segs])
        assert st == 0, leaf.last_error()
        lp = ora.stark_prove(PARAMS, ru.node_instance(leaf, npv)).tobytes()
        st, rpv = internal.witness([lp], [[NOPV, NOPV, npv]], prep_commits=[leaf_pcs], is_leaf=[1], leaf_commit=LC, internal_commit=IC)
        assert st == 0, internal.last_error()
        rp = ora.stark_prove(PARAMS, ru.node_instance(internal, rpv)).tobytes()
        assert z.verify(PARAMS, ivk, [NOPV, NOPV, rpv], rp) == 0
        return dict(proof=rp, pvs=rpv, aux=np.concatenate([cells, sibs.reshape(-1)]), cells=cells, **extra)

    kids = [child_flow(bytes([i + 1] * 32)) for i in range(3)]
    return dict(kids=kids, ivk=ivk, child_flow=child_flow, LC=LC, IC=IC)


def _witness(D, kids, **kw):
    return D.witness([k["proof"] for k in kids], [[NOPV, NOPV, k["pvs"]] for k in kids], aux=[k["aux"] for k in kids], **kw)


def test_deferral_node_and_join(ora, flows):
    kids, ivk = flows["kids"], flows["ivk"]
    D = z.RecursionCircuit(PARAMS, ivk, 4, stmt="deferral")
    assert D.n_pvs == 16
    st, dpv = _witness(D, kids)
    assert st == 0, D.last_error()
    want = ru.deferral_chain(np.zeros(8, np.uint32), [ru.deferral_claim(k["pvs"], k["cells"]) for k in kids])
    assert (dpv[:8] == 0).all() and dpv[8:].tolist() == want.tolist()
    dnode = ru.node_instance(D, dpv)
    for a in dnode[:2]:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []
    dproof = ora.stark_prove(PARAMS, dnode).tobytes()
    dvk = ru.verifying(PARAMS, dnode)
    assert z.verify(PARAMS, dvk, [NOPV, NOPV, dpv], dproof) == 0
    # the join: a root under key A (the toy key itself) + the deferral node's proof
    J = z.RecursionCircuit.join(PARAMS, ivk, PARAMS, dvk)
    assert J.n_pvs == 50 + 8
    g = kids[0]
    st, jpv = J.witness([g["proof"], dproof], [[NOPV, NOPV, g["pvs"]], [NOPV, NOPV, dpv]])
    assert st == 0, J.last_error()
    assert jpv[:50].tolist() == g["pvs"].tolist() and jpv[50:].tolist() == want.tolist()
    jnode = ru.node_instance(J, jpv)
    for a in jnode[:2]:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []
    jproof = ora.stark_prove(PARAMS, jnode).tobytes()
    assert z.verify(PARAMS, ru.verifying(PARAMS, jnode), [NOPV, NOPV, jpv], jproof) == 0
    # the two proofs the other way round, one proof only: no witness
    assert J.witness([dproof, g["proof"]], [[NOPV, NOPV, dpv], [NOPV, NOPV, g["pvs"]]])[0] == -3
    assert J.witness([g["proof"]], [[NOPV, NOPV, g["pvs"]]])[0] == -3
    # a deferral proof whose chain does not start at zero cannot be joined
    st, d3 = _witness(D, kids[:1], acc_start=np.arange(8))
    assert st == 0 and d3[:8].tolist() == list(range(8))
    assert d3[8:].tolist() == ru.deferral_chain(np.arange(8, dtype=np.uint32), [ru.deferral_claim(kids[0]["pvs"], kids[0]["cells"])]).tolist()
    d3proof = ora.stark_prove(PARAMS, ru.node_instance(D, d3)).tobytes()
    assert J.witness([g["proof"], d3proof], [[NOPV, NOPV, g["pvs"]], [NOPV, NOPV, d3]])[0] == -7


def test_deferral_node_refusals(ora, flows):
    kids, ivk = flows["kids"], flows["ivk"]
    D = z.RecursionCircuit(PARAMS, ivk, 4, stmt="deferral")
    two = kids[:2]
    assert _witness(D, two)[0] == 0
    # a flipped byte in a child proof
    bad = bytearray(two[1]["proof"])
    bad[4 * 300] ^= 1
    assert _witness(D, [two[0], dict(two[1], proof=bytes(bad))])[0] == -7
    # public-value cells that do not open in the child's final memory root; a wrong sibling
    for pos in (3, 16 + 8 * 5 + 2):
        ax = two[1]["aux"].copy()
        ax[pos] ^= 1
        assert _witness(D, [two[0], dict(two[1], aux=ax)])[0] == -7
    # a child that states other public values than its proof binds (another exe commitment: the initial memory root)
    lie = two[1]["pvs"].copy()
    lie[10] ^= 1
    assert _witness(D, [two[0], dict(two[1], pvs=lie)])[0] == -7
    # reordered children verify, but chain to another accumulator
    st, swapped = _witness(D, two[::-1])
    assert st == 0
    assert swapped[8:].tolist() != _witness(D, two)[1][8:].tolist()
    # a child flow that did not exit with code 0 (pc_end != 0)
    nz = flows["child_flow"](bytes([9] * 32), pc_end=0x200080)
    assert _witness(D, [nz])[0] == -7
    # the other witness entry points refuse a deferral circuit
    assert D.witness([two[0]["proof"]], [[NOPV, NOPV, two[0]["pvs"]]])[0] == -3


def test_deferral_over_join_children(ora, flows):
    """chunk -> batch -> bundle: the bundle's deferral node over a BATCH proof (a join)."""
    kids, ivk = flows["kids"], flows["ivk"]
    claims = [ru.deferral_claim(k["pvs"], k["cells"]) for k in kids[:2]]
    chain = ru.deferral_chain(np.zeros(8, np.uint32), claims)
    D1 = z.RecursionCircuit(PARAMS, ivk, 4, stmt="deferral")
    st, dpv = _witness(D1, kids[:2])
    assert st == 0 and dpv[8:].tolist() == chain.tolist()
    dnode = ru.node_instance(D1, dpv)
    dproof, dvk = ora.stark_prove(PARAMS, dnode).tobytes(), ru.verifying(PARAMS, dnode)
    J = z.RecursionCircuit.join(PARAMS, ivk, PARAMS, dvk)

    def batch(region_claims):   # a guest flow whose memory holds `region_claims`, joined with the deferral node over kids[:2]
        g = flows["child_flow"](bytes([0x42] * 32), region=ru.deferral_region_cells(region_claims))
        st, jpv = J.witness([g["proof"], dproof], [[NOPV, NOPV, g["pvs"]], [NOPV, NOPV, dpv]])
        assert st == 0, J.last_error()
        jnode = ru.node_instance(J, jpv)
        return dict(g, jpv=jpv, jproof=ora.stark_prove(PARAMS, jnode).tobytes(), jvk=ru.verifying(PARAMS, jnode))

    b = batch(claims)
    assert b["jpv"].size == 58 and b["jpv"][50:].tolist() == chain.tolist()
    with pytest.raises(z.ZkhipError):   # a join key needs the region's place; a plain key refuses one
        z.RecursionCircuit(PARAMS, b["jvk"], 2, stmt="deferral")
    with pytest.raises(z.ZkhipError):
        z.RecursionCircuit(PARAMS, ivk, 2, stmt="deferral", region_index=b["region_index"])
    D2 = z.RecursionCircuit(PARAMS, b["jvk"], 2, stmt="deferral", region_index=b["region_index"])
    assert D2.n_pvs == 16

    def aux(x, n_flags=None, region=None):
        reg = x["region"] if region is None else region
        n = int(reg[0]) if n_flags is None else n_flags
        return np.concatenate([x["aux"], reg, x["region_sibs"].reshape(-1), np.array([1 if k < n else 0 for k in range(63)], np.uint32)])

    def wit(x, **kw):
        return D2.witness([x["jproof"]], [[NOPV, NOPV, x["jpv"]]], aux=[aux(x, **kw)])

    st, pv2 = wit(b)
    assert st == 0, D2.last_error()
    want = ru.deferral_chain(np.zeros(8, np.uint32), [ru.deferral_claim(b["jpv"], b["cells"])])
    assert (pv2[:8] == 0).all() and pv2[8:].tolist() == want.tolist()
    node = ru.node_instance(D2, pv2)
    for a in node[:2]:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []
    assert z.verify(PARAMS, ru.verifying(PARAMS, node), [NOPV, NOPV, pv2], ora.stark_prove(PARAMS, node).tobytes()) == 0
    # a wrong claim count, no claim counted, a cell that is not the one in the batch's memory, a sibling
    assert wit(b, n_flags=1)[0] == -7 and wit(b, n_flags=3)[0] == -7 and wit(b, n_flags=0)[0] == -7
    for pos in (2 * 40, 2 * 70 + 1, 0):
        reg = b["region"].copy()
        reg[pos] ^= 1
        assert wit(b, region=reg)[0] == -7
    sib = dict(b, region_sibs=b["region_sibs"].copy())
    sib["region_sibs"][3][1] ^= 1
    assert wit(sib)[0] == -7
    # a batch whose memory states OTHER claims than the ones its deferral node verified (the join itself does not look at the memory)
    liar = batch(claims[::-1])
    assert D2.witness([liar["jproof"]], [[NOPV, NOPV, liar["jpv"]]], aux=[aux(liar)])[0] == -7
    # (a region that states NO claim -- word 0 = 0 -- has no flags the circuit accepts either: n_flags = 0 above fails on the count for this
    # batch, and on `flag 0 = 1` for any batch)


def test_fold_of_deferral_nodes(ora, flows):
    """A task with more children than one deferral node takes (a batch holds up to 45 chunks): several deferral nodes, each continuing the
    chain of the one before, a FOLD over their proofs (a node circuit with the chain as its chained state), the join over the fold."""
    kids, ivk = flows["kids"], flows["ivk"]
    claims = [ru.deferral_claim(k["pvs"], k["cells"]) for k in kids]
    zero = np.zeros(8, np.uint32)
    mid, end = ru.deferral_chain(zero, claims[:2]), ru.deferral_chain(zero, claims)
    D = z.RecursionCircuit(PARAMS, ivk, 2, stmt="deferral")
    st, d0 = _witness(D, kids[:2])
    assert st == 0 and d0[:8].tolist() == [0] * 8 and d0[8:].tolist() == mid.tolist()
    n0 = ru.node_instance(D, d0)
    p0, dvk = ora.stark_prove(PARAMS, n0).tobytes(), ru.verifying(PARAMS, n0)
    st, d1 = _witness(D, kids[2:], acc_start=mid)
    assert st == 0 and d1[:8].tolist() == mid.tolist() and d1[8:].tolist() == end.tolist()
    p1 = ora.stark_prove(PARAMS, ru.node_instance(D, d1)).tobytes()
    F = z.RecursionCircuit(PARAMS, dvk, 3, stmt=dict(start=[(2, k) for k in range(8)], end=[(2, 8 + k) for k in range(8)]))
    assert F.n_pvs == 32
    st, fpv = F.witness([p0, p1], [[NOPV, NOPV, d0], [NOPV, NOPV, d1]])
    assert st == 0, F.last_error()
    assert fpv[8:16].tolist() == [0] * 8 and fpv[16:24].tolist() == end.tolist()
    fnode = ru.node_instance(F, fpv)
    for a in fnode[:2]:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a["prep"]) == []
    fproof, fvk = ora.stark_prove(PARAMS, fnode).tobytes(), ru.verifying(PARAMS, fnode)
    assert z.verify(PARAMS, fvk, [NOPV, NOPV, fpv], fproof) == 0
    # nodes the other way round: the chain does not continue
    assert F.witness([p1, p0], [[NOPV, NOPV, d1], [NOPV, NOPV, d0]])[0] == -7
    # the join over (root, fold): the statement ends with the chain over all three children
    J = z.RecursionCircuit.join(PARAMS, ivk, PARAMS, fvk)
    g = kids[0]
    st, jpv = J.witness([g["proof"], fproof], [[NOPV, NOPV, g["pvs"]], [NOPV, NOPV, fpv]])
    assert st == 0, J.last_error()
    assert jpv.size == 58 and jpv[:50].tolist() == g["pvs"].tolist() and jpv[50:].tolist() == end.tolist()
    # a fold whose chain does not start at zero (the second node alone) cannot be joined
    st, f1 = F.witness([p1], [[NOPV, NOPV, d1]])
    assert st == 0 and f1[8:16].tolist() == mid.tolist()
    f1proof = ora.stark_prove(PARAMS, ru.node_instance(F, f1)).tobytes()
    assert J.witness([g["proof"], f1proof], [[NOPV, NOPV, g["pvs"]], [NOPV, NOPV, f1]])[0] == -7
