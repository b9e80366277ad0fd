"""CPU, world_size 2 over gloo: the N>1 path of bench.py / shard.py -- round-robin segment
assignment, all-gather of commitments, proof gather -- with real proofs (made by the oracle here,
by the HIP prover on GPUs) checked by the product's host verifier on rank 0; then rank 0 FOLDS the
gathered proofs to one root under one aggregation key (zkvm-prover_amd/aggregate.py fold_tree over the
real verifier circuits, the oracle as the node prover: what crates/integration/src/testers/batch.rs:97-107
hands to the next layer)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = (1, 0, 6, 3, 3)


def _airs(seed):
    from zkvm_prover_amd import air

    sa = air.SyntheticAir(width=16, n_free=6, n_bool=2, n_boundary=2, seed=7)  # same AIR, different witness
    tr, pv = sa.gen_trace(5, seed=seed)
    return [dict(program=sa.program(), log_height=5, width=16, n_pvs=len(pv), trace=tr, pvs=pv)]


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist

    import oracle_lib as ora
    import zkvm_prover_amd as z
    from zkvm_prover_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.assign_segments(5, world, rank)
    assert mine == list(range(rank, 5, world))
    airs = _airs(100 + rank)
    proof = ora.stark_prove(PARAMS, airs).tobytes()
    commits, proofs = shard.exchange(proof)
    assert len(commits) == world and commits[rank] == shard.commitment_of(proof)
    ok = True
    if rank == 0:
        assert len(proofs) == world and proofs[0] == proof
        for r in range(world):
            a = _airs(100 + r)
            ok &= z.verify(PARAMS, a, [a[0]["pvs"]], proofs[r]) == 0
            ok &= shard.commitment_of(proofs[r]) == commits[r]
        ok &= commits[0] != commits[1]
        # gather -> aggregate: the gathered proofs under ONE root (leaf node over both, wrapped by the internal circuit)
        import recursion_util as ru

        vk = [{k: a[k] for k in ("program", "log_height", "width", "n_pvs")} for a in _airs(100)]
        agg = ru.OracleAggregator(PARAMS, vk)
        (root, rpv), levels = agg.aggregate(proofs, [[_airs(100 + r)[0]["pvs"]] for r in range(world)])
        ok &= [len(l) for l in levels] == [1, 1]
        ok &= z.verify(PARAMS, agg.root_vk(), [ru.NOPV, ru.NOPV, rpv], root) == 0
        ok &= rpv[-16:-8].tolist() == agg.leaf_commit.tolist() and rpv[-8:].tolist() == agg.internal_commit.tolist()
        ok &= rpv[8:16].tolist() == ru.internal_accumulator([ru.leaf_accumulator([[_airs(100 + r)[0]["pvs"]] for r in range(world)])]).tolist()
        # a gathered proof that was swapped for another rank's claim has no witness
        try:
            agg.aggregate(proofs[::-1], [[_airs(100 + r)[0]["pvs"]] for r in range(world)])
            ok = False
        except AssertionError:
            pass
    else:
        assert proofs is None
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_two_rank_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_fold_tree_shape():
    """aggregate.fold_tree: leaf groups of 4, internal groups of 3, at least one internal level (the root is a proof of the internal
    circuit whatever the count) -- the shape of AggregationPlan::build under one key."""
    from zkvm_prover_amd import aggregate

    for n, want in ((1, [1, 1]), (4, [1, 1]), (5, [2, 1]), (13, [4, 2, 1]), (40, [10, 4, 2, 1])):
        root, levels = aggregate.fold_tree(list(range(n)), lambda g: ("L", tuple(g)), lambda g, leaves: ("I", leaves, tuple(g)))
        assert [len(l) for l in levels] == want and root[0] == "I"
        assert all(node[1] for node in levels[1]) and not any(node[1] for lv in levels[2:] for node in lv)

        def flat(node):
            return [x for c in node[-1] for x in (flat(c) if isinstance(c, tuple) else [c])]

        assert flat(root) == list(range(n))


def test_assign_segments():
    from zkvm_prover_amd import shard

    for world in (1, 2, 4, 8):
        seen = sorted(s for r in range(world) for s in shard.assign_segments(13, world, r))
        assert seen == list(range(13))
