#!/usr/bin/env python3
"""Generates tests/golden/proof_digests_v3.json: SHA-256 of the oracle's proof bytes for a fixed set of small AIR
sets (plain, mixed heights, bus interactions, compound messages, preprocessed tables, a small chip set).

These are REGRESSION pins produced by this repository's own oracle (no upstream vectors exist offline, SURVEY.md
8c): the GPU parity tests compare the HIP prover with the oracle, this fixture keeps the pair from drifting together.
Run from the repo root:  python3 tests/golden/gen_proof_digests.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402

from zkvm_prover_amd import air  # noqa: E402

PARAMS = (1, 0, 8, 3, 4)
NOPV = np.zeros(0, np.uint32)


def cases():
    def fib(n):
        tr, pv = air.fibonacci_trace(n)
        return dict(program=air.fibonacci_air().program(), log_height=n, width=2, n_pvs=3, trace=tr, pvs=pv)

    def syn(log_n, width, n_free, seed, n_bool=4, n_boundary=3):
        sa = air.SyntheticAir(width=width, n_free=n_free, n_bool=n_bool, n_boundary=n_boundary, seed=seed)
        tr, pv = sa.gen_trace(log_n, seed=seed + 100)
        return dict(program=sa.program(), log_height=log_n, width=width, n_pvs=len(pv), trace=tr, pvs=pv)

    s, t = air.lookup_traces(6, 4, seed=1)
    sender = dict(program=air.lookup_sender_air().program(), log_height=6, width=3, n_pvs=0, trace=s, pvs=NOPV)
    table = dict(program=air.lookup_table_air().program(), log_height=4, width=3, n_pvs=0, trace=t, pvs=NOPV)
    mt, mpv = air.bus_mix_trace(5, 3)
    mix = dict(program=air.bus_mix_air().program(), log_height=5, width=6, n_pvs=1, trace=mt, pvs=mpv)
    limb = dict(program=air.limb_air().program(), log_height=5, width=4, n_pvs=0, trace=air.limb_trace(5, 3), pvs=NOPV)
    u, m, prep = air.range_traces(6, 4, seed=1)
    user = dict(program=air.range_user_air().program(), log_height=6, width=4, n_pvs=0, trace=u, pvs=NOPV)
    rtab = dict(program=air.range_table_air().program(), log_height=4, width=1, n_pvs=0, trace=m, pvs=NOPV, prep=prep)
    return {
        "fib_5": (PARAMS, [fib(5)]),
        "fib_min": (PARAMS, [fib(1)]),
        "mixed_heights": (PARAMS, [syn(7, 40, 10, 1), fib(6), syn(4, 12, 5, 2, n_bool=2, n_boundary=2)]),
        "blowup_4": ((2, 0, 6, 3, 3), [syn(6, 20, 7, 13), fib(5)]),
        "lookup": (PARAMS, [sender, fib(5), table]),
        "bus_mix": (PARAMS, [mix]),
        "compound_messages": (PARAMS, [limb, sender, table]),
        "range_table_prep": (PARAMS, [user, fib(5), rtab]),
        "chipset_6": (PARAMS, air.ChipSet(n_chips=6, log_max=8, log_min=3, total_width=60, seed=1, log_table=2).gen(1)),
    }


def digests(ora):
    out = {}
    for name, (params, airs) in cases().items():
        proof = ora.stark_prove(params, airs)
        assert ora.stark_verify(params, airs, proof) == 0
        out[name] = {"words": int(len(proof)), "sha256": hashlib.sha256(proof.tobytes()).hexdigest()}
    return out


if __name__ == "__main__":
    import oracle_lib as ora

    path = os.path.join(HERE, "proof_digests_v3.json")
    json.dump(digests(ora), open(path, "w"), indent=1, sort_keys=True)
    print(open(path).read())
