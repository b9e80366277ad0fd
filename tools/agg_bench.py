#!/usr/bin/env python3
"""The aggregation layer at the REFERENCE'S parameters (100 queries, PoW 16 + 16, blow-up 2): 4 proofs of a segment-shaped
AIR set (19 chips + range table, tallest 2^log_max) under a leaf verifier circuit, then 3 leaf proofs under an internal one.
Reports circuit sizes, witness (host) time, trace generation + proof time on the device, per-kernel breakdown.
Usage: python tools/agg_bench.py [log_max] [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import zkvm_prover_amd as z
from zkvm_prover_amd import air

NOPV = np.zeros(0, np.uint32)


def timed(fn, reps=3):
    best = 1e9
    out = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


def main():
    log_max = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    params = z.DEFAULT_PARAMS
    torch.cuda.set_stream(torch.cuda.Stream(device=0))
    ctx = z.Context(0)
    cs = air.ChipSet(n_chips=19, log_max=log_max, log_min=max(4, log_max - 10), total_width=400, seed=0)
    res = {"params": list(params), "child": {"chips": 20, "log_max": log_max}}
    kids, proofs, pvs = [], [], []
    pk = None
    for i in range(4):
        airs = cs.gen(seed=i + 1)
        if pk is None:
            pk = z.ProvingKey(ctx, params, airs)
            res["child"]["cells"] = int(sum(a["width"] << a["log_height"] for a in airs))
        d = [ctx.upload(a["trace"].reshape(-1)) for a in airs]
        pv = [a["pvs"] for a in airs]
        dt, proof = timed(lambda: pk.prove(d, pv), reps=2 if i == 0 else 1)
        kids.append(airs), proofs.append(proof), pvs.append(pv)
        res["child"]["prove_ms"] = round(dt * 1e3, 2)
        res["child"]["proof_bytes"] = len(proof)
    vk = pk.verifying_airs()
    t0 = time.perf_counter()
    assert z.verify(params, vk, pvs[0], proofs[0]) == 0
    res["child"]["host_verify_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    child_vk, child_proofs, child_pvs, stmt = vk, proofs, pvs, None
    for level, arity in ((0, 4), (1, 3)):
        t0 = time.perf_counter()
        rc = z.RecursionCircuit(params, child_vk, arity, stmt=stmt)
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        st, npv = rc.witness(child_proofs[:arity], child_pvs[:arity])
        t_wit = time.perf_counter() - t0
        assert st == 0, rc.last_error()
        node_airs = rc.airs()
        t0 = time.perf_counter()
        npk = z.ProvingKey(ctx, params, node_airs)
        torch.cuda.synchronize()
        t_keygen = time.perf_counter() - t0
        node_pvs = [NOPV, NOPV, npv]

        def gen_and_prove():
            tr = rc.tracegen(ctx)
            return npk.prove(tr, node_pvs)

        gen_and_prove()
        t_prove, nproof = timed(gen_and_prove)
        ctx.profile_reset()
        ctx.profile_enable(True)
        gen_and_prove()
        ctx.profile_enable(False)
        top = sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1])[:12]
        nvk = npk.verifying_airs()
        assert z.verify(params, nvk, node_pvs, nproof) == 0
        cells = sum(a["width"] << a["log_height"] for a in node_airs)
        res["level%d" % level] = {"children": arity, "wires": rc.n_wires, "gate_rows": rc.n_gates, "permutations": rc.n_perms,
                                  "log_heights": [a["log_height"] for a in node_airs], "main_cells": int(cells),
                                  "build_s": round(t_build, 3), "keygen_s": round(t_keygen, 3), "witness_ms": round(t_wit * 1e3, 1),
                                  "tracegen_prove_ms": round(t_prove * 1e3, 2), "proof_bytes": len(nproof),
                                  "children_folded_per_s": round(arity / (t_wit + t_prove), 2),
                                  "kernels_ms": {k: round(v[1], 3) for k, v in top}}
        sys.stderr.write(json.dumps(res["level%d" % level]) + "\n")
        # next level: three copies of this node's proof would break the state chain only if there were a chained state
        child_vk, child_proofs, child_pvs, stmt = nvk, [nproof] * 3, [node_pvs] * 3, "node"
        npk.close()
    s = json.dumps(res, indent=1)
    print(s)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(s + "\n")


if __name__ == "__main__":
    main()
