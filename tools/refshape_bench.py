"""Proof time of an AIR set with the dimensions of the chunk proof the reference stores (air.ReferenceShapedSet: 17 AIRs of
2^1 .. 2^21 rows, cached main of width 9, two preprocessed traces, the stored proof's after-challenge widths; blow-up 4,
44 queries): one proof alone with the per-kernel profile, then `inflight` proofs in flight on contexts of their own.
Usage: python tools/refshape_bench.py [shrink] [inflight] [proofs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import zkvm_prover_amd as z
from zkvm_prover_amd import air

shrink = int(sys.argv[1]) if len(sys.argv) > 1 else 0
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n_proofs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
PARAMS = (2, 0, 44, 0, 16)
t0 = time.time()
airs = air.ReferenceShapedSet(shrink=shrink).gen()
cells = sum(a["width"] << a["log_height"] for a in airs)
print("17 AIRs, heights 2^%s, %.1f M main-trace cells (trace generation %.1f s)"
      % ([a["log_height"] for a in airs], cells / 1e6, time.time() - t0), flush=True)
pvs = [a["pvs"] for a in airs]
lanes = []
streams = [torch.cuda.Stream(device=0) for _ in range(inflight)]
for k in range(inflight):
    with torch.cuda.stream(streams[k]):   # a Python Context issues on the torch stream current at its creation
        ctx = z.Context(0)
        pk = z.ProvingKey(ctx, PARAMS, airs)
        lanes.append((ctx, pk, [ctx.upload(a["trace"].reshape(-1)) for a in airs]))
ctx, pk, d = lanes[0]
proof = pk.prove(d, pvs)
vk = pk.verifying_airs()
assert z.verify(PARAMS, vk, pvs, proof) == 0
print("proof %d bytes, verifies; v1 container %d bytes" % (len(proof), len(z.proof_to_v1(PARAMS, vk, pvs, proof))), flush=True)
ctx.profile_reset()
ctx.profile_enable(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
pk.prove(d, pvs)
torch.cuda.synchronize()
ctx.profile_enable(False)
prof = ctx.profile_read()
best = 1e9
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pk.prove(d, pvs)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print("one proof alone: %.2f ms (%.2f ns per main-trace cell)" % (best * 1e3, best * 1e9 / cells), flush=True)
for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:16]:
    print("   %-28s x%-4d %9.3f ms" % (k, cnt, ms))


# `inflight` proofs in flight, driven from one thread like bench.py: enqueue on lane k, fetch the proof it held before
per = (n_proofs + inflight - 1) // inflight
for c, p, dd in lanes:
    p.prove(dd, pvs)
torch.cuda.synchronize()
t0 = time.perf_counter()
pending = [False] * inflight
for i in range(per * inflight):
    k = i % inflight
    c, p, dd = lanes[k]
    if pending[k]:
        p.fetch()
    p.prove_async(dd, pvs)
    pending[k] = True
for k, (c, p, dd) in enumerate(lanes):
    if pending[k]:
        assert p.fetch() == proof
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": "reference-shaped chunk proof (17 AIRs, 2^1..2^%d rows, %.0f M main cells, blow-up 4, 44 queries)" % (21 - shrink, cells / 1e6),
                  "ms_alone": round(best * 1e3, 2), "inflight": inflight, "proofs": per * inflight,
                  "proofs_per_s": round(per * inflight / dt, 3), "ms_per_proof": round(dt * 1e3 / (per * inflight), 2),
                  "proof_bytes": len(proof)}))
