"""Proof time of a chunk-circuit-shaped AIR set: 42 chips of mixed heights (tallest 2^log_max), ~300 columns in
total, per-chip buses with compound messages and a shared preprocessed range table.  Usage:
python tools/chipset_bench.py [log_max]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import zkvm_prover_amd as z
from zkvm_prover_amd import air

log_max = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cs = air.ChipSet(n_chips=42, log_max=log_max, log_min=max(4, log_max - 10), total_width=300, seed=3, log_table=4)
t0 = time.time()
airs = cs.gen(2)
cells = sum(a["width"] << a["log_height"] for a in airs)
print("42 chips + range table: widths sum %d, heights %s, %.1f M trace cells (trace generation %.1f s)"
      % (sum(cs.widths), sorted(cs.heights, reverse=True)[:8], cells / 1e6, time.time() - t0))
torch.cuda.set_stream(torch.cuda.Stream(device=0))  # a stream of its own, not the legacy default stream
ctx = z.Context(0)
t0 = time.time()
pk = z.ProvingKey(ctx, z.DEFAULT_PARAMS, airs)
print("keygen %.1f s, proof %d bytes" % (time.time() - t0, pk.proof_size))
d = [ctx.upload(a["trace"].reshape(-1)) for a in airs]
pvs = [a["pvs"] for a in airs]
proof = pk.prove(d, pvs)
assert z.verify(z.DEFAULT_PARAMS, pk.verifying_airs(), pvs, proof) == 0
for it in range(2):
    ctx.profile_reset()
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pk.prove(d, pvs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
print("proof %.2f ms (%.2f ns per trace cell)" % (dt * 1e3, dt * 1e9 / cells))
best = 1e9
for it in range(5):  # without the per-kernel events
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pk.prove(d, pvs)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print("proof without profiling events %.2f ms" % (best * 1e3))
for k, (cnt, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1])[:40]:
    print("   %-28s x%-4d %9.3f ms" % (k, cnt, ms))
