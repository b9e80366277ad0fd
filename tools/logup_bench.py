"""Stage times of a proof whose AIRs talk over buses (LogUp phase), through the C ABI on one MI355X:
2^log_n lookups of (key, value) pairs into a 2^12-row table next to a 2^log_n-row AIR with six
interactions of up to eight fields.  Usage: python tools/logup_bench.py [log_n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import zkvm_prover_amd as z
from zkvm_prover_amd import air

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nopv = np.zeros(0, np.uint32)
s, t = air.lookup_traces(log_n, 12, seed=1, sender_width=16)
mt, mpv = air.bus_mix_trace(log_n, seed=2)
airs = [dict(program=air.lookup_sender_air(16).program(), log_height=log_n, width=16, n_pvs=0, trace=s, pvs=nopv),
        dict(program=air.bus_mix_air().program(), log_height=log_n, width=6, n_pvs=1, trace=mt, pvs=mpv),
        dict(program=air.lookup_table_air().program(), log_height=12, width=3, n_pvs=0, trace=t, pvs=nopv)]
ctx = z.Context(0)
pk = z.ProvingKey(ctx, z.DEFAULT_PARAMS, airs)
d = [ctx.upload(a["trace"].reshape(-1)) for a in airs]
pvs = [a["pvs"] for a in airs]
proof = pk.prove(d, pvs)
assert z.verify(z.DEFAULT_PARAMS, airs, pvs, proof) == 0
for it in range(2):
    ctx.profile_reset()
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pk.prove(d, pvs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
print("proof %.2f ms, %d bytes; 7 interactions over 2 x 2^%d rows + 1 over 2^12" % (dt * 1e3, len(proof), log_n))
for k, (cnt, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1])[:16]:
    print("   %-28s x%-4d %9.3f ms" % (k, cnt, ms))
