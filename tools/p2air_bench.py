"""Device trace generation + proof of the Poseidon2 AIR (298 columns, 282 degree-3 constraints, one permutation per
row): the trace never leaves the device.  Usage: python tools/p2air_bench.py [log_height]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import zkvm_prover_amd as z
from zkvm_prover_amd import air

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
N = 1 << log_n
torch.cuda.set_stream(torch.cuda.Stream(device=0))  # a stream of its own, not the legacy default stream
ctx = z.Context(0)
prog = air.poseidon2_air().program()
airs = [dict(program=prog, log_height=log_n, width=298, n_pvs=0)]
t0 = time.time()
pk = z.ProvingKey(ctx, z.DEFAULT_PARAMS, airs)
print("Poseidon2 AIR 2^%d x 298: keygen %.1f s, proof %d bytes" % (log_n, time.time() - t0, pk.proof_size))
# Montgomery input states straight on the device (any word < p is a valid Montgomery residue)
d_in = torch.randint(0, air.P, (N * 16,), dtype=torch.int32, device="cuda:0")
d_tr = torch.empty(298 * N, dtype=torch.int32, device="cuda:0")
NOPV = np.zeros(0, np.uint32)
for it in range(3):
    ctx.profile_reset()
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.poseidon2_air_tracegen(d_in, log_n, d_tr)
    ctx.sync()
    t1 = time.perf_counter()
    proof = pk.prove([d_tr], [NOPV])
    t2 = time.perf_counter()
    ctx.profile_enable(False)
bytes_w = 298 * N * 4 + 64 * N
print("tracegen %.3f ms (%.2f TB/s of %.2f GB written+read), proof %.2f ms" % ((t1 - t0) * 1e3, bytes_w / (t1 - t0) / 1e12, bytes_w / 1e9,
                                                                             (t2 - t1) * 1e3))
assert z.verify(z.DEFAULT_PARAMS, airs, [NOPV], proof) == 0
for k, (cnt, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1])[:14]:
    print("   %-28s x%-4d %9.3f ms" % (k, cnt, ms))
