"""Ad-hoc stage timing at BASELINE sizes (LDE + Merkle of a 2^log_n x width trace)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import zkvm_prover_amd as z

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
width = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = z.Context(0)
n = 1 << log_n
trace = torch.randint(0, z.P, (width * n,), dtype=torch.int32, device="cuda")
out = torch.empty(width * 2 * n, dtype=torch.int32, device="cuda")
for it in range(3):
    ctx.profile_reset()
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.time()
    ctx.lde_batch(trace, log_n, 1, width, 31, t_out=out)
    torch.cuda.synchronize()
    t1 = time.time()
    tree = ctx.merkle_commit([(out, log_n + 1, width)], want_root=True)
    torch.cuda.synchronize()
    t2 = time.time()
    ctx.profile_enable(False)
    print("iter %d: lde %.2f ms, merkle %.2f ms" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    for k, (cnt, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1]):
        print("   %-28s x%-3d %9.3f ms" % (k, cnt, ms))
    tree.close()
lde_bytes = 4 * n * width * 3
print("LDE algorithmic GB/s: %.1f" % (lde_bytes / (t1 - t0) / 1e9))
hash_bytes = 4 * 2 * n * width + 32 * (4 * n - 1)
print("Merkle algorithmic GB/s: %.1f ; perms/s %.3e" % (hash_bytes / (t2 - t1) / 1e9, (2 * n * ((width + 7) // 8) + 2 * n - 1) / (t2 - t1)))
