"""Timing of the LogUp / sum-check building blocks (K6, K7) at large sizes: achieved HBM GB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import zkvm_prover_amd as z

ctx = z.Context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
tabs = [torch.randint(1, z.P, (4 * n,), dtype=torch.int32, device="cuda") for _ in range(3)]
num = torch.randint(0, z.P, (n,), dtype=torch.int32, device="cuda")
r = np.array([3, 5, 7, 11], dtype=np.uint32)
for it in range(3):
    ctx.profile_reset()
    ctx.profile_enable(True)
    f = ctx.mle_fold(tabs[0], n // 2, r)
    s2 = ctx.sumcheck_round(tabs[:2], n // 2)
    s3 = ctx.sumcheck_round(tabs, n // 2)
    inv = ctx.ext_batch_inverse(tabs[0], n)
    run, tot = ctx.logup_running_sum(tabs[1], num, n)
    ctx.profile_enable(False)
    st = ctx.profile_read()
bytes_of = {"mle_fold": 16 * n + 8 * n, "ext_batch_inverse": 32 * n * 2 + 4 * n, "logup_scan": 16 * n * 3 + 16 * n}
for k, (cnt, ms) in sorted(st.items()):
    extra = ""
    if k == "mle_fold":
        extra = "%.0f GB/s" % (bytes_of[k] / ms / 1e6)
    if k == "sumcheck_round":
        extra = "k=2 and k=3 rounds: %.0f GB/s combined" % ((2 + 3) * 16 * n / ms / 1e6)
    if k == "ext_batch_inverse":
        extra = "2 calls: %.0f GB/s" % (bytes_of[k] / ms / 1e6)
    if k == "logup_scan":
        extra = "%.0f GB/s" % (bytes_of[k] / ms / 1e6)
    print("%-22s x%-3d %8.3f ms  %s" % (k, cnt, ms, extra))
