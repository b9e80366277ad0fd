"""Latency of the serial sponge: observe n words into the device transcript (what the prover does with the opened values).
Usage: python tools/transcript_bench.py [n_words]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import zkvm_prover_amd as z

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
ctx = z.Context(0)
rng = np.random.default_rng(1)
vals = rng.integers(0, 2013265921, size=n, dtype=np.uint32)
for misalign in (0, 3):
    best = 1e9
    for it in range(6):
        tr = ctx.transcript()
        if misalign:
            tr.observe(vals[:misalign])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.observe(vals)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        s = tr.sample(4)
    print("observe %d words (buffer offset %d): %.3f ms = %.2f us per permutation; sample %s" % (n, misalign, best * 1e3, best * 1e6 / (n / 8), s.tolist()))
