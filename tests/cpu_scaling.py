#!/usr/bin/env python3
"""How the optimised CPU prover (oracle/fast) scales with host threads on this machine: the bench workload at a reduced
height, OMP_NUM_THREADS swept.  Used to pick the thread count bench.py's cpu_baseline runs with."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ may drive the oracle; tools/ may not
CHILD = r"""
import sys, time, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as ora
from zkvm_prover_amd import air
s = int(sys.argv[1]); width = 300
sa = air.SyntheticAir(width=width, n_free=60, n_bool=15, n_boundary=7, seed=0)
tr, pv = sa.gen_trace(s, seed=1)
ftr, fpv = air.fibonacci_trace(s)
airs = [dict(program=sa.program(), log_height=s, width=width, n_pvs=len(pv), trace=tr, pvs=pv),
        dict(program=air.fibonacci_air().program(), log_height=s, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
ora.fast_lib()
ora.fast_stark_prove((1, 0, 100, 16, 16), airs, cap_words=1 << 22)
t0 = time.time(); ora.fast_stark_prove((1, 0, 100, 16, 16), airs, cap_words=1 << 22); print("%%.3f" %% (time.time() - t0))
""" % (ROOT, ROOT)

if __name__ == "__main__":
    log_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            print(f, open(f).read().strip())
        except OSError:
            pass
    print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for nt in (8, 16, 32, 64, 128, 256):
        if nt > (os.cpu_count() or 1):
            break
        for extra in ({}, {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}):
            env = dict(os.environ, OMP_NUM_THREADS=str(nt), **extra)
            env.pop("FAST_ORACLE_TIMING", None)
            r = subprocess.run([sys.executable, "-c", CHILD, str(log_rows)], env=env, capture_output=True, text=True)
            print("threads %3d %-40s 2^%d rows: %s s" % (nt, str(extra), log_rows, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]))
