#!/usr/bin/env python3
"""The aggregation tree through the C++ API at the reference's parameters: N segment proofs of a segment-shaped AIR set (a
counter chip that carries the chained state + 19 chips + range table, tallest 2^log_max) -> leaf nodes (4) -> internal nodes
(3) -> root, `prove_cli prove-agg` (include/zkhip_aggregation.hpp).  Prints the CLI's JSON line (proofs folded per second,
witness / device / self-verification seconds) and checks the root under root.vk.
Usage: python tools/agg_tree_bench.py [n_segments] [log_max]"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import prover_mirror_util as pm  # noqa: E402
from zkvm_prover_amd import air  # noqa: E402

P = air.P


def counter(start, log_n):
    b = air.AirBuilder(1, 2)
    x = b.var(0)
    b.when_first_row(x - b.pub(0))
    b.when_transition(b.next(0) - x - 1)
    b.when_last_row(x - b.pub(1))
    n = 1 << log_n
    tr = ((start + np.arange(n, dtype=np.int64)) % P).astype(np.uint32).reshape(1, n)
    return dict(program=b.program(), log_height=log_n, width=1, n_pvs=2, trace=tr, pvs=np.array([start % P, (start + n - 1) % P], np.uint32))


def main():
    n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    log_max = int(sys.argv[2]) if len(sys.argv) > 2 else 17
    params = (1, 0, 100, 16, 16)
    cs = air.ChipSet(n_chips=19, log_max=log_max, log_min=max(4, log_max - 10), total_width=400, seed=0)
    tmp = tempfile.mkdtemp(prefix="zkhip_agg_")
    try:
        tasks, start, first = [], 7, None
        t0 = time.time()
        for i in range(n_seg):
            airs = [counter(start, 10)] + cs.gen(seed=i + 1)
            start += (1 << 10) - 1
            first = first or airs
            d = os.path.join(tmp, "s%d" % i)
            os.mkdir(d)
            tasks.append(pm.write_task(d, airs, identifier="seg-%d" % i))
        exe, cfg = pm.write_app(tmp, first, params)
        sys.stderr.write("wrote %d segment tasks (%.1f M cells each) in %.1f s\n" % (n_seg, sum(a["width"] << a["log_height"] for a in first) / 1e6, time.time() - t0))
        out = os.path.join(tmp, "out")
        os.mkdir(out)
        r = pm.run_cli("prove-agg", exe, cfg, out, "3", "0:0/0:1", *tasks)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            sys.exit(r.returncode)
        info = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        v = pm.run_cli("verify", os.path.join(out, "root.vk"), cfg, os.path.join(out, "root.json"))
        info["root_verifies_under_root_vk"] = v.returncode == 0
        info["workload"] = "%d segments of 20 chips + counter, tallest 2^%d, 100 queries, PoW 16+16" % (n_seg, log_max)
        info.pop("root_public_values", None)
        print(json.dumps(info))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
