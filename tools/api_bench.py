#!/usr/bin/env python3
"""The headline workload through the C++ Prover API instead of bench.py's harness: writes the app (AIR programs +
openvm.toml) and ONE task (the 2^22 x 300 + 2^22 x 2 witness) to a scratch directory and runs
`prove_cli bench-many` = BatchProver::prove_repeated (include/zkhip_prover.hpp): `inflight` Provers on the GPU, the
witness uploaded once per lane, n proofs, every proof self-verified (as Prover::gen_proof_stark must,
crates/prover/src/prover/mod.rs:407-411).  Prints the CLI's JSON line.
Usage: python tools/api_bench.py [--log-rows 22] [--width 300] [--proofs 24] [--inflight 3]"""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from zkvm_prover_amd import air  # noqa: E402

TOML = """[app_fri_params.fri_params]
log_blowup = 1
log_final_poly_len = 0
num_queries = 100
commit_proof_of_work_bits = 16
query_proof_of_work_bits = 16
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--width", type=int, default=300)
    ap.add_argument("--proofs", type=int, default=24)
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    w, s = args.width, args.log_rows
    sa = air.SyntheticAir(width=w, n_free=max(4, w // 5), n_bool=min(16, max(1, w // 20)), n_boundary=min(8, max(1, w // 40)), seed=0)
    fa = air.fibonacci_air()
    t0 = time.time()
    try:
        import torch

        tr, pv = sa.gen_trace(s, seed=1000, xp="torch", device="cuda")
        tr = tr.cpu().numpy().view(np.uint32)
        del torch
    except Exception:
        tr, pv = sa.gen_trace(s, seed=1000)
    ftr, fpv = air.fibonacci_trace(s, a0=0, b0=1)
    tmp = tempfile.mkdtemp(prefix="zkhip_api_bench_")
    exe, cfg, task = os.path.join(tmp, "app.zkair"), os.path.join(tmp, "openvm.toml"), os.path.join(tmp, "task.bin")
    words = [0x58414B5A, 2]
    for prog, width, n_pvs in ((sa.program(), w, len(pv)), (fa.program(), 2, 3)):
        words += [width, n_pvs, len(prog)] + [int(x) for x in prog]
    np.array(words, dtype=np.uint32).tofile(exe)
    open(cfg, "w").write(TOML)
    with open(task, "wb") as f:
        ident = b"bench-chunk"
        f.write(struct.pack("<I", len(ident)) + ident + struct.pack("<I", 2))
        for lh, pvs, trace in ((s, pv, tr), (s, fpv, ftr)):
            head = np.concatenate([np.array([lh, len(pvs)], dtype=np.uint32), np.asarray(pvs, dtype=np.uint32)])
            body = np.ascontiguousarray(trace, dtype=np.uint32).reshape(-1)
            f.write(struct.pack("<Q", 4 * (head.size + body.size)))
            head.tofile(f)
            body.tofile(f)
    del tr
    sys.stderr.write("wrote %s (%.1f GB) in %.1f s\n" % (task, os.path.getsize(task) / 1e9, time.time() - t0))
    cli = os.path.join(ROOT, "zkvm-prover_amd", "prove_cli")
    r = subprocess.run([cli, "bench-many", exe, cfg, task, str(args.proofs), str(args.inflight), str(args.gpus)], capture_output=True, text=True)
    for fpath in (exe, cfg, task):
        os.remove(fpath)
    os.rmdir(tmp)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        sys.exit(r.returncode)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    out["workload"] = "2^%d x %d degree-3 AIR + 2^%d x 2 Fibonacci AIR, reference FRI parameters" % (s, w, s)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
