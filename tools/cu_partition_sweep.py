#!/usr/bin/env python3
"""Single-proof latency and in-flight rate of the headline workload (2^22 x 300 + Fibonacci, bench.py's instance) for the
pipelined trace commit with and without a CU partition (zkhip_set_commit_pipeline x zkhip_set_cu_partition): the LDE on a
stream masked to `side_cus` CUs, the pipeline's row sponge on the remaining CUs.  Proof bytes must not depend on either.
Usage: python tools/cu_partition_sweep.py [--log-rows 22] [--width 300] [out.json]"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import zkvm_prover_amd as z
from zkvm_prover_amd import air


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--width", type=int, default=300)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("out", nargs="?")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    log_n, width = args.log_rows, args.width
    sa = air.SyntheticAir(width=width, n_free=max(4, width // 5), n_bool=min(16, max(1, width // 20)),
                          n_boundary=min(8, max(1, width // 40)), seed=0)
    fa = air.fibonacci_air()
    airs = [dict(program=sa.program(), log_height=log_n, width=width, n_pvs=sa.n_pvs),
            dict(program=fa.program(), log_height=log_n, width=2, n_pvs=3)]
    pipes = []
    for i in range(args.inflight):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            c = z.Context(0)
            tr, pv = sa.gen_trace(log_n, seed=1000 + i, xp="torch", device=dev)
            d_trace = tr.reshape(-1).contiguous()
            del tr
            c._check(c.lib.zkhip_to_monty(c.h, d_trace.data_ptr(), d_trace.numel()))
            ftr, fpv = air.fibonacci_trace(log_n, a0=i, b0=1)
            pipes.append(dict(ctx=c, st=st, pk=z.ProvingKey(c, z.DEFAULT_PARAMS, airs), traces=[d_trace, c.upload(ftr.reshape(-1))], pvs=[pv, fpv]))
    torch.cuda.synchronize()

    def one(i):
        p = pipes[i]
        p["pk"].prove_async(p["traces"], p["pvs"])
        return p["pk"].fetch()

    ref = [hashlib.sha256(bytes(one(i))).hexdigest() for i in range(len(pipes))]
    rows = []
    for parts, side in [(0, 0), (4, 0), (4, 32), (4, 48), (4, 64), (4, 96), (6, 48), (6, 64), (8, 64), (3, 64), (2, 64), (4, 128)]:
        for p in pipes:
            p["ctx"].set_commit_pipeline(parts)
            p["ctx"].set_cu_partition(side)
        same = all(hashlib.sha256(bytes(one(i))).hexdigest() == ref[i] for i in range(len(pipes)))  # also warms the streams
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            one(0)
        alone = (time.perf_counter() - t0) / args.reps
        res = {"commit_parts": parts, "side_cus": side, "proof_bytes_unchanged": same, "ms_one_proof_alone": round(alone * 1e3, 2)}
        for nf in (2, 3):
            if nf > len(pipes):
                continue
            n_steps = 4 * nf
            t0 = time.perf_counter()
            for i in range(nf):
                pipes[i]["pk"].prove_async(pipes[i]["traces"], pipes[i]["pvs"])
            for i in range(n_steps):
                pipes[i % nf]["pk"].fetch()
                if i + nf < n_steps:
                    pipes[i % nf]["pk"].prove_async(pipes[i % nf]["traces"], pipes[i % nf]["pvs"])
            torch.cuda.synchronize()
            res["ms_per_proof_%d_inflight" % nf] = round((time.perf_counter() - t0) / n_steps * 1e3, 2)
        rows.append(res)
        sys.stderr.write(json.dumps(res) + "\n")
    out = {"device": torch.cuda.get_device_name(0), "workload": "2^%d x %d + Fibonacci, reference FRI parameters" % (log_n, width), "rows": rows}
    s = json.dumps(out, indent=1)
    print(s)
    if args.out:
        open(args.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
