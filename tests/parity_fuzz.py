"""Randomised parity campaign: HIP prover vs CPU oracle on random AIR sets (random constraint DAGs, random bus
expressions, random preprocessed matrices, random heights / blow-ups / query counts).  Test infrastructure
(uses oracle/ as the checker).  Usage: python tests/parity_fuzz.py [n_cases] [first_seed] [min_log_height] [max_log_height]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as ora
import zkvm_prover_amd as z
from zkvm_prover_amd import air

import test_gpu_logup as tl
import test_gpu_stark as ts

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lh_lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # log2 of the smallest / largest trace height drawn
lh_hi = int(sys.argv[4]) if len(sys.argv) > 4 else 8
zk = z.Context(0)
bad, t0 = 0, time.time()
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    params = (int(rng.integers(1, 4)), 0, int(rng.integers(1, 12)), int(rng.integers(0, 6)), int(rng.integers(0, 6)))
    airs = []
    for k in range(int(rng.integers(1, 5))):
        kind = int(rng.integers(0, 5))
        lh = int(rng.integers(lh_lo, lh_hi + 1))
        if kind == 4:
            # degree-5 definitions (four quotient chunks) need blow-up >= 4; otherwise a degree-3 chip
            w = int(rng.integers(8, 30))
            sa = air.SyntheticAir(width=w, n_free=5, n_bool=2, n_boundary=1, seed=seed * 10 + k, degree=5 if params[0] >= 2 else 3)
            tr, pv = sa.gen_trace(max(lh, 1), seed=seed + k)
            airs.append(dict(program=sa.program(), log_height=max(lh, 1), width=w, n_pvs=len(pv), trace=tr, pvs=pv))
            continue
        if kind == 0:
            airs.append(tl._random_bus_air(seed * 10 + k, lh))
        elif kind == 1:
            airs.append(ts._syn(max(lh, 1), int(rng.integers(8, 40)), 5, seed * 10 + k, n_bool=2, n_boundary=1))
        elif kind == 2:
            airs.append(ts._fib(max(lh, 1)))
        else:
            airs.append(tl._limb(lh, seed=seed + k))
    # every 8th case adds a requester + the Poseidon2 chip on a 24-field bus; the chip's 298 permutation columns are
    # generated ON THE DEVICE (zkhip_poseidon2_air_tracegen), the oracle generates its own
    dev_chip = None
    if case % 8 == 7:
        import torch
        import test_p2air_cpu as tp

        lu, lc = int(rng.integers(max(lh_lo, 1), lh_hi + 1)), int(rng.integers(max(lh_lo, 1), lh_hi + 1))
        n_req = int(rng.integers(2, min(1 << lu, 1 << lc) + 1))
        pair, uniq = tp.hasher_pair(log_user=lu, n_req=n_req, log_chip=lc, seed=seed, bus=int(rng.integers(20, 200)))
        airs += pair
        N = 1 << lc
        d_chip = torch.empty(299 * N, dtype=torch.int32, device="cuda:0")
        zk.poseidon2_air_tracegen(zk.upload(uniq.reshape(-1)), lc, d_chip)
        d_chip[298 * N:] = zk.upload(pair[1]["trace"][298])
        dev_chip = (len(airs) - 1, d_chip)
    # final polynomial length: any value up to the shortest trace of the set (every second case; 0 = a constant)
    if case % 2 == 1:
        lfp_max = min(4, min(a["log_height"] for a in airs))
        params = (params[0], int(rng.integers(0, lfp_max + 1)), params[2], params[3], params[4])
    # every third AIR with at least two columns gets a cached main partition of random width
    for a in airs:
        if a["width"] >= 2 and rng.integers(0, 3) == 0:
            a["program"] = air.with_cached_width(a["program"], int(rng.integers(1, a["width"])))
    # constraint degree 3 needs blow-up >= 2^1: all our AIRs have degree <= 3
    try:
        exp = ora.stark_prove(params, airs)
        pk = z.ProvingKey(zk, params, airs)
        pvs = [a["pvs"] for a in airs]
        d_traces = [zk.upload(a["trace"].reshape(-1)) for a in airs]
        if dev_chip is not None:
            d_traces[dev_chip[0]] = dev_chip[1]
        zk.set_commit_pipeline(int(rng.integers(0, 5)))  # 0 / 1 = plain, 2..4 = pipelined trace commit (tall, wide chips only)
        got = pk.prove(d_traces, pvs)
        vk = pk.verifying_airs()
        ok = got == exp.tobytes() and z.verify(params, vk, pvs, got) == 0
        if ok and params[3] == 0:  # no commit-phase proof-of-work: the proof fits the reference's v1 container
            back, pvs_back = z.proof_from_v1(params, vk, z.proof_to_v1(params, vk, pvs, got))
            ok = back == got and all(list(map(int, x)) == list(map(int, y)) for x, y in zip(pvs_back, pvs))
        pk.close()
    except Exception as e:  # noqa: BLE001
        ok = False
        print("seed", seed, "exception", repr(e)[:200])
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, params, [(a["log_height"], a["width"]) for a in airs])
print("%d cases, %d mismatches, %.1f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
