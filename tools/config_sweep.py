"""BASELINE.json configs[1] and configs[2] measured through the C ABI on one MI355X:
  #2  BabyBear NTT, N = 2^20..2^24, W in {1, 16, 300} columns (natural order in/out,
      algorithmic bytes = 2*4*N*W, ops = (N/2) log2 N butterflies per column)
  #3  Merkle-Poseidon2 commit of a 2^22 x 300 matrix (bytes = 4*N*W + 32*(2N-1), perms = N*ceil(W/8) + N-1)
Prints one JSON object; each NTT case is self-checked by inverse(forward(x)) == x (bit-exact parity against
the oracle lives in tests/test_gpu_kernels.py).  Usage: python tools/config_sweep.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import zkvm_prover_amd as z


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ctx = z.Context(0)
    ctx.use_torch_stream()
    out = {"device": torch.cuda.get_device_name(0), "ntt": [], "merkle": []}
    for log_n in range(20, 25):
        for width in (1, 16, 300):
            n = 1 << log_n
            x = torch.randint(0, z.P, (width * n,), dtype=torch.int32, device="cuda")
            ref = x.clone()
            reps = 3 if width == 300 else 10
            t_f = timed(lambda: ctx.ntt_batch(x, log_n, width), reps)
            # reps+1 forward transforms were applied; undo them and compare
            for _ in range(reps + 1):
                ctx.ntt_batch(x, log_n, width, inverse=True)
            ok = bool(torch.equal(x, ref))
            t_i = timed(lambda: ctx.ntt_batch(x, log_n, width, inverse=True), reps)
            bytes_alg = 2 * 4 * n * width
            out["ntt"].append({"log_n": log_n, "width": width, "fwd_ms": round(t_f * 1e3, 3), "inv_ms": round(t_i * 1e3, 3),
                               "fwd_alg_GBps": round(bytes_alg / t_f / 1e9, 1),
                               "fwd_butterflies_per_s": float("%.4g" % (width * (n // 2) * log_n / t_f)),
                               "roundtrip_exact": ok})
            del x, ref
            torch.cuda.empty_cache()
    log_n, width = 22, 300
    n = 1 << log_n
    m = torch.randint(0, z.P, (width * n,), dtype=torch.int32, device="cuda")
    roots = []

    def commit():
        t = ctx.merkle_commit([(m, log_n, width)], want_root=True)
        roots.append(bytes(t.root.tobytes()) if hasattr(t, "root") else b"")
        t.close()

    t_m = timed(commit, 3)
    perms = n * ((width + 7) // 8) + n - 1
    out["merkle"].append({"log_n": log_n, "width": width, "ms": round(t_m * 1e3, 3),
                          "alg_GBps": round((4 * n * width + 32 * (2 * n - 1)) / t_m / 1e9, 1),
                          "perms_per_s": float("%.4g" % (perms / t_m)), "deterministic": len(set(roots)) == 1})
    s = json.dumps(out, indent=1)
    print(s)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(s + "\n")


if __name__ == "__main__":
    main()
