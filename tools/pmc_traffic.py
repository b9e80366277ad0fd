"""Builds profiles/roundNN_pmc_traffic.json from two rocprofv3 counter passes over the same bench command:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d A -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d B -o w --output-format csv -- python3 bench.py ... (same)
    python tools/pmc_traffic.py A/f_counter_collection.csv B/w_counter_collection.csv N_PROOFS out.json
HBM traffic per kernel per proof = FETCH_FACTOR * FETCH_SIZE + WRITE_SIZE (both reported in KiB... see `correction`)."""
import collections
import csv
import json
import sys

CORRECTION = ("traffic = 2*FETCH_SIZE + WRITE_SIZE for kernels whose reads are full-wave coalesced (>=256 B per wave "
              "request): calibrated on k_bitrev_scale_tiled (reads 5.03 GB of 128-B rows per LDE, FETCH_SIZE reports "
              "2.47 GB) and consistent with MI355X_MICROARCH.md (FETCH_SIZE = 1/2 of coalesced streaming reads on gfx950). "
              "The NTT pass kernels read 32/64-byte segments; a separate calibration (profiles/round01_ntt_tile_pmc.txt) "
              "shows FETCH_SIZE is NOT halved there, so their traffic is FETCH_SIZE + WRITE_SIZE.  Counter unit: KiB.")


def load(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    f_csv, w_csv, n_proofs, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, fc = load(f_csv, "FETCH_SIZE")
    write, _ = load(w_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(fetch, key=lambda k: -(fetch[k] + write.get(k, 0))):
        if not (k.startswith("zk::") or k.startswith("void zk::") or k == "quot_jit"):
            continue
        factor = 1 if "k_ntt_pass4" in k else 2
        f_kb, w_kb = fetch[k] / n_proofs, write.get(k, 0.0) / n_proofs
        kernels[k.replace("void ", "")] = {"FETCH_SIZE_KB_per_proof": round(f_kb, 3), "WRITE_SIZE_KB_per_proof": round(w_kb, 3),
                                           "launches_per_proof": round(fc[k] / n_proofs, 2), "fetch_factor": factor,
                                           "hbm_bytes_per_proof_corrected": round((factor * f_kb + w_kb) * 1024)}
    json.dump({"source": "tools/pmc_traffic.py over two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE) of "
                         "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1` (%d proofs per run)" % n_proofs,
               "correction": CORRECTION, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:8]:
        print("%-40s %8.2f GB per proof" % (k[:40], v["hbm_bytes_per_proof_corrected"] / 1e9))


if __name__ == "__main__":
    main()
