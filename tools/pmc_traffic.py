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
              "shows FETCH_SIZE is NOT halved there.  Round 5 (VERDICT round 4 weak 3): a transform pass reads exactly what it writes (or half "
              "of it: the forward transform's first pass reads the un-extended coefficients), and round 4's factor 1 for EVERY pass booked "
              "17.4 GB of reads beside 31.6 GB of writes -- impossible.  The passes differ: the second pass of a transform reads contiguous "
              "8 KiB runs (full lines: FETCH_SIZE halved), the first reads 32 / 64-byte segments (not halved).  The factor is therefore chosen "
              "bounded PER DISPATCH: true reads lie in [FETCH_SIZE, 2 FETCH_SIZE] and are at least the pass's algorithmic reads (= its WRITE_SIZE; "
              "half of it for the forward transform's first pass; an LDE is four dispatches in a fixed order and the two counter passes run the "
              "same deterministic program).  `hbm_bytes_per_proof_corrected` is the LOWER bound, `..._upper_bound` the upper.  Counter unit: KiB.")


def load(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def per_dispatch(path, counter):
    """{kernel: [value of dispatch 0, 1, ...]} in dispatch order"""
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
    out = collections.defaultdict(list)
    for _, k, v in sorted(rows):
        out[k].append(v)
    return out


def ntt_reads(fetch_list, write_list):
    """Bounds on the reads (KiB) of the transform-pass kernel over its dispatches.  An LDE is four dispatches in a fixed order -- inverse
    pass 1, inverse pass 2, forward pass 1 (reads the un-extended coefficients: HALF of what it writes), forward pass 2 -- and every other
    pass reads exactly what it writes.  FETCH_SIZE is halved for full-line reads and not for 32 / 64-byte segments, so per dispatch the true
    reads lie in [FETCH_SIZE, 2 FETCH_SIZE] and cannot be less than the pass's algorithmic reads.  Returns (lower bound, upper bound,
    algorithmic reads, implied factor per pass kind = algorithmic reads / FETCH_SIZE)."""
    lower = upper = alg = 0.0
    f_kind, a_kind = [0.0] * 4, [0.0] * 4
    for j, (f, w) in enumerate(zip(fetch_list, write_list)):
        r_alg = w / 2 if j % 4 == 2 else w
        lower += min(max(f, r_alg), 2 * f)
        upper += 2 * f
        alg += r_alg
        f_kind[j % 4] += f
        a_kind[j % 4] += r_alg
    return lower, upper, alg, [round(a / f, 3) if f else None for a, f in zip(a_kind, f_kind)]


def main():
    f_csv, w_csv, n_proofs, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, fc = load(f_csv, "FETCH_SIZE")
    write, _ = load(w_csv, "WRITE_SIZE")
    fetch_d, write_d = per_dispatch(f_csv, "FETCH_SIZE"), per_dispatch(w_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(fetch, key=lambda k: -(fetch[k] + write.get(k, 0))):
        if not (k.startswith("zk::") or k.startswith("void zk::") or k == "quot_jit"):
            continue
        f_kb, w_kb = fetch[k] / n_proofs, write.get(k, 0.0) / n_proofs
        entry = {"FETCH_SIZE_KB_per_proof": round(f_kb, 3), "WRITE_SIZE_KB_per_proof": round(w_kb, 3), "launches_per_proof": round(fc[k] / n_proofs, 2)}
        if "k_ntt_pass4" in k and len(fetch_d[k]) == len(write_d.get(k, [])):
            lo, hi, alg, implied = ntt_reads(fetch_d[k], write_d[k])
            entry.update({"fetch_factor": "per dispatch, between 1 and 2", "implied_factor_by_pass_kind": dict(zip(("inverse_pass1", "inverse_pass2", "forward_pass1", "forward_pass2"), implied)),
                          "read_KB_per_proof_lower_bound": round(lo / n_proofs, 3), "read_KB_per_proof_upper_bound": round(hi / n_proofs, 3),
                          "algorithmic_read_KB_per_proof": round(alg / n_proofs, 3),
                          "hbm_bytes_per_proof_corrected": round((lo / n_proofs + w_kb) * 1024), "hbm_bytes_per_proof_upper_bound": round((hi / n_proofs + w_kb) * 1024),
                          "traffic_over_algorithmic": [round((lo / n_proofs + w_kb) / (alg / n_proofs + w_kb), 3), round((hi / n_proofs + w_kb) / (alg / n_proofs + w_kb), 3)]})
        else:
            factor = 1 if "k_ntt_pass4" in k else 2   # (dispatch lists of the two passes differ: round 4's rule)
            entry.update({"fetch_factor": factor, "hbm_bytes_per_proof_corrected": round((factor * f_kb + w_kb) * 1024)})
        kernels[k.replace("void ", "")] = entry
    json.dump({"source": "tools/pmc_traffic.py over two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE) of "
                         "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1` (%d proofs per run)" % n_proofs,
               "correction": CORRECTION, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:8]:
        print("%-40s %8.2f GB per proof" % (k[:40], v["hbm_bytes_per_proof_corrected"] / 1e9))


if __name__ == "__main__":
    main()
