"""Builds profiles/roundNN_pmc_valu.json from one rocprofv3 counter pass:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d D -o v --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1
    python tools/pmc_valu.py D/v_counter_collection.csv D/v_kernel_trace.csv N_PROOFS out.json
Per kernel and per proof: VALU instructions issued (wave granularity), waves, kernel time, and the time the same
instructions would take at the row-hash kernel's measured issue rate -- a whole-proof VALU roofline."""
import collections
import csv
import json
import sys


def main():
    cc, kt, n_proofs, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    valu, waves = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            valu[k] += float(r["Counter_Value"])
        elif r["Counter_Name"] == "SQ_WAVES":
            waves[k] += float(r["Counter_Value"])
    dur = collections.Counter()
    for r in csv.DictReader(open(kt)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    ours = [k for k in valu if k.startswith("zk::") or k == "quot_jit"]
    hash_k = "zk::k_hash_rows"
    rate = valu[hash_k] / (dur[hash_k] * 1e-3)  # wave-instructions per second, whole GPU
    kernels, tot_i, tot_ms = {}, 0.0, 0.0
    for k in sorted(ours, key=lambda k: -valu[k]):
        i = valu[k] / n_proofs
        kernels[k] = {"valu_wave_instr_per_proof": round(i), "waves_per_proof": round(waves[k] / n_proofs),
                      "ms_per_proof_under_pmc": round(dur[k] / n_proofs, 3), "ms_at_hash_issue_rate": round(i / rate * 1e3, 3)}
        tot_i += i
        tot_ms += i / rate * 1e3
    json.dump({"source": "tools/pmc_valu.py over rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES of `python3 bench.py --steps 1 "
                         "--warmup 0 --no-cpu-baseline --inflight 1` (%d proofs per run)" % n_proofs,
               "hash_issue_rate_wave_instr_per_s": rate,
               "total_valu_wave_instr_per_proof": round(tot_i), "total_ms_at_hash_issue_rate": round(tot_ms, 2),
               "note": "total_ms_at_hash_issue_rate is the time one proof's VALU instructions need when every SIMD issues at the "
                       "rate the row-hash kernel (the best-utilised kernel) achieves: the VALU roofline of a whole proof",
               "kernels": kernels}, open(out, "w"), indent=1)
    print("total %.3g wave-instructions per proof -> %.1f ms at the hash kernel's issue rate" % (tot_i, tot_ms))
    for k in list(kernels)[:8]:
        print("  %-36s %8.3g  %6.2f ms" % (k[:36], kernels[k]["valu_wave_instr_per_proof"], kernels[k]["ms_at_hash_issue_rate"]))


if __name__ == "__main__":
    main()
