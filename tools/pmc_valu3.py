"""profiles/round03_pmc_valu.json from rocprofv3 counter passes of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1`
(separate --pmc runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes): per kernel and per launch the VALU wave-instructions
(SQ_INSTS_VALU), waves, the SQ's VALU-active and busy cycle counters, and the kernel's duration under the profiler; for the kernels
with an ISA mix (tools/isa_mix.py) the ABSOLUTE VALU roofline: wave-instructions x model cycles per wave-instruction / (1024 SIMDs x
2.4 GHz).  Usage: python tools/pmc_valu3.py <dir with *_counter_collection.csv and one *_kernel_trace.csv> N_PROOFS out.json [mix.json ...]"""
import collections
import csv
import glob
import json
import os
import sys

SIMDS, CLOCK_HZ = 1024, 2.4e9


def kname(s):
    return s.split("(")[0].replace("void ", "")


def derive(e):
    """Counter-derived quantities of one kernel.  On gfx950 SQ_ACTIVE_INST_VALU returns the same numbers as SQ_INSTS_VALU (measured here:
    identical for every kernel), so issue utilisation comes from the instruction count and the clock instead: GRBM_GUI_ACTIVE ticks
    once per XCD cycle (8 XCDs), SQ_BUSY_CYCLES once per shader-engine cycle (32 SEs)."""
    if e.get("grbm_gui_active_per_launch") and e.get("ms_per_launch_under_pmc"):
        e["clock_ghz_from_grbm_gui_active"] = round(e["grbm_gui_active_per_launch"] / 8 / (e["ms_per_launch_under_pmc"] * 1e-3) / 1e9, 3)
        cyc = e["grbm_gui_active_per_launch"] / 8
        e["measured_cycles_per_wave_instr_per_simd"] = round(cyc * SIMDS / e["valu_wave_instr_per_launch"], 3)
    if e.get("sq_busy_cycles_per_launch") and e.get("grbm_gui_active_per_launch"):
        e["sq_busy_fraction"] = round(e["sq_busy_cycles_per_launch"] / 32 / (e["grbm_gui_active_per_launch"] / 8), 4)


def main():
    d, n_proofs, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    mixes = {}
    for p in sys.argv[4:]:
        m = json.load(open(p))
        mixes[m["kernel"]] = m
    ctr = collections.defaultdict(collections.Counter)
    launches = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            ctr[r["Counter_Name"]][k] += float(r["Counter_Value"])
            seen.add((k, r.get("Dispatch_Id")))
        if "SQ_INSTS_VALU" in {c for c in ctr}:
            pass
    dur = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[:1]:
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            launches[k] += 1
    valu = ctr["SQ_INSTS_VALU"]
    kernels = {}
    for k in sorted([k for k in valu if k.startswith("zk::") or k == "quot_jit"], key=lambda k: -valu[k])[:24]:
        n = max(1, launches[k])
        e = {"launches_per_proof": round(n / n_proofs, 2), "valu_wave_instr_per_launch": round(valu[k] / n), "waves_per_launch": round(ctr["SQ_WAVES"][k] / n),
             "ms_per_launch_under_pmc": round(dur[k] / n, 4)}
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INST_CYCLES_VMEM"):
            if k in ctr[c]:
                e[c.lower() + "_per_launch"] = round(ctr[c][k] / n)
        derive(e)
        mangled = {"zk::k_hash_rows": "_ZN2zk11k_hash_rowsEPKPKjjmPj"}.get(k)
        if mangled in mixes:
            m = mixes[mangled]
            floor_ms = e["valu_wave_instr_per_launch"] * m["cycles_per_wave_instruction_model"] / (SIMDS * CLOCK_HZ) * 1e3
            e["isa_mix"] = {"multiply_class_fraction": m["multiply_class_fraction"], "cycles_per_wave_instruction_model": m["cycles_per_wave_instruction_model"]}
            e["valu_roofline_ms_per_launch"] = round(floor_ms, 3)
            e["valu_roofline_frac_under_pmc"] = round(floor_ms / e["ms_per_launch_under_pmc"], 4)
            if "measured_cycles_per_wave_instr_per_simd" in e:
                e["issue_efficiency_at_measured_clock"] = round(m["cycles_per_wave_instruction_model"] / e["measured_cycles_per_wave_instr_per_simd"], 4)
        if mangled:
            # which compiled body these counts belong to: bench.py recomputes the hash from the library it loads (tools/code_object_hash.py)
            try:
                sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
                import code_object_hash
                e["code_sha256"], e["code_bytes"] = code_object_hash.kernel_code_sha256(
                    os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zkvm-prover_amd", "libzkhip.so"), mangled)
            except Exception as ex:   # (the counts are still worth keeping)
                e["code_sha256"] = None
                e["code_sha256_note"] = repr(ex)
        kernels[k] = e
    # the whole proof against the same absolute floor: every kernel's VALU wave-instructions priced at the row hash's model cycles
    total_instr = sum(e["launches_per_proof"] * e["valu_wave_instr_per_launch"] for e in kernels.values())
    cpi = next((e["isa_mix"]["cycles_per_wave_instruction_model"] for e in kernels.values() if "isa_mix" in e), None)
    totals = {"total_valu_wave_instr_per_proof": round(total_instr)}
    if cpi:
        totals["total_valu_roofline_ms_per_proof"] = round(total_instr * cpi / (SIMDS * CLOCK_HZ) * 1e3, 3)
    json.dump({**totals, "source": "tools/pmc_valu3.py over separate rocprofv3 --pmc passes of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                         "--inflight 1` (%d proofs per run)" % n_proofs,
               "peak": {"simds": SIMDS, "clock_hz": CLOCK_HZ},
               "note": "valu_roofline_ms_per_launch = SQ_INSTS_VALU x (4.2 cycles for the multiply-class share + 2.2 for the rest; shares from the "
                       "kernel's ISA, tools/isa_mix.py; costs from tools/ubench_valu.hip) / (1024 SIMDs x 2.4 GHz): an absolute floor, priced with "
                       "no kernel's own rate",
               "counters_seen": sorted(ctr.keys()), "kernels": kernels}, open(out, "w"), indent=1)
    for k in list(kernels)[:6]:
        print(k, kernels[k])


if __name__ == "__main__":
    main()
