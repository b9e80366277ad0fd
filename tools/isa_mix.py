#!/usr/bin/env python3
"""Dynamic VALU instruction mix of a kernel, estimated from its gfx950 assembly (hipcc -S --cuda-device-only): instructions of the
innermost loops are weighted by their trip counts (given per kernel: the round loops of the Poseidon2 permutation run 4 / 13 / 4
times), everything else once.  Classes follow the issue costs measured with tools/ubench_valu.hip on MI355X:
multiply-class (v_mad_u64_u32, v_mad_i64_i32, v_mul_lo_u32, v_mul_hi_u32, v_min_u32, v_lshl_add_u32, 64-bit shifts) 4.2 cycles per
wave64 instruction, full-rate VOP1 / VOP2 / VOP3 integer ops 2.2.
Usage: python tools/isa_mix.py <file.s> <mangled kernel name> <trip counts, comma separated> [out.json]"""
import json
import re
import sys

MUL_CLASS = ("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_min_u32", "v_max_u32", "v_lshl_add_u32", "v_lshl_add_u64",
             "v_lshlrev_b64", "v_lshrrev_b64", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_i32_i24")
CYC_MUL, CYC_FULL = 4.2, 2.2


def main():
    path, kernel, trips = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3].split(",")]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start + 1:end]
    label_at = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            loops.append((label_at[m.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    inner.sort()
    if len(inner) != len(trips):
        sys.exit("kernel has %d innermost loops, %d trip counts given" % (len(inner), len(trips)))
    weight = [1] * len(body)
    for (a, b), t in zip(inner, trips):
        for i in range(a, b + 1):
            weight[i] = t
    mul = full = 0.0
    hist = {}
    for l, w in zip(body, weight):
        m = re.match(r"^\s+(v_\w+)", l)
        if not m:
            continue
        op = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", m.group(1))
        hist[op] = hist.get(op, 0) + w
        if op.startswith(MUL_CLASS):
            mul += w
        else:
            full += w
    frac = mul / (mul + full)
    out = {"kernel": kernel, "innermost_loop_trip_counts": trips, "weighted_valu_instructions": mul + full, "multiply_class_fraction": round(frac, 4),
           "cycles_per_wave_instruction_model": round(CYC_MUL * frac + CYC_FULL * (1 - frac), 3),
           "class_costs_cycles": {"multiply_class": CYC_MUL, "full_rate": CYC_FULL},
           "top_opcodes": dict(sorted(hist.items(), key=lambda kv: -kv[1])[:12])}
    s = json.dumps(out, indent=1)
    print(s)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(s + "\n")


if __name__ == "__main__":
    main()
