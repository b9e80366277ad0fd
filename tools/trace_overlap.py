"""Which kernels share the GPU in time (rocprofv3 --kernel-trace of `bench.py --inflight 3`): fraction of the wall time
with 0 / 1 / 2 / 3 row-hash kernels active, and what runs when none is.
    python tools/trace_overlap.py D/t_kernel_trace.csv [skip_first_fraction]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
w0 = t_lo + int((t_hi - t_lo) * skip)   # steady state only
ev = []
for s, e, k in rows:
    if e <= w0:
        continue
    ev.append((max(s, w0), 1, k))
    ev.append((e, -1, k))
ev.sort()
active = collections.Counter()
hist = collections.Counter()
other_when_no_hash = collections.Counter()
idle = 0
prev = w0
for t, d, k in ev:
    dt = t - prev
    if dt > 0:
        nh = active["zk::k_hash_rows"]
        hist[nh] += dt
        if sum(active.values()) == 0:
            idle += dt
        elif nh == 0:
            for kk, c in active.items():
                if c > 0:
                    other_when_no_hash[kk] += dt
    active[k] += d
    prev = t
tot = t_hi - w0
print("steady-state window %.1f ms; idle %.1f %%" % (tot / 1e6, 100.0 * idle / tot))
for n in sorted(hist):
    print("  %d row-hash kernels active: %5.1f %%" % (n, 100.0 * hist[n] / tot))
print("  with no row hash active, time share of:")
for k, v in other_when_no_hash.most_common(8):
    print("     %-40s %5.1f %%" % (k[:40], 100.0 * v / tot))
