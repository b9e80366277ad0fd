"""GPU busy time vs wall time of the last proof in a rocprofv3 kernel trace (how much of a proof is launch gaps):
    rocprofv3 --kernel-trace -d D -o t --output-format csv -- python3 tools/chipset_bench.py 20
    python tools/trace_gaps.py D/t_kernel_trace.csv [min_gap_ms_between_proofs]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
split_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
# proofs are separated by host-side gaps (verification, timing code) of milliseconds
groups, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - max(x[1] for x in cur[-8:]) > split_ns:
        groups.append(cur)
        cur = []
    cur.append(b)
groups.append(cur)
g = max(groups[-3:], key=len) if len(groups) >= 3 else groups[-1]
t0, t1 = g[0][0], max(x[1] for x in g)
# union of busy intervals
busy, end = 0, t0
for s, e, _ in g:
    if e <= end:
        continue
    busy += e - max(s, end)
    end = e
gaps = sorted(((b[0] - a[1]) for a, b in zip(g, g[1:]) if b[0] > a[1]), reverse=True)
print("kernels %d, wall %.2f ms, busy %.2f ms (%.1f %%), idle %.2f ms; median gap %.1f us, gaps > 20 us: %d (%.2f ms)"
      % (len(g), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), (t1 - t0 - busy) / 1e6,
         (gaps[len(gaps) // 2] / 1e3 if gaps else 0), sum(1 for x in gaps if x > 20000), sum(x for x in gaps if x > 20000) / 1e6))
