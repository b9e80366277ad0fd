#!/usr/bin/env python3
"""Per-kernel statistics (calls, total / average / min / max duration) from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace
--stats` without --output-format csv writes <name>_results.db): the same table as rocprofv3's *_kernel_stats.csv.
Usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("""select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                          from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                          group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    out = [["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]]
    for name, calls, tot, mn, mx in rows:
        out.append([name, calls, tot, round(tot / calls, 1), round(100.0 * tot / total, 3), mn, mx])
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)
    sys.stderr.write("%d kernels, %d launches, %.3f ms of kernel time over a span of %.3f ms\n"
                     % (len(rows), sum(r[1] for r in rows), total / 1e6, (span[1] - span[0]) / 1e6))


if __name__ == "__main__":
    main()
