#!/usr/bin/env python3
"""rocprofv3's default output is a rocpd SQLite database: write the per-kernel statistics of one as the CSV that
`rocprofv3 --stats --output-format csv` produces (what scripts/kstats.py reads).
    python scripts/rocpd_stats.py <results.db> <kernel_stats.csv>"""
import csv
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
per = {}
for name, dur in db.execute("select name, duration from kernels"):
    per.setdefault(name, []).append(dur)
total = sum(sum(v) for v in per.values())
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / total, 4), min(v), max(v),
                    statistics.pstdev(v) if len(v) > 1 else 0.0])
print(f"{len(per)} kernels, {total / 1e6:.2f} ms of kernel time")
