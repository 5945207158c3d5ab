#!/usr/bin/env python3
"""Export the per-kernel summary of a rocprofv3 rocpd database (what `--stats` tabulates) to CSV.
    python tools/prof_export.py gpurun_out/prof/smap_results.db profiles/r1_xxx_kernel_stats.csv"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", r[4], r[5], f"{100.0 * r[2] / tot:.2f}"])
print(f"{len(rows)} kernels, total {tot / 1e6:.3f} ms -> {sys.argv[2]}")
