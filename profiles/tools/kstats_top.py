#!/usr/bin/env python3
"""kstats_top.py <rocprofv3 *kernel_stats.csv> [n]: the n kernels with the most time — calls, average, share."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[:n]:
    print("%-100s calls %6s avg %9.2f us  %5s %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
