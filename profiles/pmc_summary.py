#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per launch.
HBM bytes per launch = FETCH_SIZE * 1024 * 2 (gfx950 wide-read correction, MI355X_MICROARCH.md §HBM)
                     + WRITE_SIZE * 1024 (uncalibrated)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Kernel Name") or "?"
            cn = r.get("Counter_Name") or r.get("Counter Name")
            cv = float(r.get("Counter_Value") or r.get("Counter Value") or 0)
            a = acc[name][cn]
            a[0] += cv
            a[1] += 1
print(f"{'kernel':80s} {'counter':14s} {'launches':>9s} {'mean/launch':>14s} {'KB->bytes(corr)':>18s}")
for name, cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
    for cn, (tot, n) in cs.items():
        mean = tot / max(n, 1)
        corr = mean * 1024 * (2 if cn == "FETCH_SIZE" else 1)
        print(f"{name[:80]:80s} {cn:14s} {n:9d} {mean:14.2f} {corr:18.0f}")
