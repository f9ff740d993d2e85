#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 kernel trace: the dispatches behind the N-th last launch of a marker kernel (one per step),
by kernel: launches per step, mean us, us per step; plus the idle time between consecutive dispatches.

  python profiles/tools/trace_steps.py <..._kernel_trace.csv> <marker kernel substring> [steps = 32]
"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
marker, steps = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 32
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
assert len(marks) > steps, (len(marks), steps)
sel = rows[marks[-steps - 1]:marks[-1]]   # `steps` whole steps
t0, t1 = int(sel[0]["Start_Timestamp"]), int(rows[marks[-1]]["Start_Timestamp"])
by = collections.OrderedDict()
busy = 0
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"]
    k = k[:k.index("(")] if "(" in k else k
    e = by.setdefault(k, [0, 0]); e[0] += 1; e[1] += d
    busy += d
print(f"{steps} steps, {(t1 - t0) / steps / 1e3:.1f} us per step, {busy / steps / 1e3:.1f} us in kernels, {len(sel) / steps:.0f} launches per step")
for k, (n, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[:90]:90s} {n / steps:6.1f} per step  {d / n / 1e3:8.2f} us each  {d / steps / 1e3:8.1f} us per step  {100.0 * d / (t1 - t0):5.1f} %")
