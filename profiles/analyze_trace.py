#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel stats + inter-kernel gaps inside the AR loop."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "--grid" in sys.argv:
            name = f'{name[:48]} g=({r["Grid_Size_X"]},{r["Grid_Size_Y"]},{r["Grid_Size_Z"]}) wg={r["Workgroup_Size_X"]} lds={r["LDS_Block_Size"]} v={r["VGPR_Count"]}+{r["Accum_VGPR_Count"]}'
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
dur = defaultdict(list)
for s, e, n in rows:
    dur[n].append(e - s)
print(f"{'kernel':110s} {'calls':>7s} {'avg_us':>8s} {'min_us':>8s} {'total_ms':>9s}")
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:110]:110s} {len(d):7d} {sum(d)/len(d)/1e3:8.2f} {min(d)/1e3:8.2f} {sum(d)/1e6:9.2f}")
# gaps between consecutive kernels (only short ones = back-to-back graph nodes)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
short = [g for g in gaps if -5000 < g < 20000]
if short:
    short.sort()
    print(f"\nback-to-back gaps: n={len(short)} mean={sum(short)/len(short)/1e3:.2f}us median={short[len(short)//2]/1e3:.2f}us p90={short[int(len(short)*0.9)]/1e3:.2f}us")
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"kernel busy {busy/1e6:.1f} ms over span {span/1e6:.1f} ms")
