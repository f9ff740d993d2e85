#!/usr/bin/env python3
"""Per-stream timeline of a rocprofv3 --kernel-trace CSV of the 3-runner bench: how much of the codec's time is overlapped by another
runner's self-attention, and how the kernel time adds up against the wall clock.
  overlap_timeline.py kernel_trace.csv
Kernels are classed by name (codec: resunit / conv_b3p / convt_b3 / conv1d / snake_split / dac_embed; attention: attn_kernel / attn_walk;
gemm: gemm_tile / gemm16; other).  Interval arithmetic on [start, end) per class over the busiest 60 % of the trace (the timed steps)."""
import csv, sys
from collections import defaultdict

CODEC = ("resunit", "conv_b3p", "convt_b3", "conv1d", "convt1d", "snake_split", "dac_embed", "conv1x1")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    cls = "codec" if any(k in n for k in CODEC) else "attn" if ("attn_kernel" in n or "attn_walk" in n or "attn_rows" in n) else "gemm" if "gemm" in n else "other"
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls, r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo, hi = t0 + (t1 - t0) * 0.35, t0 + (t1 - t0) * 0.95      # skip loading / warm-up
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def inter(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


by = defaultdict(list)
for s, e, c, q, st in rows:
    by[c].append((s, e))
span = rows[-1][1] - rows[0][0]
print(f"window {span / 1e9:.3f} s, {len(rows)} dispatches on {len(set(r[3] for r in rows))} hardware queues")
U = {c: union(v) for c, v in by.items()}
for c in ("codec", "attn", "gemm", "other"):
    if c in by:
        print(f"  {c:6s} sum of kernel durations {sum(e - s for s, e in by[c]) / 1e9:7.3f} s   time with at least one such kernel running {length(U[c]) / 1e9:7.3f} s")
dec = union(by.get("attn", []) + by.get("gemm", []) + by.get("other", []))
both = inter(U.get("codec", []), U.get("attn", []))
print(f"  codec running: {length(U.get('codec', [])) / 1e9:.3f} s; of it with a self-attention kernel of another runner also running: {length(both) / 1e9:.3f} s "
      f"= {100 * length(both) / max(1, length(U.get('codec', []))):.1f} %")
print(f"  any kernel running {length(union([(s, e) for s, e, *_ in rows])) / 1e9:.3f} s of the {span / 1e9:.3f} s window; sum of all kernel durations {sum(e - s for s, e, *_ in rows) / 1e9:.3f} s")
# what co-running costs: mean duration of the attention launches that overlap a codec kernel vs those that do not
ca = U.get("codec", [])
import bisect
starts = [s for s, _ in ca]
ov, free = [], []
for s, e in by.get("attn", []):
    k = bisect.bisect_right(starts, e) - 1
    hit = any(a < e and b > s for a, b in ca[max(0, k - 2):k + 1])
    (ov if hit else free).append(e - s)
if ov and free:
    print(f"  self-attention launches: {len(free)} outside codec time, mean {sum(free) / len(free) / 1e3:.1f} us; {len(ov)} overlapping a codec kernel, mean {sum(ov) / len(ov) / 1e3:.1f} us")
