#!/usr/bin/env python3
"""HBM bytes per decode STEP (or per launch of one kernel family) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an eager run:
   pmc_step.py NAME once_per_step_kernel_substring fetch.csv write.csv [family_substring]
Sums the counters over every dispatch of the run and divides by the number of dispatches of the kernel that runs once per step (so prefill /
warm-up forwards count as steps: keep them few).  With family_substring: bytes per launch of the kernels whose name contains it instead.
bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half the bytes of wide reads on gfx950).
Appends {NAME: ...} to $PMC_JSON_OUT (profiles/pmc_traffic_secondary.json)."""
import csv, json, os, sys

name, once, fcsv, wcsv = sys.argv[1:5]
fam = sys.argv[5] if len(sys.argv) > 5 else None


SKIP = int(os.environ.get("PMC_SKIP_STEPS", "0"))   # leave out everything before the (SKIP + 1)-th step marker: loads, encoder passes, prefill, warm-up


def total(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if (r.get("Counter_Name") or r.get("Counter Name")) != counter:
            continue
        rows.append((int(r.get("Dispatch_Id") or r.get("Dispatch Id") or len(rows)), r.get("Kernel_Name") or r.get("Kernel Name") or "",
                     float(r.get("Counter_Value") or r.get("Counter Value") or 0)))
    rows.sort()
    tot, steps, famtot, famn, seen = 0.0, 0, 0.0, 0, 0
    for _, k, v in rows:
        seen += once in k
        if seen <= SKIP:
            continue
        tot += v
        steps += once in k
        if fam and fam in k:
            famtot += v
            famn += 1
    return tot, steps, famtot, famn


f, sf, ff, nf = total(fcsv, "FETCH_SIZE")
w, sw, fw, nw = total(wcsv, "WRITE_SIZE")
if fam:
    rec = {"per": f"launch of kernels matching '{fam}'", "launches": nf, "fetch_kb": ff / max(nf, 1), "write_kb": fw / max(nw, 1),
           "hbm_bytes": ff / max(nf, 1) * 2048 + fw / max(nw, 1) * 1024}
else:
    rec = {"per": f"decode step (dispatches of '{once}' counted as steps)", "steps": sf, "fetch_kb": f / max(sf, 1), "write_kb": w / max(sw, 1),
           "hbm_bytes": f / max(sf, 1) * 2048 + w / max(sw, 1) * 1024}
rec["method"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate eager passes; bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024"
print(name, json.dumps(rec))
out = os.environ.get("PMC_JSON_OUT")
if out:
    d = json.load(open(out)) if os.path.exists(out) else {}
    d[name] = rec
    json.dump(d, open(out, "w"), indent=1)
