#!/usr/bin/env python3
"""Per kernel name: the mean of every counter in the rocprofv3 --pmc counter_collection CSVs given (one file per counter group)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError as e:
        print(f"# {path}: {e}"); continue
    for r in rows:
        name = r.get("Kernel_Name", "")
        if "gemm_tile_kernel" not in name: continue
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for name, ctrs in acc.items():
    print(name)
    for c, (s, n) in sorted(ctrs.items()):
        print(f"    {c:32s} mean per dispatch {s / n:16.1f}   ({n} dispatches)")
    g = {c: s / n for c, (s, n) in ctrs.items()}
    if "SQ_WAVE_CYCLES" in g and "SQ_WAVES" not in g: pass
    if "SQ_LDS_BANK_CONFLICT" in g and g.get("SQ_LDS_IDX_ACTIVE"):
        print(f"    -> LDS bank-conflict cycles / LDS active cycles = {g['SQ_LDS_BANK_CONFLICT'] / g['SQ_LDS_IDX_ACTIVE']:.3f}")
    if "SQ_WAIT_INST_ANY" in g and g.get("SQ_WAIT_ANY") is not None and g.get("SQ_ACTIVE_INST_ANY"):
        tot = g["SQ_WAIT_INST_ANY"] + g["SQ_WAIT_ANY"] + g["SQ_ACTIVE_INST_ANY"]
        print(f"    -> of wave cycles: parked at s_waitcnt / barrier {g['SQ_WAIT_ANY'] / tot:.2f}, issue stalls {g['SQ_WAIT_INST_ANY'] / tot:.2f}, issuing {g['SQ_ACTIVE_INST_ANY'] / tot:.2f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and g.get("SQ_BUSY_CYCLES"):
        print(f"    -> MFMA busy cycles / SQ busy cycles = {g['SQ_VALU_MFMA_BUSY_CYCLES'] / g['SQ_BUSY_CYCLES']:.3f}")
