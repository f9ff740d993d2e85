#!/usr/bin/env python3
"""pmc_kernel_summary.py <kernel-name regex> <counter_collection.csv ...>: per kernel (template arguments shortened) the mean of every counter per dispatch,
and the derived shares (wave cycles parked / stalled / issuing, MFMA busy, LDS conflicts)."""
import csv, re, sys, collections
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[2:]:
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError as e:
        print(f"# {path}: {e}"); continue
    for r in rows:
        name = r.get("Kernel_Name", "")
        if not pat.search(name): continue
        name = name[:110]
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for name, ctrs in sorted(acc.items()):
    g = {c: s / n for c, (s, n) in ctrs.items()}
    n = max(v[1] for v in ctrs.values())
    print(f"{name}   ({n} dispatches)")
    line = []
    if g.get("SQ_WAIT_ANY") is not None and g.get("SQ_ACTIVE_INST_ANY"):
        tot = g["SQ_WAIT_INST_ANY"] + g["SQ_WAIT_ANY"] + g["SQ_ACTIVE_INST_ANY"]
        line.append(f"wave cycles: parked {g['SQ_WAIT_ANY'] / tot:.2f}, issue stalls {g['SQ_WAIT_INST_ANY'] / tot:.2f}, issuing {g['SQ_ACTIVE_INST_ANY'] / tot:.2f}")
    if g.get("SQ_VALU_MFMA_BUSY_CYCLES") and g.get("SQ_BUSY_CYCLES"):
        line.append(f"MFMA busy / SQ busy {g['SQ_VALU_MFMA_BUSY_CYCLES'] / g['SQ_BUSY_CYCLES']:.2f} (of 4 SIMDs x ... per SE)")
    if g.get("SQ_INSTS_MFMA") and g.get("SQ_INSTS_VALU"):
        line.append(f"VALU per MFMA {(g['SQ_INSTS_VALU'] - g['SQ_INSTS_MFMA']) / g['SQ_INSTS_MFMA']:.1f}, LDS per MFMA {g.get('SQ_INSTS_LDS', 0) / g['SQ_INSTS_MFMA']:.2f}")
    if g.get("SQ_LDS_IDX_ACTIVE"):
        line.append(f"LDS conflict share {g.get('SQ_LDS_BANK_CONFLICT', 0) / g['SQ_LDS_IDX_ACTIVE']:.3f}")
    if g.get("SQ_WAVE_CYCLES") and g.get("SQ_INSTS_MFMA") and g.get("SQ_WAVES"):
        line.append(f"quad-cycles per MFMA per wave {g['SQ_WAVE_CYCLES'] / g['SQ_INSTS_MFMA']:.1f}")
    for l in line: print("    " + l)
