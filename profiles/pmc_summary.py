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


# ---- optional: per-kernel-class JSON for bench.py's roofline.traffic ---------------------------------------
import json, os
CLASS = [("attn_short_kernel", "attn_cross"), ("attn_rows_kernel", "attn_self"), ("attn_walk_kernel", "attn_self"), ("attn_kernel", "attn_self"), ("attn_combine", "attn_self"), ("gemm_tile_kernel", "gemm_qkv"), ("gemm16_kernel", "gemm_qkv"),
         ("ln_rows", "ln"), ("pack_", "pack"), ("resunit_b3_kernel", "dac_resunit"), ("resunit_t7_kernel", "dac_resunit"), ("conv_b3p_kernel<7", "dac_conv7"), ("conv1d_mfma_b3_kernel", "dac_conv7"),
         ("conv1d_mfma_kernel<7", "dac_conv7"), ("conv_b3p_kernel<1", "dac_conv1"), ("snake_split_kernel", "dac_conv1"), ("conv1d_mfma_kernel<1", "dac_conv1"),
         ("conv1x1_direct_kernel", "dac_conv1"), ("convt_b3_kernel", "dac_convt"), ("convt1d_mfma_kernel", "dac_convt"),
         ("conv1d_cout1_kernel", "dac_final"), ("dac_embed", "dac_embed"), ("embed_rows_kernel", "embed")]   # first match wins
if os.environ.get("PMC_JSON_OUT"):
    agg = {}
    for name, cs in acc.items():
        for pat, cls in CLASS:
            if pat in name:
                a = agg.setdefault(cls, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                for cn, (tot, n) in cs.items():
                    if cn in a:
                        a[cn][0] += tot
                        a[cn][1] += n
                break
    out = {"workload": json.loads(os.environ.get("PMC_WORKLOAD", "{}")), "kernels": {},
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 "
                     "(FETCH_SIZE counts half the bytes of wide coalesced reads on gfx950, MI355X_MICROARCH.md)"}
    for cls, a in agg.items():
        f = a["FETCH_SIZE"][0] / max(a["FETCH_SIZE"][1], 1)
        w = a["WRITE_SIZE"][0] / max(a["WRITE_SIZE"][1], 1)
        out["kernels"][cls] = {"fetch_kb_per_launch": f, "write_kb_per_launch": w, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
                               "launches": a["FETCH_SIZE"][1]}
    json.dump(out, open(os.environ["PMC_JSON_OUT"], "w"), indent=1)
