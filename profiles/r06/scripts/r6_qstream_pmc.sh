#!/bin/bash
# HBM-side fetch of qgemv_stream_kernel on the LM head (512 MB of codes, past the 256 MB Infinity Cache) beside the bench's plain read of the same bytes:
# FETCH_SIZE per dispatch (rocprofv3 --pmc, its own run with --kernel-trace only; on gfx950 it reports half the bytes of a wide streaming read: doubled below)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfgname in w8d2rt1 w8d2rt1lds; do
  rm -rf /tmp/pmc_qs
  QSTREAM_SHAPE=head QSTREAM_KS=1 QSTREAM_CFG=$cfgname timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_qs -- $R/profiles/qstream_bench 8 > $O/qstream_pmc_$cfgname.log 2>&1
  f=$(find /tmp/pmc_qs -name "*counter_collection.csv" | head -1)
  python - "$f" $cfgname <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "fill" in k or "ref_" in k or "fold" in k: continue
    # FETCH_SIZE is in KB on this rocprofv3; x 2: the gfx950 correction for wide streaming reads
    print(f"{sys.argv[2]:12s} {k:60s} dispatches {len(v):4d}  FETCH_SIZE mean {sum(v)/len(v):12.1f} KB  x 2 = {2*sum(v)/len(v)/1e3:8.1f} MB")
PY
done 2>&1 | tee $O/qstream_fetch_size.txt
