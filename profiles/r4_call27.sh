mkdir -p gpurun_out/r4
O=$GRAFT_REPO_ROOT/gpurun_out/r4
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_kk -- python $GRAFT_REPO_ROOT/profiles/kokoro_bench.py > $O/kokoro_kt.log 2>&1; cp "$(find /tmp/kt_kk -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_kokoro_r4.csv)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4/kernel_stats_kokoro_r4.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:18]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
