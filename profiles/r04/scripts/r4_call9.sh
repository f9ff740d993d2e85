mkdir -p gpurun_out/r4
O=$GRAFT_REPO_ROOT/gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_dia.py tests/test_gpu_orpheus.py -q 2>&1 | grep -E "passed|failed|^E " | tail -4
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_dia -- python $GRAFT_REPO_ROOT/profiles/dia_bench.py 64 > $O/dia_kt.log 2>&1; f=$(find /tmp/kt_dia -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_dia_call9.csv)
grep -E "ms per step|decoder step" $O/dia_kt.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4/kernel_stats_dia_call9.csv')))
for r in rows[:22]:
    print(f"{r['Name'][:75]:75s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  {r['Percentage']}%")
PY
