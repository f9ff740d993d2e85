mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profkk -- python $R/profiles/kokoro_bench.py > $R/gpurun_out/r2/prof_kk.log 2>&1
cd $R; f=$(find /tmp/profkk -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r2/kernel_stats_kokoro_82m_linear_mfma.csv; python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:18]: print('%6.2f%% %9.1f us x %5s  %s'%(100*int(r['TotalDurationNs'])/tot, float(r['AverageNs'])/1e3, r['Calls'], r['Name'][:100]))
PY
