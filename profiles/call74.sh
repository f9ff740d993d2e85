mkdir -p gpurun_out/r2

export TMPDIR=/tmp
R=$PWD
cd /tmp; rm -rf /tmp/profkk2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profkk2 -- python $R/profiles/kokoro_bench.py > $R/gpurun_out/r2/prof_kk.log 2>&1
cd $R; tail -2 gpurun_out/r2/prof_kk.log; f=$(find /tmp/profkk2 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r2/kernel_stats_kokoro_82m_end_of_round.csv; python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms %.2f'%(tot/1e6))
for r in rows[:9]: print('%6.2f%% %9.1f us x %5s  %s'%(100*int(r['TotalDurationNs'])/tot, float(r['AverageNs'])/1e3, r['Calls'], r['Name'][:70]))
PY
