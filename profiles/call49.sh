mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profdac -- python $R/profiles/dac_bench.py 248 2 --batch=64 > $R/gpurun_out/r2/prof_dac.log 2>&1
cd $R; f=$(find /tmp/profdac -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r2/kernel_stats_dac_b64.csv; python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:16]: print('%6.2f%% %9.1f us x %4s  %s'%(100*int(r['TotalDurationNs'])/tot, float(r['AverageNs'])/1e3, r['Calls'], r['Name'][:90]))
PY
