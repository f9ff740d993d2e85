export TMPDIR=/tmp
R=$PWD
for v in off on; do
cd /tmp; rm -rf /tmp/profkk_$v
if [ $v = off ]; then export TTS_KK_NO1X1=1; else unset TTS_KK_NO1X1; fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profkk_$v -- python $R/profiles/kokoro_bench.py > /dev/null 2>&1
cd $R; f=$(find /tmp/profkk_$v -name "*kernel_stats.csv" | head -1); python - "$f" $v <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print(sys.argv[2], 'total kernel ms %.2f'%(tot/1e6))
for r in rows:
    if 'kk_conv1' in r['Name']: print('   %8.1f us x %4s  %6.2f ms  %s'%(float(r['AverageNs'])/1e3, r['Calls'], int(r['TotalDurationNs'])/1e6, r['Name'][:40]))
PY
done
