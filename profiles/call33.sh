mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$PWD
cd /tmp && DIA_BENCH_UTTERANCES=4 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/profdia2 -- python $R/profiles/dia_bench.py 16 > $R/gpurun_out/r2/prof_dia2.log 2>&1
cd $R; f=$(find /tmp/profdia2 -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
# only the last 16 lock-step steps: take the last N dispatches covering them: find kernels after the last dia_embed with grid y=4
idx=[i for i,r in enumerate(rows) if 'dia_embed' in r['Kernel_Name'] and int(r['Grid_Size_Y'])==4]
start=idx[-8]; end=idx[-1]
d=collections.defaultdict(lambda:[0,0])
for r in rows[start:end]:
    n=r['Kernel_Name'][:60]+' g='+r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z']
    d[n][0]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); d[n][1]+=1
tot=sum(v[0] for v in d.values())
span=int(rows[end]['Start_Timestamp'])-int(rows[start]['Start_Timestamp'])
print('7 steps: kernel time %.3f ms, span %.3f ms'%(tot/1e6, span/1e6))
for n,v in sorted(d.items(), key=lambda kv:-kv[1][0])[:16]: print('%6.1f us x %4d  %5.1f%%  %s'%(v[0]/v[1]/1e3, v[1], 100*v[0]/tot, n))
PY
