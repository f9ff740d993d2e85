mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$PWD
timeout 300 python profiles/b1_prof.py 512 2>&1 | tail -1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/profb1 -- python $R/profiles/b1_prof.py 256 > $R/gpurun_out/r2/prof_b1.log 2>&1
cd $R; tail -2 gpurun_out/r2/prof_b1.log; f=$(find /tmp/profb1 -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'feed_kernel' in r['Kernel_Name']]
start=idx[-33]; end=idx[-1]
d=collections.defaultdict(lambda:[0,0])
for r in rows[start:end]:
    n=r['Kernel_Name'][:70]+' g='+r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z']+' wg='+r['Workgroup_Size_X']
    d[n][0]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); d[n][1]+=1
tot=sum(v[0] for v in d.values())
span=int(rows[end]['Start_Timestamp'])-int(rows[start]['Start_Timestamp'])
print('32 steps: kernel time %.3f ms, span %.3f ms, launches/step %.1f'%(tot/1e6, span/1e6, (end-start)/32))
for n,v in sorted(d.items(), key=lambda kv:-kv[1][0])[:20]: print('%6.1f us x %4d  %5.1f%%  %s'%(v[0]/v[1]/1e3, v[1], 100*v[0]/tot, n))
PY
