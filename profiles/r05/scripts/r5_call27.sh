# round 5, call 27: kernel-trace stats of the two one-sequence chains with the kernels of the end of the round (eager launches: rocprofv3 crashes in the capture of these loops)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B1_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b1 -- python $R/profiles/b1_prof.py 512 > $O/b1_prof_call27.log 2>&1; grep -E "ms/step" $O/b1_prof_call27.log | tee $O/b1_prof_call27.txt; tail -3 $O/b1_prof_call27.log | cut -c1-300
f=$(find /tmp/kt_b1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_b1_final.csv
ORPHEUS_BENCH_GREEDY_ONLY=1 TTS_HIP_LLAMA_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_or -- python $R/profiles/orpheus_bench.py > $O/orpheus_prof_call27.log 2>&1; grep -E "ms/step" $O/orpheus_prof_call27.log | cut -c1-140 | tee $O/orpheus_prof_call27.txt; tail -3 $O/orpheus_prof_call27.log | cut -c1-300
f=$(find /tmp/kt_or -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_orpheus_final.csv
python - <<'PY'
import csv
for name in ("b1", "orpheus"):
    rows = list(csv.DictReader(open(f"/root/repo/gpurun_out/r5/kernel_stats_{name}_final.csv")))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(name)
    for r in rows[:12]:
        print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:8.2f} pct {100*float(r['TotalDurationNs'])/tot:5.1f}")
PY
