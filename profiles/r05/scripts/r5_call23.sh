# round 5, call 23: which kernels make a Kokoro-82M generation (400 phoneme ids): kernel-trace stats of profiles/kokoro_bench.py
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kk -- python $R/profiles/kokoro_bench.py 2>&1 | grep -E "phoneme|error" | tee $O/kokoro_bench_call23.txt
f=$(find /tmp/prof_kk -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_kokoro_call23.csv
python - "$f" <<'PY' | tee -a $O/kokoro_bench_call23.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over the whole run (2 x durations + 2 x generation at 64 ids, the same at 400 ids)")
for r in rows[:28]:
    print(f"{r['Name'][:105]:105s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {100*float(r['TotalDurationNs'])/tot:5.1f}")
PY
