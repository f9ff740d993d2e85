# round 5, call 15: which kernels make the Dia step (4 utterances x 2 rows): rocprofv3 kernel-trace stats of profiles/dia_bench.py
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dia -- python $R/profiles/dia_bench.py 64 2>&1 | grep -E " ms|error" | tee $O/dia_bench_call15.txt
f=$(find /tmp/prof_dia -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats_dia_call15.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
PY
t=$(find /tmp/prof_dia -name "*kernel_trace.csv" | head -1)
python $R/profiles/tools/trace_steps.py "$t" dia_embed_kernel 32 | tee $O/dia_step_kernels_call15.txt
