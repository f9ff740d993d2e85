# round 6, end of round: the whole GPU suite, smoke(), the driver's bench command (every section under the default time budget), kernel-trace stats of
# one runner's pass (the roofline's kernel: average duration per launch must agree with the line's HIP-event figure)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "passed|failed|^E |^FAILED|rror|streams identical|relative logit error by cached" | tail -20 | tee $O/gpu_tests_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke_final.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6/bench_driver_flags.json') if l.startswith('{')][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "steps", d["steps"], "warmup", d["warmup"])
print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","frac","traffic","avg_launch_us") if k in d["roofline"]})
print("time_budget", d["time_budget"])
lu=d.get("long_utterances",{})
for k in ("uniform","ragged_stream","ragged","uniform_same_mix"):
    print(k, {kk: vv for kk, vv in lu.get(k,{}).items() if kk not in ("note","workload_note")})
print("b1", d.get("decode_step_batch1",{}).get("steps_1024"))
print("e2e", d.get("generate_batch1_end_to_end",{}).get("top_k_50"))
print("cpu", d.get("cpu_baseline"))
for n,v in d.get("secondary",{}).items(): print(n, v.get("value"), v.get("unit"), v.get("ms_per_decode_step"), v.get("error"))
PY
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_bench -- python $R/bench.py --steps 2 --warmup 1 --streams 1 --no-long --no-secondary --no-e2e --no-step-sweep --no-cpu-baseline --no-roofline > $O/bench_kt.log 2>&1; f=$(find /tmp/kt_bench -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_bench_b1024_s1.csv)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6/kernel_stats_bench_b1024_s1.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']}%")
PY
