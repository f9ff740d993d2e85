# round 4, end-of-round validation: full GPU suite, smoke(), the default bench line, kernel-trace stats of one runner, the 3-runner per-stream timeline
mkdir -p gpurun_out/r4
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 > $O/gpu_tests_final.txt; cat $O/gpu_tests_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 2400 python bench.py --time-budget-s 0 > $O/bench_default_final.json 2> $O/bench_default_final.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_default_final.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
print('roofline', {k: d['roofline'].get(k) for k in ('kernel', 'bound', 'achieved', 'frac', 'traffic', 'avg_launch_us')})
for f in d['roofline_families']: print('  ', f['kernel'][:60], f['bound'], f['achieved'], f['frac'], f.get('traffic'), f['share_of_kernel_time'])
print('long', json.dumps(d.get('long_utterances'))[:900])
print('b1', d.get('decode_step_batch1', {}).get('steps_1024'))
print('e2e', json.dumps(d.get('generate_batch1_end_to_end'))[:600])
print('secondary', {k: (v.get('value'), v.get('ms_per_decode_step'), (v.get('roofline') or {}).get('traffic')) for k, v in d.get('secondary', {}).items()})
print('cpu', d.get('cpu_baseline'))
PY
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/bench.py --batch 1024 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-step-sweep --no-long --no-secondary --no-e2e > $O/kt_bench_s1.log 2>&1; cp "$(find /tmp/kt1 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_bench_b1024_s1.csv; head -12 $O/kernel_stats_bench_b1024_s1.csv | cut -c1-150)
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt3 -- python $R/bench.py --batch 1024 --streams 3 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-step-sweep --no-long --no-secondary --no-e2e > $O/kt_bench_s3.log 2>&1; f=$(find /tmp/kt3 -name '*kernel_trace.csv' | head -1); python $R/profiles/overlap_timeline.py "$f" > $O/timeline_3runners.txt 2>&1; cat $O/timeline_3runners.txt)
