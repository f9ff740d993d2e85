# (call 33: after the generation-graph key fix; bench under rocgdb so that a fault names its kernel)
# state after the batch-1 chain + epilogue / straight-line load work: full GPU suite, default bench line, kernel-trace stats of one runner,
# batch-1 timeline, codec class table
mkdir -p gpurun_out/r3
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
ulimit -c 0
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -12 > gpurun_out/r3/gpu_tests_call33.txt
cat gpurun_out/r3/gpu_tests_call33.txt
timeout 1800 rocgdb -batch -ex 'set pagination off' -ex 'set confirm off' -ex run -ex 'info threads' -ex bt -ex 'x/12i $pc-24' --args python bench.py > gpurun_out/r3/bench_default_call33.out 2> gpurun_out/r3/bench_default_call33.log
grep '^{"metric' gpurun_out/r3/bench_default_call33.out > gpurun_out/r3/bench_default_call33.json; grep -v 'New Thread\|exited\]\|^{"metric' gpurun_out/r3/bench_default_call33.out | tail -30 | cut -c1-300
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_default_call33.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
print('roofline', {k: d['roofline'].get(k) for k in ('kernel', 'bound', 'achieved', 'frac', 'traffic', 'avg_launch_us')})
for f in d['roofline_families']: print('  ', f['kernel'][:60], f['bound'], f['achieved'], f['frac'], f.get('fp32_equivalent_TFLOPs'), f['share_of_kernel_time'])
print('long', json.dumps(d.get('long_utterances'))[:500])
print('b1', d['decode_step_batch1'])
print('secondary', {k: (v.get('value'), v.get('ms_per_decode_step')) for k, v in d['secondary'].items()})
kc = d['kernel_classes']; tot = sum(v['ms'] for v in kc.values())
for k, v in sorted(kc.items(), key=lambda kv: -kv[1]['ms']): print(f"  {k:16s} {v['ms']:8.1f} ms {v['launches']:6d} {100 * v['ms'] / tot:5.1f}%")
PY
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --batch 1024 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-step-sweep --no-long --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r3/kt_bench.log 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r3/kernel_stats_bench_b1024_s1.csv; head -12 $GRAFT_REPO_ROOT/gpurun_out/r3/kernel_stats_bench_b1024_s1.csv | cut -c1-160)
timeout 300 python profiles/b1_chain.py 2>&1 | grep -v Warning | tail -24 > gpurun_out/r3/b1_chain_call33.txt; cat gpurun_out/r3/b1_chain_call33.txt
timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | tail -8 > gpurun_out/r3/dac_classes_call33.txt; cat gpurun_out/r3/dac_classes_call33.txt
