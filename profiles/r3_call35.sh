# end of round (after attn_short was restored to one row per wave): full GPU suite, smoke(), default bench line with the final library
mkdir -p gpurun_out/r3
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
ulimit -c 0
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -12 > gpurun_out/r3/gpu_tests_call35.txt
cat gpurun_out/r3/gpu_tests_call35.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 1800 python bench.py > gpurun_out/r3/bench_default_call35.json 2> gpurun_out/r3/bench_default_call35.log
tail -3 gpurun_out/r3/bench_default_call35.log | cut -c1-300
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_default_call35.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
print('roofline', {k: d['roofline'].get(k) for k in ('kernel', 'bound', 'achieved', 'frac', 'traffic', 'avg_launch_us')})
for f in d['roofline_families']: print('  ', f['kernel'][:70], f['bound'], f['achieved'], f['frac'], f.get('fp32_equivalent_TFLOPs'), f['share_of_kernel_time'])
print('long', json.dumps(d.get('long_utterances'))[:400])
print('b1', d['decode_step_batch1']['steps_1024'])
print('secondary', {k: (v.get('value'), v.get('ms_per_decode_step')) for k, v in d['secondary'].items()})
PY
