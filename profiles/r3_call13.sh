mkdir -p gpurun_out/r3
bash profiles/r3_pmc.sh > gpurun_out/r3/pmc_run.txt 2>&1
tail -45 gpurun_out/r3/pmc_run.txt
cp gpurun_out/r3/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -12 > gpurun_out/r3/gpu_tests_call13.txt
cat gpurun_out/r3/gpu_tests_call13.txt
timeout 1500 python bench.py > gpurun_out/r3/bench_default_call13.json 2> gpurun_out/r3/bench_default_call13.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_default_call13.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
print('roofline', {k: d['roofline'].get(k) for k in ('kernel', 'bound', 'achieved', 'frac', 'traffic', 'algorithmic_bytes_per_launch')})
for f in d['roofline_families']: print('  ', f['kernel'][:60], f['bound'], f['achieved'], f['frac'], f.get('fp32_equivalent_TFLOPs'), f.get('traffic'), f['share_of_kernel_time'])
print('long', json.dumps(d.get('long_utterances'))[:700])
print('b1', d['decode_step_batch1']['steps_1024'])
PY
