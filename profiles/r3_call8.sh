mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -12 > gpurun_out/r3/gpu_tests_call8.txt
cat gpurun_out/r3/gpu_tests_call8.txt
timeout 1500 python bench.py > gpurun_out/r3/bench_default_call8.json 2> gpurun_out/r3/bench_default_call8.log
tail -5 gpurun_out/r3/bench_default_call8.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_default_call8.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'bound', 'achieved', 'frac')})
print('long', json.dumps(d.get('long_utterances'))[:900])
for k, v in (d.get('secondary') or {}).items():
    print(k, json.dumps({x: v.get(x) for x in ('value', 'error', 'ms_per_decode_step', 'roofline')})[:500])
PY
