mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_runner.py -q -k "generate_stream" 2>&1 | grep -E "passed|failed|^E " | tail -4
timeout 1500 python bench.py --steps 1 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline --no-secondary --no-e2e > gpurun_out/r4/bench_long.json 2> gpurun_out/r4/bench_long.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_long.json').read().strip().split('\n')[-1])
print('value', d['value'])
print(json.dumps(d.get('long_utterances'), indent=1)[:2500])
PY
tail -3 gpurun_out/r4/bench_long.log
