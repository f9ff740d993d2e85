# round 4, call 1: the four-unit library on the GPU: full suite (incl. the new hygiene tests), then a short headline line for a same-box baseline
mkdir -p gpurun_out/r4
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "passed|failed|^E |^FAILED|rror|first divergence|worst|identical to" | tail -14 > gpurun_out/r4/gpu_tests_call1.txt
cat gpurun_out/r4/gpu_tests_call1.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-step-sweep --no-cpu-baseline --no-long --no-secondary > gpurun_out/r4/bench_short_call1.json 2> gpurun_out/r4/bench_short_call1.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_short_call1.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'attn', d['roofline']['frac'])
for f in d.get('roofline_families', []): print('  ', f['kernel'][:60], f['bound'], f['achieved'], f['frac'], f.get('share_of_kernel_time'))
print(json.dumps(d.get('kernel_classes', {}))[:1500])
PY
