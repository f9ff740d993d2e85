mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_parler.py -q -x 2>&1 | grep -E "passed|failed|^E " | tail -4
timeout 300 python profiles/dec_overlap.py 1024 96 2>&1 | head -3
timeout 600 python bench.py --steps 2 --warmup 1 --no-step-sweep --no-cpu-baseline --no-long --no-secondary --no-e2e > gpurun_out/r4/bench_short_call12.json 2> gpurun_out/r4/bench_short_call12.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_short_call12.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'attn', d['roofline']['frac'], d['roofline'].get('avg_launch_us'))
k = d.get('kernel_classes', {})
for n in ('ln', 'attn_self', 'gemm_qkv', 'gemm_fc1', 'attn_cross'): print(n, k.get(n))
PY
