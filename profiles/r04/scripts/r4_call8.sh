mkdir -p gpurun_out/r4
bash profiles/r04/scripts/r4_pmc_secondary.sh > gpurun_out/r4/pmc_secondary_run.txt 2>&1; tail -12 gpurun_out/r4/pmc_secondary_run.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_runner.py -q -k "two_rank" 2>&1 | grep -E "passed|failed|^E " | tail -5
timeout 900 python bench.py --steps 1 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline --no-long > gpurun_out/r4/bench_e2e_sec.json 2> gpurun_out/r4/bench_e2e_sec.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_e2e_sec.json').read().strip().split('\n')[-1])
print('value', d['value'])
print(json.dumps(d.get('generate_batch1_end_to_end'), indent=1))
for k, v in d.get('secondary', {}).items():
    print(k, v.get('value'), v.get('ms_per_decode_step'), v.get('x_real_time'), json.dumps(v.get('roofline', v.get('error')))[:400])
PY
