mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_dia.py -x -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
timeout 900 python bench.py --workload dia --steps 2 --warmup 1 > gpurun_out/r2/bench_dia.json 2> gpurun_out/r2/bench_dia.log; echo rc=$?; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2/bench_dia.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_decode_step','encode_ms_per_utterance','dac_ms_per_pass','x_real_time_per_gpu')}, d['roofline']['frac'], d.get('cpu_baseline'))
PY
