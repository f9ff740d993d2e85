mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_dac.py tests/test_gpu_snac.py tests/test_gpu_runner.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -5
bash profiles/call49.sh 2>&1 | grep -E "direct|<1, 2, 2, 2"
timeout 1200 python bench.py > gpurun_out/r2/bench_default_v2.json 2> gpurun_out/r2/bench_default_v2.log; echo rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2/bench_default_v2.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])
PY
