mkdir -p gpurun_out/r2
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|error" | tail -4
timeout 600 python bench.py --workload orpheus --steps 2 --warmup 1 > gpurun_out/r2/bench_orpheus.json 2> gpurun_out/r2/bench_orpheus.log; echo "orpheus rc=$?"
timeout 300 python profiles/orpheus_bench.py 2>&1 | tail -5 > gpurun_out/r2/orpheus_sampled.log; cat gpurun_out/r2/orpheus_sampled.log
timeout 900 python bench.py > gpurun_out/r2/bench_final.json 2> gpurun_out/r2/bench_final.log; echo "default rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2/bench_orpheus.json')); print('orpheus', d['value'], d.get('ms_per_decode_step'), (d.get('roofline') or {}).get('frac'))
d=json.load(open('gpurun_out/r2/bench_final.json')); print('default', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['decode_step_batch1']['steps_1024'])
PY
