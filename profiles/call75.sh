mkdir -p gpurun_out/r2
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|error" | tail -4
timeout 600 python bench.py --workload kokoro --steps 2 --warmup 1 > gpurun_out/r2/bench_kokoro.json 2> gpurun_out/r2/bench_kokoro.log; echo "kokoro rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r2/bench_kokoro.json')); print('kokoro', d['value'], d.get('cpu_baseline',{}).get('value'))"
timeout 600 python profiles/kokoro_bench.py 2>&1 | tail -2
