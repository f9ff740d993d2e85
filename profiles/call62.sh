timeout 900 python -m pytest tests/test_gpu_dia.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -3
timeout 900 python bench.py --workload dia --steps 2 --warmup 1 > gpurun_out/bench_dia.json 2> gpurun_out/bench_dia.log; python -c "
import json; d=json.load(open('gpurun_out/bench_dia.json')); print(d['value'], d['ms_per_decode_step'], d['roofline']['frac'])"
