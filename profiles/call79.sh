mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parler.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -2
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-step-sweep > gpurun_out/r2/b_ln_split.json 2> gpurun_out/r2/b_ln_split.log
python -c "
import json; d=json.load(open('gpurun_out/r2/b_ln_split.json')); print('ln per-variant kernels:', d['value'], d['roofline']['achieved'])"
