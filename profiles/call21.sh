mkdir -p gpurun_out/r2
timeout 200 python profiles/dac_bench.py 248 3 --batch=32 --prof 2>&1 | grep -E "^batch|dac_" > gpurun_out/r2/dac_polysin.log; cat gpurun_out/r2/dac_polysin.log
timeout 900 python -m pytest tests/test_gpu_dac.py tests/test_gpu_snac.py tests/test_gpu_runner.py tests/test_gpu_orpheus.py tests/test_gpu_dia.py -q 2>&1 | tail -4
timeout 500 python bench.py --no-cpu-baseline --no-roofline --no-step-sweep --steps 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default bench', d['value'], d['ms_per_step'])"
