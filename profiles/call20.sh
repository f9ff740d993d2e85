mkdir -p gpurun_out/r2
timeout 200 python profiles/dac_bench.py 248 3 --batch=32 --prof 2>&1 | grep -E "^batch|dac_" > gpurun_out/r2/dac_fastsin.log; cat gpurun_out/r2/dac_fastsin.log
timeout 600 python -m pytest tests/test_gpu_dac.py -q 2>&1 | tail -12
