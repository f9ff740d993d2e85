mkdir -p gpurun_out/r2
timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | tail -6 | tee gpurun_out/r2/dac_conv7_waves4.log
timeout 900 python -m pytest tests/test_gpu_dac.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
