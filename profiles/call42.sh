mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py tests/test_gpu_snac.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
timeout 300 python profiles/orpheus_bench.py 2>&1 | tail -4 | tee gpurun_out/r2/orpheus_sampled.log
