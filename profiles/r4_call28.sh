mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py -q 2>&1 | grep -E "passed|failed|^E |^FAILED" | tail -5
