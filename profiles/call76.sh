mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_kokoro.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -3
timeout 300 python profiles/kokoro_bench.py 2>&1 | tail -2
