mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_dia.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
bash profiles/call33.sh
