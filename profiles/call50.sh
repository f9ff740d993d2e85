mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_dac.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -5
bash profiles/call49.sh
