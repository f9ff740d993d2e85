timeout 25 python -m pytest tests/test_gpu_dac.py -x -q -k "bf16x3" 2>&1 | grep -E "passed|failed|^E " | tail -4
