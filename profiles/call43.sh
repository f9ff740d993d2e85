timeout 900 python -m pytest tests/test_gpu_dia.py -x -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -30
