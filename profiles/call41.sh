timeout 900 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py -x -q 2>&1 | tail -15
