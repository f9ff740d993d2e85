timeout 900 python -m pytest tests/test_gpu_kokoro.py -x -q 2>&1 | tail -4
