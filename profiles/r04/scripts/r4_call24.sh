timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py -q -k "mid_flight or generate_stream or compaction" 2>&1 | grep -E "passed|failed|^E " | tail -4
