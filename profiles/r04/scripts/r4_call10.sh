mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py -q -x -k "mid_flight or generate_stream or compaction or lockstep or device_resident" 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -14
