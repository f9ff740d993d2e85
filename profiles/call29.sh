mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_kokoro.py -q -s 2>&1 | grep -E "passed|failed|kokoro-82m|Error|assert" | tail -6
timeout 300 python profiles/kokoro_bench.py 2>&1 | tail -3
TTS_HIP_KOKORO_MFMA=0 timeout 300 python profiles/kokoro_bench.py 2>&1 | tail -2
