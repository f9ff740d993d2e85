mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_kokoro.py -q -s 2>&1 | grep -E "passed|failed|kokoro-82m|Error|assert|stuck" | tail -6
timeout 200 python profiles/kokoro_bench.py 2>&1 | tail -3
TTS_HIP_KOKORO_LSTM_SPLIT=0 timeout 200 python profiles/kokoro_bench.py 2>&1 | tail -2
