mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests/test_gpu_dia.py tests/test_gpu_kokoro.py -q -s > gpurun_out/r2/t_call8.log 2>&1; grep -E "passed|failed|kokoro own|FAILED|Error|assert" gpurun_out/r2/t_call8.log | tail -15
