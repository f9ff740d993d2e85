mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests/test_gpu_dia.py tests/test_gpu_orpheus.py tests/test_gpu_kokoro.py tests/test_gpu_runner.py -q -s > gpurun_out/r2/t_call7.log 2>&1; grep -E "passed|failed|kokoro own|FAILED|Error" gpurun_out/r2/t_call7.log | tail -15
timeout 300 python profiles/dia_bench.py > gpurun_out/r2/dia_batch.log 2>&1; tail -6 gpurun_out/r2/dia_batch.log
