mkdir -p gpurun_out/r4
{
timeout 600 python -m pytest tests/test_gpu_kokoro.py tests/test_gpu_dac.py -q -s 2>&1 | grep -E "passed|failed|^E |kokoro-82m" | tail -8
timeout 300 python profiles/kokoro_bench.py 2>&1 | tail -3
} > gpurun_out/r4/kokoro_b3.txt 2>&1
cat gpurun_out/r4/kokoro_b3.txt
