mkdir -p gpurun_out/r4
{
timeout 600 python -m pytest tests/test_gpu_kokoro.py tests/test_gpu_snac.py tests/test_gpu_dac.py -q 2>&1 | grep -E "passed|failed|^E " | tail -4
timeout 300 python profiles/kokoro_bench.py 2>&1 | tail -6
} > gpurun_out/r4/kokoro_epilogue.txt 2>&1
cat gpurun_out/r4/kokoro_epilogue.txt
