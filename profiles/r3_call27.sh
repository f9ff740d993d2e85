# phase-structured epilogue of conv_b3p_kernel + snake_vec in snake_split_kernel: parity, codec class table; then batch-1 chain default
mkdir -p gpurun_out/r3
{
B3_KNOBS="2" timeout 300 python profiles/b3_check.py 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_dac.py tests/test_gpu_upstream.py -q -x 2>&1 | tail -3
timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | tail -8
} > gpurun_out/r3/conv1_epilogue_call27.txt 2>&1
cat gpurun_out/r3/conv1_epilogue_call27.txt
