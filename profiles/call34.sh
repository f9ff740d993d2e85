mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_dia.py -x -q 2>&1 | tail -5
DIA_BENCH_UTTERANCES=4 timeout 300 python profiles/dia_bench.py 32 2>&1 | tail -6
TTS_HIP_GEMV_STREAM=0 DIA_BENCH_UTTERANCES=4 timeout 300 python profiles/dia_bench.py 32 2>&1 | tail -4
timeout 200 ./profiles/gemv_bench 2 16 > gpurun_out/r2/gemv_bench_r2_r16.log 2>&1; grep -E "gemm16|BEST" gpurun_out/r2/gemv_bench_r2_r16.log
