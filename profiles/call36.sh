mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_dia.py tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
DIA_BENCH_UTTERANCES=4 timeout 300 python profiles/dia_bench.py 32 2>&1 | tail -5
bash profiles/call33.sh
