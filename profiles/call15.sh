mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py tests/test_gpu_dia.py tests/test_gpu_snac.py -q > gpurun_out/r2/t_call15.log 2>&1; tail -5 gpurun_out/r2/t_call15.log
timeout 200 python profiles/orpheus_bench.py 2>&1 | tail -2
TTS_HIP_ATTN_SPLIT=1 timeout 200 python profiles/orpheus_bench.py 2>&1 | tail -2
timeout 200 python profiles/dia_bench.py 2>&1 | tail -4
