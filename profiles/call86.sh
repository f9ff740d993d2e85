mkdir -p gpurun_out/r2
S=$(date +%s)
timeout 150 python -m pytest tests/test_gpu_parler.py tests/test_gpu_dia.py tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py tests/test_gpu_sampler.py tests/test_gpu_t5.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -4
echo "elapsed $(( $(date +%s) - S )) s"
