mkdir -p gpurun_out/r2
S=$(date +%s)
timeout 170 python -m pytest tests/test_gpu_kokoro.py tests/test_gpu_snac.py tests/test_gpu_dac.py tests/test_gpu_runner.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -4
echo "elapsed $(( $(date +%s) - S )) s"
