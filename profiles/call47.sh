mkdir -p gpurun_out/r2
for p in 0 1; do echo "== TTS_HIP_DAC_PAD=$p"; TTS_HIP_DAC_PAD=$p timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | tail -6; done | tee gpurun_out/r2/dac_row_stride.log
timeout 900 python -m pytest tests/test_gpu_dac.py tests/test_gpu_snac.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -5
