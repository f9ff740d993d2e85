mkdir -p gpurun_out/r2
timeout 600 python profiles/dbg_dac.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_dac.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -5
for p in 0 1; do echo "== TTS_HIP_DAC_CONV1_DIRECT=$p"; TTS_HIP_DAC_CONV1_DIRECT=$p timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | tail -6; done | tee gpurun_out/r2/dac_conv1_direct.log
