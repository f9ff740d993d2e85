mkdir -p gpurun_out/r2
for v in 0 1 2; do echo "== TTS_HIP_DAC_C192=$v"; TTS_HIP_DAC_C192=$v timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | grep -E "batch=|conv7"; done | tee gpurun_out/r2/dac_c192.log
TTS_HIP_DAC_C192=1 timeout 900 python -m pytest tests/test_gpu_dac.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -3
