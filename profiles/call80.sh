mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_dac.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -2
for t in 1 2; do echo "== ALPHA_TAB=$t"; TTS_HIP_DAC_ALPHA_TAB=$t timeout 200 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_" ; done > gpurun_out/r2/convt_occ2.txt 2>&1
cat gpurun_out/r2/convt_occ2.txt
