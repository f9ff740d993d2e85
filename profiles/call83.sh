mkdir -p gpurun_out/r2
for p in 0 1 2; do echo "== TTS_HIP_DAC_PRIO=$p"; TTS_HIP_DAC_PRIO=$p timeout 200 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_conv7|dac_conv1"; done > gpurun_out/r2/dac_setprio.txt 2>&1
cat gpurun_out/r2/dac_setprio.txt
