mkdir -p gpurun_out/r2
for v in 3 4 5 20 30 300 400; do echo "variant $v"; TTS_HIP_DAC_VARIANT=$v timeout 200 python profiles/dac_bench.py 248 2 --batch=32 --prof 2>&1 | grep -E "^batch|dac_conv7"; done >> gpurun_out/r2/dac_variants2.log 2>&1; cat gpurun_out/r2/dac_variants2.log
