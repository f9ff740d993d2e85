mkdir -p gpurun_out/r2
for v in 0 1 2 10 100 200 111 112 211; do echo "variant $v"; TTS_HIP_DAC_VARIANT=$v timeout 200 python profiles/dac_bench.py 248 2 --batch=32 --prof 2>&1 | grep -E "^batch|dac_conv7"; done > gpurun_out/r2/dac_variants.log 2>&1; cat gpurun_out/r2/dac_variants.log
