mkdir -p gpurun_out/r2
{ timeout 40 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|Error|error" | tail -6
  echo "== dac_bench TTS_HIP_DAC_BF16X3=1"; TTS_HIP_DAC_BF16X3=1 timeout 30 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_conv7|rror"; } > gpurun_out/r2/dac_bf16x3.txt 2>&1
cat gpurun_out/r2/dac_bf16x3.txt
