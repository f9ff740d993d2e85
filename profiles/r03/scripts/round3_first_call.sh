#!/bin/bash
# First GPU call of round 3 (about 90 s of box time): the bf16x3 convolution experiment beyond what round 2 could run.
#   1. parity of knob 2 (96-channel tile, never run) and of the three untested 64-channel tile shapes against the oracle;
#   2. codec pass of 64 utterances per (knob, variant): conv7 family time and pass time;
#   3. the DAC test file with the experiment forced on for every context (what a default flip would have to pass).
mkdir -p gpurun_out/r3
{
for k in 1 2; do for v in 0 1 2 3; do
  echo "== TTS_HIP_DAC_BF16X3=$k TTS_HIP_DAC_B3_VARIANT=$v"
  TTS_HIP_DAC_B3_VARIANT=$v B3_KNOBS="$k" timeout 60 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|rror" | tail -3
  TTS_HIP_DAC_BF16X3=$k TTS_HIP_DAC_B3_VARIANT=$v timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_conv7|rror"
done; done
echo "== test_gpu_dac.py with TTS_HIP_DAC_BF16X3=2 in the environment"
TTS_HIP_DAC_BF16X3=2 timeout 200 python -m pytest tests/test_gpu_dac.py tests/test_gpu_runner.py -q 2>&1 | grep -E "passed|failed|^E |^FAILED" | tail -12
} > gpurun_out/r3/b3_first_call.txt 2>&1
cat gpurun_out/r3/b3_first_call.txt
