# one tap per k-step in the wide k = 7 convs (7 k-steps per 16 channels instead of 8): parity and timing per tile variant
mkdir -p gpurun_out/r3
{
B3_KNOBS="2" timeout 120 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|rror" | tail -2
timeout 300 python -m pytest tests/test_gpu_dac.py -q -x 2>&1 | tail -2
for cfg in "1 0" "1 10" "1 12" "1 2" "0 2"; do
set -- $cfg
echo "== TTS_HIP_DAC_TAP7=$1 variant $2"
TTS_HIP_DAC_TAP7=$1 TTS_HIP_DAC_B3_VARIANT=$2 timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_conv7|rror"
done
} > gpurun_out/r3/tap7_call16.txt 2>&1
cat gpurun_out/r3/tap7_call16.txt
