mkdir -p gpurun_out/r3
{
for pl in 1 0; do
echo "== parity TTS_HIP_DAC_PLANES=$pl"
TTS_HIP_DAC_PLANES=$pl B3_KNOBS="2" timeout 120 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|rror" | tail -3
done
echo "== test_gpu_dac.py (planes on)"
timeout 300 python -m pytest tests/test_gpu_dac.py -q -x 2>&1 | tail -3
for cfg in "1 2" "1 0" "0 2"; do
set -- $cfg
echo "== TTS_HIP_DAC_PLANES=$1 variant $2"
TTS_HIP_DAC_PLANES=$1 TTS_HIP_DAC_B3_VARIANT=$2 timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_|rror"
done
} > gpurun_out/r3/planes_call9.txt 2>&1
cat gpurun_out/r3/planes_call9.txt
