mkdir -p gpurun_out/r3
{
echo "== parity"
B3_KNOBS="0 2" timeout 120 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|rror" | tail -3
timeout 300 python -m pytest tests/test_gpu_dac.py -q -x 2>&1 | tail -3
for k in 0 2; do
echo "== TTS_HIP_DAC_BF16X3=$k (variant 2)"
TTS_HIP_DAC_BF16X3=$k TTS_HIP_DAC_B3_VARIANT=2 timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_|rror"
done
} > gpurun_out/r3/snakevec_call4.txt 2>&1
cat gpurun_out/r3/snakevec_call4.txt
