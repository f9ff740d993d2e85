# resunit_t7_kernel (one tap per k-step in the fused residual units): parity and timing against resunit_b3_kernel
mkdir -p gpurun_out/r3
{
B3_KNOBS="2" timeout 120 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|rror" | tail -2
timeout 300 python -m pytest tests/test_gpu_dac.py -q -x 2>&1 | tail -2
for t in 1 0; do
echo "== TTS_HIP_DAC_TAP7=$t"
TTS_HIP_DAC_TAP7=$t timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_|rror"
done
} > gpurun_out/r3/resunit_t7_call17.txt 2>&1
cat gpurun_out/r3/resunit_t7_call17.txt
