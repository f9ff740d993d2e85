mkdir -p gpurun_out/r3
{
echo "== parity (fused residual units on by default)"
B3_KNOBS="0" timeout 120 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|rror" | tail -3
echo "== test_gpu_dac.py"
timeout 300 python -m pytest tests/test_gpu_dac.py -q -x 2>&1 | tail -5
for f in 1 0; do
echo "== TTS_HIP_DAC_FUSE=$f"
TTS_HIP_DAC_FUSE=$f timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_|rror"
done
} > gpurun_out/r3/fuse_call2.txt 2>&1
cat gpurun_out/r3/fuse_call2.txt
