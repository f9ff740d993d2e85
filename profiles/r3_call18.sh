# k = 1 convs of the wide classes: 256 x 256 / 128 x 512 tiles against 128 x 256
mkdir -p gpurun_out/r3
{
for v in 1 0; do
echo "== TTS_HIP_DAC_K1_VARIANT=$v"
TTS_HIP_DAC_K1_VARIANT=$v B3_KNOBS="2" timeout 120 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|rror" | tail -1
TTS_HIP_DAC_K1_VARIANT=$v timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_conv1|rror"
done
} > gpurun_out/r3/k1_tiles_call18.txt 2>&1
cat gpurun_out/r3/k1_tiles_call18.txt
