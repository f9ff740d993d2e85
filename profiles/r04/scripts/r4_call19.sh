mkdir -p gpurun_out/r4
{
timeout 600 python -m pytest tests/test_gpu_dac.py tests/test_gpu_runner.py -q -k "not two_rank and not generate_stream" 2>&1 | grep -E "passed|failed|^E |^FAILED" | tail -5
echo "== planes-input transposed convs (default)"
timeout 120 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | grep -E "batch=|dac_"
echo "== fp32-input transposed convs (tune dac_convt_planes=0)"
timeout 120 python profiles/dac_bench.py 248 3 --batch=64 --prof --tune=dac_convt_planes=0 2>&1 | grep -E "batch=|dac_"
} > gpurun_out/r4/convt_planes.txt 2>&1
cat gpurun_out/r4/convt_planes.txt
