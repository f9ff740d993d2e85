mkdir -p gpurun_out/r2
timeout 60 python -m pytest tests/test_gpu_dac.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -3
{ echo "== f16 tensors, conv lambdas inlined"; timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --f16 --prof 2>&1 | grep -E "batch=|dac_"; } > gpurun_out/r2/dac_f16_inlined.txt 2>&1
cat gpurun_out/r2/dac_f16_inlined.txt
