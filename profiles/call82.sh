mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_dac.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -4
{ echo "== tile embed"; timeout 200 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_"; } > gpurun_out/r2/dac_embed_tile.txt 2>&1
cat gpurun_out/r2/dac_embed_tile.txt
