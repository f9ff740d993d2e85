mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_dac.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -2
{ echo "== f32"; timeout 200 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_"
  echo "== f16"; timeout 200 python profiles/dac_bench.py 248 2 --batch=64 --f16 --prof 2>&1 | grep -E "batch=|dac_"; } > gpurun_out/r2/convt_occ2b.txt 2>&1
cat gpurun_out/r2/convt_occ2b.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-step-sweep > gpurun_out/r2/b_convt_occ2.json 2> gpurun_out/r2/b_convt_occ2.log
python -c "
import json; d=json.load(open('gpurun_out/r2/b_convt_occ2.json')); print('convT 2 waves/SIMD:', d['value'], d['roofline']['achieved'])"
