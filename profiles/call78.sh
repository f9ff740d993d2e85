mkdir -p gpurun_out/r2
for kb in 48; do
TTS_HIP_DAC_LDS_RESERVE_KB=$kb timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-step-sweep > gpurun_out/r2/b_reserve$kb.json 2> gpurun_out/r2/b_reserve$kb.log
python -c "
import json; d=json.load(open('gpurun_out/r2/b_reserve$kb.json')); print('reserve $kb KB:', d['value'], d['roofline']['achieved'])"
done
