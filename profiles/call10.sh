mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_dia.py -q > gpurun_out/r2/t_call10.log 2>&1; tail -2 gpurun_out/r2/t_call10.log
for w in dia orpheus kokoro; do timeout 600 python bench.py --workload $w --steps 2 --warmup 1 > gpurun_out/r2/bench_$w.json 2> gpurun_out/r2/bench_$w.log; echo "$w rc=$?"; tail -2 gpurun_out/r2/bench_$w.log | cut -c1-300; python -c "
import json
d=json.load(open('gpurun_out/r2/bench_$w.json'))
print(d['value'], d['ms_per_step'], d.get('ms_per_decode_step'), d.get('roofline'), d.get('cpu_baseline'), d.get('by_length'))
"; done
