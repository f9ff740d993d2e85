mkdir -p gpurun_out/r2
timeout 300 python bench.py --model small --batch 6 --streams 2 --audio-steps 40 --prompt-len 6 --steps 1 --warmup 1 > gpurun_out/r2/b6_small.json 2> gpurun_out/r2/b6_small.log; echo rc=$?; tail -3 gpurun_out/r2/b6_small.log; head -c 600 gpurun_out/r2/b6_small.json; echo
timeout 600 python -m pytest tests/test_gpu_runner.py -x -q > gpurun_out/r2/t_call6.log 2>&1; tail -4 gpurun_out/r2/t_call6.log
timeout 600 python bench.py > gpurun_out/r2/b6_default.json 2> gpurun_out/r2/b6_default.log; echo rc=$?; tail -2 gpurun_out/r2/b6_default.log; python -c "
import json
d=json.load(open('gpurun_out/r2/b6_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['share_of_kernel_time'])
for r in d['roofline_families']: print('  ', r['kernel'][:40], r['bound'], r['achieved'], r['frac'], r['avg_launch_us'], r['share_of_kernel_time'])
print(d['decode_step_batch1']); print(d['cpu_baseline'])
"
