mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_kokoro.py -q -s > gpurun_out/r2/t_call12.log 2>&1; grep -E "passed|failed|kokoro-82m|Error" gpurun_out/r2/t_call12.log | tail -5
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-step-sweep --steps 2 "$@" > gpurun_out/r2/ov_$name.json 2> gpurun_out/r2/ov_$name.log; python -c "
import json; d=json.load(open('gpurun_out/r2/ov_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
run b384s2 --batch 384 --streams 2
run b256s2 --batch 256 --streams 2
run b512s1 --batch 512 --streams 1
run b256s3 --batch 256 --streams 3
