mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py tests/test_gpu_orpheus.py tests/test_gpu_kokoro.py tests/test_gpu_gemv_rows.py -x -q > gpurun_out/r2/t_call4.log 2>&1; tail -4 gpurun_out/r2/t_call4.log
for cfg in "384 1" "128 1"; do set -- $cfg; timeout 300 python bench.py --batch $1 --streams $2 --no-cpu-baseline > gpurun_out/r2/b4_$1x$2.json 2> gpurun_out/r2/b4_$1x$2.log; python -c "
import json,sys
d=json.load(open('gpurun_out/r2/b4_$1x$2.json'))
print('$1x$2', d['value'], d['ms_per_decode_step'], d['phase_ms'], {k:(round(v['ms']/v['launches']*1e3,2)) for k,v in d.get('kernel_classes',{}).items()})
"; done
timeout 120 python profiles/orpheus_bench.py > gpurun_out/r2/orpheus_defaults.log 2>&1; tail -2 gpurun_out/r2/orpheus_defaults.log
