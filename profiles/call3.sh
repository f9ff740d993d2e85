mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parler.py -x -q -k "many_rows or every_tile_shape or lockstep or full_size" > gpurun_out/r2/t_tile.log 2>&1; tail -5 gpurun_out/r2/t_tile.log
for cfg in "128 3" "384 1" "256 1" "192 2"; do set -- $cfg; timeout 300 python bench.py --batch $1 --streams $2 --no-cpu-baseline > gpurun_out/r2/b_tile_$1x$2.json 2> gpurun_out/r2/b_tile_$1x$2.log; python -c "
import json,sys
d=json.load(open('gpurun_out/r2/b_tile_$1x$2.json'))
print('$1x$2', d['value'], d['ms_per_decode_step'], d['phase_ms'], {k:(v['ms'],v['launches']) for k,v in d.get('kernel_classes',{}).items()})
"; done
