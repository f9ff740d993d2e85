mkdir -p gpurun_out/r2
for cfg in "512 3" "448 3" "512 2"; do set -- $cfg; timeout 600 python bench.py --batch $1 --streams $2 --steps 2 --warmup 1 --no-cpu-baseline --no-step-sweep > gpurun_out/r2/b_$1x$2.json 2> gpurun_out/r2/b_$1x$2.log; python - $1 $2 <<'PY'
import json,sys
try:
    d=json.load(open(f'gpurun_out/r2/b_{sys.argv[1]}x{sys.argv[2]}.json')); print(sys.argv[1:], d['value'], d['ms_per_step'])
except Exception as e: print(sys.argv[1:], 'failed', e)
PY
done
