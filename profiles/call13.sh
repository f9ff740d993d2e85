mkdir -p gpurun_out/r2
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-roofline --no-step-sweep --steps 2 "$@" > gpurun_out/r2/ov_$name.json 2> gpurun_out/r2/ov_$name.log; python -c "
import json; d=json.load(open('gpurun_out/r2/ov_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
run b384s3 --batch 384 --streams 3
run b512s2 --batch 512 --streams 2
run b512s3 --batch 512 --streams 3
run b256s4 --batch 256 --streams 4
