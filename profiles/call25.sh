mkdir -p gpurun_out/r2
run() { name=$1; shift; env "$@" timeout 500 python bench.py --no-cpu-baseline --no-step-sweep --steps 2 ${BARGS} > gpurun_out/r2/ov_$name.json 2> gpurun_out/r2/ov_$name.log; python -c "
import json; d=json.load(open('gpurun_out/r2/ov_$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['kernel'][:30], r['achieved'], r['frac'], r['avg_launch_us'])"; }
BARGS="--batch 384 --streams 3"
run g128 TTS_HIP_DAC_GROUP=128
run g384 TTS_HIP_DAC_GROUP=384
