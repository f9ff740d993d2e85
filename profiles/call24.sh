mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_dac.py tests/test_gpu_runner.py -q 2>&1 | grep -E "passed|failed" | tail -2
run() { name=$1; shift; env "$@" timeout 500 python bench.py --no-cpu-baseline --no-step-sweep --steps 2 ${BARGS} > gpurun_out/r2/ov_$name.json 2> gpurun_out/r2/ov_$name.log; python -c "
import json; d=json.load(open('gpurun_out/r2/ov_$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['kernel'][:30], r['achieved'], r['frac'], r['avg_launch_us'], r['timing_source'][:30])"; }
BARGS="--batch 384 --streams 3"
run g32 X=1
run g64 TTS_HIP_DAC_GROUP=64
run g16 TTS_HIP_DAC_GROUP=16
BARGS="--batch 512 --streams 3"
run g32_512x3 X=1
