mkdir -p gpurun_out/r2
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-step-sweep --steps 2 ${BARGS} > gpurun_out/r2/ov_$name.json 2> gpurun_out/r2/ov_$name.log; python -c "
import json; d=json.load(open('gpurun_out/r2/ov_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
BARGS="--batch 192 --streams 2"
run s2_base X=1
run s2_res40 TTS_HIP_DAC_LDS_RESERVE_KB=40
run s2_res40_shallow TTS_HIP_DAC_LDS_RESERVE_KB=40 TTS_HIP_TILE_DEEP=0
run s2_res72_shallow TTS_HIP_DAC_LDS_RESERVE_KB=72 TTS_HIP_TILE_DEEP=0
run s2_shallow TTS_HIP_TILE_DEEP=0
BARGS="--batch 128 --streams 3"
run s3_base X=1
run s3_res40_shallow TTS_HIP_DAC_LDS_RESERVE_KB=40 TTS_HIP_TILE_DEEP=0
BARGS="--batch 384 --streams 1"
run s1_shallow TTS_HIP_TILE_DEEP=0
