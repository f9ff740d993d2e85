#!/bin/bash
# fused residual units: weight stages by global_load_lds (dac_wdma=1) against registers + ds_write_b128 (dac_wdma=0)
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_gpu_dac.py -q -x 2>&1 | tail -3
for w in 0 1; do
  echo "== dac_wdma=$w"
  python profiles/dac_bench.py 248 3 --batch=64 --prof --tune=dac_wdma=$w 2>&1 | grep -v "^$" | grep -i "batch=\|resunit\|RESUNIT"
done
} > gpurun_out/r6/wdma.txt 2>&1
cat gpurun_out/r6/wdma.txt
