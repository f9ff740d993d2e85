mkdir -p gpurun_out/r2/pmc
export TMPDIR=/tmp
R=$PWD
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -- python $R/profiles/dac_bench.py 248 1 --batch=32 --no-warmup > $R/gpurun_out/r2/pmc/dac_$ctr.log 2>&1
  f=$(find /tmp/pmc_$ctr -name "*counter_collection.csv" | head -1); cp "$f" $R/gpurun_out/r2/pmc/dac_$ctr.csv
done
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_busy -- python $R/profiles/dac_bench.py 248 1 --batch=32 --no-warmup > $R/gpurun_out/r2/pmc/dac_busy.log 2>&1
f=$(find /tmp/pmc_busy -name "*counter_collection.csv" | head -1); cp "$f" $R/gpurun_out/r2/pmc/dac_busy.csv
f=$(find /tmp/pmc_busy -name "*kernel_trace.csv" | head -1); cp "$f" $R/gpurun_out/r2/pmc/dac_busy_trace.csv
cd $R
python profiles/pmc_summary.py gpurun_out/r2/pmc/dac_FETCH_SIZE.csv gpurun_out/r2/pmc/dac_WRITE_SIZE.csv > gpurun_out/r2/pmc/pmc_fetch_write_dac_group32.txt; head -12 gpurun_out/r2/pmc/pmc_fetch_write_dac_group32.txt | cut -c1-160
python profiles/pmc_summary.py gpurun_out/r2/pmc/dac_busy.csv > gpurun_out/r2/pmc/pmc_mfma_busy_dac_group32.txt; head -24 gpurun_out/r2/pmc/pmc_mfma_busy_dac_group32.txt | cut -c1-140
ls -la gpurun_out/r2/pmc/
