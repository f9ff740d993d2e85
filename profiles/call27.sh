mkdir -p gpurun_out/r2/pmc
export TMPDIR=/tmp
R=$PWD
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc64_$ctr -- python $R/profiles/dac_bench.py 248 1 --batch=64 --no-warmup > $R/gpurun_out/r2/pmc/dac64_$ctr.log 2>&1
  f=$(find /tmp/pmc64_$ctr -name "*counter_collection.csv" | head -1); cp "$f" $R/gpurun_out/r2/pmc/dac64_$ctr.csv
done
cd $R
python profiles/pmc_summary.py gpurun_out/r2/pmc/dac64_FETCH_SIZE.csv gpurun_out/r2/pmc/dac64_WRITE_SIZE.csv > gpurun_out/r2/pmc/pmc_fetch_write_dac_group64.txt; head -8 gpurun_out/r2/pmc/pmc_fetch_write_dac_group64.txt | cut -c1-150
