# XCD-aware channel-tile order of conv_b3p_kernel / convt_b3_kernel: parity, timing, FETCH_SIZE of a codec pass
mkdir -p gpurun_out/r3
{
B3_KNOBS="2" timeout 120 python profiles/b3_check.py 2>&1 | grep -E "BF16X3|rror" | tail -2
timeout 300 python -m pytest tests/test_gpu_dac.py -q -x 2>&1 | tail -2
timeout 60 python profiles/dac_bench.py 248 2 --batch=64 --prof 2>&1 | grep -E "batch=|dac_|rror"
} > gpurun_out/r3/xcd_tiles_call14.txt 2>&1
cat gpurun_out/r3/xcd_tiles_call14.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/pmc; mkdir -p $O
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_dac_$ctr -- python $R/profiles/dac_bench.py 248 1 --batch=64 --no-warmup > $O/dac_$ctr.log 2>&1
  f=$(find /tmp/pmc_dac_$ctr -name "*counter_collection.csv" | head -1); cp "$f" $O/dac2_$ctr.csv
done
cd $R; python profiles/pmc_summary.py $O/dac2_FETCH_SIZE.csv $O/dac2_WRITE_SIZE.csv > $O/pmc_fetch_write_dac_xcd_tiles.txt; rm -f $O/dac2_*.csv
grep -E "conv_b3p|convt_b3|resunit" $O/pmc_fetch_write_dac_xcd_tiles.txt | cut -c1-150
