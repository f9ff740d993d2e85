mkdir -p gpurun_out/r2/pmc
export TMPDIR=/tmp
R=$PWD
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  DIA_BENCH_UTTERANCES=4 timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcdia_$ctr -- python $R/profiles/dia_bench.py 8 > $R/gpurun_out/r2/pmc/dia_$ctr.log 2>&1
  f=$(find /tmp/pmcdia_$ctr -name "*counter_collection.csv" | head -1); cp "$f" $R/gpurun_out/r2/pmc/dia_$ctr.csv
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcdac_$ctr -- python $R/profiles/dac_bench.py 248 1 --batch=64 --no-warmup > $R/gpurun_out/r2/pmc/dac64_$ctr.log 2>&1
  f=$(find /tmp/pmcdac_$ctr -name "*counter_collection.csv" | head -1); cp "$f" $R/gpurun_out/r2/pmc/dac64_$ctr.csv
done
cd $R
python profiles/pmc_summary.py gpurun_out/r2/pmc/dia_FETCH_SIZE.csv gpurun_out/r2/pmc/dia_WRITE_SIZE.csv > gpurun_out/r2/pmc/pmc_fetch_write_dia_lockstep4.txt; grep -E "gemv_stream|attn_gqa_split|rms_fold" gpurun_out/r2/pmc/pmc_fetch_write_dia_lockstep4.txt | cut -c1-170
python profiles/pmc_summary.py gpurun_out/r2/pmc/dac64_FETCH_SIZE.csv gpurun_out/r2/pmc/dac64_WRITE_SIZE.csv > gpurun_out/r2/pmc/pmc_fetch_write_dac_group64_v2.txt; grep -E "direct|<1, 2, 2, 2" gpurun_out/r2/pmc/pmc_fetch_write_dac_group64_v2.txt | cut -c1-170
